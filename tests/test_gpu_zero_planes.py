"""Known-zero planes (lh_state zero bits): a launch neither reads a theta_i plane the library
knows to be all zeros (rhs_kernel / column_stepper_kernel NOICE) nor stores the identically zero
d theta_i (right_hand_side.jl:182, :359).  Everything here is a BITWISE comparison against the
general path (LH_TUNE zero=0: theta_i read, the d theta_i plane cleared at every launch), plus
the bookkeeping of the bits across upload / fill / copy / device-pointer hand-out."""
import ctypes as C

import numpy as np
import pytest

import case_model as M
import parity_cases as pc

pytestmark = pytest.mark.gpu
O = pc.O

NOICE_CASES = ["c2_richards_f64", "c2_richards_f32", "c3_coupled_f32", "c3_coupled_f64",
               "c4_richards_f64_128", "c5_percol_f64", "c1_dirichlet_f64", "single_cell_f64"]


def _run(case, tuning, *, mode, nsteps=0, dt=0.0):
    """Tendency (mode 'rhs' / 'rhs_dt') or stepped state ('step') with the given LH_TUNE spec."""
    import torch
    with pc.GpuModel(case) as g:
        F = g.F
        F.check(g.L.lh_set_tuning(g.ctx, tuning.encode()), g.ctx)
        Y, Ya = g.prognostic_and_aux()
        out = {}
        if mode in ("rhs", "rhs_dt"):
            dY = g.state(0)
            if mode == "rhs":
                g.rhs(Y, Ya, dY)
            else:
                tdt = torch.zeros(1, device="cuda",
                                  dtype=torch.float64 if case.dtype == np.float64 else torch.float32)
                F.check(g.L.lh_rhs_stable_dt(g.ctx, 0.0, Y, Ya, dY, 0.5, tdt.data_ptr()), g.ctx)
                F.check(g.L.lh_synchronize(g.ctx), g.ctx)
                out["dt"] = np.array([tdt.item()])
            out.update(g.tendencies(dY))
        else:
            F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, dt, nsteps, None), g.ctx)
            m = case.om.model
            if m != M.MODEL_HEAT:
                out["vl"] = g.download(Y, F.LH_VAR_VARTHETA_L)
                out["ti"] = g.download(Y, F.LH_VAR_THETA_I)
            if m != M.MODEL_RICHARDS:
                out["rhoe"] = g.download(Y, F.LH_VAR_RHOE_INT)
        assert g.status() == 0
        return out


def _loaded_hip_runtime():
    """The HIP runtime this process already has mapped (never a second copy)."""
    with open("/proc/self/maps") as fh:
        paths = {ln.split()[-1] for ln in fh if "libamdhip64.so" in ln}
    assert paths, "no HIP runtime mapped"
    return C.CDLL(sorted(paths)[0])


@pytest.mark.parametrize("name", NOICE_CASES)
@pytest.mark.parametrize("mode", ["rhs", "rhs_dt"])
def test_noice_tendency_is_bitwise_the_general_kernel(name, mode):
    case = pc.make_case(name, ncols=None if name != "c1_dirichlet_f64" else 5)
    assert not np.any(case.ti)      # GpuModel.upload turns the all-zero field into a fill
    a = _run(case, "zero=1", mode=mode)
    b = _run(case, "zero=0", mode=mode)
    for k in b:
        assert np.array_equal(a[k], b[k]), (name, mode, k)
    assert not np.any(a["ti"])


@pytest.mark.parametrize("name", ["c2_richards_f64", "c3_coupled_f32", "c5_percol_f64", "c4_richards_f64_128"])
@pytest.mark.parametrize("engine", ["persist=0", "persist=2"])
def test_noice_stepper_is_bitwise_the_general_kernels(name, engine):
    case = full = pc.make_case(name, ncols=300)
    dt = O.stable_dt(full.om, full.vl, full.ti, full.rhoe, 0.2, full.T_aux)
    a = _run(case, engine + ",zero=1", mode="step", nsteps=7, dt=dt)
    b = _run(case, engine + ",zero=0", mode="step", nsteps=7, dt=dt)
    for k in b:
        assert np.array_equal(a[k], b[k]), (name, engine, k)


def test_zero_bits_follow_every_writer():
    """upload / non-zero fill / copy / device pointer clear the bit (the general kernel runs and
    sees the ice); a zero fill sets it again; garbage uploaded into dY's theta_i plane is gone
    after rhs! (d theta_i = 0 is part of the result)."""
    case = pc.make_case("mixed_factors_f64")            # has ice
    noice = full = pc.make_case("c3_coupled_f64", ncols=130)
    want = pc.run_oracle_rhs(full)
    with pc.GpuModel(noice) as g:
        F = g.F
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        garbage = np.full((noice.ncols, noice.om.nlev), 7.5)
        g.upload(dY, F.LH_VAR_THETA_I, garbage)
        g.rhs(Y, Ya, dY)
        got = g.tendencies(dY)
        assert not np.any(got["ti"])
        pc.assert_tendencies_close(full, got, want)
        # now put ice into Y: the launch must see it (bit cleared by the upload)
        ice = np.zeros((noice.ncols, noice.om.nlev))
        ice[:, 10:20] = 0.05
        g.upload(Y, F.LH_VAR_THETA_I, ice)
        g.rhs(Y, Ya, dY)
        import dataclasses
        iced = dataclasses.replace(full, ti=ice)
        got = g.tendencies(dY)
        pc.assert_tendencies_close(iced, got, pc.run_oracle_rhs(iced))
        assert np.max(np.abs(got["vl"] - want["vl"])) > 0          # and it matters
        # a zero fill restores the no-ice state, a non-zero fill is seen
        F.check(g.L.lh_state_fill(g.ctx, Y, F.LH_VAR_THETA_I, 0.0), g.ctx)
        g.rhs(Y, Ya, dY)
        pc.assert_tendencies_close(full, g.tendencies(dY), want)
        F.check(g.L.lh_state_fill(g.ctx, Y, F.LH_VAR_THETA_I, 0.03), g.ctx)
        g.rhs(Y, Ya, dY)
        filled = dataclasses.replace(full, ti=np.full_like(full.ti, 0.03))
        pc.assert_tendencies_close(filled, g.tendencies(dY), pc.run_oracle_rhs(filled))
        # lh_state_copy carries the bit with the data
        Y2 = g.state(0)
        F.check(g.L.lh_state_copy(g.ctx, Y2, Y), g.ctx)
        g.rhs(Y2, Ya, dY)
        pc.assert_tendencies_close(filled, g.tendencies(dY), pc.run_oracle_rhs(filled))
        # writing through the device pointer is seen as well
        import torch
        F.check(g.L.lh_state_fill(g.ctx, Y, F.LH_VAR_THETA_I, 0.0), g.ctx)
        p, ls, cs = C.c_void_p(), C.c_int64(), C.c_int64()
        F.check(g.L.lh_state_device_ptr(g.ctx, Y, F.LH_VAR_THETA_I, C.byref(p), C.byref(ls), C.byref(cs)), g.ctx)
        F.check(g.L.lh_synchronize(g.ctx), g.ctx)
        assert cs.value == 1
        n = noice.om.nlev * ls.value
        host = np.zeros((noice.om.nlev, ls.value))
        host[:, :noice.ncols] = 0.03
        t = torch.from_numpy(host.reshape(-1))
        assert t.numel() == n
        hip = _loaded_hip_runtime()
        rc = hip.hipMemcpy(C.c_void_p(p.value), C.c_void_p(t.data_ptr()), C.c_size_t(n * 8), 1)
        assert rc == 0
        g.rhs(Y, Ya, dY)
        pc.assert_tendencies_close(filled, g.tendencies(dY), pc.run_oracle_rhs(filled))
    # a case with real ice is untouched by all of this
    got = pc.run_gpu_rhs(case)
    pc.assert_tendencies_close(case, got, pc.run_oracle_rhs(case))


def test_a_kept_device_pointer_stays_trusted_by_nobody():
    """A zero-copy host keeps the pointer lh_state_device_ptr gave it.  Whatever the library writes
    into that plane afterwards (a zero fill, a copy of a zero plane, the d theta_i = 0 clear), a
    later write through the OLD pointer must be seen by the next launch -- the plane stays
    'exposed' until lh_state_release_ptr; afterwards a zero fill is trusted again."""
    import dataclasses
    import torch
    full = pc.make_case("c3_coupled_f64", ncols=130)
    want = pc.run_oracle_rhs(full)
    filled = dataclasses.replace(full, ti=np.full_like(full.ti, 0.03))
    want_filled = pc.run_oracle_rhs(filled)
    hip = _loaded_hip_runtime()
    with pc.GpuModel(full) as g:
        F = g.F
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        p, ls, cs = C.c_void_p(), C.c_int64(), C.c_int64()
        F.check(g.L.lh_state_device_ptr(g.ctx, Y, F.LH_VAR_THETA_I, C.byref(p), C.byref(ls), C.byref(cs)), g.ctx)
        n = full.om.nlev * ls.value
        host = np.zeros((full.om.nlev, ls.value))
        host[:, :full.ncols] = 0.03
        t = torch.from_numpy(host.reshape(-1))

        def write_through_old_pointer():
            F.check(g.L.lh_synchronize(g.ctx), g.ctx)
            assert hip.hipMemcpy(C.c_void_p(p.value), C.c_void_p(t.data_ptr()), C.c_size_t(n * 8), 1) == 0

        # pointer first, THEN a zero fill, then the write through the old pointer
        F.check(g.L.lh_state_fill(g.ctx, Y, F.LH_VAR_THETA_I, 0.0), g.ctx)
        g.rhs(Y, Ya, dY)
        pc.assert_tendencies_close(full, g.tendencies(dY), want)
        write_through_old_pointer()
        g.rhs(Y, Ya, dY)
        pc.assert_tendencies_close(filled, g.tendencies(dY), want_filled)
        # the same after a copy of an all-zero state over it
        Z = g.state(0)
        F.check(g.L.lh_state_copy(g.ctx, Y, Z), g.ctx)
        for var, a in ((F.LH_VAR_VARTHETA_L, full.vl), (F.LH_VAR_RHOE_INT, full.rhoe)):
            g.upload(Y, var, a)
        write_through_old_pointer()
        g.rhs(Y, Ya, dY)
        pc.assert_tendencies_close(filled, g.tendencies(dY), want_filled)
        # an exposed d theta_i plane of a tendency state: dirtied through the pointer between two
        # launches, it is zero again after each of them
        q = C.c_void_p()
        F.check(g.L.lh_state_device_ptr(g.ctx, dY, F.LH_VAR_THETA_I, C.byref(q), None, None), g.ctx)
        for _ in range(2):
            F.check(g.L.lh_synchronize(g.ctx), g.ctx)
            assert hip.hipMemcpy(C.c_void_p(q.value), C.c_void_p(t.data_ptr()), C.c_size_t(n * 8), 1) == 0
            g.rhs(Y, Ya, dY)
            assert not np.any(g.tendencies(dY)["ti"])
        # released: a zero fill is trusted again (the launch no longer reads the plane, so even a
        # -- now illegal -- write through the old pointer would go unseen: that is the contract)
        F.check(g.L.lh_state_release_ptr(g.ctx, Y, F.LH_VAR_THETA_I), g.ctx)
        F.check(g.L.lh_state_fill(g.ctx, Y, F.LH_VAR_THETA_I, 0.0), g.ctx)
        g.rhs(Y, Ya, dY)
        pc.assert_tendencies_close(full, g.tendencies(dY), want)
        assert g.status() == 0


@pytest.mark.parametrize("name", ["mixed_smooth_f64", "mixed_smooth_f32", "mixed_factors_f64"])
def test_column_order_does_not_change_a_column(name):
    """theta_i is static, so callers may order the columns ice-free first (workloads.ice_sorted_order):
    waves are then all-ice-free or all-icy.  Columns are independent and a lane's result does not
    depend on what shares its wave: the reordered ensemble gives every column the same bits --
    tendency, step bound and stepped state."""
    import torch
    W = pc._w
    case = pc.make_case(name)
    order = W.ice_sorted_order(case.ti)
    assert 0 < np.count_nonzero(np.any(case.ti != 0, axis=1)) < case.ncols      # a mixed ensemble
    sorted_case = W.reorder_columns(case, order)
    out = []
    for c in (case, sorted_case):
        with pc.GpuModel(c) as g:
            F = g.F
            Y, Ya = g.prognostic_and_aux()
            dY = g.state(0)
            tdt = torch.zeros(1, device="cuda", dtype=torch.float64 if c.dtype == np.float64 else torch.float32)
            F.check(g.L.lh_rhs_stable_dt(g.ctx, 0.0, Y, Ya, dY, 0.4, tdt.data_ptr()), g.ctx)
            res = g.tendencies(dY)
            F.check(g.L.lh_synchronize(g.ctx), g.ctx)
            res["dt"] = np.array([tdt.item()])
            F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, 0.1 * float(tdt.item()), 4, None), g.ctx)
            res["Y_vl"], res["Y_re"] = g.download(Y, F.LH_VAR_VARTHETA_L), g.download(Y, F.LH_VAR_RHOE_INT)
            out.append(res)
    a, b = out
    for k in a:
        want = a[k] if k == "dt" else a[k][order]
        assert np.array_equal(want, b[k], equal_nan=True), (name, k)
