"""theta(z,t) parity: the device SSPRK33 stepper (lh_step_ssprk33: three fused
RHS+stage kernels per step) against the oracle's SSPRK33 on the same inputs.

Tolerance: BASELINE.json asks for theta(z,t) within 1e-6 relative of the
reference.  Asserted here: 1e-6 (the north-star bound) for every case, and the
much tighter bound the implementation actually meets -- 1e-11 relative to the
field's scale for Float64, 2e-5 for Float32 (a few hundred eps32 after tens of
steps; see assert_state_close for why 1e-6 is not meaningful in Float32) -- so a
regression shows up long before the north-star bound is at risk.
"""
import ctypes as C
import math

import numpy as np
import pytest

import case_model as M
import parity_cases as pc

pytestmark = pytest.mark.gpu
O = pc.O


def gpu_steps(case, dt, nsteps, bcv=None, math_mode=None):
    with pc.GpuModel(case, math_mode) as g:
        F = g.F
        Y, Ya = g.prognostic_and_aux()
        p = None
        if bcv is not None:
            bcv = np.ascontiguousarray(bcv, dtype=np.float64)
            p = bcv.ctypes.data_as(C.POINTER(C.c_double))
        F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, dt, nsteps, p), g.ctx)
        out = {}
        m = case.om.model
        if m != M.MODEL_HEAT:
            out["vl"] = g.download(Y, F.LH_VAR_VARTHETA_L)
            out["ti"] = g.download(Y, F.LH_VAR_THETA_I)
        if m != M.MODEL_RICHARDS:
            out["rhoe"] = g.download(Y, F.LH_VAR_RHOE_INT)
        assert g.status() == 0
        return out


def cpu_steps(case, dt, nsteps, bcv=None):
    cp = lambda a: None if a is None else a.copy()
    vl, ti, re = cp(case.vl), cp(case.ti), cp(case.rhoe)
    O.ssprk33(case.om, dt, nsteps, vl=vl, ti=ti, rhoe=re, T_aux=case.T_aux, bc_stage_values=bcv)
    out = {}
    if case.om.model != M.MODEL_HEAT:
        out["vl"], out["ti"] = vl, ti
    if case.om.model != M.MODEL_RICHARDS:
        out["rhoe"] = re
    return out


def stable_dt(case, courant=0.2):
    return O.stable_dt(case.om, case.vl, case.ti, case.rhoe, courant, case.T_aux)


def assert_state_close(case, got, want, tight):
    for k in want:
        g = got[k].astype(np.float64)
        w = want[k].astype(np.float64)
        scale = np.max(np.abs(w)) if k != "ti" else 1.0
        # north-star bound: 1e-6 relative.  Asserted for Float64 only: 1e-6 is 8
        # eps(Float32), less than two Float32 evaluations of the reference's own
        # arithmetic differ by after tens of steps (the oracle in Float32 vs
        # Float64 is ~1e-5 apart on these cases), so for Float32 the bound is `tight`.
        if case.dtype == np.float64:
            assert np.all(np.abs(g - w) <= 1e-6 * np.maximum(np.abs(w), 1e-3 * scale)), k
        # what the implementation meets
        assert np.max(np.abs(g - w)) <= tight * scale, (k, float(np.max(np.abs(g - w)) / scale))


@pytest.mark.parametrize("name,nsteps", [("c2_richards_f64", 40), ("c4_richards_f64_128", 20),
                                         ("c3_coupled_f64", 30), ("c5_percol_f64", 25),
                                         ("mixed_smooth_f64", 30), ("c1_dirichlet_f64", 60),
                                         ("richards_viscosity_f64", 30),
                                         # one Dirichlet component per face / per-column Dirichlet values: the stepper
                                         # evaluates those face states once per stage / once per call
                                         ("mixed_smooth_f64_hyddir", 30), ("mixed_smooth_f64_pcdir", 30),
                                         ("c1_dirichlet_f64_pcdir", 60)])
def test_theta_zt_matches_oracle_f64(name, nsteps):
    case = pc.make_case(name, ncols=None if name != "c1_dirichlet_f64" else 3)
    dt = stable_dt(case)
    assert math.isfinite(dt) and dt > 0
    got = gpu_steps(case, dt, nsteps)
    want = cpu_steps(case, dt, nsteps)
    assert_state_close(case, got, want, 1e-11)
    # the state moved: the comparison is not vacuous
    key = "vl" if "vl" in want else "rhoe"
    assert np.max(np.abs(want[key] - getattr(case, "vl" if key == "vl" else "rhoe"))) > 0


@pytest.mark.parametrize("name,nsteps", [("c3_coupled_f32", 30), ("c2_richards_f32", 40),
                                         ("mixed_smooth_f32", 20), ("heat_dirichlet_f32", 30)])
def test_theta_zt_matches_oracle_f32(name, nsteps):
    case = pc.make_case(name)
    dt = stable_dt(case)
    got = gpu_steps(case, dt, nsteps)
    want = cpu_steps(case, dt, nsteps)
    # measured on MI355X: 2e-7 ... 1.0e-6 of the field scale (profiles/round2_f32_bound.json,
    # round3_f32_drift.json); the bound leaves a factor 3, so a 10x regression fails
    assert_state_close(case, got, want, 3e-6)


def test_time_dependent_dirichlet_stage_values():
    """Dirichlet closures are evaluated by the host at the SSPRK33 stage times
    t, t+dt, t+dt/2 and handed over as numbers (heat_test_interface.jl:36-37)."""
    case = pc.make_case("heat_dirichlet_f64", ncols=64)
    dt = stable_dt(case)
    nsteps = 50
    t = dt * np.arange(nsteps)
    ts = np.stack([t, t + dt, t + dt / 2], axis=1)
    bcv = np.zeros((nsteps, 3, 2, 2))
    bcv[:, :, M.FACE_BOTTOM, M.COMP_ENERGY] = 290.0 + 5.0 * np.cos(2 * math.pi * ts / (40 * dt))
    bcv[:, :, M.FACE_TOP, M.COMP_ENERGY] = 280.0 - 3.0 * np.sin(2 * math.pi * ts / (25 * dt))
    got = gpu_steps(case, dt, nsteps, bcv)
    want = cpu_steps(case, dt, nsteps, bcv)
    assert_state_close(case, got, want, 1e-11)
    # and they differ from the constant-BC run (the stage values are really used)
    const = cpu_steps(case, dt, nsteps)
    assert np.max(np.abs(const["rhoe"] - want["rhoe"])) > 1e-3 * np.max(np.abs(want["rhoe"]))


def test_stepper_equals_three_rhs_calls():
    """One fused device step == the Shu-Osher combination of three lh_rhs calls."""
    case = pc.make_case("c3_coupled_f64", ncols=200)
    dt = stable_dt(case)
    got = gpu_steps(case, dt, 1)
    with pc.GpuModel(case) as g:
        F = g.F
        Y, _ = g.prognostic_and_aux()
        dY = g.state(0)
        U = g.state(0)
        vars_ = (F.LH_VAR_VARTHETA_L, F.LH_VAR_THETA_I, F.LH_VAR_RHOE_INT)
        get = lambda st: [g.download(st, v) for v in vars_]
        put = lambda st, arrs: [g.upload(st, v, a) for v, a in zip(vars_, arrs)]
        u0 = get(Y)
        g.rhs(Y, None, dY)
        k1 = get(dY)
        u1 = [a + dt * k for a, k in zip(u0, k1)]
        put(U, u1)
        g.rhs(U, None, dY)
        k2 = get(dY)
        u2 = [(3 * a + b + dt * k) * 0.25 for a, b, k in zip(u0, u1, k2)]
        put(U, u2)
        g.rhs(U, None, dY)
        k3 = get(dY)
        u3 = [(a + 2 * b + 2 * dt * k) * (1.0 / 3.0) for a, b, k in zip(u0, u2, k3)]
    for name, a in zip(("vl", "ti", "rhoe"), u3):
        scale = max(np.max(np.abs(a)), 1e-300)
        assert np.max(np.abs(got[name] - a)) <= 4e-16 * scale, name


# ---------------------------------------------------------------------------
# the reference's own integration tests, run on the device through the C ABI
# ---------------------------------------------------------------------------

def expected_equilibrium(z, z_interface, nu, S_s=1e-3, alpha=2.6, n=2.0, m=0.5):
    return np.where(z < z_interface, -S_s * (z - z_interface) + nu,
                    nu * (1 + (alpha * np.maximum(z - z_interface, 0.0)) ** n) ** (-m))


def test_heat_analytic_on_device():
    """test/SoilModel/heat_test_interface.jl:1-100 (MSE < 1e-6 vs the closed form)."""
    sp = M.default_soil(nu=0.495, nu_ss_gravel=0.1, nu_ss_om=0.1, nu_ss_quartz=0.1,
                        rho_c_ds=0.43314518988433487, kappa_solid=8.0, kappa_sat_unfrozen=0.57,
                        kappa_sat_frozen=2.29)
    n, dt, tf, A, omega = 60, 1e-4, 2.0, 5.0, 2 * math.pi
    bc = {(M.FACE_TOP, M.COMP_ENERGY): (M.BC_DIRICHLET, 0.0),
          (M.FACE_BOTTOM, M.COMP_ENERGY): (M.BC_DIRICHLET, A)}
    om = M.CaseModel(M.MODEL_HEAT, n, 0.0, 1.0, soil=sp, bc=bc)
    e = om.earth
    rho_c_s = sp.rho_c_ds
    rhoe0 = rho_c_s * (0.0 - e.T_0)
    ncols = 2
    case = pc.Case("heat_analytic", om, np.float64, ncols, vl=np.zeros((ncols, n)),
                   ti=np.zeros((ncols, n)), rhoe=np.full((ncols, n), rhoe0))
    nsteps = int(round(tf / dt))
    t = dt * np.arange(nsteps)
    ts = np.stack([t, t + dt, t + dt / 2], axis=1)
    bcv = np.zeros((nsteps, 3, 2, 2))
    bcv[:, :, M.FACE_BOTTOM, M.COMP_ENERGY] = A * np.cos(omega * ts)
    got = gpu_steps(case, dt, nsteps, bcv)
    z, _ = O.grid(0.0, 1.0, n)
    s = math.sqrt(omega / 2) * (1 + 1j)
    analytic = np.real((np.exp(s * (1 - z)) - np.exp(-s * (1 - z))) * A * np.exp(1j * omega * tf)
                       / (np.exp(s) - np.exp(-s)))
    T = e.T_0 + got["rhoe"][0] / rho_c_s
    assert np.mean((analytic - T) ** 2) < 1e-6
    # and theta(z,t) parity with the oracle after 20000 steps
    want = cpu_steps(case, dt, nsteps, bcv)
    assert np.max(np.abs(got["rhoe"] - want["rhoe"])) <= 1e-9 * np.max(np.abs(want["rhoe"]))


def test_coupled_equilibrium_on_device():
    """test/SoilModel/coupled.jl:1-120 in full on the device: n = 20 on (-2, 0),
    zero-flux BCs, 32 days at dt = 20 s (138 240 steps, 414 720 fused launches);
    the reference's two assertions as written (:117-118), conservation, and
    theta(z,t) parity with the oracle."""
    sp, vg = pc.coupled_soil()
    n, dt = 20, 20.0
    nsteps = int(60 * 60 * 24 * 32 / dt)
    om = M.CaseModel(M.MODEL_COUPLED, n, -2.0, 0.0, soil=sp, vg=vg,
                       bc=pc._flux_bcs(energy=0.0, hydrology=0.0))
    z, _ = O.grid(-2.0, 0.0, n)
    e = om.earth
    vl = np.full((1, n), 0.495)
    rho_c_s = sp.rho_c_ds + 0.495 * (e.cp_l * e.rho_liq)
    rhoe = (rho_c_s * ((289.0 + 5.0 * z) - e.T_0))[None, :]
    case = pc.Case("coupled_eq", om, np.float64, 1, vl=vl, ti=np.zeros((1, n)), rhoe=rhoe)
    got = gpu_steps(case, dt, nsteps)
    want = cpu_steps(case, dt, nsteps)
    # the reference's assertions (:117-118)
    assert math.sqrt(np.mean(got["vl"][0] - expected_equilibrium(z, -0.3, 0.5)) ** 2.0) < 1e-3
    rcs = sp.rho_c_ds + got["vl"][0] * (e.cp_l * e.rho_liq)
    temp = e.T_0 + got["rhoe"][0] / rcs
    assert math.sqrt(np.mean(temp - 284.0) ** 2.0) < 1e-3
    # theta(z,t) parity after 138 240 steps
    assert np.max(np.abs(got["vl"] - want["vl"]) / np.abs(want["vl"])) < 1e-6        # north star
    assert np.max(np.abs(got["vl"] - want["vl"])) < 1e-9
    assert np.max(np.abs(got["rhoe"] - want["rhoe"])) < 1e-8 * np.max(np.abs(want["rhoe"]))
    # zero-flux BCs and one flux per face: the device conserves mass and energy
    assert abs(got["vl"].sum() - vl.sum()) < 1e-11 * vl.sum()
    assert abs(got["rhoe"].sum() - rhoe.sum()) < 1e-11 * abs(rhoe.sum())
    assert np.all(got["ti"] == 0.0)


def test_richards_equilibrium_on_device():
    """test/SoilModel/richards_equation.jl:1-95 in full: 36 days at dt = 100 s on the
    device, the reference's assertion as written (:94)."""
    sp = M.default_soil(nu=0.495, S_s=1e-3)
    vg = M.default_vg(n=2.0, alpha=2.6, Ksat=0.0443 / 3600 / 100, theta_r=0.0)
    n, dt = 50, 100.0
    nsteps = int(60 * 60 * 24 * 36 / dt)
    om = M.CaseModel(M.MODEL_RICHARDS, n, -10.0, 0.0, soil=sp, vg=vg,
                       bc=pc._flux_bcs(hydrology=0.0))
    z, _ = O.grid(-10.0, 0.0, n)
    case = pc.Case("richards_eq", om, np.float64, 1, vl=np.full((1, n), 0.494),
                   ti=np.zeros((1, n)))
    got = gpu_steps(case, dt, nsteps)
    assert math.sqrt(np.mean(got["vl"][0] - expected_equilibrium(z, -0.56, 0.495)) ** 2.0) < 1e-4
    assert abs(got["vl"].sum() - case.vl.sum()) < 1e-11 * case.vl.sum()
    want = cpu_steps(case, dt, nsteps)
    assert np.max(np.abs(got["vl"] - want["vl"]) / np.abs(want["vl"])) < 1e-6


def test_sand_infiltration_on_device():
    """test/SoilModel/richards_equation.jl:98-173 setup (Dirichlet top, free
    drainage bottom), 10 of the 48 minutes; the reference's comparison data is a
    download and unavailable offline, so the check is parity with the oracle."""
    sp = M.default_soil(nu=0.287, S_s=1e-3)
    vg = M.default_vg(n=3.96, alpha=2.7, Ksat=34 / 3600 / 100, theta_r=0.075)
    bc = {(M.FACE_TOP, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, 0.267),
          (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_FREE_DRAINAGE, 0.0)}
    om = M.CaseModel(M.MODEL_RICHARDS, 150, -1.5, 0.0, soil=sp, vg=vg, bc=bc)
    case = pc.Case("sand", om, np.float64, 4, vl=np.full((4, 150), 0.1), ti=np.zeros((4, 150)))
    got = gpu_steps(case, 0.25, 2400)
    want = cpu_steps(case, 0.25, 2400)
    assert np.max(np.abs(got["vl"] - want["vl"]) / np.abs(want["vl"])) < 1e-6
    assert np.max(np.abs(got["vl"] - want["vl"])) < 1e-9
    assert got["vl"][0, -1] > 0.25 and got["vl"][0, 0] < 0.1001


# ---------------------------------------------------------------------------
# BASELINE full sizes: size-independent properties + sampled parity
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("workload", ["c2", "c3"])
def test_full_size_properties(workload):
    import bench
    N = 1_000_000
    case = bench.build_case(workload, N, 0)
    n = case.om.nlev
    with pc.GpuModel(case) as g:
        F = g.F
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        g.rhs(Y, Ya, dY)
        d = g.tendencies(dY)
        assert g.status() == 0
        # zero-flux BCs: the column sum of each tendency is a telescoping sum of face
        # fluxes = F_top - F_bottom = 0 up to rounding of the differences
        for k, scale_src in (("vl", None), ("rhoe", None)):
            if k not in d:
                continue
            col = np.abs(d[k].astype(np.float64).sum(axis=1))
            mag = np.abs(d[k].astype(np.float64)).sum(axis=1) + 1e-300
            tol = 64 * np.finfo(case.dtype).eps
            assert np.max(col / mag) < tol * n, (k, float(np.max(col / mag)))
        assert np.all(d["ti"] == 0)                       # d theta_i == 0 identically
        # sampled parity against the oracle: 4096 columns spread over the batch
        idx = np.linspace(0, N - 1, 4096).astype(np.int64)
        import dataclasses
        sl = lambda a: None if a is None else np.ascontiguousarray(a[idx])
        sub = dataclasses.replace(case, ncols=len(idx), vl=sl(case.vl), ti=sl(case.ti),
                                  rhoe=sl(case.rhoe), T_aux=sl(case.T_aux))
        want = pc.run_oracle_rhs(sub, nthreads=8)
        got = {k: v[idx] for k, v in d.items()}
        cw = 16.0 if case.dtype == np.float32 else 4.0
        pc.assert_tendencies_close(sub, got, want, cw, label="[full size]")
        # three device steps conserve mass (and energy) per column
        m0 = case.vl.astype(np.float64).sum(axis=1)
        dt = O.stable_dt(sub.om, sub.vl, sub.ti, sub.rhoe, 0.2)
        F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, dt, 3, None), g.ctx)
        vl1 = g.download(Y, F.LH_VAR_VARTHETA_L).astype(np.float64)
        rel = np.abs(vl1.sum(axis=1) - m0) / m0
        assert np.max(rel) < (1e-13 if case.dtype == np.float64 else 2e-6)
        assert np.max(np.abs(vl1 - case.vl)) > 0          # it moved
        # the same three steps by the other engine (three fused-stage launches per step instead
        # of the persistent column stepper) from the same start: bitwise the same state
        Y2, _ = g.prognostic_and_aux()
        F.check(g.L.lh_set_tuning(g.ctx, b"persist=0"), g.ctx)
        F.check(g.L.lh_step_ssprk33(g.ctx, Y2, Ya, 0.0, dt, 3, None), g.ctx)
        np.testing.assert_array_equal(g.download(Y2, F.LH_VAR_VARTHETA_L).astype(np.float64), vl1)
        if workload == "c3":
            np.testing.assert_array_equal(g.download(Y2, F.LH_VAR_RHOE_INT), g.download(Y, F.LH_VAR_RHOE_INT))


def test_full_size_c5_per_column_parameters():
    """BASELINE config 5 at its per-GPU size: 1e6 columns x 128 levels, per-column van Genuchten
    parameters and porosity, per-column infiltration flux at the top, free drainage at the bottom.
    Size-independent properties on EVERY column + sampled parity against the oracle."""
    import bench
    import dataclasses
    N = 1_000_000
    case = bench.build_case("c5", N, 0)
    om = case.om
    n = om.nlev
    dz = (om.zmax - om.zmin) / n
    with pc.GpuModel(case) as g:
        F = g.F
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        g.rhs(Y, Ya, dY)
        d = g.tendencies(dY)
        assert g.status() == 0
        assert np.all(d["ti"] == 0)
        # the column sum of the tendency telescopes to the two boundary fluxes:
        # dz sum_i d vl_i = -(F_top - F_bottom), F_top = -0.5 Ksat_c (prescribed),
        # F_bottom = -K(bottom cell) (free drainage, boundary_conditions.jl:328-356)
        diag = g.state(0b1111)
        F.check(g.L.lh_diagnostics(g.ctx, Y, Ya, diag), g.ctx)
        K0 = np.empty(N)
        F.check(g.L.lh_download_level(g.ctx, diag, F.LH_DIAG_K, 0, K0.ctypes.data_as(C.c_void_p)), g.ctx)
        f_top = np.asarray(om.percol_bc[(M.FACE_TOP, M.COMP_HYDROLOGY)], dtype=np.float64)
        lhs = dz * d["vl"].astype(np.float64).sum(axis=1)
        rhs = -(f_top - (-K0))
        scale = dz * np.abs(d["vl"].astype(np.float64)).sum(axis=1) + np.abs(f_top) + np.abs(K0)
        assert np.max(np.abs(lhs - rhs) / scale) < 64 * n * np.finfo(np.float64).eps
        # sampled parity: 4096 columns spread over the batch, with their own parameters
        idx = np.linspace(0, N - 1, 4096).astype(np.int64)
        sub_om = dataclasses.replace(om, percol={k: v[idx] for k, v in om.percol.items()},
                                     percol_bc={k: v[idx] for k, v in om.percol_bc.items()})
        sub = dataclasses.replace(case, om=sub_om, ncols=len(idx), vl=np.ascontiguousarray(case.vl[idx]),
                                  ti=np.ascontiguousarray(case.ti[idx]))
        want = pc.run_oracle_rhs(sub, nthreads=8)
        pc.assert_tendencies_close(sub, {k: v[idx] for k, v in d.items()}, want, 4.0, label="[c5 full size]")
        # three device steps: every column's water changes by exactly the boundary fluxes' work
        # to rounding is not a closed form (K_bottom moves); what is: both engines agree bitwise
        dt = O.stable_dt(sub.om, sub.vl, sub.ti, None, 0.2)
        F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, dt, 3, None), g.ctx)
        vl1 = g.download(Y, F.LH_VAR_VARTHETA_L)
        assert np.all(np.isfinite(vl1)) and np.max(np.abs(vl1 - case.vl)) > 0
        Y2, _ = g.prognostic_and_aux()
        F.check(g.L.lh_set_tuning(g.ctx, b"persist=0"), g.ctx)
        F.check(g.L.lh_step_ssprk33(g.ctx, Y2, Ya, 0.0, dt, 3, None), g.ctx)
        np.testing.assert_array_equal(g.download(Y2, F.LH_VAR_VARTHETA_L), vl1)


def test_stable_dt_matches_oracle_and_device_dt_stepping():
    """lh_stable_dt against the oracle's rule, and adaptive stepping with dt kept in
    device memory (lh_stable_dt_device -> [RCCL min all-reduce] ->
    lh_step_ssprk33_device_dt) against stepping with the same dt passed by value."""
    import torch
    for name in ("c2_richards_f64", "c3_coupled_f32", "mixed_smooth_f64", "c1_dirichlet_f64",
                 "c5_percol_f64", "heat_dirichlet_f64", "richards_viscosity_f64"):
        case = pc.make_case(name, ncols=None if name != "c1_dirichlet_f64" else 5)
        want = O.stable_dt(case.om, case.vl, case.ti, case.rhoe, 0.5, case.T_aux)
        with pc.GpuModel(case) as g:
            F = g.F
            Y, Ya = g.prognostic_and_aux()
            dt = C.c_double()
            F.check(g.L.lh_stable_dt(g.ctx, Y, Ya, 0.5, C.byref(dt)), g.ctx)
            rel = 1e-11 if case.dtype == np.float64 else 1e-4
            assert abs(dt.value - want) <= rel * want, (name, dt.value, want)
            if name not in ("c2_richards_f64", "c3_coupled_f32"):
                continue
            # adaptive loop, dt never leaves the device
            tdt = torch.zeros(1, device="cuda",
                              dtype=torch.float64 if case.dtype == np.float64 else torch.float32)
            dts = []
            for _ in range(5):
                F.check(g.L.lh_stable_dt_device(g.ctx, Y, Ya, 0.2, tdt.data_ptr()), g.ctx)
                _pkg().partition.global_min_dt(tdt)          # no-op without a process group
                F.check(g.L.lh_step_ssprk33_device_dt(g.ctx, Y, Ya, 0.0, tdt.data_ptr(), None), g.ctx)
                F.check(g.L.lh_synchronize(g.ctx), g.ctx)
                dts.append(float(tdt.item()))
            var = F.LH_VAR_VARTHETA_L
            a = g.download(Y, var)
        # same sequence of dt values passed from the host
        with pc.GpuModel(case) as g2:
            Y2, Ya2 = g2.prognostic_and_aux()
            for d in dts:
                g2.F.check(g2.L.lh_step_ssprk33(g2.ctx, Y2, Ya2, 0.0, d, 1, None), g2.ctx)
            b = g2.download(Y2, var)
        assert np.array_equal(a, b), name
        assert all(d > 0 and math.isfinite(d) for d in dts)


def _pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.mark.parametrize("name", ["c2_richards_f64", "c3_coupled_f32", "c3_coupled_f64",
                                  "mixed_smooth_f64", "c1_dirichlet_f64", "c5_percol_f64",
                                  "heat_dirichlet_f64", "richards_viscosity_f64", "single_cell_f64",
                                  "mixed_smooth_f32"])
def test_fused_rhs_stable_dt(name):
    """lh_rhs_stable_dt = lh_rhs (same tendencies, bit for bit) + the stable-step bound of
    lh_stable_dt / the oracle's rule from the same pass.  The fused bound is accumulated in
    Float32 whatever the working type (a safety estimate under a Courant factor: lh_closures.hpp
    slope32), so it agrees with the Float64 rule to Float32 rounding: 1e-6 relative."""
    import torch
    case = pc.make_case(name, ncols=None if name != "c1_dirichlet_f64" else 7)
    want_dt = O.stable_dt(case.om, case.vl, case.ti, case.rhoe, 0.5, case.T_aux)
    with pc.GpuModel(case) as g:
        F = g.F
        Y, Ya = g.prognostic_and_aux()
        dA, dB = g.state(0), g.state(0)
        g.rhs(Y, Ya, dA)
        tdt = torch.full((1,), -1.0, device="cuda",
                         dtype=torch.float64 if case.dtype == np.float64 else torch.float32)
        F.check(g.L.lh_rhs_stable_dt(g.ctx, 0.0, Y, Ya, dB, 0.5, tdt.data_ptr()), g.ctx)
        F.check(g.L.lh_synchronize(g.ctx), g.ctx)
        a, b = g.tendencies(dA), g.tendencies(dB)
        for k in a:
            assert np.array_equal(a[k], b[k]), (name, k)
        sep = C.c_double()
        F.check(g.L.lh_stable_dt(g.ctx, Y, Ya, 0.5, C.byref(sep)), g.ctx)
    got = float(tdt.item())
    rel = 1e-6 if case.dtype == np.float64 else 2e-4
    assert abs(got - want_dt) <= rel * want_dt, (name, got, want_dt)
    assert abs(got - sep.value) <= rel * want_dt, (name, got, sep.value)


# ------------------------------------------------------ the Float32 theta(z,t) bound
def _as_f64_problem(case32):
    """The same problem in Float64: the Float32-rounded inputs AND parameters (FT(x) of the model
    constructors), so that what differs from the Float32 runs is the arithmetic alone."""
    import dataclasses
    r = lambda x: float(np.float32(x))
    rd = lambda obj: dataclasses.replace(obj, **{f.name: r(getattr(obj, f.name))
                                                  for f in dataclasses.fields(obj)
                                                  if isinstance(getattr(obj, f.name), float)})
    om = case32.om
    om64 = dataclasses.replace(om, earth=rd(om.earth), soil=rd(om.soil), vg=rd(om.vg), cf=rd(om.cf),
                               zmin=r(om.zmin), zmax=r(om.zmax),
                               bc={k: (kind, r(v)) for k, (kind, v) in om.bc.items()})
    up = lambda a: None if a is None else a.astype(np.float64)
    return dataclasses.replace(case32, om=om64, dtype=np.float64, vl=up(case32.vl), ti=up(case32.ti),
                               rhoe=up(case32.rhoe), T_aux=up(case32.T_aux))


F32_BOUND_RESULTS = {}


@pytest.mark.parametrize("name,nsteps", [("c3_coupled_f32", 30), ("c2_richards_f32", 40),
                                         ("mixed_smooth_f32", 20)])
def test_f32_theta_zt_is_as_close_to_f64_as_the_reference_arithmetic_is(name, nsteps):
    """north_star: theta(z,t) within 1e-6 relative of the reference.  In Float32 the reference's
    OWN arithmetic is not that close to the exact trajectory: this test measures three states
    after the same steps -- HIP Float32 (H32), the oracle in Float32 (O32 = the reference's
    arithmetic, pow for pow) and the oracle in Float64 on the Float32-rounded problem (O64) -- and
    asserts that the HIP path is no farther from O64 than 1.5 x the reference arithmetic is:
        max|H32 - O64| <= 1.5 max|O32 - O64|      per prognostic field.
    The distances are printed (and recorded in BASELINE.md): they are the Float32 tolerance
    statement for BASELINE config C3."""
    case32 = pc.make_case(name)
    case64 = _as_f64_problem(case32)
    dt = float(np.float32(stable_dt(case32)))
    H32 = gpu_steps(case32, dt, nsteps)
    O32 = cpu_steps(case32, dt, nsteps)
    O64 = cpu_steps(case64, dt, nsteps)
    rec = {}
    for k in O64:
        if k == "ti":
            assert np.array_equal(H32[k], case32.ti)            # theta_i never moves
            continue
        w = O64[k]
        scale = np.max(np.abs(w))
        dH = float(np.max(np.abs(H32[k].astype(np.float64) - w)) / scale)
        dO = float(np.max(np.abs(O32[k].astype(np.float64) - w)) / scale)
        dHO = float(np.max(np.abs(H32[k].astype(np.float64) - O32[k].astype(np.float64))) / scale)
        rec[k] = {"H32_vs_O64": dH, "O32_vs_O64": dO, "H32_vs_O32": dHO}
        assert dO > 0 and dH <= 1.5 * dO, (name, k, rec[k])
        # and the state moved: the comparison is not vacuous
        ref0 = getattr(case64, "vl" if k == "vl" else "rhoe")
        assert np.max(np.abs(w - ref0)) > 100 * dO * scale, (name, k)
    F32_BOUND_RESULTS[name] = rec
    print(f"\nF32_BOUND {name} steps={nsteps} dt={dt:.6g} " + " ".join(
        f"{k}: H32-O64={v['H32_vs_O64']:.3g} O32-O64={v['O32_vs_O64']:.3g} H32-O32={v['H32_vs_O32']:.3g}"
        for k, v in rec.items()))
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "f32_bound.json"), "w") as fh:
            json.dump(F32_BOUND_RESULTS, fh, indent=1)


def test_f32_drift_over_2000_steps_stays_with_the_reference_arithmetic():
    """The Float32 statement of the previous test over a LONG horizon: 2000 SSPRK33 steps of the
    coupled configuration C3 (6000 evaluations of f), checked every 200 steps --
        max|H32 - O64| <= 1.5 max|O32 - O64|      per prognostic field at every checkpoint,
    i.e. the HIP Float32 path drifts from the Float64 trajectory no faster than the reference's own
    Float32 arithmetic does.  Both growth curves are recorded (gpurun_out/f32_drift.json ->
    profiles/round3_f32_drift.json)."""
    import json
    import os
    case32 = pc.make_case("c3_coupled_f32", ncols=192)
    case64 = _as_f64_problem(case32)
    dt = float(np.float32(stable_dt(case32)))
    chunk, nchunks = 200, 10
    curves = {"dt": dt, "steps": [], "fields": {}}
    with pc.GpuModel(case32) as g:
        F = g.F
        Y, Ya = g.prognostic_and_aux()
        o32 = {k: getattr(case32, k).copy() for k in ("vl", "ti", "rhoe")}
        o64 = {k: getattr(case64, k).copy() for k in ("vl", "ti", "rhoe")}
        for c in range(nchunks):
            F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, dt, chunk, None), g.ctx)
            O.ssprk33(case32.om, dt, chunk, vl=o32["vl"], ti=o32["ti"], rhoe=o32["rhoe"], nthreads=8)
            O.ssprk33(case64.om, dt, chunk, vl=o64["vl"], ti=o64["ti"], rhoe=o64["rhoe"], nthreads=8)
            H = {"vl": g.download(Y, F.LH_VAR_VARTHETA_L), "rhoe": g.download(Y, F.LH_VAR_RHOE_INT)}
            curves["steps"].append((c + 1) * chunk)
            for k in ("vl", "rhoe"):
                w = o64[k]
                scale = np.max(np.abs(w))
                dH = float(np.max(np.abs(H[k].astype(np.float64) - w)) / scale)
                dO = float(np.max(np.abs(o32[k].astype(np.float64) - w)) / scale)
                curves["fields"].setdefault(k, {"H32_vs_O64": [], "O32_vs_O64": []})
                curves["fields"][k]["H32_vs_O64"].append(dH)
                curves["fields"][k]["O32_vs_O64"].append(dO)
                assert dO > 0 and dH <= 1.5 * dO, (k, (c + 1) * chunk, dH, dO)
        assert g.status() == 0
        assert np.array_equal(g.download(Y, F.LH_VAR_THETA_I), case32.ti)
    # the state moved far more than the distances compared
    assert np.max(np.abs(o64["vl"] - case64.vl)) > 1e3 * curves["fields"]["vl"]["O32_vs_O64"][-1] * np.max(np.abs(o64["vl"]))
    print("\nF32_DRIFT " + json.dumps(curves))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "f32_drift.json"), "w") as fh:
            json.dump(curves, fh, indent=1)


# --------------------------------- property pins for the BC branches no reference vector covers
def _rhs_vl(case):
    return pc.run_gpu_rhs(case)["vl"]


def test_free_drainage_uniform_column_has_a_still_bottom_cell():
    """boundary_conditions.jl:328-356: FreeDrainage sets the bottom flux to -K(centre state).
    In a column of uniform vartheta_l every interior face carries -K as well (unit head gradient),
    so the bottom cell's tendency vanishes (to the rounding of (psi + z_1) - (psi + z_0) = dz),
    whatever the top does; with a zero-flux top the top cell drains at exactly K/dz.  BASELINE C5's
    bottom boundary, pinned by property."""
    n, N = 64, 130
    c = np.arange(N)
    vg_n = 1.4 + 2.6 * pc.uhash(c, 2, n)
    Ksat = 10.0 ** (-7.0 + 3.0 * pc.uhash(c, 4, n))
    nu = 0.3 + 0.25 * pc.uhash(c, 6, n)
    om = M.CaseModel(M.MODEL_RICHARDS, n, -1.28, 0.0,
                     bc={(M.FACE_TOP, M.COMP_HYDROLOGY): (M.BC_FLUX, 0.0),
                         (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_FREE_DRAINAGE, 0.0)},
                     percol=dict(vg_n=vg_n, vg_Ksat=Ksat, nu=nu))
    vl = np.repeat((0.6 * nu)[:, None], n, axis=1)
    case = pc.Case("fd_uniform", om, np.float64, N, vl=vl, ti=np.zeros((N, n)))
    d = _rhs_vl(case)
    diag = pc.run_gpu_diagnostics(case)
    K = diag["K"][:, 0]
    dz = 1.28 / n
    eps = np.finfo(np.float64).eps
    assert np.all(K > 0)
    # interior cells and the bottom cell: |d| <= rounding of the head difference
    bound = 8 * eps * K * (np.abs(diag["psi"][:, 0]) + 1.28) / dz / dz
    assert np.all(np.abs(d[:, :-1]) <= bound[:, None])
    # the top cell: -(0 - (-K))/dz
    assert np.allclose(d[:, -1], -K / dz, rtol=1e-12, atol=0)


@pytest.mark.parametrize("consistent", [True, False])
def test_dirichlet_hydrostatic_column(consistent):
    """Dirichlet-as-flux, boundary_conditions.jl:371-401 (BASELINE C1's boundaries).  A hydrostatic
    column whose two Dirichlet face values are the hydrostatic face values has zero flux through
    every face when the bottom sign is the consistent one (lh_set_bottom_sign_consistent(1)):
    |d vartheta_l| <= rounding everywhere.  As the reference WRITES the bottom face
    (:395-398: the whole expression is negated, gravity term included) the bottom flux is
    -K_f (psi_f - psi_c + dz/2)/(dz/2) negated = +2 K_f for a hydrostatic pair, so the bottom
    cell alone gets -(0 - 2 K_f)/dz = +2 K_f/dz: the closed form of the quirk, on the device."""
    import __graft_entry__ as ge
    P = ge.load_package().parameterizations
    lh = ge.load_package()
    n = 64
    zmin, zmax, zi = -1.28, 0.0, -2.0                     # water table below the column: all unsaturated
    hm = lh.vanGenuchten(np.float64)                      # loam defaults
    nu, S_s = 0.43, 1e-3
    zc, zf = pc.grid_np(zmin, zmax, n)
    vl = P.hydrostatic_profile(hm, zc, zi, nu, S_s)[None, :]
    top = float(P.hydrostatic_profile(hm, np.float64(zmax), zi, nu, S_s))
    bot = float(P.hydrostatic_profile(hm, np.float64(zmin), zi, nu, S_s))
    om = M.CaseModel(M.MODEL_RICHARDS, n, zmin, zmax,
                     bc={(M.FACE_TOP, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, top),
                         (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, bot)},
                     consistent_bottom_sign=consistent)
    case = pc.Case("dirichlet_hydrostatic", om, np.float64, 1, vl=vl, ti=np.zeros((1, n)))
    d = _rhs_vl(case)[0]
    diag = pc.run_gpu_diagnostics(case)
    K, psi = diag["K"][0], diag["psi"][0]
    dz = (zmax - zmin) / n
    eps = np.finfo(np.float64).eps
    # hydrostatic: psi + z is the water-table height at every centre
    assert np.allclose(psi + zc, zi, rtol=0, atol=1e-12)
    # rounding bound of one face flux: K * (eps-level error of the head difference) / dz;
    # the head values are O(|psi| + |z|)
    face_err = 16 * eps * np.max(K) * (np.max(np.abs(psi)) + abs(zmin)) / (dz / 2)
    if consistent:
        assert np.max(np.abs(d)) <= 2 * face_err / dz
    else:
        K_f = float(P.hydraulic_conductivity(hm, P.effective_saturation(nu, np.float64(bot), 0.0)))
        assert np.max(np.abs(d[1:])) <= 2 * face_err / dz
        assert d[0] == pytest.approx(2.0 * K_f / dz, rel=1e-9)
    # the oracle says the same, cell for cell
    pc.assert_tendencies_close(case, {"vl": d[None, :], "ti": np.zeros((1, n))}, pc.run_oracle_rhs(case), 4.0)
