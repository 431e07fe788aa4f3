"""lh_tune_placement: moving the written state to another set of plane slots
changes where the launch writes, never what it computes.

Checked through the C ABI on the device: the tendency after tuning is bitwise the
tendency before; tuning the stepper's stage state leaves Y bitwise untouched and
the stepped result bitwise equal to an untuned context's; the status word is
restored; invalid arguments are rejected; the host mirror's make_rhs / Simulation
call it on their own for large ensembles only.
"""
import ctypes as C

import numpy as np
import pytest

import case_model as M
import parity_cases as pc

pytestmark = pytest.mark.gpu
O = pc.O


def _fields(g, st, case):
    F, m = g.F, case.om.model
    out = {}
    if m != M.MODEL_HEAT:
        out["vl"] = g.download(st, F.LH_VAR_VARTHETA_L)
        out["ti"] = g.download(st, F.LH_VAR_THETA_I)
    if m != M.MODEL_RICHARDS:
        out["rhoe"] = g.download(st, F.LH_VAR_RHOE_INT)
    return out


@pytest.mark.parametrize("flags", [0, 1])       # 1 = LH_PLACE_MOVE_INPUT
@pytest.mark.parametrize("name,ncols", [("c2_richards_f64", 70_000), ("c3_coupled_f32", 50_000),
                                        ("c5_percol_f64", 3_000)])
def test_tendency_identical_after_tuning(name, ncols, flags):
    case = pc.make_case(name, ncols)
    with pc.GpuModel(case) as g:
        F, L = g.F, g.L
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        g.rhs(Y, Ya, dY)
        before = _fields(g, dY, case)
        y_before = _fields(g, Y, case)
        b, a = C.c_float(), C.c_float()
        F.check(L.lh_tune_placement(g.ctx, Y, Ya, dY, 5, flags, C.byref(b), C.byref(a)), g.ctx)
        assert 0 < a.value <= b.value          # never reports a slower placement than it started from
        g.rhs(Y, Ya, dY)
        after = _fields(g, dY, case)
        for k in before:
            np.testing.assert_array_equal(before[k], after[k], err_msg=k)
        for k, v in _fields(g, Y, case).items():
            np.testing.assert_array_equal(v, y_before[k], err_msg=k)
        assert g.status() == 0
        # the handle still answers for its (possibly new) planes
        p, ls, cs = C.c_void_p(), C.c_int64(), C.c_int64()
        var = F.LH_VAR_VARTHETA_L
        F.check(L.lh_state_device_ptr(g.ctx, dY, var, C.byref(p), C.byref(ls), C.byref(cs)), g.ctx)
        assert p.value and cs.value == 1 and ls.value >= ncols
        # a second round and a fresh state keep working (slots were returned to the arenas)
        F.check(L.lh_tune_placement(g.ctx, Y, Ya, dY, 3, 0, None, None), g.ctx)
        dY2 = g.state(0)
        g.rhs(Y, Ya, dY2)
        for k, v in _fields(g, dY2, case).items():
            np.testing.assert_array_equal(v, before[k], err_msg=k)


@pytest.mark.parametrize("flags", [0, 1])
@pytest.mark.parametrize("name,ncols,dt", [("c2_richards_f64", 40_000, 20.0), ("c3_coupled_f32", 30_000, 5.0)])
def test_stepper_stage_state_tuning(name, ncols, dt, flags):
    case = pc.make_case(name, ncols)
    res = []
    for tune in (False, True):
        with pc.GpuModel(case) as g:
            F, L = g.F, g.L
            F.check(L.lh_set_tuning(g.ctx, b"persist=0"), g.ctx)   # the fused-stage stepper owns a stage state
            Y, Ya = g.prognostic_and_aux()
            y0 = _fields(g, Y, case)
            if tune:
                F.check(L.lh_tune_placement(g.ctx, Y, Ya, None, 4, flags, None, None), g.ctx)
                for k, v in _fields(g, Y, case).items():     # dt = 0 trial stages: Y untouched
                    np.testing.assert_array_equal(v, y0[k], err_msg=k)
            F.check(L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, dt, 5, None), g.ctx)
            res.append(_fields(g, Y, case))
            assert g.status() == 0
    for k in res[0]:
        np.testing.assert_array_equal(res[0][k], res[1][k], err_msg=k)
        assert not np.array_equal(res[0][k], y0[k]) or k == "ti"   # the steps did something


def test_status_word_survives_tuning():
    case = pc.make_case("c2_richards_f64", 2_000)
    case.vl = case.vl.copy()
    case.vl[7, 3] = np.nan
    with pc.GpuModel(case) as g:
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        g.rhs(Y, Ya, dY)                              # raises the non-finite flag
        g.F.check(g.L.lh_tune_placement(g.ctx, Y, Ya, dY, 3, 0, None, None), g.ctx)
        assert g.status() == 1                        # still there, exactly once
        assert g.status() == 0
    case = pc.make_case("c2_richards_f64", 2_000)
    with pc.GpuModel(case) as g:
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        g.F.check(g.L.lh_tune_placement(g.ctx, Y, Ya, dY, 3, 0, None, None), g.ctx)
        assert g.status() == 0


def test_rejects_bad_arguments():
    case = pc.make_case("c3_coupled_f32", 500)
    with pc.GpuModel(case) as g:
        F, L = g.F, g.L
        Y, Ya = g.prognostic_and_aux()
        assert L.lh_tune_placement(g.ctx, Y, Ya, Y, 0, 0, None, None) == F.LH_EINVAL
        assert b"must not be Y" in L.lh_last_error(g.ctx)
        assert L.lh_tune_placement(g.ctx, None, Ya, None, 0, 0, None, None) != 0
        wrong = g.state(0b0001)                       # a one-plane state is not a coupled dY
        assert L.lh_tune_placement(g.ctx, Y, Ya, wrong, 0, 0, None, None) != 0
        assert L.lh_tune_placement(None, Y, Ya, None, 0, 0, None, None) == F.LH_EINVAL
        assert L.lh_tune_placement(g.ctx, Y, Ya, None, 0, 0x80, None, None) == F.LH_EINVAL


def test_host_mirror_tunes_on_request_and_large_ensembles_only(monkeypatch):
    lh = pc._pkg()
    soil = lh.soil
    calls = []
    real = lh._ffi.lib().lh_tune_placement

    class Spy:
        def __call__(self, *a):
            calls.append(a)
            return real(*a)
    lib = lh._ffi.lib()
    monkeypatch.setattr(lib, "lh_tune_placement", Spy(), raising=False)

    def model(ncolumns):
        domain = lh.Column(lh.Float64, zlim=(-1.0, 0.0), nelements=16, ncolumns=ncolumns)
        bc = lh.SoilColumnBC(top=lh.SoilComponentBC(hydrology=lh.VerticalFlux(0.0)),
                             bottom=lh.SoilComponentBC(hydrology=lh.VerticalFlux(0.0)))
        return lh.SoilModel(lh.Float64, domain=domain, energy_model=lh.PrescribedTemperatureModel(),
                            hydrology_model=lh.SoilHydrologyModel(lh.Float64), boundary_conditions=bc,
                            soil_param_set=lh.SoilParams(lh.Float64), earth_param_set=lh.EarthParameterSet())

    def ic(z, m):
        return {"ϑ_l": 0.2 + 0.0 * z, "θ_i": 0.0 * z}
    small = model(64)
    Y, Ya = lh.initialize_states(small, ic, 0.0)
    dY = Y.similar()
    monkeypatch.delenv("LH_PLACEMENT_TUNE", raising=False)
    monkeypatch.setattr(soil, "PLACEMENT_TUNE_MIN_PLANE_BYTES", 1)
    rhs = lh.make_rhs(small)
    rhs(dY, Y, Ya, 0.0)
    assert calls == []                                # implicit tuning is opt-in (LH_PLACEMENT_TUNE=1)
    monkeypatch.setenv("LH_PLACEMENT_TUNE", "1")
    monkeypatch.setattr(soil, "PLACEMENT_TUNE_MIN_PLANE_BYTES", 32 << 20)
    rhs = lh.make_rhs(small)
    rhs(dY, Y, Ya, 0.0)
    assert calls == []                                # ... and for large ensembles only
    monkeypatch.setattr(soil, "PLACEMENT_TUNE_MIN_PLANE_BYTES", 1)
    rhs = lh.make_rhs(small)
    rhs(dY, Y, Ya, 0.0)
    first = dY.get("ϑ_l").copy()
    rhs(dY, Y, Ya, 0.0)
    assert len(calls) == 1                            # once per (Y, dY) pair
    assert calls[0][5] == 0                           # never LH_PLACE_MOVE_INPUT by itself
    np.testing.assert_array_equal(first, dY.get("ϑ_l"))
    sim = lh.Simulation(small, lh.SSPRK33(), Y_init=Y, dt=1.0, tspan=(0.0, 3.0), Ya_init=Ya)
    lh.run(sim)
    assert len(calls) == 2
    monkeypatch.setenv("LH_PLACEMENT_TUNE", "0")
    rhs = lh.make_rhs(small)
    rhs(dY, Y, Ya, 0.0)
    assert len(calls) == 2
    b, a = lh.tune_placement(small, Y, Ya, dY, max_candidates=3)
    assert 0 < a <= b
