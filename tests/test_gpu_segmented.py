"""Level-segmented launches (small ensembles: blockIdx.y cuts every column into
segments, rhs_kernel CFG::SEG) must be BITWISE the unsegmented launch: the same
closures of the same cells, the shared faces evaluated twice from the same inputs.

Through the C ABI, for every model / BC / factor combination of the parity cases:
tendency, fused tendency + stable dt, and the fused SSPRK33 stepper, with the
unsegmented launch (LH_TUNE seg=-1) as the reference and segment lengths that
divide the column, do not divide it, are 1, and the automatic choice.
"""
import ctypes as C

import numpy as np
import pytest

import case_model as M
import parity_cases as pc

pytestmark = pytest.mark.gpu
O = pc.O

CASES = ["c1_dirichlet_f64", "c2_richards_f64", "c2_richards_f32", "c3_coupled_f32", "c3_coupled_f64",
         "c5_percol_f64", "heat_dirichlet_f64", "heat_dirichlet_f32", "mixed_factors_f64",
         "mixed_smooth_f32", "richards_viscosity_f64"]
SEGS = [b"seg=8,persist=0", b"seg=5,persist=0", b"seg=1,persist=0", b"seg=0,persist=0"]   # seg=0: the library's own choice
# (persist=0: the fused-stage launches, not the persistent column stepper, do the stepping)


def _fields(g, st, case):
    F, m = g.F, case.om.model
    out = {}
    if m != M.MODEL_HEAT:
        out["vl"] = g.download(st, F.LH_VAR_VARTHETA_L)
        out["ti"] = g.download(st, F.LH_VAR_THETA_I)
    if m != M.MODEL_RICHARDS:
        out["rhoe"] = g.download(st, F.LH_VAR_RHOE_INT)
    return out


def _run(case, tune, nsteps, dt):
    with pc.GpuModel(case) as g:
        F, L = g.F, g.L
        F.check(L.lh_set_tuning(g.ctx, tune), g.ctx)
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        g.rhs(Y, Ya, dY)
        res = {"rhs": _fields(g, dY, case)}
        import torch
        dev = torch.zeros(1, device="cuda", dtype=torch.float64 if case.dtype == np.float64 else torch.float32)
        if case.om.model != M.MODEL_HEAT or True:
            F.check(L.lh_rhs_stable_dt(g.ctx, 0.0, Y, Ya, dY, 0.4, C.c_void_p(dev.data_ptr())), g.ctx)
            F.check(L.lh_synchronize(g.ctx), g.ctx)
            res["rhs4"] = _fields(g, dY, case)
            res["dt"] = {"dt": dev.cpu().numpy().copy()}
        g.status()
        F.check(L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, dt, nsteps, None), g.ctx)
        res["step"] = _fields(g, Y, case)
        return res


@pytest.mark.parametrize("name", CASES)
def test_segmented_launch_is_bitwise_the_unsegmented_one(name):
    case = pc.make_case(name)
    dt = 1e-3 if "mixed_factors" in name else 0.5
    ref = _run(case, b"seg=-1,persist=0", 3, dt)
    for tune in SEGS:
        got = _run(case, tune, 3, dt)
        for part in ref:
            for k in ref[part]:
                np.testing.assert_array_equal(got[part][k], ref[part][k],
                                              err_msg=f"{name} {tune.decode()} {part}.{k}")


def test_single_column_uses_segments_and_matches_oracle():
    """The reference's own shape: ONE column.  The library cuts it into segments by
    itself; the result is the oracle's (and the unsegmented launch's)."""
    case = pc.make_case("c1_dirichlet_f64")
    assert case.ncols == 1
    got = pc.run_gpu_rhs(case)
    want = pc.run_oracle_rhs(case)
    pc.assert_tendencies_close(case, got, want, 4.0)


@pytest.mark.parametrize("name", ["c1_dirichlet_f64", "c3_coupled_f32", "mixed_smooth_f64"])
def test_graph_replayed_steps_equal_plain_launches(name):
    """Small ensembles replay blocks of 16 fused steps as a hipGraph (lh_step_ssprk33 with
    constant boundary values, >= 64 steps): same kernels, same order, same bits -- also
    across a step count that is not a multiple of the block."""
    case = pc.make_case(name)
    res = []
    for tune in (b"graph=0,persist=0", b"graph=1,persist=0"):
        with pc.GpuModel(case) as g:
            g.F.check(g.L.lh_set_tuning(g.ctx, tune), g.ctx)
            Y, Ya = g.prognostic_and_aux()
            g.F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, 0.25, 150, None), g.ctx)
            res.append(_fields(g, Y, case))
            assert g.status() == 0
    for k in res[0]:
        np.testing.assert_array_equal(res[0][k], res[1][k], err_msg=k)


# tall and ragged columns: 65..128 levels step as ONE wavefront with two adjacent cells per lane
# (column_stepper_wave_kernel, CW = 2: odd counts leave the top lane half empty, 97 and 66 leave
# whole lanes empty), more than 128 levels as one thread per cell with workgroup barriers
TALL = ["c4_richards_f64_128", "c2_richards_f64_n97", "c2_richards_f64_n66", "c2_richards_f32_n127",
        "mixed_smooth_f64_n101", "mixed_smooth_f32_n128", "c2_richards_f64_n150", "mixed_smooth_f32_n131",
        "c2_richards_f64_n3", "mixed_smooth_f64_n2"]


# Dirichlet faces whose face-state closures are (not) constants of a call: the wave stepper evaluates
# them once per call or once per stage (face_state_is_static), per-column boundary values included
FACES = ["mixed_smooth_f64_hyddir", "mixed_smooth_f32_endir", "mixed_smooth_f64_pcdir", "c1_dirichlet_f64_pcdir",
         "mixed_smooth_f64_n101_hyddir", "heat_dirichlet_f64_pcdir", "richards_viscosity_f64_pcdir"]


@pytest.mark.parametrize("name", CASES + ["single_cell_f64"] + TALL + FACES)
def test_persistent_column_stepper_is_bitwise_the_fused_stages(name):
    """Ensembles of few columns step inside ONE launch (workgroup = column, thread = cell,
    column_stepper_kernel).  Same closures, same face expressions, same stage updates as
    the fused-stage launches: the stepped state must be bitwise equal -- with constant and
    with per-stage boundary values."""
    case = pc.make_case(name)
    dt = 1e-3 if "mixed_factors" in name else 0.5
    nsteps = 7
    rng = np.random.default_rng(5)
    base = np.zeros((2, 2))
    for (f, comp), (kind, v) in case.om.bc.items():
        base[f, comp] = v
    bcv = np.broadcast_to(base, (nsteps, 3, 2, 2)) * (1.0 + 1e-3 * rng.standard_normal((nsteps, 3, 1, 1)))
    bcv = np.ascontiguousarray(bcv, dtype=np.float64)
    for use_bcv in (False, True):
        res = []
        for tune in (b"persist=0,seg=-1", b"persist=1"):
            with pc.GpuModel(case) as g:
                g.F.check(g.L.lh_set_tuning(g.ctx, tune), g.ctx)
                Y, Ya = g.prognostic_and_aux()
                p = bcv.ctypes.data_as(C.POINTER(C.c_double)) if use_bcv else None
                g.F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, dt, nsteps, p), g.ctx)
                res.append((_fields(g, Y, case), g.status()))
        assert res[0][1] == res[1][1]
        for k in res[0][0]:
            np.testing.assert_array_equal(res[0][0][k], res[1][0][k], err_msg=f"{name} bcv={use_bcv} {k}")


def test_engine_choice_for_large_ensembles():
    """lh_step_engine: large ensembles step in the persistent column stepper from 3 steps per call on;
    with a Dirichlet face only where the face state's closures are constants of the call (constant
    boundary values, nothing else they read moves); the Float64 coupled model with conductivity factors
    stays with the fused stages; LH_TUNE persist= overrides."""
    def engine(name, nsteps, per_stage, tune=b"", ncols=40000):
        case = pc.make_case(name, ncols=ncols)
        with pc.GpuModel(case) as g:
            g.F.check(g.L.lh_set_tuning(g.ctx, tune), g.ctx)
            return g.L.lh_step_engine(g.ctx, nsteps, per_stage)
    F = pc._pkg()._ffi
    ST, FU = F.LH_ENGINE_COLUMN_STEPPER, F.LH_ENGINE_FUSED_STAGES
    assert engine("c2_richards_f64", 30, 0) == ST              # flux boundaries
    assert engine("c2_richards_f64", 2, 0) == FU               # too few steps to pay the tile I/O
    assert engine("c2_richards_f64", 2, 0, ncols=100) == ST    # small ensembles: always
    assert engine("c1_dirichlet_f64", 30, 0) == ST             # Dirichlet faces, constant values
    assert engine("c1_dirichlet_f64", 30, 1) == FU             # ... per-stage values: closures every stage
    assert engine("richards_viscosity_f64", 30, 0) == ST       # T prescribed: static
    assert engine("mixed_smooth_f32", 30, 0) == ST
    assert engine("mixed_smooth_f32_hyddir", 30, 0) == FU      # viscosity sees the moving T of the boundary cell
    assert engine("mixed_smooth_f32_endir", 30, 0) == FU       # kappa of the face state sees the moving vartheta_l
    assert engine("mixed_smooth_f64", 30, 0) == FU             # Float64 closures bound both engines
    assert engine("heat_dirichlet_f64", 30, 0) == ST           # heat only: the prescribed water fields are static
    assert engine("c5_percol_f64", 30, 0) == ST                # per-column parameters and flux values
    assert engine("c2_richards_f64", 30, 0, tune=b"persist=0") == FU
    assert engine("mixed_smooth_f64", 30, 1, tune=b"persist=2") == ST


def test_column_height_sweep_is_bitwise_in_both_engines():
    """Every column height class of the steppers -- 1..5 cells, around 32 / 64 / 96 / 128 levels (one
    wave with one or two cells per lane, ragged top lanes, whole lanes empty), beyond 128 (one thread
    per cell) -- with flux, Dirichlet and per-column Dirichlet boundaries: five steps are the same bits
    as the fused stages."""
    bad = []
    for fam in ("mixed_smooth_f64", "mixed_smooth_f32", "c2_richards_f64", "mixed_smooth_f64_pcdir"):
        for n in (1, 2, 3, 4, 5, 17, 31, 32, 33, 63, 64, 65, 66, 95, 96, 97, 127, 128, 129, 130, 160):
            name = f"{fam[:-6]}_n{n}_pcdir" if fam.endswith("_pcdir") else f"{fam}_n{n}"
            case = pc.make_case(name, ncols=37)
            res = []
            for tune in (b"persist=0,seg=-1", b"persist=2"):
                with pc.GpuModel(case) as g:
                    g.F.check(g.L.lh_set_tuning(g.ctx, tune), g.ctx)
                    Y, Ya = g.prognostic_and_aux()
                    g.F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, 0.5, 5, None), g.ctx)
                    res.append((_fields(g, Y, case), g.status()))
            same = res[0][1] == res[1][1] and all(np.array_equal(res[0][0][k], res[1][0][k], equal_nan=True)
                                                  for k in res[0][0])
            if not same:
                bad.append(name)
    assert not bad, bad
