"""Randomised parity over wide state and parameter ranges (tools/fuzz_closures.py): ice from none
to 98 % of the pore space, liquid from 1e-9 of it to 15 % oversaturated and below theta_r, per-column
van Genuchten n from 1.15 (clay-like) to 6, alpha, Ksat over four decades, porosity, residual water,
both conductivity factors on and off, Richards and coupled, Float64 and Float32.  Every closure value
(K, psi, kappa, T of lh_diagnostics) and every tendency must sit inside the tolerance model of
tests/parity_cases.py against the oracle.  What is NOT compared is stated where it is masked: cells
whose K is rounding noise in the reference's own formula, and the reference's Float32 overflow of
S^(-1/m)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_randomised_closures_and_tendencies_stay_inside_the_tolerance_model():
    import fuzz_closures
    worst, failed = fuzz_closures.run(nseeds=2, verbose=False)
    assert failed == 0 and worst <= 1.0, (worst, failed)


def test_fused_step_bound_survives_values_beyond_the_float32_range():
    """The fused launch accumulates its step bound in Float32.  A bone-dry clay-like column
    (n = 1.15) has |psi| ~ 1e60 next to a K that underflows to 0 in Float32: 0 x Inf must not turn
    into a NaN or a zero step -- the bound is the Float64 sweep's (lh_stable_dt) to Float32 rounding.
    (Where the rule itself leaves the Float32 range -- a wet cell next to such a dry one gives
    D = K dpsi/dvl ~ 1e54 -- both routes return a step of practically zero, 0 and 1e-66: not compared.)"""
    import ctypes as C

    import numpy as np
    import torch

    import case_model as M
    import parity_cases as pc
    n, N = 32, 200
    vg_n = np.full(N, 2.0)
    vg_n[17] = 1.15
    om = M.CaseModel(M.MODEL_RICHARDS, n, -1.6, 0.0, bc=pc._flux_bcs(hydrology=0.0),
                     percol=dict(vg_n=vg_n))
    nu = om.soil.nu
    c = np.arange(N)[:, None]
    vl = nu * (0.3 + 0.5 * pc.uhash(c, np.arange(n)[None, :], 1000))
    vl[17, :] = 1e-9 * nu                  # the clay-like column: bone dry throughout
    case = pc.Case("dry_clay", om, np.float64, N, vl=vl, ti=np.zeros((N, n)))
    psi = pc.O.diagnostics(om, case.vl, case.ti)["psi"]
    assert np.abs(psi[17, 0]) > 1e45       # beyond the Float32 range
    with pc.GpuModel(case) as g:
        F = g.F
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        tdt = torch.full((1,), -1.0, device="cuda", dtype=torch.float64)
        F.check(g.L.lh_rhs_stable_dt(g.ctx, 0.0, Y, Ya, dY, 0.5, tdt.data_ptr()), g.ctx)
        sep = C.c_double()
        F.check(g.L.lh_stable_dt(g.ctx, Y, Ya, 0.5, C.byref(sep)), g.ctx)
    got = float(tdt.item())
    assert sep.value > 1e-3                # a sane bound, set by the ordinary columns
    assert np.isfinite(got) and got > 0 and abs(got - sep.value) <= 1e-6 * sep.value, (got, sep.value)


@pytest.mark.parametrize("dtype_name", ["f64", "f32"])
def test_randomised_states_step_bitwise_equal_in_both_engines(dtype_name):
    """The persistent column stepper and the fused-stage launches are the same arithmetic: on the random
    ensembles above (ice, oversaturated and bone-dry cells, liquid below theta_r -- some of whose
    tendencies are NaN in the reference, too --, per-column parameters, factors on and off, a Dirichlet
    top face) three SSPRK33 steps give the same bits, NaNs in the same cells, and the same status."""
    import numpy as np

    import case_model as M
    import fuzz_closures
    import parity_cases as pc
    dtype = np.float64 if dtype_name == "f64" else np.float32
    for model in (M.MODEL_COUPLED, M.MODEL_RICHARDS):
        for factors in (False, True):
            for percol in (False, True):
                case = fuzz_closures.fuzz_case(9007, dtype, factors, percol, model)
                res = []
                for tune in (b"persist=0,seg=-1", b"persist=2"):
                    with pc.GpuModel(case) as g:
                        g.F.check(g.L.lh_set_tuning(g.ctx, tune), g.ctx)
                        Y, Ya = g.prognostic_and_aux()
                        g.F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, 1e-4, 3, None), g.ctx)
                        out = {"vl": g.download(Y, g.F.LH_VAR_VARTHETA_L)}
                        if model == M.MODEL_COUPLED:
                            out["rhoe"] = g.download(Y, g.F.LH_VAR_RHOE_INT)
                        res.append((out, g.status()))
                assert res[0][1] == res[1][1], (model, factors, percol)
                for k in res[0][0]:
                    np.testing.assert_array_equal(res[0][0][k], res[1][0][k], err_msg=f"{dtype_name} {model} {factors} {percol} {k}")
