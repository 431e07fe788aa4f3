"""lh_boundary_fluxes: boundary_fluxes(X, bc::SoilComponentBC, face, model, cs, t)
(boundary_conditions.jl:470-489) for every column -- the two SetValue fluxes of the tendency launch --
against the oracle's restatement of the same lines, for every boundary-condition kind incl. the
bottom-sign quirk of the hydrology Dirichlet flux (:395-398) and its opt-out, and against the
tendency itself: dvartheta_l of the boundary cell is the difference of the interior face flux and
the returned boundary flux."""
import ctypes as C
import dataclasses

import numpy as np
import pytest

import case_model as M
import parity_cases as pc

pytestmark = pytest.mark.gpu
O = pc.O

CASES = ["c1_dirichlet_f64", "c2_richards_f64", "c3_coupled_f32", "c3_coupled_f64", "c5_percol_f64",
         "heat_dirichlet_f64", "heat_dirichlet_f32", "mixed_factors_f64", "mixed_smooth_f32",
         "richards_viscosity_f64", "single_cell_f64", "mixed_smooth_f64_n101"]


def _gpu_fluxes(g, Y, Ya, face):
    n = g.case.ncols
    fe, fw = np.empty(n), np.empty(n)
    g.F.check(g.L.lh_boundary_fluxes(g.ctx, Y, Ya, 0.0, face, fe.ctypes.data_as(C.POINTER(C.c_double)),
                                     fw.ctypes.data_as(C.POINTER(C.c_double))), g.ctx)
    return fe, fw


def _flux_tolerances(case, face, Cw):
    """Absolute tolerance per column of the two boundary fluxes: Cw eps propagated through the
    closures of the boundary cell and of the Dirichlet face state with the tolerance model of the
    tendency tests (parity_cases.closure_tolerances)."""
    om = case.om
    n = om.nlev
    ib = 0 if face == M.FACE_BOTTOM else n - 1
    dzb = (om.zmax - om.zmin) / n / 2
    eps = float(np.finfo(case.dtype).eps) * Cw
    f8 = lambda a: np.asarray(a, np.float64)
    diag_c = O.diagnostics(om, case.vl, case.ti, case.rhoe, case.T_aux)
    tol_c = pc.closure_tolerances(case, diag_c, Cw)
    ke = om.bc.get((face, M.COMP_ENERGY), (M.BC_NONE, 0.0))
    kh = om.bc.get((face, M.COMP_HYDROLOGY), (M.BC_NONE, 0.0))
    # the Dirichlet face state: the boundary cell with vartheta_l (and T) replaced by the face value
    vl_f = case.vl.copy()
    if kh[0] == M.BC_DIRICHLET and om.model != M.MODEL_HEAT:
        vl_f[:, ib] = kh[1]
    case_f = dataclasses.replace(case, vl=vl_f)
    diag_f = O.diagnostics(om, vl_f, case.ti, case.rhoe, case.T_aux)
    tol_f = pc.closure_tolerances(case_f, diag_f, Cw)
    N = case.ncols
    te, tw = np.zeros(N), np.zeros(N)
    if kh[0] == M.BC_FREE_DRAINAGE:
        tw = f8(tol_c["K"][:, ib])
    elif kh[0] == M.BC_DIRICHLET and om.model != M.MODEL_HEAT:
        Kf, pf, pc_ = f8(diag_f["K"][:, ib]), f8(diag_f["psi"][:, ib]), f8(diag_c["psi"][:, ib])
        g = np.abs(pf - pc_) + dzb
        tw = (f8(tol_f["K"][:, ib]) * g + Kf * (f8(tol_f["psi"][:, ib]) + f8(tol_c["psi"][:, ib]))) / dzb + eps * Kf * g / dzb
    if ke[0] == M.BC_DIRICHLET and om.model != M.MODEL_RICHARDS:
        kf, Tc = f8(diag_f["kappa"][:, ib]), f8(diag_c["T"][:, ib])
        gT = np.abs(ke[1] - Tc)
        te = (f8(tol_f["kappa"][:, ib]) * gT + kf * f8(tol_c["T"][:, ib])) / dzb + eps * kf * gT / dzb
    return te, tw


@pytest.mark.parametrize("name", CASES)
def test_boundary_fluxes_match_the_oracle(name):
    case = pc.make_case(name, ncols=None if name != "c1_dirichlet_f64" else 7)
    om = case.om
    Cw = 4.0 if case.dtype == np.float64 else 16.0
    with pc.GpuModel(case) as g:
        Y, Ya = g.prognostic_and_aux()
        for face in (M.FACE_BOTTOM, M.FACE_TOP):
            fe, fw = _gpu_fluxes(g, Y, Ya, face)
            we, ww = O.boundary_fluxes(om, face, case.vl, case.ti, case.rhoe, case.T_aux)
            te, tw = _flux_tolerances(case, face, Cw)
            for got, want, tol, comp in ((fe, we, te, "energy"), (fw, ww, tw, "water")):
                want = np.asarray(want, np.float64)
                nan = np.isnan(want)      # NoBC components: `nothing` in the reference
                assert np.array_equal(np.isnan(got), nan), (name, face, comp)
                if nan.all():
                    continue
                kind = om.bc.get((face, M.COMP_ENERGY if comp == "energy" else M.COMP_HYDROLOGY), (M.BC_NONE, 0))[0]
                d = np.abs(got - want)
                if kind == M.BC_FLUX:        # VerticalFlux: the value itself (per-column values rounded to FT once)
                    assert np.all(d == 0), (name, face, comp)
                    continue
                assert np.all(d <= tol), (name, face, comp, float((d / np.maximum(tol, 1e-300)).max()))
                # and a plain statistic next to the model: most columns agree to a few ulp of the flux
                rel = d / np.maximum(np.abs(want), 1e-300)
                share = float(np.mean(rel <= 32 * np.finfo(case.dtype).eps))
                assert share >= 0.75, (name, face, comp, share)
        assert g.status() == 0


def test_bottom_dirichlet_sign_quirk_and_its_opt_out():
    """boundary_conditions.jl:395-398 as written gives +K_f (psi_f - psi_c + dz)/dz at the bottom;
    the consistent form is K_f (psi_f - psi_c - dz)/dz: they differ by exactly 2 K_f."""
    case = pc.make_case("c1_dirichlet_f64", ncols=5)
    diag = O.diagnostics(case.om, case.vl, case.ti, case.rhoe, case.T_aux)
    res = {}
    for flag in (0, 1):
        om = dataclasses.replace(case.om, consistent_bottom_sign=flag)
        c2 = dataclasses.replace(case, om=om)
        with pc.GpuModel(c2) as g:
            Y, Ya = g.prognostic_and_aux()
            res[flag] = _gpu_fluxes(g, Y, Ya, M.FACE_BOTTOM)[1]
            want = O.boundary_fluxes(om, M.FACE_BOTTOM, case.vl, case.ti)[1]
            assert np.max(np.abs(res[flag] - want)) <= 64 * np.finfo(np.float64).eps * np.max(np.abs(want))
    # K at the Dirichlet face state vartheta_l = 0.20 == the initial state: K_f = K of the bottom cell
    Kf = diag["K"][:, 0]
    assert np.allclose(res[0] - res[1], 2.0 * Kf, rtol=1e-12, atol=0)


@pytest.mark.parametrize("name", ["c5_percol_f64", "c3_coupled_f64", "mixed_smooth_f64", "c1_dirichlet_f64"])
def test_the_boundary_cell_tendency_is_built_from_the_returned_flux(name):
    """dvartheta_l[0] = -(F_1 - F_0)/dz and dvartheta_l[n-1] = -(F_n - F_{n-1})/dz with F_0, F_n the
    returned boundary fluxes and F_1, F_{n-1} from the diagnostics K, psi (and the heat analogue)."""
    case = pc.make_case(name, ncols=None if name != "c1_dirichlet_f64" else 7)
    om = case.om
    n = om.nlev
    dz = (om.zmax - om.zmin) / n
    zc, _ = pc.grid_np(om.zmin, om.zmax, n)
    with pc.GpuModel(case) as g:
        F = g.F
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        g.rhs(Y, Ya, dY)
        tend = g.tendencies(dY)
        fb = _gpu_fluxes(g, Y, Ya, M.FACE_BOTTOM)
        ft = _gpu_fluxes(g, Y, Ya, M.FACE_TOP)
    diag = pc.run_gpu_diagnostics(case)
    K, psi = diag["K"].astype(np.float64), diag["psi"].astype(np.float64)
    h = psi + zc[None, :]
    F1 = -0.5 * (K[:, 0] + K[:, 1]) * (h[:, 1] - h[:, 0]) / dz
    Fm = -0.5 * (K[:, -2] + K[:, -1]) * (h[:, -1] - h[:, -2]) / dz
    d0 = -(F1 - fb[1]) / dz
    dn = -(ft[1] - Fm) / dz
    scale = (np.abs(F1) + np.abs(fb[1]) + np.abs(Fm) + np.abs(ft[1])).max() / dz
    tol = 64 * float(np.finfo(case.dtype).eps) * scale
    assert np.max(np.abs(d0 - tend["vl"][:, 0])) <= tol
    assert np.max(np.abs(dn - tend["vl"][:, -1])) <= tol
