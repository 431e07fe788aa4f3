"""Shared parity machinery: synthetic cases, a runner for the HIP
path THROUGH THE C ABI, a runner for the CPU oracle, and the tolerance model.

Test infrastructure, never imported by the product.  The oracle is imported lazily by the
checker legs only (see `O`).

Tolerance model (why not one global rtol): a tendency is a difference of face
fluxes, each a product of a conductivity and a head gradient, so agreement
between two correct implementations is limited by (a) the conditioning of the
van Genuchten closures -- K has 1 - (1 - S^(1/m))^m, psi has S^(-1/m) - 1, both
cancelling -- and (b) the cancellation psi_i - psi_(i-1) + dz.  `tendency_tolerance`
propagates C*eps(FT) through exactly those expressions; C is stated per test.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import sys
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import case_model as M  # noqa: E402  (plain descriptions; imports nothing from oracle/)


class _LazyOracle:
    """`O.rhs`, `O.ssprk33`, ... -- oracle/oracle_py.py, imported on first use only: building
    a case or running the HIP path never loads (or even imports) the oracle; the checker legs
    (run_oracle_rhs, the tolerance model) do."""

    def __getattr__(self, name):
        import oracle_py
        return getattr(oracle_py, name)


O = _LazyOracle()

# Case generation and the C-ABI driver are product-side measurement code (bench.py uses them):
# landhydrology.jl_amd/workloads.py.  Re-exported here under the names the tests have always used.
import __graft_entry__ as _g  # noqa: E402

_w = _g.load_package().workloads
SEED, splitmix64, uhash, Case = _w.SEED, _w.splitmix64, _w.uhash, _w.Case
COUPLED_K_SOLID, COUPLED_KSAT_UNFROZEN, COUPLED_KSAT_FROZEN = (_w.COUPLED_K_SOLID, _w.COUPLED_KSAT_UNFROZEN,
                                                               _w.COUPLED_KSAT_FROZEN)
coupled_soil, grid_np, sigmoid, wetting_front, make_case = (_w.coupled_soil, _w.grid_np, _w.sigmoid,
                                                            _w.wetting_front, _w.make_case)
_flux_bcs = _w._flux_bcs
GpuModel, run_gpu_rhs, run_gpu_diagnostics = _w.GpuModel, _w.run_gpu_rhs, _w.run_gpu_diagnostics


def run_oracle_rhs(case: Case, nthreads: int = 1):
    return O.rhs(case.om, case.vl, case.ti, case.rhoe, case.T_aux, nthreads=nthreads)


# --------------------------------------------------------------- GPU runner

def _pkg():
    import __graft_entry__ as g
    return g.load_package()


def _percol(case: Case, key, scalar):
    v = case.om.percol.get(key)
    return np.full(case.ncols, scalar) if v is None else np.asarray(v, dtype=np.float64)


def closure_tolerances(case: Case, diag, Cw: float):
    """Absolute tolerances for K, psi, T, kappa between two correct
    implementations in precision eps(FT): C*eps propagated through the
    cancelling sub-expressions of the closures."""
    om = case.om
    eps = float(np.finfo(case.dtype).eps) * Cw
    f8 = lambda a: np.asarray(a, dtype=np.float64)
    K, psi, T, kap = f8(diag["K"]), f8(diag["psi"]), f8(diag["T"]), f8(diag["kappa"])
    vl, ti = f8(case.vl), f8(case.ti)
    nu = _percol(case, "nu", om.soil.nu)[:, None]
    thr = _percol(case, "vg_theta_r", om.vg.theta_r)[:, None]
    n = _percol(case, "vg_n", om.vg.n)[:, None]
    Ksat = _percol(case, "vg_Ksat", om.vg.Ksat)[:, None]
    m = 1.0 - 1.0 / n
    tiny = float(np.finfo(case.dtype).eps)
    vls = np.maximum(vl, thr + tiny)
    S = (vls - thr) / (nu - thr)
    Se = (vls - thr) / (nu - ti - thr)
    with np.errstate(all="ignore"):
        t = np.where(S < 1, S ** (1.0 / m), 1.0)
        inner = np.where(S < 1, 1.0 - (1.0 - t) ** m, 1.0)
        condK = 1.0 + 2.0 / np.maximum(inner, 1e-300) + np.abs(np.log(np.maximum(S, 1e-300))) / m
        # the inner cancellation w = 1 - S^(1/m): an eps-rounding of t = S^(1/m) (which the
        # reference's own correctly rounded pow has) moves w by eps t / w relatively, and
        # K ~ (1 - w^m)^2 by 2 m w^m / inner of that -- large next to saturation (t -> 1, w -> 0)
        w = np.where(S < 1, 1.0 - t, 1.0)
        condK = condK + np.where(S < 1, 2.0 * m * t * np.maximum(w, 1e-300) ** (m - 1.0)
                                 / np.maximum(inner, 1e-300), 0.0)
        # and S itself is computed (a difference and a quotient: one to two roundings in any
        # implementation): the same chain amplifies ITS relative error by another 1/m
        # (d ln t / d ln S), which matters for small m (clay-like n < 1.3) next to saturation
        condK = condK + np.where(S < 1, 2.0 * t * np.maximum(w, 1e-300) ** (m - 1.0)
                                 / np.maximum(inner, 1e-300), 0.0)
        u = np.where(Se < 1, Se ** (-1.0 / m) - 1.0, 0.0)
        condpsi = np.where(Se < 1, 1.0 + (u + 1.0) / (n * np.maximum(u, 1e-300)) +
                           np.abs(np.log(np.maximum(Se, 1e-300))) / (m * n), 2.0)
        # Se itself is a computed quantity (a difference and a quotient, or their logarithms: one
        # to two roundings in any implementation), and psi amplifies its relative error by
        # |d ln psi / d ln Se| = (u + 1) / (u n m) -- 1/m times the term above, which only covers a
        # rounding of S^(-1/m); it matters for small m (clay-like n < 1.3) next to saturation
        condpsi = condpsi + np.where(Se < 1, (u + 1.0) / (n * m * np.maximum(u, 1e-300)), 0.0)
    # conductivity factors: exp(gamma (T - T_ref)) and 10^(-Omega f_i) are powers too
    cf = om.cf
    if cf.viscosity_kind:
        condK = condK + np.abs(cf.gamma * (T - cf.T_ref))
    if cf.impedance_kind:
        with np.errstate(all="ignore"):
            tl = np.minimum(vl, nu - ti)
            f_i = np.where(tl + ti != 0, ti / (tl + ti), 0.0)
        condK = condK + cf.Omega * np.abs(f_i) * np.log(10.0)
    absK = eps * np.abs(K) * np.minimum(condK, 1e12) + 1e-300
    # psi near Se -> 1 from below is O(eps^(1/n)); give it that floor
    alpha = _percol(case, "vg_alpha", om.vg.alpha)[:, None]
    floor = (eps * 4) ** (1.0 / n) / alpha * (np.abs(Se - 1.0) < 4 * eps)
    abspsi = eps * np.abs(psi) * np.minimum(condpsi, 1e12) + floor + 1e-300
    absT = eps * (np.abs(T) + np.abs(T - om.earth.T_0)) * 2.0
    abskap = eps * np.abs(kap) * 8.0
    return dict(K=absK, psi=abspsi, T=absT, kappa=abskap)


def tendency_tolerance(case: Case, Cw: float = 16.0):
    """Per-cell absolute tolerance for each tendency (dict like the outputs)."""
    om = case.om
    diag = O.diagnostics(om, case.vl, case.ti, case.rhoe, case.T_aux)
    tol = closure_tolerances(case, diag, Cw)
    eps = float(np.finfo(case.dtype).eps) * Cw
    f8 = lambda a: np.asarray(a, dtype=np.float64)
    K, psi, T, kap = f8(diag["K"]), f8(diag["psi"]), f8(diag["T"]), f8(diag["kappa"])
    n = om.nlev
    dz = (om.zmax - om.zmin) / n
    zc, _ = O.grid(om.zmin, om.zmax, n)
    zc = zc[None, :]
    water = om.model != M.MODEL_HEAT
    heat = om.model != M.MODEL_RICHARDS
    e = om.earth
    rhocp_l = e.cp_l * e.rho_liq
    E = rhocp_l * (T - e.T_0) * K
    absE = rhocp_l * (tol["T"] * np.abs(K) + np.abs(T - e.T_0) * tol["K"])
    abs_h = tol["psi"] + eps * np.abs(zc)
    N = case.ncols
    dFw = np.zeros((N, n + 1))
    dFe = np.zeros((N, n + 1))
    if n > 1:
        g = (psi[:, 1:] - psi[:, :-1] + (zc[:, 1:] - zc[:, :-1])) / dz
        Kb = 0.5 * (K[:, 1:] + K[:, :-1])
        if water:
            dFw[:, 1:n] = (0.5 * (tol["K"][:, 1:] + tol["K"][:, :-1]) * np.abs(g)
                           + Kb * (abs_h[:, 1:] + abs_h[:, :-1]) / dz + eps * np.abs(Kb * g))
        if heat:
            gT = (T[:, 1:] - T[:, :-1]) / dz
            kb = 0.5 * (kap[:, 1:] + kap[:, :-1])
            dFe[:, 1:n] = (0.5 * (tol["kappa"][:, 1:] + tol["kappa"][:, :-1]) * np.abs(gT)
                           + kb * (tol["T"][:, 1:] + tol["T"][:, :-1]) / dz + eps * np.abs(kb * gT))
            if water:
                Eb = 0.5 * (E[:, 1:] + E[:, :-1])
                dFe[:, 1:n] += (0.5 * (absE[:, 1:] + absE[:, :-1]) * np.abs(g)
                                + np.abs(Eb) * (abs_h[:, 1:] + abs_h[:, :-1]) / dz
                                + eps * np.abs(Eb * g))
    # boundary faces: evaluate the face closures with the oracle
    for face, ci, k in ((M.FACE_BOTTOM, 0, 0), (M.FACE_TOP, n - 1, n)):
        kh, vh = om.bc.get((face, M.COMP_HYDROLOGY), (M.BC_NONE, 0.0))
        ke, ve = om.bc.get((face, M.COMP_ENERGY), (M.BC_NONE, 0.0))
        pch = om.percol_bc.get((face, M.COMP_HYDROLOGY))
        vh = np.asarray(pch if pch is not None else vh, dtype=np.float64) + np.zeros(N)
        dzb = dz / 2
        vl_f = case.vl.copy()
        re_f = None if case.rhoe is None else case.rhoe.copy()
        if water and kh == M.BC_DIRICHLET:
            vl_f[:, ci] = vh.astype(case.dtype)
        fc = Case("face", om, case.dtype, N, vl=vl_f, ti=case.ti, rhoe=re_f, T_aux=case.T_aux)
        dgf = O.diagnostics(om, vl_f, case.ti, case.rhoe, case.T_aux)
        tf = closure_tolerances(fc, dgf, Cw)
        Kf, psif, kapf = f8(dgf["K"])[:, ci], f8(dgf["psi"])[:, ci], f8(dgf["kappa"])[:, ci]
        if water:
            if kh == M.BC_FLUX:
                dFw[:, k] = eps * np.abs(vh)
            elif kh == M.BC_FREE_DRAINAGE:
                dFw[:, k] = tol["K"][:, ci]
            elif kh == M.BC_DIRICHLET:
                gb = (psif - psi[:, ci] + dzb) / dzb
                gb2 = (psif - psi[:, ci] - dzb) / dzb
                gmax = np.maximum(np.abs(gb), np.abs(gb2))
                dFw[:, k] = (tf["K"][:, ci] * gmax + Kf * (tf["psi"][:, ci] + tol["psi"][:, ci]) / dzb
                             + eps * Kf * gmax)
        if heat:
            if ke == M.BC_FLUX:
                dFe[:, k] = eps * abs(ve)
            elif ke == M.BC_DIRICHLET:
                gTb = (ve - T[:, ci]) / dzb
                dFe[:, k] = (tf["kappa"][:, ci] * np.abs(gTb) + kapf * (tol["T"][:, ci] + eps * abs(ve)) / dzb
                             + eps * np.abs(kapf * gTb))
    out = {}
    if water:
        out["vl"] = (dFw[:, 1:] + dFw[:, :-1]) / dz
        out["ti"] = np.zeros((N, n))
    if heat:
        out["rhoe"] = (dFe[:, 1:] + dFe[:, :-1]) / dz
    return out


# A plain statistic next to the tolerance model (which is wide on ill-conditioned cells by design):
# the share of cells whose tendency agrees to PLAIN_REL of the field's largest |tendency| (a
# tendency is a difference of face fluxes: next to its own size a cell's error is a statement about
# the cancellation, next to the field scale it is a statement about the implementation).
# Measured on MI355X (profiles/round3_plain_statistic.txt): 1.0 for every Float64 case at 1e-13; Float32
# at 1e-6 (8 eps) 0.65 ... 1.0, so Float32 is stated at 1e-5.  The floors below would catch a silent
# widening of the tolerance model.  Measured on MI355X (profiles/round3_plain_statistic.txt): >= 0.999 for every
# Float64 case, >= 0.99 for Float32; the floors below would catch a silent widening of the model.
PLAIN_REL = {np.dtype(np.float64): 1e-13, np.dtype(np.float32): 1e-5}
PLAIN_SHARE_MIN = {np.dtype(np.float64): 0.999, np.dtype(np.float32): 0.97}


def plain_statistic(case: Case, got, want):
    """{field: share of cells within PLAIN_REL[dtype] relative}"""
    out = {}
    for k in want:
        if k == "ti":
            continue
        g = np.asarray(got[k], dtype=np.float64)
        w = np.asarray(want[k], dtype=np.float64)
        ref = np.max(np.abs(w))
        out[k] = float(np.mean(np.abs(g - w) <= PLAIN_REL[np.dtype(case.dtype)] * ref)) if w.size else 1.0
    return out


def assert_tendencies_close(case: Case, got, want, Cw: Optional[float] = None, label="", plain: bool = False):
    if Cw is None:
        Cw = 4.0 if np.dtype(case.dtype) == np.float64 else 16.0
    tol = tendency_tolerance(case, Cw)
    if plain:
        for k, share in plain_statistic(case, got, want).items():
            assert share >= PLAIN_SHARE_MIN[np.dtype(case.dtype)], (
                f"{case.name}{label}:{k}: only {share:.4f} of the cells agree to "
                f"{PLAIN_REL[np.dtype(case.dtype)]:g} relative (floor {PLAIN_SHARE_MIN[np.dtype(case.dtype)]})")
    for k in want:
        g = np.asarray(got[k], dtype=np.float64)
        w = np.asarray(want[k], dtype=np.float64)
        assert g.shape == w.shape, (k, g.shape, w.shape)
        assert np.all(np.isfinite(g)), f"{case.name}:{k} non-finite"
        if k == "ti":
            assert np.all(g == 0.0), "d theta_i must be identically zero"
            continue
        err = np.abs(g - w)
        allowed = tol[k] + Cw * float(np.finfo(case.dtype).eps) * np.abs(w)
        bad = err > allowed
        if bad.any():
            idx = np.unravel_index(np.argmax(err / allowed), err.shape)
            raise AssertionError(
                f"{case.name}{label}:{k}: {bad.sum()} of {bad.size} cells outside tolerance; worst at "
                f"{idx}: got {g[idx]:.17g} want {w[idx]:.17g} err {err[idx]:.3g} allowed "
                f"{allowed[idx]:.3g}")


def error_summary(case: Case, got, want, Cw: float = 16.0):
    """max(err/allowed) per tendency -- how much of the tolerance is used."""
    tol = tendency_tolerance(case, Cw)
    out = {}
    for k in want:
        if k == "ti":
            continue
        g = np.asarray(got[k], dtype=np.float64)
        w = np.asarray(want[k], dtype=np.float64)
        allowed = tol[k] + Cw * float(np.finfo(case.dtype).eps) * np.abs(w)
        out[k] = float(np.max(np.abs(g - w) / allowed))
    return out
