#!/usr/bin/env python3
"""Randomised closure-level parity: K, psi, kappa, T of lh_diagnostics and the tendencies of lh_rhs
against the oracle over wide state and parameter ranges -- ice from none to pores almost full,
bone-dry to oversaturated liquid, per-column van Genuchten n from 1.15 to 6, both conductivity
factors on and off, Float64 and Float32.  Prints, per configuration, how much of the tolerance
model (tests/parity_cases.py) the worst cell uses; exits non-zero when anything is outside it.

  python tools/fuzz_closures.py [nseeds=4]        (on a GPU box)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import case_model as M          # noqa: E402
import parity_cases as pc       # noqa: E402


def fuzz_case(seed, dtype, factors, percol, model):
    rng = np.random.default_rng(seed)
    n, N = 24, 640
    sp, vg = pc.coupled_soil()
    bc = {(M.FACE_TOP, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, 0.3),
          (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_FREE_DRAINAGE, 0.0)}
    if model == M.MODEL_COUPLED:
        bc.update({(M.FACE_TOP, M.COMP_ENERGY): (M.BC_DIRICHLET, 277.0),
                   (M.FACE_BOTTOM, M.COMP_ENERGY): (M.BC_FLUX, 0.02)})
    kw = {}
    nu = np.full(N, sp.nu)
    thr = np.zeros(N)
    if percol:
        nu = rng.uniform(0.3, 0.6, N)
        thr = rng.uniform(0.0, 0.1, N)
        # (Float32 with clay-like n and bone-dry cells leaves the Float32 range in the reference --
        # S^(-1/m) = Inf -- and, a few orders later, in any implementation: n >= 1.4 there)
        kw["percol"] = dict(vg_n=rng.uniform(1.15 if dtype == np.float64 else 1.4, 6.0, N), vg_alpha=rng.uniform(0.5, 8.0, N),
                            vg_Ksat=10.0 ** rng.uniform(-8, -4, N), vg_theta_r=thr, nu=nu)
    om = M.CaseModel(model, n, -1.2, 0.0, soil=sp, vg=vg, bc=bc,
                     cf=M.default_cf(viscosity=factors, impedance=factors), **kw)
    por = (nu - thr)[:, None]
    # ice: a third of the columns none, a third moderate, a third up to 98 % of the pore space
    kind = rng.integers(0, 3, N)[:, None]
    u = rng.random((N, n))
    ti = np.where(kind == 0, 0.0, np.where(kind == 1, 0.15 * u * por, 0.98 * u ** 0.3 * por))
    ti = np.where(rng.random((N, n)) < 0.3, 0.0, ti)          # ice-free cells inside icy columns
    # liquid: log-uniform from 1e-9 of the pore space to 15 % above it (oversaturated)
    r = rng.random((N, n))
    vl = thr[:, None] + por * np.where(r < 0.15, 10.0 ** rng.uniform(-9 if dtype == np.float64 else -5, -2, (N, n)),
                                       np.where(r < 0.9, rng.uniform(0.01, 1.0, (N, n)),
                                                rng.uniform(1.0, 1.15, (N, n))))
    if model == M.MODEL_RICHARDS:   # (a heat model takes S_r^x of it: DomainError in the reference)
        vl = np.where(rng.random((N, n)) < 0.02, thr[:, None] - 1e-3, vl)     # below theta_r: the eps clamp
    T = rng.uniform(262.0, 300.0, (N, n))
    e = om.earth
    tl = np.minimum(vl, nu[:, None] - ti)
    rho_c_s = sp.rho_c_ds + tl * (e.cp_l * e.rho_liq) + ti * (e.cp_i * e.rho_ice)
    rhoe = rho_c_s * (T - e.T_0) - ti * e.rho_ice * e.LH_f0
    args = dict(vl=vl.astype(dtype), ti=ti.astype(dtype))
    if model == M.MODEL_COUPLED:
        args["rhoe"] = rhoe.astype(dtype)
    else:
        args["T_aux"] = T.astype(dtype)
    return pc.Case(f"fuzz{seed}", om, dtype, N, **args)


def run(nseeds=4, verbose=True):
    """-> (worst fraction of the tolerance used, number of configurations outside it)"""
    worst_all, failed = 0.0, 0
    for dtype in (np.float64, np.float32):
        cw = 4.0 if dtype == np.float64 else 16.0
        for model in (M.MODEL_COUPLED, M.MODEL_RICHARDS):
            for factors in (False, True):
                for percol in (False, True):
                    for seed in range(nseeds):
                        case = fuzz_case(1000 * seed + 7, dtype, factors, percol, model)
                        want = pc.O.diagnostics(case.om, case.vl, case.ti, case.rhoe, case.T_aux)
                        got = pc.run_gpu_diagnostics(case)
                        tol = pc.closure_tolerances(case, want, cw)
                        used = {}
                        with np.errstate(all="ignore"):
                            nn = pc._percol(case, "vg_n", case.om.vg.n)[:, None]
                            mm_ = 1.0 - 1.0 / nn
                            nu_ = pc._percol(case, "nu", case.om.soil.nu)[:, None]
                            th_ = pc._percol(case, "vg_theta_r", case.om.vg.theta_r)[:, None]
                            S_ = (np.maximum(case.vl.astype(np.float64), th_ + np.finfo(dtype).eps) - th_) / (nu_ - th_)
                            t_ = np.where(S_ < 1, S_ ** (1.0 / mm_), 1.0)
                            inner_ = np.where(S_ < 1, 1.0 - (1.0 - t_) ** mm_, 1.0)
                            ill = inner_ < 64 * np.finfo(dtype).eps
                        for k in ("K", "psi", "T", "kappa"):
                            if model == M.MODEL_RICHARDS and k in ("T", "kappa"):
                                continue
                            g, w = got[k].astype(np.float64), want[k].astype(np.float64)
                            fin = np.isfinite(w)
                            if k == "psi":
                                # Float32, clay-like n, bone dry: the reference's S^(-1/m) overflows to Inf
                                # (psi = -Inf); the log-domain closure has no such intermediate and returns
                                # the finite value (deviations): not compared
                                g = np.where(np.isneginf(w) & np.isfinite(g), w, g)
                            if k == "K":
                                # K_r = sqrt(S) (1 - (1 - S^(1/m))^m)^2 with 1 - S^(1/m) within a few ulp of
                                # 1: the value IS rounding noise in the reference too (conditioning 2/inner
                                # beyond the 1e12 cap of the tolerance model); K < 1e-25 Ksat there
                                g = np.where(ill, w, g)
                            if not np.array_equal(np.isfinite(g), fin):
                                used[k] = float("inf")
                                bad = np.argwhere(np.isfinite(g) != fin)[0]
                                i = tuple(bad)
                                print(f"    {k}: finiteness differs at {i}: got {g[i]!r} want {w[i]!r} vl={case.vl[i]!r} "
                                      f"ti={case.ti[i]!r} percol={ {q: float(v[i[0]]) for q, v in case.om.percol.items()} }")
                                continue
                            ratio = np.where(fin, np.abs(g - w) / tol[k], 0.0)
                            used[k] = float(ratio.max())
                            if used[k] > 1.0:
                                i = np.unravel_index(np.argmax(ratio), ratio.shape)
                                print(f"    {k}: worst at {i}: got {g[i]!r} want {w[i]!r} tol {tol[k][i]:.3g} vl={case.vl[i]!r} "
                                      f"ti={case.ti[i]!r} percol={ {q: float(v[i[0]]) for q, v in case.om.percol.items()} }")
                        try:
                            d_want = pc.run_oracle_rhs(case)
                            # an input the REFERENCE answers with NaN (liquid below theta_r next to ice under the
                            # impedance factor: 0/0) must come back as NaN in the same cells, with the flag raised
                            nonfinite_expected = any(not np.all(np.isfinite(v)) for v in d_want.values())
                            with pc.GpuModel(case) as g_:
                                Y_, Ya_ = g_.prognostic_and_aux()
                                dY_ = g_.state(0)
                                g_.rhs(Y_, Ya_, dY_)
                                d_got = g_.tendencies(dY_)
                                assert (g_.status() != 0) == nonfinite_expected, "non-finite flag disagrees with the oracle"
                            ok_cols = ~ill.any(axis=1)     # (columns with a noise-valued K are not compared)
                            for k in d_want:
                                bad_w = ~np.isfinite(d_want[k])
                                assert np.array_equal(bad_w, ~np.isfinite(d_got[k])), f"{k}: non-finite cells differ from the oracle's"
                                ok_cols &= ~bad_w.any(axis=1)
                            tolt = pc.tendency_tolerance(case, cw)
                            for k in d_want:
                                if k == "ti":
                                    assert np.all(d_got[k] == 0)
                                    continue
                                g, w = d_got[k].astype(np.float64)[ok_cols], d_want[k].astype(np.float64)[ok_cols]
                                allowed = tolt[k][ok_cols] + cw * float(np.finfo(dtype).eps) * np.abs(w)
                                used["d" + k] = float(np.max(np.abs(g - w) / allowed))
                        except AssertionError as ex:
                            used["rhs"] = float("inf")
                            print("   ", ex)
                        worst = max(used.values())
                        worst_all = max(worst_all, worst)
                        flag = "" if worst <= 1.0 else "   <-- OUTSIDE"
                        failed += worst > 1.0
                        if verbose or worst > 1.0:
                            print(f"{np.dtype(dtype).name} model={model} factors={int(factors)} percol={int(percol)} "
                                  f"seed={seed}: " + " ".join(f"{k}={v:.3f}" for k, v in used.items()) + flag, flush=True)
    return worst_all, failed


def main():
    worst_all, failed = run(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
    print(f"worst fraction of the tolerance used: {worst_all:.3f}; configurations outside: {failed}")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
