#!/usr/bin/env python3
"""Worst cells of the closure-level parity (lh_diagnostics vs the oracle) for one case.
usage (GPU box): python tools/worst_closure.py CASE [field]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import parity_cases as pc  # noqa: E402

name = sys.argv[1]
field = sys.argv[2] if len(sys.argv) > 2 else "K"
case = pc.make_case(name)
got = pc.run_gpu_diagnostics(case)
want = pc.O.diagnostics(case.om, case.vl, case.ti, case.rhoe, case.T_aux)
Cw = 4.0 if case.dtype == np.float64 else 16.0
tol = pc.closure_tolerances(case, want, Cw)
err = np.abs(got[field].astype(np.float64) - want[field].astype(np.float64))
ratio = err / tol[field]
order = np.argsort(ratio.ravel())[::-1][:12]
nu = case.om.soil.nu
for f in order:
    c, i = np.unravel_index(f, ratio.shape)
    vl, ti = float(case.vl[c, i]), float(case.ti[c, i])
    S = max(vl, case.om.vg.theta_r + 2.2e-16) / nu
    print(f"col {c} lev {i}: vl={vl:.17g} ti={ti:.6g} S={S:.6g} got={float(got[field][c, i]):.17g} "
          f"want={float(want[field][c, i]):.17g} relerr={err[c, i] / abs(float(want[field][c, i]) or 1):.3g} err/tol={ratio[c, i]:.3f}")
