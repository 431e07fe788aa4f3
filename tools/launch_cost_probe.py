#!/usr/bin/env python3
"""Fixed cost of one column launch: lh_rhs time against the number of levels (1e6 Richards Float64
columns, zero-flux BCs, known-zero ice), HIP events around 50 back-to-back launches.
The intercept of the straight line is what a launch costs before and after its level loop
(dispatch, table staging, first loads, drain)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: F401,E402  (before any HIP library is loaded)
import case_model as M      # noqa: E402
import parity_cases as pc   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rows = []
for n in (1, 2, 4, 8, 16, 32, 64, 128):
    om = M.CaseModel(M.MODEL_RICHARDS, n, -0.02 * n, 0.0, bc=pc._flux_bcs(hydrology=0.0))
    vl = np.full((N, n), 0.3) + 0.05 * pc.uhash(np.arange(N)[:, None], np.arange(n)[None, :], 1000)
    case = pc.Case("probe", om, np.float64, N, vl=vl, ti=np.zeros((N, n)))
    with pc.GpuModel(case) as g:
        F, L, ctx = g.F, g.L, g.ctx
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        for _ in range(10):
            g.rhs(Y, Ya, dY)
        best = 1e9
        for _ in range(3):
            F.check(L.lh_timer_start(ctx), ctx)
            for _ in range(50):
                F.check(L.lh_rhs(ctx, 0.0, Y, Ya, dY), ctx)
            ms = C.c_float()
            F.check(L.lh_timer_stop(ctx, C.byref(ms)), ctx)
            best = min(best, ms.value / 50)
    rows.append((n, best))
    print(f"nlev {n:4d}: {best * 1e3:8.1f} us per launch", flush=True)
x = np.array([r[0] for r in rows[3:]], float)
y = np.array([r[1] for r in rows[3:]], float)
b, a = np.polyfit(x, y, 1)
print(f"fit over nlev >= 8: {a * 1e3:.1f} us + {b * 1e3:.2f} us per level")
