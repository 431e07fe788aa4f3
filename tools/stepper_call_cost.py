#!/usr/bin/env python3
"""Per-call cost of the persistent column stepper: time of one lh_step_ssprk33 call against its number of
steps (T(n) = a + b n: a = table staging + getting the level-fastest registers from and to the
column-fastest planes, b = the step).   usage (GPU box): tools/stepper_call_cost.py [workload ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch
import bench
import parity_cases as pc

for wl in (sys.argv[1:] or ["c2", "c3", "c4"]):
    case = bench.build_case(wl, int(os.environ.get("NCOLS", "1000000")), 0)
    with pc.GpuModel(case) as g:
        F, L, ctx = g.F, g.L, g.ctx
        F.check(L.lh_set_tuning(ctx, ("persist=2" + os.environ.get("TUNE_EXTRA", "")).encode()), ctx)
        Y, Ya = g.prognostic_and_aux()
        ns_list = [1, 2, 3, 5, 10, 30, 100]
        ts = []
        for ns in ns_list:
            F.check(L.lh_step_ssprk33(ctx, Y, Ya, 0.0, 1e-3, ns, None), ctx)
            F.check(L.lh_synchronize(ctx), ctx)
            best = 1e9
            for _ in range(3):
                ms = C.c_float()
                F.check(L.lh_timer_start(ctx), ctx)      # HIP events on the context's own stream
                F.check(L.lh_step_ssprk33(ctx, Y, Ya, 0.0, 1e-3, ns, None), ctx)
                F.check(L.lh_timer_stop(ctx, C.byref(ms)), ctx)
                best = min(best, ms.value)
            ts.append(best)
        b, a = np.polyfit(ns_list, ts, 1)
        print(f"{wl}: " + "  ".join(f"n={n}: {t:.3f} ms" for n, t in zip(ns_list, ts)) + f"   => T(n) = {a:.3f} + {b:.4f} n ms", flush=True)
