#!/usr/bin/env python3
"""Launch-bound regime: time lh_step_ssprk33 for small ensembles (steps/s)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch  # noqa: F401
import bench
import parity_cases as pc

SIZES = os.environ.get("SIZES")
CASES = [tuple(x.split(":")) for x in SIZES.split(",")] if SIZES else \
    [("c2", 1), ("c2", 1024), ("c3", 1), ("c3", 16384), ("c2", 65536)]
for wl, ncols in CASES:
    ncols = int(ncols)
    case = bench.build_case(wl, ncols, 0)
    g = pc.GpuModel(case)
    F = g.F
    Y, Ya = g.prognostic_and_aux()
    nsteps = int(os.environ.get("NSTEPS", "3000"))
    F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, 1e-6, min(200, nsteps), None), g.ctx)
    F.check(g.L.lh_synchronize(g.ctx), g.ctx)
    calls = int(os.environ.get("CALLS", "1"))
    for _ in range(3 if calls > 1 else 0):
        F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, 1e-6, nsteps, None), g.ctx)
    F.check(g.L.lh_synchronize(g.ctx), g.ctx)
    t0 = time.perf_counter()
    for _ in range(calls):
        F.check(g.L.lh_step_ssprk33(g.ctx, Y, Ya, 0.0, 1e-6, nsteps, None), g.ctx)
    enq = (time.perf_counter() - t0) / calls
    F.check(g.L.lh_synchronize(g.ctx), g.ctx)
    el = (time.perf_counter() - t0) / calls
    print(f"{wl} ncols={ncols:6d}: {el / nsteps * 1e6:8.2f} us per step ({el / nsteps / 3 * 1e6:.2f} us per stage launch; host enqueue {enq / nsteps * 1e6:.2f} us per step)", flush=True)
    g.close()
