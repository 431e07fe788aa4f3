#!/usr/bin/env python3
"""Which slot pairings of a plane arena are fast?  One arena of LH_TUNE=arena=N slots per
model; Y takes the first slots, then NOUT output states follow; time lh_rhs(Y -> dY_k) for
every k (and, with SWAP=1, lh_rhs(Y_k -> dY_0) for input states at the later slots).
usage: [LH_TUNE=arena=16] tools/placement_combos.py [workload=c2] [nmodels=3] [nout=7]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch  # noqa: F401  (before the HIP library, as bench.py)
import bench
import parity_cases as pc

workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
nmodels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nout = int(sys.argv[3]) if len(sys.argv) > 3 else 7
case = bench.build_case(workload, int(os.environ.get("NCOLS", "1000000")), 0)
F = pc._pkg()._ffi


def ptr(g, st):
    p, ls, cs = C.c_void_p(), C.c_int64(), C.c_int64()
    var = F.LH_VAR_VARTHETA_L if case.om.model != 1 else F.LH_VAR_RHOE_INT
    F.check(g.L.lh_state_device_ptr(g.ctx, st, var, C.byref(p), C.byref(ls), C.byref(cs)), g.ctx)
    return p.value


models = []
for k in range(nmodels):
    g = pc.GpuModel(case)
    Y, Ya = g.prognostic_and_aux()
    outs = [g.state(0) for _ in range(nout)]
    models.append((g, Y, Ya, outs))
res = {}
for rnd in range(3):
    for k, (g, Y, Ya, outs) in enumerate(models):
        for j, dY in enumerate(outs):
            for _ in range(4):
                g.rhs(Y, Ya, dY)
            F.check(g.L.lh_timer_start(g.ctx), g.ctx)
            for _ in range(25):
                g.rhs(Y, Ya, dY)
            ms = C.c_float()
            F.check(g.L.lh_timer_stop(g.ctx, C.byref(ms)), g.ctx)
            res.setdefault((k, j), []).append(ms.value / 25)
for k, (g, Y, Ya, outs) in enumerate(models):
    base = ptr(g, Y)
    print(f"model {k} (Y at {base:#x}): " + "  ".join(
        f"+{(ptr(g, o) - base) / 2**20:.0f}MiB {min(res[(k, j)]):.4f}" for j, o in enumerate(outs)))
