#!/usr/bin/env python3
"""Map lh_rhs launch time over (input state, output state) placements inside one context:
NIN input states and NOUT output states allocated interleaved (I0 O0..Ok I1 ...), every
input timed against every output.  Shows whether "fast" is a property of single planes,
of pairs, or of distance.
usage: [LH_TUNE=arena=N] tools/placement_map.py [workload=c2] [nin=4] [nout=16]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch  # noqa: F401
import bench
import parity_cases as pc

workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
nin = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nout = int(sys.argv[3]) if len(sys.argv) > 3 else 16
case = bench.build_case(workload, int(os.environ.get("NCOLS", "1000000")), 0)
F = pc._pkg()._ffi
g = pc.GpuModel(case)
Y, Ya = g.prognostic_and_aux()
var = F.LH_VAR_VARTHETA_L if case.om.model != 1 else F.LH_VAR_RHOE_INT


def ptr(st):
    p, ls, cs = C.c_void_p(), C.c_int64(), C.c_int64()
    F.check(g.L.lh_state_device_ptr(g.ctx, st, var, C.byref(p), C.byref(ls), C.byref(cs)), g.ctx)
    return p.value


ins, outs = [Y], []
per = nout // nin
for k in range(nin):
    if k:
        s = g.state(0)
        F.check(g.L.lh_state_copy(g.ctx, s, Y), g.ctx)
        ins.append(s)
    outs += [g.state(0) for _ in range(per)]
base = min(ptr(s) for s in ins + outs)
for _ in range(20):
    g.rhs(ins[0], Ya, outs[0])
print("inputs at  MiB:", [round((ptr(s) - base) / 2**20) for s in ins])
print("outputs at MiB:", [round((ptr(s) - base) / 2**20) for s in outs])
tab = np.zeros((len(ins), len(outs)))
for rnd in range(2):
    for i, a in enumerate(ins):
        for j, b in enumerate(outs):
            for _ in range(2):
                g.rhs(a, Ya, b)
            F.check(g.L.lh_timer_start(g.ctx), g.ctx)
            for _ in range(8):
                g.rhs(a, Ya, b)
            ms = C.c_float()
            F.check(g.L.lh_timer_stop(g.ctx, C.byref(ms)), g.ctx)
            t = ms.value / 8
            tab[i, j] = t if rnd == 0 else min(tab[i, j], t)
for i in range(len(ins)):
    print(f"in{i}: " + " ".join(f"{t:.3f}" for t in tab[i]))
