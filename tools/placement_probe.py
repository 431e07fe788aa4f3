#!/usr/bin/env python3
"""Probe: kernel time of lh_rhs vs plane addresses (one process, several models)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import bench, parity_cases as pc
use_torch = "--torch" in sys.argv
stream = None
if use_torch:
    import torch
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); stream = ts.cuda_stream
case = bench.build_case("c2", 1_000_000, 0)
models = []
for pad in os.environ.get("PADS", "266240,266240,0,266240").split(","):
    os.environ["LH_TUNE"] = f"pad={pad}"
    g = pc.GpuModel(case, stream=stream)
    Y, Ya = g.prognostic_and_aux(); dY = g.state(0)
    addrs = []
    for st, vars_ in ((Y, (0, 1)), (dY, (0, 1))):
        for v in vars_:
            p = C.c_void_p(); g.F.check(g.L.lh_state_device_ptr(g.ctx, st, v, C.byref(p), None, None), g.ctx)
            addrs.append(p.value)
    models.append((pad, g, Y, Ya, dY, addrs))
for rnd in range(2):
    for pad, g, Y, Ya, dY, addrs in models:
        for _ in range(5): g.rhs(Y, Ya, dY)
        g.F.check(g.L.lh_timer_start(g.ctx), g.ctx)
        for _ in range(40): g.rhs(Y, Ya, dY)
        ms = C.c_float(); g.F.check(g.L.lh_timer_stop(g.ctx, C.byref(ms)), g.ctx)
        print(f"round {rnd} pad={pad:>8s} {ms.value/40:.4f} ms  addrs " + " ".join(f"{a:#x}" for a in addrs) + "  low21: " + " ".join(f"{a & 0x1fffff:#x}" for a in addrs), flush=True)
