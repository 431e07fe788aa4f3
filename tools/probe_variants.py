#!/usr/bin/env python3
"""lh_stream_probe under launch-shape overrides (LH_TUNE) on the C2 planes: xcd map, nontemporal policy."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import __graft_entry__ as g

pkg = g.load_package()
F, W = pkg._ffi, pkg.workloads
case = W.make_case(sys.argv[1] if len(sys.argv) > 1 else "c2_richards_f64", ncols=1_000_000)
with W.GpuModel(case) as gm:
    L, ctx = gm.L, gm.ctx
    Y, Ya = gm.prognostic_and_aux()
    dY = gm.state(0)
    m = case.om.model
    rm = {0: 0b001, 1: 0b100, 2: 0b101}[m]
    for rep in range(2):
        for tune in (b"", b"xcd=0", b"nt=0", b"xcd=0,nt=0"):
            F.check(L.lh_set_tuning(ctx, tune), ctx)
            ms = C.c_float()
            F.check(L.lh_stream_probe(ctx, Y, rm, dY, rm, 40, C.byref(ms)), ctx)
            print(f"{tune.decode() or 'default':12s} {ms.value:.4f} ms")
