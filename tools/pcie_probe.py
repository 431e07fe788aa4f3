#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary: lh_upload of the prognostic planes + lh_rhs +
lh_download of the tendencies, per evaluation (what a host that keeps its state in
host memory would see; never bench.py's `value`)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch  # noqa: F401
import bench
import parity_cases as pc

case = bench.build_case("c2", 1_000_000, 0)
g = pc.GpuModel(case)
F = g.F
Y, Ya = g.prognostic_and_aux()
dY = g.state(0)
n = case.om.nlev
for layout, arr in (("level-fastest [ncols][nlev] (parent(field) per column)", np.ascontiguousarray(case.vl)),
                    ("column-fastest [nlev][ncols] (the library's own)", np.ascontiguousarray(case.vl.T))):
    out = np.empty_like(arr)
    ls, cs = (1, n) if arr.shape[0] == case.ncols else (case.ncols, 1)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        F.check(g.L.lh_upload(g.ctx, Y, F.LH_VAR_VARTHETA_L, arr.ctypes.data, ls, cs), g.ctx)
        F.check(g.L.lh_upload(g.ctx, Y, F.LH_VAR_THETA_I, arr.ctypes.data, ls, cs), g.ctx)
        g.rhs(Y, Ya, dY)
        F.check(g.L.lh_download(g.ctx, dY, F.LH_VAR_VARTHETA_L, out.ctypes.data, ls, cs), g.ctx)
        F.check(g.L.lh_download(g.ctx, dY, F.LH_VAR_THETA_I, out.ctypes.data, ls, cs), g.ctx)
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    gb = 4 * arr.nbytes / 1e9
    print(f"{layout}: {t * 1e3:.1f} ms per evaluation (2 planes up, 2 down = {gb:.2f} GB, "
          f"{gb / t:.1f} GB/s) -> {case.ncols * n / t:.3g} cell-updates/s")
