#!/usr/bin/env python3
"""Cost of a TIME-DEPENDENT prescribed temperature (Richards + viscosity factor, 1e6 columns): the
reference's rhs! re-evaluates T_profile(z, t) at every stage time (right_hand_side.jl:37-42), so the
host refreshes Ya before every stage launch (lh_ssprk33_stage).  Three ways, per SSPRK33 step:
  constant   no refresh (Ya uploaded once)
  profile    lh_upload_profile of nlev numbers before every stage (asynchronous, pinned staging)
  plane      lh_upload of the broadcast plane before every stage (what round 2 did)
usage: tools/aux_refresh_probe.py [ncols]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (before any HIP library is loaded)
import __graft_entry__ as g

pkg = g.load_package()
F, W = pkg._ffi, pkg.workloads
ncols = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
case = W.make_case("richards_viscosity_profile_f64", ncols=ncols)
n = case.om.nlev
with W.GpuModel(case) as gm:
    L, ctx = gm.L, gm.ctx
    Y, Ya = gm.prognostic_and_aux()
    U = gm.state(0)
    dt = 1e-6
    prof = np.ascontiguousarray(case.T_aux[0])
    plane = np.ascontiguousarray(case.T_aux)

    def step(refresh):
        for stage in (1, 2, 3):
            if refresh == "profile":
                F.check(L.lh_upload_profile(ctx, Ya, F.LH_VAR_T, prof.ctypes.data), ctx)
            elif refresh == "plane":
                F.check(L.lh_upload(ctx, Ya, F.LH_VAR_T, plane.ctypes.data, 1, n), ctx)
            F.check(L.lh_ssprk33_stage(ctx, stage, Y, U, Ya, dt, None), ctx)

    res = {}
    for mode, reps in (("constant", 40), ("profile", 40), ("plane", 3), ("constant", 40), ("profile", 40)):
        if mode != "plane":   # (a plane upload leaves a plane: back to the profile for the other modes)
            F.check(L.lh_upload_profile(ctx, Ya, F.LH_VAR_T, prof.ctypes.data), ctx)
        for _ in range(3):
            step(mode)
        F.check(L.lh_synchronize(ctx), ctx)
        t0 = time.perf_counter()
        for _ in range(reps):
            step(mode)
        F.check(L.lh_synchronize(ctx), ctx)
        res.setdefault(mode, []).append((time.perf_counter() - t0) / reps * 1e3)
    print(f"{ncols} columns x {n} levels, Richards + viscosity (f64), ms per SSPRK33 step (wall clock, 3 stage launches):")
    for k, v in res.items():
        print(f"  {k:9s} " + " ".join(f"{x:8.3f}" for x in v))
    c, p = min(res["constant"]), min(res["profile"])
    print(f"  time-dependent profile / constant = {p / c:.3f}; plane upload / constant = {min(res['plane']) / c:.1f}")
