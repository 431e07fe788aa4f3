#!/usr/bin/env python3
"""Time lh_rhs for a list of launch shapes (LH_TUNE) in ONE process, interleaved
rounds (cdna guide rule 24).  Needs a library built with `make TUNING=1` for the
cpl/pf/nt variants; block= works in every build.
usage: tools/tune_rhs.py [workload] [cfg ...]      cfg like "block=128,pf=2,nt=1"
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import bench
import parity_cases as pc

workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfgs = sys.argv[2:] or ["", "block=128", "block=64", "pf=2", "pf=3", "pf=4", "nt=1", "pf=2,nt=1",
                        "pf=4,nt=1", "cpl=2", "cpl=2,pf=2", "cpl=2,nt=1", "cpl=2,pf=2,nt=1",
                        "block=128,pf=2", "block=128,cpl=2,pf=2"]
ncols = int(os.environ.get("NCOLS", "1000000"))
case = bench.build_case(workload, ncols, 0)
nlev = case.om.nlev
bpc = bench.WORKLOADS[workload][1]
nmodels = int(os.environ.get("NMODELS", "2"))   # separate allocations: placement effects
models = []
pads = os.environ.get("PADS", "").split(",") if os.environ.get("PADS") else [None] * nmodels
nmodels = len(pads)
for k in range(nmodels):
    if pads[k] is not None:
        os.environ["LH_TUNE"] = f"{os.environ.get('PADKEY', 'pad')}={pads[k]}"
    g = pc.GpuModel(case)
    Y, Ya = g.prognostic_and_aux()
    dY = g.state(0)
    models.append((g, Y, Ya, dY))
var = g.F.LH_VAR_VARTHETA_L if case.om.model != 1 else g.F.LH_VAR_RHOE_INT
res = {(cfg, k): [] for cfg in cfgs for k in range(nmodels)}
same = {}
ref = None
for rnd in range(3):
    for cfg in cfgs:
        for k, (g, Y, Ya, dY) in enumerate(models):
            F, L, ctx = g.F, g.L, g.ctx
            F.check(L.lh_set_tuning(ctx, cfg.encode()), ctx)
            for _ in range(5):
                g.rhs(Y, Ya, dY)
            if rnd == 0 and k == 0:
                out = g.download(dY, var)
                ref = out if ref is None else ref
                same[cfg] = bool(np.array_equal(out, ref))
            F.check(L.lh_timer_start(ctx), ctx)
            reps = 40
            for _ in range(reps):
                g.rhs(Y, Ya, dY)
            ms = C.c_float()
            F.check(L.lh_timer_stop(ctx, C.byref(ms)), ctx)
            res[(cfg, k)].append(ms.value / reps)
print(f"workload {workload}: {ncols} cols x {nlev} lev, {bpc} B/cell; {nmodels} separately allocated models, pads={pads}")
for cfg in cfgs:
    line = f"  {cfg or '(default)':28s}"
    for k in range(nmodels):
        t = sorted(res[(cfg, k)])
        med = t[len(t) // 2]
        gbs = ncols * nlev * bpc / (med * 1e-3) / 1e9
        line += f" | m{k}: med {med:.4f} min {t[0]:.4f} ms {gbs:6.0f} GB/s {gbs/80:.1f}%"
    print(line + f" | same={same[cfg]}")
