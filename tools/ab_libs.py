#!/usr/bin/env python3
"""Same-process A/B of library builds: one model per library on the same inputs,
interleaved timing rounds (device/placement variance exceeds most code effects,
so separate runs cannot be compared).
usage: tools/ab_libs.py <workload> <name|path>[@LH_TUNE string] ...     name -> lib/variants/liblandhydro_hip_<name>.so,
                                                        "product" -> the in-tree library
"""
import ctypes as C
import os
import sys

os.environ["LH_ALLOW_MISSING_SYMBOLS"] = "1"   # older library builds lack the newest entry points
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch   # before any HIP library is loaded (as bench.py does)
import bench
import parity_cases as pc

workload = sys.argv[1]
names = sys.argv[2:]
ncols = int(os.environ.get("NCOLS", "1000000"))
mode = os.environ.get("MODE", "rhs")     # rhs | rhs_dt | step | persist
case = bench.build_case(workload, ncols, 0)
bpc = bench.WORKLOADS[workload][1]
F = pc._pkg()._ffi
product = F.LIB_PATH
models = []
copies = int(os.environ.get("COPIES", "1"))   # models per library, allocated round-robin: separates
lib_names = names                             # the placement effect from the code effect
names = [f"{nm}#{k}" for k in range(copies) for nm in lib_names] if copies > 1 else names
for nm_full in names:
    nm = nm_full.split("#")[0]
    if "@" in nm:                                  # name@tune: per-model LH_TUNE (read at lh_create)
        nm, os.environ["LH_TUNE"] = nm.split("@", 1)
    path = product if nm == "product" else (nm if os.path.sep in nm else os.path.join(
        os.path.dirname(product), "variants", f"liblandhydro_hip_{nm}.so"))
    F._lib, F.LIB_PATH = None, path
    g = pc.GpuModel(case)
    Y, Ya = g.prognostic_and_aux()
    dY = g.state(0)
    if os.environ.get("TUNE"):                     # every model at its measured-best placement
        tb, ta = C.c_float(), C.c_float()
        F.check(g.L.lh_tune_placement(g.ctx, Y, Ya, dY if mode != "step" else None, int(os.environ["TUNE"]),
                                      F.LH_PLACE_MOVE_INPUT, C.byref(tb), C.byref(ta)), g.ctx)
        print(f"  tuned {nm_full}: {tb.value:.4f} -> {ta.value:.4f} ms", flush=True)
    models.append((nm_full, g, Y, Ya, dY))
var = F.LH_VAR_VARTHETA_L if case.om.model != 1 else F.LH_VAR_RHOE_INT
res = {nm: [] for nm in names}
outs = {}
dbuf = torch.zeros(2, dtype=torch.float64, device="cuda")     # stable-dt output (one FT value)
for rnd in range(4):
    for nm, g, Y, Ya, dY in models:
        L, ctx = g.L, g.ctx

        def once():
            if mode == "rhs":
                g.rhs(Y, Ya, dY)
            elif mode == "rhs_dt":
                F.check(L.lh_rhs_stable_dt(ctx, 0.0, Y, Ya, dY, 0.4, C.c_void_p(dbuf.data_ptr())), ctx)
            elif mode == "persist":   # 30 steps in one call: the engine lh_step_ssprk33 chooses (persistent column stepper)
                F.check(L.lh_step_ssprk33(ctx, Y, Ya, 0.0, 1e-9, 30, None), ctx)
            else:   # 3 fused stage kernels per step; tiny dt keeps the state where it is
                F.check(L.lh_step_ssprk33(ctx, Y, Ya, 0.0, 1e-9, 1, None), ctx)
        for _ in range(5 if mode != "persist" else 1):
            once()
        if rnd == 0 and mode in ("rhs", "rhs_dt"):
            outs[nm] = g.download(dY, var)
        F.check(L.lh_timer_start(ctx), ctx)
        reps = {"step": 14, "persist": 3}.get(mode, 40)
        for _ in range(reps):
            once()
        ms = C.c_float()
        F.check(L.lh_timer_stop(ctx, C.byref(ms)), ctx)
        res[nm].append(ms.value / reps / {"step": 3, "persist": 30}.get(mode, 1))   # step: per evaluation; persist: per step
print(f"workload {workload} mode {mode}: {ncols} cols x {case.om.nlev} lev, {bpc} B/cell")
ref = outs.get(names[0])
def plane_ptrs(g, st, nvars):
    out = []
    for v in range(nvars):
        p, ls, cs = C.c_void_p(), C.c_int64(), C.c_int64()
        if g.L.lh_state_device_ptr(g.ctx, st, v, C.byref(p), C.byref(ls), C.byref(cs)) == 0 and p.value:
            out.append(p.value)
    return out
if os.environ.get("SHOWPTR"):
    for nm, g, Y, Ya, dY in models:
        ps = plane_ptrs(g, Y, 3) + plane_ptrs(g, dY, 3)
        print(f"  {nm:16s} " + " ".join(f"{p:#014x}" for p in ps))
for nm in names:
    best = min(res[nm])
    gbs = bpc * ncols * case.om.nlev / (best * 1e-3) / 1e9
    eq = "" if ref is None else f"  same_as_first={bool(np.array_equal(outs[nm], ref, equal_nan=True))}"
    print(f"  {nm:16s} " + " ".join(f"{x:.4f}" for x in res[nm]) + f"  best {best:.4f} ms  {gbs:7.1f} GB/s "
          f"{gbs / 80:.1f}%{eq}")
