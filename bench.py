#!/usr/bin/env python3
"""bench.py -- batched Richards RHS throughput on MI355X.

Metric (BASELINE.json): column-cell updates/s (RHS evals).  A "step" is ONE
lh_rhs launch over one batch: `rhs!(dY, Y, Ya, t)` for every column.  Workload at
every N: BASELINE config C2 per GPU -- 1e6 independent 64-layer Richards columns,
Float64, loam van Genuchten, zero-flux BCs, synthetic wetting-front state
 -- i.e. weak scaling, columns block-partitioned over ranks, no
data-path collective.  Every third evaluation (= once per SSPRK33 step) is the fused
lh_rhs_stable_dt launch at every N; for N > 1 its one-value result is
min-all-reduced over RCCL inside the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the driver contract).

tests/parity_cases.py supplies the synthetic-input generator (pure numpy) and the C-ABI
harness (GpuModel); the CPU oracle under oracle/ is loaded and executed in the
cpu_baseline leg only (tests/test_abi_cpu.py checks that input generation does not).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 measured copy


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-placement-tune", action="store_true",
                    help="skip lh_tune_placement (measure the first placement the allocator gives)")
    ap.add_argument("--ncols", type=int, default=1_000_000, help="columns per GPU")
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                         "the multi-process path on a single GPU)")
    ap.add_argument("--stepper", action="store_true", default=True,
                    help="also time the device SSPRK33 stepper, both engines (extra JSON fields; default)")
    ap.add_argument("--no-stepper", dest="stepper", action="store_false")
    return ap.parse_args()


WORKLOADS = {
    # name -> (case name in tests/parity_cases.py, algorithmic bytes per cell-update, dtype tag)
    "c2": ("c2_richards_f64", 32.0, "f64"),
    "c3": ("c3_coupled_f32", 24.0, "f32"),
    "c4": ("c4_richards_f64_128", 32.0, "f64"),
    "c5": ("c5_percol_f64", 32.5, "f64"),
    # SURVEY 8(f)-3 at scale: ice lenses + saturated zones + both conductivity factors, Dirichlet
    # top / free-drainage bottom (coupled, 48 levels); viscosity factor with prescribed T (Richards,
    # 50 levels: one more plane read)
    "f3c32": ("mixed_smooth_f32", 24.0, "f32"),
    "f3c64": ("mixed_smooth_f64", 48.0, "f64"),
    "f3v64": ("richards_viscosity_f64", 40.0, "f64"),
}


def build_case(workload, ncols, col_offset):
    """Synthetic inputs, generated chunk-wise on the host (counter-based hash, so
    every rank builds exactly its own block of the global ensemble)."""
    import parity_cases as pc
    name = WORKLOADS[workload][0]
    return pc.make_case(name, ncols=ncols, col_offset=col_offset)


def _usable_cores():
    """Cores this process may really use: the affinity mask, cut down to the cgroup CPU
    quota when there is one (a GPU box hands out a 16-core share per GPU of a 128-core
    host; 128 OpenMP threads on that share just time the scheduler)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                tok = fh.read().split()
            if path.endswith("cpu.max"):
                if tok[0] != "max":
                    n = min(n, max(1, int(int(tok[0]) / int(tok[1]) + 0.5)))
            else:
                q = int(tok[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                        n = min(n, max(1, int(q / int(fh.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("LH_CPU_THREADS")
    return int(env) if env else n


def cpu_baseline(case, seconds):
    """The oracle (kind 'port': a scalar C restatement of the reference's Julia
    path; the reference itself cannot run here) timed on a bounded sample of the
    same workload, all host cores via OpenMP over columns."""
    import dataclasses
    import oracle_py as O
    import parity_cases as pc
    cores = _usable_cores()
    threads = max(1, min(cores, O.max_threads()))
    n = case.om.nlev
    sample = min(case.ncols, 20000)
    sl = lambda a: None if a is None else np.ascontiguousarray(a[:sample])
    om = case.om
    if om.percol or om.percol_bc:
        om = dataclasses.replace(om, percol={k: v[:sample] for k, v in om.percol.items()},
                                 percol_bc={k: v[:sample] for k, v in om.percol_bc.items()})
    sub = dataclasses.replace(case, om=om, ncols=sample, vl=sl(case.vl), ti=sl(case.ti),
                              rhoe=sl(case.rhoe), T_aux=sl(case.T_aux))
    pc.run_oracle_rhs(sub, nthreads=threads)            # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        pc.run_oracle_rhs(sub, nthreads=threads)
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds or reps >= 2000:
            break
    # the same restatement on ONE thread (the reference has no threading): ~2 s of it
    r1, t1 = 0, time.perf_counter()
    while True:
        pc.run_oracle_rhs(sub, nthreads=1)
        r1 += 1
        el1 = time.perf_counter() - t1
        if el1 >= min(2.0, seconds) or r1 >= 50:
            break
    import shutil
    julia = shutil.which("julia")
    return {"value": sample * n * reps / el, "unit": "cell-updates/s", "cores": threads,
            "single_thread_value": sample * n * r1 / el1,
            "kind": "port",
            "reference_julia": ("julia found at %s but LandHydrology.jl's un-vendored dependencies "
                                "(ClimaCore, CLIMAParameters, OrdinaryDiffEq) are not shipped with this "
                                "repository: reference CPU path not timed" % julia) if julia
            else "reference CPU path: unavailable on this host (no julia binary)",
            "sample": f"{reps} RHS evals of the first {sample} columns x {n} levels of the same "
                      f"workload, OpenMP over columns on {threads} threads ({el:.1f} s)"}


def main():
    a = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    assert world == a.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {a.gpus}"

    import __graft_entry__ as g
    import parity_cases as pc
    pkg = g.load_package()
    F = pkg._ffi

    # weak scaling: rank r owns the block [r*ncols, (r+1)*ncols) of the global ensemble
    lo, hi = pkg.partition.block_range(world * a.ncols, rank, world)
    case = build_case(a.workload, hi - lo, lo)
    nlev = case.om.nlev
    # one explicit HIP stream shared by torch (events, collectives' stream
    # dependencies) and the library: torch's default stream is the NULL handle,
    # which the C ABI reads as "create your own"
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    gm = pc.GpuModel(case, stream=tstream.cuda_stream)
    Y, Ya = gm.prognostic_and_aux()
    dY = gm.state(0)
    L, ctx = gm.L, gm.ctx

    # one-off setup, as a user of the ensemble API gets it on the first rhs! call:
    # the library places the written state in HBM by measurement (lh_tune_placement)
    placement = None
    if not a.no_placement_tune:
        b4, af = C.c_float(), C.c_float()
        try:
            F.check(L.lh_tune_placement(ctx, Y, Ya, dY, 0, F.LH_PLACE_MOVE_INPUT, C.byref(b4), C.byref(af)), ctx)
            placement = {"kernel_ms_first_placement": b4.value, "kernel_ms_chosen": af.value}
        except Exception as e:      # noqa: BLE001  (measure the first-come placement then)
            placement = {"error": repr(e)}

    # device scalar for the stable-dt min all-reduce (FT-sized, torch-owned)
    tdt = torch.zeros(1, device="cuda", dtype=torch.float64 if case.dtype == np.float64 else torch.float32)

    # Every rank does the same work at every N (weak scaling): every third evaluation --
    # once per SSPRK33 step -- is lh_rhs_stable_dt, which also leaves this rank's
    # stable-step bound in device memory from the same pass; for N > 1 that one FT
    # value is min-all-reduced over RCCL (no host round trip, no second sweep).
    def rhs_step(i):
        if i % 3 == 2:
            F.check(L.lh_rhs_stable_dt(ctx, 0.0, Y, Ya, dY, 0.5, tdt.data_ptr()), ctx)
            pkg.partition.global_min_dt(tdt)     # no-op for a single rank
        else:
            F.check(L.lh_rhs(ctx, 0.0, Y, Ya, dY), ctx)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        rhs_step(i)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(a.steps):
        rhs_step(i)
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    assert gm.status() == 0, "non-finite tendency during the bench"

    # kernel-only average duration at N = 1-style: HIP events on the launch stream
    # around a back-to-back run of the dominant kernel alone (no collectives)
    kreps = max(20, min(a.steps, 200))
    torch.cuda.synchronize()
    F.check(L.lh_timer_start(ctx), ctx)          # hipEventRecord on the launch stream
    for _ in range(kreps):
        F.check(L.lh_rhs(ctx, 0.0, Y, Ya, dY), ctx)
    ms = C.c_float()
    F.check(L.lh_timer_stop(ctx, C.byref(ms)), ctx)
    kern_ms = ms.value / kreps

    cells = a.ncols * nlev
    bytes_per_cell = WORKLOADS[a.workload][1]
    # HBM bytes per launch from the PMC passes of this same command (rocprofv3
    # cannot run inside the timed process): profiles/pmc_traffic.json, written by
    # tools/gpu_profile.sh + tools/summarize_prof.py; null when sizes differ
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            tr = json.load(fh).get(a.workload)
        if tr and tr["ncols"] == a.ncols and tr["nlev"] == nlev:
            traffic = tr["total_bytes"]
    except (OSError, ValueError, KeyError):
        pass
    # the same bytes as one launch moved by a plain device-to-device copy (hipMemcpyAsync through
    # torch): the "measured copy ceiling" SURVEY 8(d) asks to report beside the spec peak
    copy_gbs = None
    try:
        nb = int(cells * bytes_per_cell / 2)
        src = torch.empty(nb, dtype=torch.uint8, device="cuda")
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            dst.copy_(src)
        c1.record()
        torch.cuda.synchronize()
        copy_gbs = 2 * nb / (c0.elapsed_time(c1) / 10 * 1e-3) / 1e9
        del src, dst
    except RuntimeError:
        pass
    value = world * cells * a.steps / wall
    achieved = cells * bytes_per_cell / (kern_ms * 1e-3) / 1e9
    out = {
        "metric": "column-cell updates/sec (RHS evals)",
        "value": value,
        "unit": "cell-updates/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": WORKLOADS[a.workload][2],
        "data": "synthetic",
        "config": {"workload": f"{a.workload.upper()}: {a.ncols} independent {nlev}-layer "
                               f"{'coupled water+heat' if case.om.model == 2 else 'Richards'} columns per GPU "
                               f"({WORKLOADS[a.workload][2]}), one lh_rhs launch per step",
                   "columns_per_gpu": a.ncols, "levels": nlev,
                   "partition": (f"block over {world} rank(s); " if world > 1 else "single GPU; ") +
                   "every 3rd eval also yields the rank's stable dt (fused)" +
                   (f"; {'RCCL' if a.backend == 'nccl' else a.backend} min all-reduce of that value"
                    if world > 1 else "")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel_ms": kern_ms, "kernel_reps": kreps, "bytes_per_cell": bytes_per_cell,
                     "algorithmic_bytes_per_launch": cells * bytes_per_cell,
                     "measured_copy_gbs": copy_gbs,
                     "frac_of_measured_copy": (achieved / copy_gbs) if copy_gbs else None},
        "placement_tuning": placement,
    }
    if a.stepper:
        # lh_step_ssprk33 (3 RHS evaluations + stage updates per step): as the library runs it
        # (persistent column stepper from 3 steps per call on, unless a Dirichlet face makes
        # the fused stages the better engine for a large ensemble), and with the three
        # fused-stage launches per step forced (LH_TUNE persist=0; stage state placed by measurement)
        def time_steps(ns, calls):
            F.check(L.lh_step_ssprk33(ctx, Y, Ya, 0.0, 1e-3, ns, None), ctx)
            torch.cuda.synchronize()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(calls):
                F.check(L.lh_step_ssprk33(ctx, Y, Ya, 0.0, 1e-3, ns, None), ctx)
            s1.record()
            torch.cuda.synchronize()
            ms = s0.elapsed_time(s1) / (ns * calls)
            return {"ms_per_step": ms, "steps_per_call": ns, "cell_updates_per_s": 3 * cells / (ms * 1e-3)}
        try:    # extra fields only: never at the price of the headline line
            out["ssprk33"] = time_steps(30, 2)            # the library's own choice of engine
            F.check(L.lh_set_tuning(ctx, b"persist=0"), ctx)
            if not a.no_placement_tune:
                F.check(L.lh_tune_placement(ctx, Y, Ya, None, 0, F.LH_PLACE_MOVE_INPUT, None, None), ctx)
            out["ssprk33_fused_stages"] = time_steps(10, 1)
            F.check(L.lh_set_tuning(ctx, b""), ctx)
        except Exception as e:      # noqa: BLE001
            out["ssprk33_error"] = repr(e)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(case, a.cpu_seconds)
    gm.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
