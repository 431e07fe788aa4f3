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

With --gpus N > 1 and no launcher in the environment (no WORLD_SIZE) bench.py starts the N
ranks itself -- a child `python -m torch.distributed.run` spawned BEFORE this process touches the
GPU -- relays rank 0's JSON line and fails loudly if fewer than N ranks ran.

Prints ONE JSON line on rank 0 (see the driver contract).  How to read its numbers:
  value, ms_per_step          the timed loop (K steps between barriers, wall clock, max over ranks)
  ms_per_step_median/_min     per-step HIP-event durations inside that same loop
  roofline.frac               bytes the launches really move x K / wall / 8 TB/s  (= the headline rate)
  roofline.kernel_frac        the dominant kernel alone: bytes moved / its mean HIP-event duration
                              among the timed steps / 8 TB/s
  roofline.bytes_per_cell     SURVEY 8(d)'s contract figure (every prognostic read, every tendency
                              written); bytes_moved_per_cell is smaller when the library knows a
                              plane is zero (no theta_i read, no d theta_i = 0 store)
  roofline.traffic            HBM bytes per launch of the dominant kernel from committed rocprofv3 PMC
                              passes of the SAME command line (profiles/pmc_traffic.json), else null

The synthetic-input generator (pure numpy) and the C-ABI driver (GpuModel) are part of the package
(landhydrology.jl_amd/workloads.py): nothing under tests/ is imported.  The CPU oracle under oracle/
is loaded and executed in the cpu_baseline leg only (tests/test_abi_cpu.py checks both).
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

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 measured copy


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--placement-tune", action="store_true",
                    help="call lh_tune_placement in setup (off by default: the Richards Float64 kernels are "
                         "bound by VALU issue and gain 0-2 %%; the Float32 coupled launch gains ~9 %%, DESIGN.md 4.3)")
    ap.add_argument("--no-placement-tune", action="store_true", help="(default; kept for old command lines)")
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
    ap.add_argument("--no-known-zero", action="store_true",
                    help="LH_TUNE zero=0: read theta_i and clear the d theta_i plane at every launch "
                         "(the traffic of SURVEY 8(d)'s byte contract)")
    ap.add_argument("--no-contract-regime", action="store_true",
                    help="skip the extra block that times the same launch with the contract traffic (zero=0) after "
                         "the headline loop (profiling runs: one kernel instantiation per run)")
    ap.add_argument("--no-step-events", action="store_true",
                    help="no HIP events between the timed steps (kernel statistics then come from a "
                         "separate back-to-back block)")
    return ap.parse_args()


WORKLOADS = {
    # name -> (case name of workloads.make_case, algorithmic bytes per cell-update, dtype tag)
    # BASELINE config 1 as an ensemble: the reference's own single 64-layer column with constant-head
    # (Dirichlet) top and bottom faces, replicated --ncols times
    "c1": ("c1_dirichlet_f64", 32.0, "f64"),
    "c2": ("c2_richards_f64", 32.0, "f64"),
    "c3": ("c3_coupled_f32", 24.0, "f32"),
    "c4": ("c4_richards_f64_128", 32.0, "f64"),
    "c5": ("c5_percol_f64", 32.5, "f64"),
    # SURVEY 8(f)-3 at scale: ice lenses + saturated zones + both conductivity factors, Dirichlet
    # top / free-drainage bottom (coupled, 48 levels); viscosity factor with prescribed T (Richards,
    # 50 levels: one more plane read)
    "f3c32": ("mixed_smooth_f32", 24.0, "f32"),
    "f3c64": ("mixed_smooth_f64", 48.0, "f64"),
    # ... the same ensembles with their columns ordered ice-free first (workloads.ice_sorted_order): waves
    # are then all-ice-free or all-icy instead of mixed
    "f3c32s": ("mixed_smooth_f32_icesorted", 24.0, "f32"),
    "f3c64s": ("mixed_smooth_f64_icesorted", 48.0, "f64"),
    "f3v64": ("richards_viscosity_f64", 40.0, "f64"),
    # ... and with the prescribed temperature level-uniform, as the reference's T_profile(z, t) is:
    # nlev numbers (lh_upload_profile), read from LDS -- no T plane is streamed
    "f3v64p": ("richards_viscosity_profile_f64", 32.0, "f64"),
}


def build_case(workload, ncols, col_offset):
    """Synthetic inputs, generated chunk-wise on the host (counter-based hash, so
    every rank builds exactly its own block of the global ensemble)."""
    import __graft_entry__ as g
    name = WORKLOADS[workload][0]
    return g.load_package().workloads.make_case(name, ncols=ncols, col_offset=col_offset)


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
    sys.path.insert(0, os.path.join(ROOT, "oracle"))     # the checker: this leg only
    import oracle_py as O
    oracle_rhs = lambda c, nthreads: O.rhs(c.om, c.vl, c.ti, c.rhoe, c.T_aux, nthreads=nthreads)
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
    oracle_rhs(sub, threads)            # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle_rhs(sub, threads)
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds or reps >= 2000:
            break
    # the same restatement on ONE thread (the reference has no threading): ~2 s of it
    r1, t1 = 0, time.perf_counter()
    while True:
        oracle_rhs(sub, 1)
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


def planes_streamed(case):
    """(planes read, planes written, NOICE) by one tendency launch of this case: the prognostic
    planes of the model, minus a theta_i plane the library knows to be zero (kernels without
    conductivity factors), and never the identically zero d theta_i."""
    import __graft_entry__ as g
    M = g.load_package().case_model
    om = case.om
    factors = bool(om.cf.viscosity_kind or om.cf.impedance_kind)
    noice = (case.ti is None or not case.ti.any()) and not factors
    if om.model == M.MODEL_RICHARDS:
        nr = 1 + (0 if noice else 1) + (1 if (om.cf.viscosity_kind and not case.aux_profile) else 0)   # a level-uniform T is no plane
        nw = 1
    elif om.model == M.MODEL_HEAT:
        nr, nw = 2 + (0 if noice else 1), 1
    else:
        nr, nw = 2 + (0 if noice else 1), 2
    return nr, nw, noice


def spawn_ranks(a):
    """--gpus N without a launcher: start the N ranks as a child torchrun (this process has not
    touched the GPU and never will), relay rank 0's line, fail if anything is missing."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    other = [ln for ln in r.stdout.splitlines() if not ln.startswith("{")]
    if other:
        print("\n".join(other), file=sys.stderr)
    if r.returncode != 0 or not lines:
        print(f"bench.py: the {a.gpus}-rank child run failed (exit code {r.returncode})", file=sys.stderr)
        return r.returncode or 1
    line = json.loads(lines[-1])
    if line.get("n_gpus") != a.gpus:
        print(f"bench.py: asked for {a.gpus} ranks, the line reports {line.get('n_gpus')}", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {a.gpus}: refusing to report a line "
                         f"for a rank count that is not the one asked for")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and world > ndev:
        raise SystemExit(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node shows {ndev} "
                         f"(use --backend gloo to rehearse the multi-process path on fewer)")
    torch.cuda.set_device(local_rank % max(1, ndev))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import __graft_entry__ as g
    pkg = g.load_package()
    pc = pkg.workloads   # the synthetic cases and the C-ABI driver (nothing under tests/ is imported)
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
    L, ctx = gm.L, gm.ctx
    if a.no_known_zero:
        F.check(L.lh_set_tuning(ctx, b"zero=0"), ctx)
    Y, Ya = gm.prognostic_and_aux()     # an all-zero theta_i is a fill: the library knows it is zero
    dY = gm.state(0)
    nr, nw, noice = planes_streamed(case)
    if a.no_known_zero:
        nr, nw, noice = nr + (1 if noice else 0), nw + (1 if case.om.model != 1 else 0), False

    # The one collective of the path: the min all-reduce of the ranks' stable-step bounds.  Over
    # RCCL it lives BEHIND the C ABI (lh_comm_init: lh_rhs_stable_dt then delivers the global
    # minimum, ncclAllReduce on the context's stream); torch.distributed only ships the 128-byte
    # id, and carries the collective itself only in the gloo rehearsal (ranks sharing one GPU
    # cannot form an RCCL communicator).
    native_comm = world > 1 and a.backend == "nccl"
    comm_note = None
    if native_comm:
        # every rank must take the same route: agree on the outcome of lh_comm_init before using it
        ok = 1
        try:
            pkg.partition.attach_native_comm(ctx, rank, world)
        except Exception as e:      # noqa: BLE001
            ok, comm_note = 0, repr(e)
            print(f"bench.py rank {rank}: lh_comm_init failed ({e!r}); falling back to torch.distributed "
                  f"for the min all-reduce if any rank failed", file=sys.stderr)
        flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if ok:
                F.check(L.lh_comm_destroy(ctx), ctx)
            native_comm = False
            comm_note = comm_note or "another rank failed to attach the native communicator"

    # optional one-off setup (--placement-tune): the library places the written state in HBM by
    # measurement (lh_tune_placement); the default line is measured on the first-come placement
    placement = None
    if a.placement_tune:
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
    # value is min-all-reduced (no host round trip, no second sweep).
    def rhs_step(i):
        if i % 3 == 2:
            F.check(L.lh_rhs_stable_dt(ctx, 0.0, Y, Ya, dY, 0.5, tdt.data_ptr()), ctx)
            if not native_comm:
                pkg.partition.global_min_dt(tdt)     # torch path (gloo rehearsal); no-op for one rank
        else:
            F.check(L.lh_rhs(ctx, 0.0, Y, Ya, dY), ctx)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    cells = (hi - lo) * nlev
    esize = np.dtype(case.dtype).itemsize
    bytes_per_cell = WORKLOADS[a.workload][1]                       # SURVEY 8(d) contract
    percol_extra = bytes_per_cell - round(bytes_per_cell / esize) * esize   # C5's per-column block
    moved_per_cell = (nr + nw) * esize + percol_extra
    bytes_moved = cells * moved_per_cell

    # Ancillary measurements FIRST (order: stream probe, contract-traffic regime, then the W warm-up
    # steps and the K timed steps): they need the same resident state, and the ~90 launches they issue
    # bring the device to its sustained clock before the headline loop (a box starts at idle clocks:
    # the first tens of launches of a short run are 10-20 % slower, see per_step_ms).

    # the ceiling of this access pattern on these very planes: the library's no-arithmetic probe
    # (reads up to nr planes of Y, writes nw planes of dY; it reports its own plane counts)
    probe = None
    try:
        F.check(L.lh_synchronize(ctx), ctx)
        y_planes = {0: [0, 1], 1: [2], 2: [0, 2, 1]}[case.om.model]      # theta_i last
        d_planes = {0: [0, 1], 1: [2], 2: [0, 2, 1]}[case.om.model]
        rp, wp = y_planes[:nr], d_planes[:nw]
        pm, wm = sum(1 << v for v in rp), sum(1 << v for v in wp)
        pms = C.c_float()
        F.check(L.lh_stream_probe(ctx, Y, pm, dY, wm, 40, C.byref(pms)), ctx)
        probe = {"gbs": cells * esize * (len(rp) + len(wp)) / (pms.value * 1e-3) / 1e9, "ms": pms.value,
                 "planes_read": len(rp), "planes_written": len(wp)}
    except Exception as e:      # noqa: BLE001
        probe = {"error": repr(e)}

    # The SURVEY 8(d) byte contract as traffic: the same launch with the known-zero planes switched
    # off (theta_i read, d theta_i = 0 stored: LH_TUNE zero=0), timed in THIS process -- both byte
    # regimes from one run, one box, one clock.
    contract = None
    if not a.no_known_zero and not a.no_contract_regime:
        try:
            F.check(L.lh_set_tuning(ctx, b"zero=0"), ctx)
            for _ in range(5):
                F.check(L.lh_rhs(ctx, 0.0, Y, Ya, dY), ctx)
            nk = 60     # (a fixed block, whatever --steps: a measurement of its own)
            F.check(L.lh_timer_start(ctx), ctx)
            for _ in range(nk):
                F.check(L.lh_rhs(ctx, 0.0, Y, Ya, dY), ctx)
            cms = C.c_float()
            F.check(L.lh_timer_stop(ctx, C.byref(cms)), ctx)
            c_ms = cms.value / nk
            c_bytes = cells * bytes_per_cell
            contract = {"kernel_ms": c_ms, "launches_timed": nk, "bytes_per_launch": c_bytes,
                        "achieved": c_bytes / (c_ms * 1e-3) / 1e9, "frac": c_bytes / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "is": "lh_rhs with LH_TUNE zero=0 (every prognostic plane read, every tendency plane "
                              "written: the bytes of SURVEY 8(d)'s contract really moved), HIP events around a "
                              "back-to-back block in this same process; the memset of the d theta_i plane is part of it"}
        except Exception as e:      # noqa: BLE001
            contract = {"error": repr(e)}
        finally:
            F.check(L.lh_set_tuning(ctx, b""), ctx)

    for i in range(a.warmup):
        rhs_step(i)
    barrier()
    step_events = not a.no_step_events
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)] if step_events else []
    t0 = time.perf_counter()
    for i in range(a.steps):
        if step_events:
            evs[i].record()              # on tstream = the stream the launches go to
        rhs_step(i)
    if step_events:
        evs[a.steps].record()
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    assert gm.status() == 0, "non-finite tendency during the bench"
    dt_seen = float(tdt.item()) if a.steps >= 3 or a.warmup >= 3 else None

    # per-step durations inside the timed loop (HIP events on the launch stream)
    stats = {}
    if step_events:
        d = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps)])
        plain = d[np.arange(a.steps) % 3 != 2]
        fused = d[np.arange(a.steps) % 3 == 2]
        stats = {"ms_per_step_median": float(np.median(d)), "ms_per_step_min": float(d.min()),
                 "ms_per_step_events_mean": float(d.mean()),
                 # the first steps of the timed loop one by one (every third is the fused-dt launch):
                 # what a clock ramp or a slow box looks like from inside the run
                 "per_step_ms": [round(float(x), 4) for x in d[:32]]}
        kern_ms, kern_n = float(plain.mean()), int(plain.size)
        kern_med, kern_min = float(np.median(plain)), float(plain.min())
        fused_ms = float(fused.mean()) if fused.size else None
        kern_src = "per-step HIP events inside the timed loop (the lh_rhs steps)"
    else:
        kern_n = max(20, min(a.steps, 200))
        torch.cuda.synchronize()
        F.check(L.lh_timer_start(ctx), ctx)
        for _ in range(kern_n):
            F.check(L.lh_rhs(ctx, 0.0, Y, Ya, dY), ctx)
        ms = C.c_float()
        F.check(L.lh_timer_stop(ctx, C.byref(ms)), ctx)
        kern_ms, kern_med, kern_min, fused_ms = ms.value / kern_n, None, None, None
        kern_src = "HIP events around a separate back-to-back block of lh_rhs launches (after the timed loop)"

    ms_per_step = wall / a.steps * 1e3
    achieved = bytes_moved / (ms_per_step * 1e-3) / 1e9
    kernel_gbs = bytes_moved / (kern_ms * 1e-3) / 1e9

    # N > 1: what every rank measured, and what the one collective costs on its own (the in-stream
    # min all-reduce of one FT value, timed by HIP events around a block of lh_allreduce_min calls)
    ranks = None
    if world > 1:
        mine = torch.tensor([kern_ms, fused_ms if fused_ms is not None else float("nan"), ms_per_step],
                            device="cuda", dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ar_ms = None
        if native_comm:
            for _ in range(5):
                F.check(L.lh_allreduce_min(ctx, tdt.data_ptr()), ctx)
            F.check(L.lh_timer_start(ctx), ctx)
            for _ in range(50):
                F.check(L.lh_allreduce_min(ctx, tdt.data_ptr()), ctx)
            ams = C.c_float()
            F.check(L.lh_timer_stop(ctx, C.byref(ams)), ctx)
            ar_ms = ams.value / 50
        ranks = {"kernel_ms": [float(x[0]) for x in allr], "fused_dt_kernel_ms": [float(x[1]) for x in allr],
                 "ms_per_step": [float(x[2]) for x in allr],
                 "allreduce_min_ms": ar_ms,
                 "allreduce_share_of_a_3_eval_step": (ar_ms / (3 * kern_ms)) if ar_ms else None,
                 "is": "per-rank HIP-event means of the timed loop (rank order); allreduce_min_ms: one "
                       "ncclAllReduce(min, 1 value) on the launch stream, back-to-back block on rank 0"}

    # HBM bytes per launch from committed PMC passes of this same command line (rocprofv3 cannot
    # run inside the timed process): profiles/pmc_traffic.json, written by tools/gpu_profile.sh
    traffic, traffic_source, traffic_command, valu_per_cell, valu_busy = None, None, None, None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            tr = json.load(fh).get(a.workload + ("_nozero" if a.no_known_zero else ""))
        if tr and tr["ncols"] == a.ncols and tr["nlev"] == nlev and tr.get("known_zero", True) != a.no_known_zero:
            traffic, traffic_source, traffic_command = tr["total_bytes"], tr.get("source"), tr.get("command")
            valu_per_cell = tr.get("valu_per_cell")
            valu_busy = tr.get("valu_busy")
        # ... and of the contract-traffic regime (the "<workload>_nozero" PMC passes)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            trz = json.load(fh).get(a.workload + "_nozero")
        if contract and "kernel_ms" in contract and trz and trz["ncols"] == a.ncols and trz["nlev"] == nlev:
            contract["traffic"] = trz["total_bytes"]
            contract["traffic_source"] = trz.get("source")
            contract["valu_per_cell"] = trz.get("valu_per_cell")
    except (OSError, ValueError, KeyError):
        pass

    value = world * cells * a.steps / wall
    out = {
        "metric": "column-cell updates/sec (RHS evals)",
        "value": value,
        "unit": "cell-updates/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": ms_per_step,
        **stats,
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
                   ((f"; min all-reduce of that value: " +
                     ("RCCL inside the library (lh_comm_init, ncclAllReduce on the launch stream)" if native_comm
                      else f"torch.distributed {a.backend}" + (" (rehearsal on a shared GPU)" if a.backend == "gloo"
                                                                else f" (native communicator unavailable: {comm_note})"))) if world > 1 else "")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "frac_is": "bytes_moved_per_launch / ms_per_step / peak (the timed loop, wall clock)",
                     "traffic": traffic, "traffic_source": traffic_source,
                     "traffic_is": "PROFILED, not measured by this run: per-launch HBM bytes of the same kernel on the "
                                   "same workload/size from committed rocprofv3 PMC passes (null when they differ)",
                     "traffic_command": traffic_command,
                     "bytes_per_cell": bytes_per_cell, "bytes_moved_per_cell": moved_per_cell,
                     "planes_read": nr, "planes_written": nw, "theta_i_known_zero": bool(noice),
                     "bytes_moved_per_launch": bytes_moved,
                     "algorithmic_bytes_per_launch": cells * bytes_per_cell,
                     "kernel_ms": kern_ms, "kernel_ms_median": kern_med, "kernel_ms_min": kern_min,
                     "kernel_launches_timed": kern_n, "kernel_ms_is": kern_src,
                     "kernel_achieved": kernel_gbs, "kernel_frac": kernel_gbs / HBM_PEAK_GBS,
                     "kernel_frac_contract_bytes": cells * bytes_per_cell / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "fused_dt_kernel_ms": fused_ms,
                     "stream_probe": probe,
                     "kernel_frac_of_stream_probe": (kernel_gbs / probe["gbs"]) if probe and "gbs" in probe else None,
                     "valu_per_cell": valu_per_cell,
                     "valu_busy_profiled": valu_busy,
                     "valu_is": "PROFILED (same committed PMC passes as traffic): VALU instructions per cell-update, and the "
                                "fraction of the launch the SIMDs spend issuing them (4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / "
                                "GRBM_GUI_ACTIVE per XCD) -- the Float64 kernels are bound by this, not by HBM"},
        "order": "stream probe (43 launches), contract-traffic regime (65 launches), W warm-up steps, K timed steps, "
                 "then the optional stepper blocks",
        "contract_traffic": contract,
        "ranks": ranks,
        "stable_dt_seen": dt_seen,
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
            eng = L.lh_step_engine(ctx, ns, 0)
            return {"ms_per_step": ms, "steps_per_call": ns, "cell_updates_per_s": 3 * cells / (ms * 1e-3),
                    "engine": {F.LH_ENGINE_FUSED_STAGES: "fused stages", F.LH_ENGINE_COLUMN_STEPPER: "column stepper"}.get(eng, eng)}
        try:    # extra fields only: never at the price of the headline line
            out["ssprk33"] = time_steps(30, 2)            # the library's own choice of engine
            F.check(L.lh_set_tuning(ctx, b"persist=0"), ctx)
            if a.placement_tune:
                F.check(L.lh_tune_placement(ctx, Y, Ya, None, 0, F.LH_PLACE_MOVE_INPUT, None, None), ctx)
            out["ssprk33_fused_stages"] = time_steps(10, 1)
            F.check(L.lh_set_tuning(ctx, b""), ctx)
        except Exception as e:      # noqa: BLE001
            out["ssprk33_error"] = repr(e)
    if a.stepper:
        # The adaptive time loop as a host would run it, nothing leaving the device: per step
        # lh_rhs_stable_dt (tendency + this rank's step bound; with a communicator the library's RCCL
        # min follows on the same stream) -> lh_step_ssprk33_device_dt (dt read from device memory).
        # At N > 1 this is the one place the collective sits on the critical path of real stepping.
        try:
            def adaptive(nsteps):
                for _ in range(nsteps):
                    F.check(L.lh_rhs_stable_dt(ctx, 0.0, Y, Ya, dY, 0.2, tdt.data_ptr()), ctx)
                    if not native_comm:
                        pkg.partition.global_min_dt(tdt)
                    F.check(L.lh_step_ssprk33_device_dt(ctx, Y, Ya, 0.0, tdt.data_ptr(), None), ctx)
            adaptive(3)
            barrier()
            t1 = time.perf_counter()
            adaptive(20)
            barrier()
            w1 = time.perf_counter() - t1
            if world > 1:
                tt = torch.tensor([w1], device="cuda", dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                w1 = float(tt.item())
            out["adaptive_ssprk33_per_step_calls"] = {
                "ms_per_step": w1 / 20 * 1e3, "steps": 20, "courant": 0.2, "last_dt": float(tdt.item()),
                "is": "lh_rhs_stable_dt (+ min all-reduce over ranks) + lh_step_ssprk33_device_dt per step "
                      "(four evaluations of f), wall clock, max over ranks"}
            # the same loop as ONE call: f(Y) evaluated once per step (stage 2 formed from (Y, f(Y)))
            tel = torch.zeros_like(tdt)
            F.check(L.lh_step_ssprk33_adaptive(ctx, Y, Ya, 0.0, 0.2, 0.0, 3, tdt.data_ptr(), tel.data_ptr()), ctx)
            barrier()
            t2 = time.perf_counter()
            F.check(L.lh_step_ssprk33_adaptive(ctx, Y, Ya, 0.0, 0.2, 0.0, 20, tdt.data_ptr(), tel.data_ptr()), ctx)
            barrier()
            w2 = time.perf_counter() - t2
            if world > 1:
                tt = torch.tensor([w2], device="cuda", dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                w2 = float(tt.item())
            out["adaptive_ssprk33"] = {
                "ms_per_step": w2 / 20 * 1e3, "steps": 20, "courant": 0.2, "last_dt": float(tdt.item()),
                "elapsed": float(tel.item()),
                "cell_updates_per_s": 3 * cells / (w2 / 20),
                "is": "lh_step_ssprk33_adaptive: per step the tendency + step bound in one launch, the min all-reduce "
                      "over ranks inside the library, stages 2 and 3 (three evaluations of f); wall clock, max over ranks"
                      + ("" if native_comm or world == 1 else
                         "; NO communicator attached in this rehearsal: every rank steps with its local bound")}
        except Exception as e:      # noqa: BLE001
            out["adaptive_ssprk33_error"] = repr(e)
    if native_comm:
        F.check(L.lh_comm_destroy(ctx), ctx)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(case, a.cpu_seconds)
    gm.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
