"""The multi-GPU path THROUGH THE HIP LIBRARY: block partition + the one collective.

A GPU box for these tests has ONE MI355X, so what can run here is
  * the native RCCL communicator with one rank (lh_comm_init .. ncclAllReduce .. lh_comm_destroy),
  * two CONTEXTS in one process over disjoint column blocks of one ensemble, stepping adaptively
    through lh_rhs_stable_dt -> min of the two device words -> lh_step_ssprk33_device_dt, checked
    BITWISE against the single-context run of the whole ensemble,
  * bench.py starting its own ranks (--gpus 2 --backend gloo: two processes sharing the GPU; an
    RCCL communicator cannot be formed by ranks on one device, so the collective rides gloo).
The 8-GPU form (one rank per GPU, RCCL min inside the library) is what the driver runs.
"""
import ctypes as C
import dataclasses
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import case_model as M
import parity_cases as pc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _slice_case(case, lo, hi):
    sl = lambda a: None if a is None else np.ascontiguousarray(a[lo:hi])
    om = case.om
    if om.percol or om.percol_bc:
        om = dataclasses.replace(om, percol={k: v[lo:hi] for k, v in om.percol.items()},
                                 percol_bc={k: v[lo:hi] for k, v in om.percol_bc.items()})
    return dataclasses.replace(case, om=om, ncols=hi - lo, vl=sl(case.vl), ti=sl(case.ti),
                               rhoe=sl(case.rhoe), T_aux=sl(case.T_aux))


def _state_arrays(g, Y):
    F, m = g.F, g.case.om.model
    out = {}
    if m != M.MODEL_HEAT:
        out["vl"] = g.download(Y, F.LH_VAR_VARTHETA_L)
    if m != M.MODEL_RICHARDS:
        out["rhoe"] = g.download(Y, F.LH_VAR_RHOE_INT)
    return out


def test_block_range_matches_the_host_helper():
    import __graft_entry__ as ge
    pkg = ge.load_package()
    L = pkg._ffi.lib()
    for N, W in ((8, 1), (10, 3), (1_000_003, 8), (8_000_000, 8)):
        for r in range(W):
            lo, hi = C.c_int64(), C.c_int64()
            assert L.lh_block_range(N, r, W, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == pkg.partition.block_range(N, r, W)
    lo, hi = C.c_int64(), C.c_int64()
    assert L.lh_block_range(3, 0, 4, C.byref(lo), C.byref(hi)) == pkg._ffi.LH_EINVAL
    assert L.lh_block_range(8, 8, 8, C.byref(lo), C.byref(hi)) == pkg._ffi.LH_EINVAL


@pytest.mark.parametrize("name", ["c2_richards_f64", "c3_coupled_f32"])
def test_native_rccl_communicator_single_rank(name):
    """lh_comm_* end to end on the one GPU there is: the all-reduce really goes through
    ncclAllReduce (a 1-rank communicator), and leaves the local minimum untouched."""
    import torch
    case = pc.make_case(name, ncols=700)
    tdtype = torch.float64 if case.dtype == np.float64 else torch.float32
    with pc.GpuModel(case) as g:
        F, L, ctx = g.F, g.L, g.ctx
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        t0 = torch.zeros(1, device="cuda", dtype=tdtype)
        F.check(L.lh_rhs_stable_dt(ctx, 0.0, Y, Ya, dY, 0.5, t0.data_ptr()), ctx)
        F.check(L.lh_synchronize(ctx), ctx)
        base = g.tendencies(dY)
        r, n = C.c_int32(-1), C.c_int32(-1)
        F.check(L.lh_comm_info(ctx, C.byref(r), C.byref(n)), ctx)
        assert (r.value, n.value) == (0, 1)
        ident = (C.c_ubyte * F.LH_COMM_ID_BYTES)()
        F.check(L.lh_comm_unique_id(ident), None)
        assert any(ident)
        F.check(L.lh_comm_init(ctx, 0, 1, ident), ctx)
        with pytest.raises(F.ModelError):          # a second communicator is refused
            F.check(L.lh_comm_init(ctx, 0, 1, ident), ctx)
        with pytest.raises(F.ModelError):
            F.check(L.lh_comm_init(ctx, 2, 2, ident), ctx)
        t1 = torch.zeros(1, device="cuda", dtype=tdtype)
        F.check(L.lh_rhs_stable_dt(ctx, 0.0, Y, Ya, dY, 0.5, t1.data_ptr()), ctx)
        host = C.c_double()
        F.check(L.lh_stable_dt(ctx, Y, Ya, 0.5, C.byref(host)), ctx)
        F.check(L.lh_allreduce_min(ctx, t1.data_ptr()), ctx)
        F.check(L.lh_synchronize(ctx), ctx)
        assert t1.item() == t0.item() and t0.item() > 0
        assert abs(host.value - t0.item()) <= (1e-6 if case.dtype == np.float64 else 2e-4) * t0.item()
        again = g.tendencies(dY)
        for k in base:
            assert np.array_equal(base[k], again[k])
        F.check(L.lh_comm_destroy(ctx), ctx)
        F.check(L.lh_comm_destroy(ctx), ctx)       # idempotent
        F.check(L.lh_comm_info(ctx, C.byref(r), C.byref(n)), ctx)
        assert (r.value, n.value) == (0, 1)


@pytest.mark.parametrize("name,split", [("c2_richards_f64", (1000, 2)), ("c5_percol_f64", (999, 2)),
                                        ("c3_coupled_f32", (1001, 3))])
def test_contexts_over_column_blocks_step_adaptively_like_one(name, split):
    """Ranks' contexts in one process: every block runs fused-dt -> [min over blocks] ->
    lh_step_ssprk33_device_dt; the concatenation is bitwise the single-context run."""
    import torch
    N, W = split
    whole = pc.make_case(name, ncols=N)
    tdtype = torch.float64 if whole.dtype == np.float64 else torch.float32
    nsteps = 6

    def run(cases):
        gs = [pc.GpuModel(c) for c in cases]
        try:
            st = []
            for g in gs:
                Y, Ya = g.prognostic_and_aux()
                st.append((Y, Ya, g.state(0), torch.zeros(1, device="cuda", dtype=tdtype)))
            dts = []
            for _ in range(nsteps):
                for g, (Y, Ya, dY, t) in zip(gs, st):
                    g.F.check(g.L.lh_rhs_stable_dt(g.ctx, 0.0, Y, Ya, dY, 0.2, t.data_ptr()), g.ctx)
                for g in gs:
                    g.F.check(g.L.lh_synchronize(g.ctx), g.ctx)
                # the collective: the minimum over the blocks, left in every block's device word
                gmin = torch.min(torch.cat([t for *_, t in st]))
                for *_, t in st:
                    t.copy_(gmin)
                torch.cuda.synchronize()
                dts.append(float(gmin.item()))
                for g, (Y, Ya, dY, t) in zip(gs, st):
                    g.F.check(g.L.lh_step_ssprk33_device_dt(g.ctx, Y, Ya, 0.0, t.data_ptr(), None), g.ctx)
            outs = [_state_arrays(g, Y) for g, (Y, *_rest) in zip(gs, st)]
            for g in gs:
                assert g.status() == 0
            return outs, dts
        finally:
            for g in gs:
                g.close()

    import __graft_entry__ as ge
    part = ge.load_package().partition
    blocks = [_slice_case(whole, *part.block_range(N, r, W)) for r in range(W)]
    (ref,), dts_ref = run([whole])
    outs, dts = run(blocks)
    assert dts == dts_ref and all(d > 0 for d in dts) and len(set(dts)) > 1   # adaptive, and identical
    for k in ref:
        cat = np.concatenate([o[k] for o in outs], axis=0)
        assert np.array_equal(cat, ref[k]), (name, k)
        assert np.max(np.abs(ref[k] - getattr(whole, "vl" if k == "vl" else "rhoe"))) > 0   # it moved


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment must run TWO ranks and say
    so (the gloo rehearsal: both ranks share this box's one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--steps", "9", "--warmup", "3", "--ncols", "20000", "--no-cpu-baseline", "--no-stepper"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 9 and line["value"] > 0
    assert "gloo" in line["config"]["partition"]
    assert line["stable_dt_seen"] and line["stable_dt_seen"] > 0
    # a rank count the launcher does not deliver is refused, not reported
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                         "--ncols", "2000", "--no-cpu-baseline", "--no-stepper"],
                        env=env2, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r2.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (r2.stderr + r2.stdout)


def test_bench_two_ranks_over_native_rccl_when_two_gpus_are_visible():
    """The real thing -- two ranks, two GPUs, the library's own RCCL communicator carrying the min
    all-reduce -- wherever the box has a second GPU (the builder's box has one: skipped there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the RCCL path needs one GPU per rank")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "9", "--warmup", "3",
                        "--ncols", "200000", "--no-cpu-baseline", "--no-stepper"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "RCCL inside the library" in line["config"]["partition"], line["config"]["partition"]
    assert line["stable_dt_seen"] and line["stable_dt_seen"] > 0


@pytest.mark.parametrize("name", ["c2_richards_f64", "c3_coupled_f32", "mixed_smooth_f64", "c5_percol_f64",
                                  "heat_dirichlet_f64", "richards_viscosity_f64", "mixed_smooth_f32", "single_cell_f64"])
def test_adaptive_stepper_is_bitwise_the_per_step_calls(name):
    """lh_step_ssprk33_adaptive (f(Y) + step bound in one launch, stage 2 formed from (Y, f(Y)) in
    registers: three evaluations of f per step) against the sequence it replaces, lh_rhs_stable_dt +
    lh_step_ssprk33_device_dt per step (four): the same dt values, the same state, bit for bit, and
    the elapsed time it accumulates on the device is the sum of those dt."""
    import torch
    case = pc.make_case(name, ncols=None if name != "c5_percol_f64" else 700)
    tdtype = torch.float64 if case.dtype == np.float64 else torch.float32
    nsteps, courant = 4, 0.3
    with pc.GpuModel(case) as g:
        F, L, ctx = g.F, g.L, g.ctx
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        t = torch.zeros(1, device="cuda", dtype=tdtype)
        dts = []
        for _ in range(nsteps):
            F.check(L.lh_rhs_stable_dt(ctx, 0.0, Y, Ya, dY, courant, t.data_ptr()), ctx)
            F.check(L.lh_step_ssprk33_device_dt(ctx, Y, Ya, 0.0, t.data_ptr(), None), ctx)
            F.check(L.lh_synchronize(ctx), ctx)
            dts.append(float(t.item()))
        ref = _state_arrays(g, Y)
    with pc.GpuModel(case) as g:
        F, L, ctx = g.F, g.L, g.ctx
        Y, Ya = g.prognostic_and_aux()
        t = torch.zeros(1, device="cuda", dtype=tdtype)
        el = torch.zeros(1, device="cuda", dtype=tdtype)
        F.check(L.lh_step_ssprk33_adaptive(ctx, Y, Ya, 0.0, courant, 0.0, nsteps, t.data_ptr(), el.data_ptr()), ctx)
        F.check(L.lh_synchronize(ctx), ctx)
        got = _state_arrays(g, Y)
        assert float(t.item()) == dts[-1]
        acc = np.zeros(1, dtype=case.dtype)
        for d in dts:
            acc += case.dtype(d)
        assert float(el.item()) == float(acc[0])
        assert g.status() == 0
        # a cap below the bound is honoured
        cap = 0.25 * dts[-1]
        F.check(L.lh_step_ssprk33_adaptive(ctx, Y, Ya, 0.0, courant, cap, 1, t.data_ptr(), None), ctx)
        F.check(L.lh_synchronize(ctx), ctx)
        assert float(t.item()) == float(case.dtype(cap))
    for k in ref:
        assert np.array_equal(ref[k], got[k]), (name, k)
    assert all(d > 0 for d in dts)     # (a heat-only column with prescribed water has a constant bound)


def test_adaptive_stepper_is_bitwise_the_per_step_calls_at_full_size():
    """The same identity on BASELINE config 2 at its full size (1e6 columns: the one-lane-per-column
    launches, not the level-segmented ones of the small cases above)."""
    import torch

    import bench
    case = bench.build_case("c2", 1_000_000, 0)
    with pc.GpuModel(case) as g:
        F, L, ctx = g.F, g.L, g.ctx
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        t = torch.zeros(1, device="cuda", dtype=torch.float64)
        for _ in range(2):
            F.check(L.lh_rhs_stable_dt(ctx, 0.0, Y, Ya, dY, 0.3, t.data_ptr()), ctx)
            F.check(L.lh_step_ssprk33_device_dt(ctx, Y, Ya, 0.0, t.data_ptr(), None), ctx)
        ref = g.download(Y, F.LH_VAR_VARTHETA_L)
        dt_ref = float(t.item())
        Y2, _ = g.prognostic_and_aux()
        F.check(L.lh_step_ssprk33_adaptive(ctx, Y2, Ya, 0.0, 0.3, 0.0, 2, t.data_ptr(), None), ctx)
        got = g.download(Y2, F.LH_VAR_VARTHETA_L)
        assert float(t.item()) == dt_ref and dt_ref > 0
    assert np.array_equal(ref, got) and np.max(np.abs(ref - case.vl)) > 0
