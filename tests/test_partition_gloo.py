"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups.

Columns are independent, so a block-partitioned run must reproduce the
single-rank result bit for bit; the only exchange is the stable-dt `min`
all-reduce, which is order-independent.  The compute here is the CPU oracle (the
HIP path needs a GPU); what is under test is the host-side partition logic the
GPU ranks use: block ranges, per-rank synthetic blocks of one global ensemble
(counter-based hash), and the collective."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as g
import parity_cases as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_range_covers_everything():
    part = g.load_package().partition
    for N, W in ((8, 1), (8, 2), (10, 3), (1_000_003, 8), (8_000_000, 8)):
        ranges = [part.block_range(N, r, W) for r in range(W)]
        assert ranges[0][0] == 0 and ranges[-1][1] == N
        for (a, b), (c, d) in zip(ranges, ranges[1:]):
            assert b == c and b > a
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        part.block_range(3, 0, 4)
    with pytest.raises(ValueError):
        part.block_range(8, 8, 8)


def _worker(rank, world, port, N, name, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    part = g.load_package().partition
    lo, hi = part.block_range(N, rank, world)
    case = pc.make_case(name, ncols=hi - lo, col_offset=lo)       # this rank's block
    d = pc.run_oracle_rhs(case)
    dt_local = pc.O.stable_dt(case.om, case.vl, case.ti, case.rhoe, 0.5)
    t = torch.tensor([dt_local], dtype=torch.float64)
    part.global_min_dt(t)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), lo=lo, hi=hi, dt=t.numpy(), dt_local=dt_local,
             **{k: v for k, v in d.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "c2_richards_f64"), (3, "c3_coupled_f32"),
                                        (2, "c5_percol_f64")])
def test_block_partition_reproduces_single_rank(world, name, tmp_path):
    N = 301
    port = 29600 + (os.getpid() % 300) + world
    mp.start_processes(_worker, args=(world, port, N, name, str(tmp_path)), nprocs=world,
                       join=True, start_method="spawn")
    whole = pc.make_case(name, ncols=N, col_offset=0)
    ref = pc.run_oracle_rhs(whole)
    dt_ref = pc.O.stable_dt(whole.om, whole.vl, whole.ti, whole.rhoe, 0.5)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert parts[0]["lo"] == 0 and parts[-1]["hi"] == N
    for k in ref:
        cat = np.concatenate([p[k] for p in parts], axis=0)
        assert np.array_equal(cat, ref[k]), k                 # bitwise
    dts = [float(p["dt"][0]) for p in parts]
    assert all(d == dts[0] for d in dts)                      # every rank holds the global min
    assert dts[0] == min(float(p["dt_local"]) for p in parts) == dt_ref
