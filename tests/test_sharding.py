"""Multi-GPU path on CPU: utterances are block-split over ranks with no data-path collective; the optional result
gather is an all_gather.  world_size 2 over gloo, with the ORACLE standing in for the device kernels (tests only)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from misonet_amd.pipeline import shard_range, run_sharded


def test_shard_range_partitions():
    for n in (0, 1, 7, 16, 128, 129):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(128, 3, 8) == (48, 64)                   # BASELINE config 5: 8 ranks x 16 utterances
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _inputs(u):
    r = np.random.default_rng(50 + u)
    return (r.standard_normal((6, 8, 129)) + 1j * r.standard_normal((6, 8, 129))).astype(np.complex64)


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from misonet_amd import weights as W
    from oracle import miso_oracle
    torch.set_num_threads(1)
    sd = W.make_state_dict(W.miso1_spec(), 0)

    def process(lo, hi):
        if hi == lo:
            return torch.zeros((0, 2, 8, 129), dtype=torch.complex64)
        x = torch.from_numpy(np.stack([_inputs(u) for u in range(lo, hi)]))
        return torch.cat([miso_oracle.miso1_forward(x[i:i + 1], sd) for i in range(hi - lo)])

    out = run_sharded(process, n_items, rank, world, gather=True)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # the bench's max-over-ranks timing reduction
    if rank == 0:
        q.put((out.numpy(), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [3, 4])
def test_two_rank_shard_and_gather(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + n_items + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, tmax = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0
    from misonet_amd import weights as W
    from oracle import miso_oracle
    sd = W.make_state_dict(W.miso1_spec(), 0)
    assert got.shape == (n_items, 2, 8, 129)
    for u in range(n_items):
        ref = miso_oracle.miso1_forward(torch.from_numpy(_inputs(u)[None]), sd).numpy()[0]
        assert np.linalg.norm(got[u] - ref) / np.linalg.norm(ref) < 1e-5


def test_bench_shards_configs4_as_8_x_16():
    """BASELINE configs[4] = batch 128 over 8 ranks: bench.py gives rank r the utterances shard_range(128, r, 8) = [16 r,
    16 r + 16) (global index = the synthetic generator's seed), and names the workload configs[4] on the rank-0 line."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    seen = []
    for r in range(8):
        u = bench.rank_utterances(r, 8, 16)
        assert u == list(range(*shard_range(128, r, 8))) == list(range(16 * r, 16 * r + 16))
        seen += u
    assert seen == list(range(128))
    assert bench.rank_utterances(1, 2, 16) == list(range(16, 32))
    w8 = bench.workload_name(8, 16)
    assert w8.startswith("BASELINE configs[4]") and "batch 128" in w8 and "8 x 16" in w8
    assert bench.workload_name(1, 16).startswith("BASELINE configs[3]")
    assert "configs[4] sharding at 2 ranks" in bench.workload_name(2, 16)


def _worker8(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)

    def process(lo, hi):          # stand-in for the device pipeline: one complex "spectrogram" per utterance, value = its index
        return torch.stack([torch.full((2, 3, 5), complex(u, -u), dtype=torch.complex64) for u in range(lo, hi)])

    out = run_sharded(process, 128, rank, world, gather=True)
    t = torch.tensor([10.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((out.numpy(), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_shard_and_gather_order():
    """The 8-rank launch of configs[4] over gloo: every rank processes ITS 16 of 128 utterances, the all_gather returns
    them in global order, the elapsed-time reduction is the max over ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    got, tmax = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 17.0
    assert got.shape == (128, 2, 3, 5)
    assert np.array_equal(got[:, 0, 0, 0], np.arange(128) * (1 - 1j))
