"""
N > 1 path on CPU: chains sharded over the ranks of a world_size-2 (and 3, ragged) gloo job, evaluated
with the oracle standing in for the GPU entry point, gathered and compared with the single-process result.
The GPU job (`bench.py --gpus N`) runs the same helpers over RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from gptools_amd import sharding  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(chains, h, w):
    import gp_oracle as oracle

    rng = np.random.default_rng(11)
    cov = oracle.gp_periodic_exp_quad_cov_rfft2(h, w, 1.2, [2.0, 3.0], [float(h), float(w)]) + 1e-2
    loc = rng.normal(size=(h, w))
    y = rng.normal(size=(chains, h, w))
    return torch.from_numpy(y), torch.from_numpy(loc), torch.from_numpy(cov)


def _oracle_batch(y, loc, cov):
    """Stand-in with the signature and output layout of fft.gp_rfft2_lpdf_vjp: (lpdf[B], gy[B,H,W], gc[B,H,Wf])."""
    import gp_oracle as oracle

    lps, gys, gcs = [], [], []
    for k in range(y.shape[0]):
        lp, gy, gc = oracle.gp_rfft2_lpdf_vjp(y[k].numpy(), loc.numpy(), cov.numpy())
        lps.append(lp)
        gys.append(gy)
        gcs.append(gc)
    h, w = loc.shape
    return (torch.tensor(lps, dtype=torch.float64).reshape(-1),
            torch.from_numpy(np.array(gys).reshape(-1, h, w)),
            torch.from_numpy(np.array(gcs).reshape(-1, h, w // 2 + 1)))


def _worker(rank, world, port, chains, h, w, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        y, loc, cov = _inputs(chains, h, w)
        values, (lp_local, gy_local, gc_local) = sharding.sharded_eval(_oracle_batch, y, shared=(loc, cov))
        lo, hi = sharding.shard_bounds(chains, world, rank)
        slowest = sharding.max_over_ranks(1.0 + rank)
        dist.barrier()
        q.put((rank, values.numpy(), lo, hi, gy_local.numpy(), gc_local.numpy(), slowest))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,chains", [(2, 4), (2, 5), (3, 4), (2, 1)])
def test_sharded_chains_match_single_process(world, chains):
    h, w = 6, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, chains, h, w, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    y, loc, cov = _inputs(chains, h, w)
    lp_ref, gy_ref, gc_ref = _oracle_batch(y, loc, cov)
    covered = np.zeros(chains, dtype=int)
    for rank, values, lo, hi, gy_local, gc_local, slowest in results:
        # every rank sees all per-chain values, in chain order, bit-identical to the single-process run
        np.testing.assert_array_equal(values, lp_ref.numpy())
        np.testing.assert_array_equal(gy_local, gy_ref.numpy()[lo:hi])
        np.testing.assert_array_equal(gc_local, gc_ref.numpy()[lo:hi])
        covered[lo:hi] += 1
        assert slowest == float(world)  # max over ranks of 1 + rank
    assert (covered == 1).all()  # each chain evaluated exactly once


def test_shard_bounds_properties():
    for total in range(0, 20):
        for world in range(1, 9):
            bounds = [sharding.shard_bounds(total, world, r) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == total
            for (_, hi), (lo, _) in zip(bounds[:-1], bounds[1:]):
                assert hi == lo
            sizes = sharding.shard_sizes(total, world)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == total
            assert sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 2, 2)
    with pytest.raises(ValueError):
        sharding.shard_bounds(-1, 2, 0)


def test_single_process_passthrough():
    x = torch.arange(6.0).reshape(3, 2)
    assert sharding.world_and_rank() == (1, 0)
    assert sharding.local_shard(x).data_ptr() == x.data_ptr()
    assert sharding.gather_chain_values(x, 3) is x
    assert sharding.max_over_ranks(0.25) == 0.25
    with pytest.raises(ValueError):
        sharding.gather_chain_values(x, 4)
