"""
Slab decomposition of one lattice over several ranks on CPU: world_size-2 and -4 gloo jobs run
gptools_amd.slab's exchange protocol (all-to-all there and back, all-reduces) with NumPy local passes and
must reproduce the oracle's single-process value and gradients. The GPU test (test_gpu_slab.py) runs the
same protocol with the HIP passes.
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
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gptools_amd import slab  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(h, w):
    import gp_oracle as oracle

    rng = np.random.default_rng(5)
    cov = oracle.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.1, [2.0, 3.0], [float(h), float(w)]) + 1e-2
    return rng.normal(size=(h, w)), rng.normal(size=(h, w)), cov


def _worker(rank, world, port, h, w, q, chunks=None):
    import torch.distributed as dist
    from slab_stages_np import NumpyStages

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        y, loc, cov = _inputs(h, w)
        rows = h // world
        sl = slice(rank * rows, (rank + 1) * rows)
        out_cov = torch.full((h, w // 2 + 1), np.nan, dtype=torch.float64)
        lp, gy, mn = slab.gp_rfft2_lpdf_vjp_slab(torch.from_numpy(y[sl].copy()), torch.from_numpy(loc[sl].copy()),
                                                 torch.from_numpy(cov), h, w, comm=slab.DistComm(),
                                                 stages=NumpyStages(), out_cov=out_cov, chunks=chunks)
        dist.barrier()
        q.put((rank, float(lp), gy.numpy(), float(mn), out_cov.numpy()))
    finally:
        dist.destroy_process_group()


# chunks: None = the default (up to 4 row chunks per exchange, each its own asynchronous all-to-all), 1 = one exchange
@pytest.mark.parametrize("world,h,w,chunks", [(2, 8, 8, None), (2, 12, 20, None), (4, 16, 24, None), (2, 12, 20, 1),
                                              (2, 32, 16, 4)])
def test_slab_protocol_matches_oracle(world, h, w, chunks):
    import gp_oracle as oracle

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, h, w, q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    y, loc, cov = _inputs(h, w)
    lp_ref, gy_ref, gc_ref = oracle.gp_rfft2_lpdf_vjp(y, loc, cov)
    rows, nc = slab.slab_shapes(h, w, world)
    gc = np.full_like(gc_ref, np.nan)
    for rank, lp, gy, mn, oc in results:
        np.testing.assert_allclose(lp, lp_ref, rtol=1e-12)
        np.testing.assert_allclose(gy, gy_ref[rank * rows:(rank + 1) * rows], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(mn, 0.5 * h * w * cov.min(), rtol=1e-14)
        cols = list(range(rank * nc, (rank + 1) * nc)) + ([w // 2] if rank == 0 else [])
        gc[:, cols] = oc[:, cols]
        untouched = np.setdiff1d(np.arange(w // 2 + 1), cols)
        assert np.isnan(oc[:, untouched]).all()  # a rank writes only its own columns of d/dcov
    np.testing.assert_allclose(gc, gc_ref, rtol=1e-10, atol=1e-13)


def test_slab_single_rank_and_shape_errors():
    import gp_oracle as oracle
    from slab_stages_np import NumpyStages

    h, w = 6, 10
    y, loc, cov = _inputs(h, w)
    lp, gy, mn = slab.gp_rfft2_lpdf_vjp_slab(torch.from_numpy(y), torch.from_numpy(loc), torch.from_numpy(cov), h, w,
                                             stages=NumpyStages())
    lp_ref, gy_ref, _ = oracle.gp_rfft2_lpdf_vjp(y, loc, cov)
    np.testing.assert_allclose(float(lp), lp_ref, rtol=1e-12)
    np.testing.assert_allclose(gy.numpy(), gy_ref, rtol=1e-10, atol=1e-13)
    with pytest.raises(ValueError):
        slab.slab_shapes(6, 10, 4)  # 6 rows do not split over 4 ranks
    with pytest.raises(ValueError):
        slab.slab_shapes(8, 10, 4)  # 5 packed columns do not split either
    with pytest.raises(ValueError):
        slab.gp_rfft2_lpdf_vjp_slab(torch.from_numpy(y[:2]), torch.from_numpy(loc), torch.from_numpy(cov), h, w,
                                    stages=NumpyStages())


def test_exchange_chunks():
    assert slab.exchange_chunks(512, 8) == 4          # default with peers
    assert slab.exchange_chunks(512, 1) == 1          # nothing to overlap with
    assert slab.exchange_chunks(6, 2) == 3            # 4 does not divide 6 rows into even chunks
    assert slab.exchange_chunks(4, 4) == 2
    assert slab.exchange_chunks(2, 2) == 1
    assert slab.exchange_chunks(4096, 8, chunks=16) == 16
    assert slab.exchange_chunks(10, 2, chunks=4) == 1  # 10 rows: only one even chunk works (5 is odd)
