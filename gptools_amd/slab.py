"""
One lattice over several GPUs: slab decomposition of `gp_rfft2_lpdf` + gradients (BASELINE
config 5). The reference is single-process; this is the "lattice larger than one device" row of the scope.

Rank g of G owns rows [g H/G, (g+1) H/G) of y, loc and d/dy (row slab) and packed columns
[g W'/G, (g+1) W'/G) of the half spectrum, W' = W/2 (column slab; packed column 0 carries DC + i Nyquist, so
the W/2 + 1 frequencies split evenly). One evaluation:

    rows forward (local)  ->  all-to-all  ->  column pass (local: column FFT, log-density terms, F/c,
    inverse column FFT)  ->  all-to-all  ->  rows inverse (local)  ;  all-reduce(sum) lpdf, all-reduce(min)

The local passes are the HIP kernels of the single-GPU path (C ABI `gpt_slab_*`); the exchanges are
`torch.distributed.all_to_all_single` (RCCL over xGMI with the nccl backend; gloo in the CPU tests, where the
tests plug in their own local passes). `cov` / `d/dcov` keep the lattice's own H x (W/2+1) layout on every
rank; a rank touches only its columns (and column W/2 on rank 0).

Supported sizes: power-of-two heights and widths in 256..32768. Lines of more than 8192 (columns) / 16384 (rows)
points, i.e. config 5's 32768 x 32768, take the four-step route of csrc/gpt_long.hip.
"""
from __future__ import annotations

import math
import threading

import torch

from . import fft as _fft


class HipStages:
    """The three local passes on the GPU (device tensors, the context's stream)."""

    def rows_fwd(self, y_rows, loc_rows, height, width):
        rows = y_rows.shape[0]
        spec = torch.empty(rows, width // 2, dtype=torch.complex128, device=y_rows.device)
        ctx = _fft.get_context(y_rows.device)
        ctx.call("gpt_slab_rows_fwd", _fft._p(y_rows), _fft._p(loc_rows), rows, height, width, _fft._p(spec),
                 width // 2)
        return spec

    def cols_lpdf_vjp(self, spec_cols, cov, out_cov, height, width, col0):
        ncols = spec_cols.shape[1]
        lp = torch.empty(1, dtype=torch.float64, device=spec_cols.device)
        mn = torch.empty(1, dtype=torch.float64, device=spec_cols.device)
        ctx = _fft.get_context(spec_cols.device)
        ctx.call("gpt_slab_cols_lpdf_vjp", _fft._p(spec_cols), ncols, _fft._p(cov), _fft._p(out_cov), height, width,
                 col0, ncols, _fft._p(lp), _fft._p(mn))
        return lp, mn

    def rows_inv(self, spec_rows, height, width, alpha):
        rows = spec_rows.shape[0]
        out = torch.empty(rows, width, dtype=torch.float64, device=spec_rows.device)
        ctx = _fft.get_context(spec_rows.device)
        ctx.call("gpt_slab_rows_inv", _fft._p(spec_rows), width // 2, rows, height, width, float(alpha), _fft._p(out))
        return out


    def whole_lattice(self, y, loc, cov, out_cov, height, width, alpha):
        """All three passes on one device (world size 1): no exchange, and the spectrum rows are padded by 8 columns
        so that the column pass does not walk a power-of-two stride."""
        ld = width // 2 + 8
        spec = torch.empty(height, ld, dtype=torch.complex128, device=y.device)
        lp = torch.empty(1, dtype=torch.float64, device=y.device)
        mn = torch.empty(1, dtype=torch.float64, device=y.device)
        out = torch.empty(height, width, dtype=torch.float64, device=y.device)
        ctx = _fft.get_context(y.device)
        ctx.call("gpt_slab_rows_fwd", _fft._p(y), _fft._p(loc), height, height, width, _fft._p(spec), ld)
        ctx.call("gpt_slab_cols_lpdf_vjp", _fft._p(spec), ld, _fft._p(cov), _fft._p(out_cov), height, width, 0,
                 width // 2, _fft._p(lp), _fft._p(mn))
        ctx.call("gpt_slab_rows_inv", _fft._p(spec), ld, height, height, width, float(alpha), _fft._p(out))
        return lp, out, mn


class DistComm:
    """Exchanges over a torch.distributed process group (nccl = RCCL on the GPUs, gloo on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def all_to_all(self, send):  # send[p] goes to rank p; returns recv with recv[p] from rank p
        real = torch.view_as_real(send.contiguous())
        if real.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # rehearsals on a box without RCCL peers: gloo exchanges host tensors only
            host = real.cpu()
            out = torch.empty_like(host)
            self.dist.all_to_all_single(out, host, group=self.group)
            return torch.view_as_complex(out.to(real.device))
        out = torch.empty_like(real)
        self.dist.all_to_all_single(out, real, group=self.group)
        return torch.view_as_complex(out)

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_min(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return t


class SingleComm:
    """World of one: every exchange is the identity."""

    world, rank = 1, 0

    def all_to_all(self, send):
        return send

    def all_reduce_sum(self, t):
        return t

    def all_reduce_min(self, t):
        return t


class ThreadComm:
    """G ranks simulated by G threads of one process (slabs of one lattice on ONE device): used by the GPU
    tests to run the real kernels on every slab shape, and usable to evaluate a lattice slab by slab."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.slots = [None] * world
            self.barrier = threading.Barrier(world)

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world

    def _exchange(self, value):
        s = self.shared
        if value.is_cuda:
            torch.cuda.current_stream(value.device).synchronize()
        s.slots[self.rank] = value
        s.barrier.wait()
        got = list(s.slots)
        s.barrier.wait()
        return got

    def all_to_all(self, send):
        return torch.stack([blk[self.rank] for blk in self._exchange(send)])

    def all_reduce_sum(self, t):
        return torch.stack(self._exchange(t)).sum(dim=0)

    def all_reduce_min(self, t):
        return torch.stack(self._exchange(t)).min(dim=0).values


def slab_shapes(height, width, world):
    """(rows per rank, packed columns per rank); raises if the lattice does not split evenly."""
    wc = width // 2
    if width % 2 or height % world or wc % world or (height // world) % 2:
        raise ValueError(f"a {height} x {width} lattice does not split into {world} even row / column slabs")
    return height // world, wc // world


def gp_rfft2_lpdf_vjp_slab(y_rows, loc_rows, cov_rfft2, height, width, comm=None, stages=None, out_cov=None):
    """Value and gradients of `gp_rfft2_lpdf` (fft2.stan:189-192) for one lattice split over `comm.world` ranks.

    y_rows, loc_rows: this rank's rows, shape (H / G, W). cov_rfft2: (H, W/2 + 1), valid at least in this rank's
    packed columns [g W'/G, (g+1) W'/G) (and in column W/2 on rank 0). Returns (lpdf, d/dy of this rank's rows,
    min over the lattice of n c / 2); `out_cov` (H, W/2+1), if given, receives d/dcov in this rank's columns.
    d/dloc = -d/dy. lpdf and the minimum are identical on all ranks."""
    comm = comm or SingleComm()
    stages = stages or HipStages()
    world, rank = comm.world, comm.rank
    rows, nc = slab_shapes(height, width, world)
    if tuple(y_rows.shape) != (rows, width) or tuple(loc_rows.shape) != (rows, width):
        raise ValueError(f"rank {rank} must pass its {rows} x {width} rows of y and loc")
    if tuple(cov_rfft2.shape) != (height, width // 2 + 1):
        raise ValueError("cov_rfft2 keeps the lattice's own (height, width / 2 + 1) layout on every rank")
    n = height * width
    if world == 1 and isinstance(stages, HipStages):
        lp, grad_rows, mn = stages.whole_lattice(y_rows.contiguous(), loc_rows.contiguous(), cov_rfft2.contiguous(),
                                                 out_cov, height, width, -1.0 / n)
        return (lp - 0.5 * n * math.log(2.0 * math.pi)).reshape(()), grad_rows, mn.reshape(())
    # rows -> packed half spectrum of my rows, cut into the column blocks of the ranks
    spec = stages.rows_fwd(y_rows.contiguous(), loc_rows.contiguous(), height, width)      # (rows, W')
    send = spec.view(rows, world, nc).permute(1, 0, 2).contiguous()                       # (G, rows, nc)
    cols = comm.all_to_all(send).reshape(height, nc)                                      # all rows, my columns
    lp, mn = stages.cols_lpdf_vjp(cols, cov_rfft2.contiguous(), out_cov, height, width, rank * nc)
    back = comm.all_to_all(cols.view(world, rows, nc))                                    # (G, rows, nc): my rows
    spec = back.permute(1, 0, 2).contiguous().view(rows, world * nc)
    grad_rows = stages.rows_inv(spec, height, width, -1.0 / n)
    lp = comm.all_reduce_sum(lp.clone()) - 0.5 * n * math.log(2.0 * math.pi)
    mn = comm.all_reduce_min(mn.clone())
    return lp.reshape(()), grad_rows, mn.reshape(())
