"""
One lattice over several GPUs: slab decomposition of `gp_rfft2_lpdf` + gradients (BASELINE
config 5). The reference is single-process; this is the "lattice larger than one device" row of the scope.

Rank g of G owns rows [g H/G, (g+1) H/G) of y, loc and d/dy (row slab) and packed columns
[g W'/G, (g+1) W'/G) of the half spectrum, W' = W/2 (column slab; packed column 0 carries DC + i Nyquist, so
the W/2 + 1 frequencies split evenly). One evaluation:

    rows forward (local)  ->  all-to-all  ->  column pass (local: column FFT, log-density terms, F/c,
    inverse column FFT)  ->  all-to-all  ->  rows inverse (local)  ;  all-reduce(sum) lpdf, all-reduce(min)

The local passes are the HIP kernels of the single-GPU path (C ABI `gpt_slab_*`); the exchanges are
`torch.distributed` all-to-alls (RCCL over xGMI with the nccl backend; gloo in the CPU tests, where the tests plug in
their own local passes), issued per chunk of rows and asynchronously, so that they overlap the row transforms. Nothing is replicated and nothing is copied between layouts: the row kernels
write / read the all-to-all's own (rank, rows, columns) buffers, and a rank holds only its W/(2G) columns of `cov` /
`d/dcov` (`gp_rfft2_lpdf_vjp_slab_owned`; rank 0 also holds column W/2, which travels with packed column 0).

Supported sizes: power-of-two heights and widths in 256..32768. Lines of more than 8192 (columns) / 16384 (rows)
points, i.e. config 5's 32768 x 32768, take the four-step route of csrc/gpt_long.hip.
"""
from __future__ import annotations

import math
import threading

import torch

from . import fft as _fft


class SlabBuffers:
    """Exchange and output buffers of a slab evaluation, allocated once per shape and handed out again on every later
    call (a sampler evaluates the same lattice thousands of times). With buffers, the tensors an evaluation returns are
    these buffers: they are valid until the next evaluation that uses the same `SlabBuffers`."""

    def __init__(self):
        self._t = {}

    def get(self, name, shape, dtype, device):
        key = (name, tuple(int(s) for s in shape), dtype, str(device))
        t = self._t.get(key)
        if t is None:
            t = self._t[key] = torch.empty(key[1], dtype=dtype, device=device)
        return t


def _empty(stages, name, shape, dtype, device):
    """A buffer from the stages' `SlabBuffers` when they have one, a fresh tensor otherwise."""
    pool = getattr(stages, "buffers", None)
    return pool.get(name, shape, dtype, device) if pool is not None else torch.empty(tuple(shape), dtype=dtype, device=device)


class HipStages:
    """The three local passes on the GPU (device tensors, the context's stream), on the layouts of the exchange: the row
    passes write / read the (ranks, rows, columns per rank) buffers the all-to-all sends and receives, the column pass
    works on the rank's own columns of cov and d/dcov. `buffers`: a `SlabBuffers` to allocate everything once per shape
    (bench.py --slab, sampler loops); None allocates per call."""

    def __init__(self, buffers=None):
        self.buffers = buffers

    def rows_fwd_blocked(self, y_rows, loc_rows, height, width, world, tag=0):
        rows, nc = y_rows.shape[0], width // 2 // world
        send = _empty(self, f"send{tag}", (world, rows, nc), torch.complex128, y_rows.device)
        ctx = _fft.get_context(y_rows.device)
        ctx.call("gpt_slab_rows_fwd_blocked", _fft._p(y_rows), _fft._p(loc_rows), rows, height, width, _fft._p(send), nc)
        return send

    def cols_lpdf_vjp_owned(self, spec_cols, cov_cols, cov_nyquist, height, width, col0, need_cov_grad=True):
        """spec_cols (H, nc) in place; cov_cols (H, nc) = this rank's columns of cov; cov_nyquist (H,) = column W/2, on
        the rank with col0 == 0 only. Returns (lpdf partial, min partial, d/dcov of the columns or None, d/dcov of
        column W/2 or None)."""
        ncols = spec_cols.shape[1]
        dev = spec_cols.device
        lp = _empty(self, "lp", (1,), torch.float64, dev)
        mn = _empty(self, "mn", (1,), torch.float64, dev)
        out_cols = _empty(self, "out_cols", cov_cols.shape, torch.float64, dev) if need_cov_grad else None
        edge = None
        if col0 == 0:
            edge = _empty(self, "edge", (4, height), torch.float64, dev)
            edge[0].copy_(cov_cols[:, 0])
            edge[1].copy_(cov_nyquist)
        ctx = _fft.get_context(dev)
        ctx.call("gpt_slab_cols_lpdf_vjp_owned", _fft._p(spec_cols), ncols, _fft._p(cov_cols), _fft._p(out_cols),
                 cov_cols.shape[1], _fft._p(edge), height, width, col0, ncols, _fft._p(lp), _fft._p(mn))
        out_nyq = None
        if col0 == 0 and need_cov_grad:
            out_cols[:, 0].copy_(edge[2])
            out_nyq = _empty(self, "out_nyq", (height,), torch.float64, dev)
            out_nyq.copy_(edge[3])
        return lp, mn, out_cols, out_nyq

    def rows_inv_blocked(self, recv, height, width, alpha, out=None):
        world, rows, nc = recv.shape
        if out is None:
            out = _empty(self, "grad_rows", (rows, width), torch.float64, recv.device)
        ctx = _fft.get_context(recv.device)
        ctx.call("gpt_slab_rows_inv_blocked", _fft._p(recv), nc, rows, height, width, float(alpha), _fft._p(out))
        return out

    # --- the plain-layout calls (spec[rows][ld], whole-lattice cov): kept for callers with their own exchange
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
        spec = _empty(self, "spec", (height, ld), torch.complex128, y.device)
        lp = _empty(self, "lp", (1,), torch.float64, y.device)
        mn = _empty(self, "mn", (1,), torch.float64, y.device)
        out = _empty(self, "grad_rows", (height, width), torch.float64, y.device)
        ctx = _fft.get_context(y.device)
        ctx.call("gpt_slab_rows_fwd", _fft._p(y), _fft._p(loc), height, height, width, _fft._p(spec), ld)
        ctx.call("gpt_slab_cols_lpdf_vjp", _fft._p(spec), ld, _fft._p(cov), _fft._p(out_cov), height, width, 0,
                 width // 2, _fft._p(lp), _fft._p(mn))
        ctx.call("gpt_slab_rows_inv", _fft._p(spec), ld, height, height, width, float(alpha), _fft._p(out))
        return lp, out, mn


class _Done:
    """Handle of an exchange that has already completed."""

    def wait(self):
        return None


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

    def all_to_all_start(self, send_parts, recv_parts):
        """Asynchronous exchange of one chunk: send_parts[p] goes to rank p, recv_parts[p] is filled from rank p; every
        part is a contiguous tensor (views into larger buffers are fine). Returns a handle; wait() orders the current
        stream behind the exchange. With RCCL the transfer runs on the communicator's own stream, behind what the
        current stream has queued so far, so kernels launched after this call overlap it."""
        if self.dist.get_backend(self.group) == "nccl":
            return self.dist.all_to_all([torch.view_as_real(t) for t in recv_parts],
                                        [torch.view_as_real(t) for t in send_parts], group=self.group, async_op=True)
        got = self.all_to_all(torch.stack(list(send_parts)))  # gloo (CPU tests, rehearsals): synchronous
        for dst, src in zip(recv_parts, got):
            dst.copy_(src)
        return _Done()

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

    def all_to_all_start(self, send_parts, recv_parts):
        recv_parts[0].copy_(send_parts[0])
        return _Done()

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

    def all_to_all_start(self, send_parts, recv_parts):
        s = self.shared
        cuda = send_parts[0].is_cuda
        if cuda:
            torch.cuda.current_stream(send_parts[0].device).synchronize()
        s.slots[self.rank] = list(send_parts)
        s.barrier.wait()
        for dst, parts in zip(recv_parts, list(s.slots)):
            dst.copy_(parts[self.rank])
        if cuda:
            torch.cuda.current_stream(send_parts[0].device).synchronize()
        s.barrier.wait()  # (everyone has copied: the slots and the send buffers may be reused)
        return _Done()

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


def own_columns(cov_rfft2, rank, world):
    """A rank's share of a whole-lattice (H, W/2 + 1) spectrum: (its packed columns as a contiguous (H, W/(2 G)) array,
    column W/2 on rank 0 or None). Column W/2 (Nyquist) rides with packed column 0 (DC + i Nyquist), so rank 0 owns both."""
    wf = cov_rfft2.shape[-1]
    nc = (wf - 1) // world
    cols = cov_rfft2[:, rank * nc:(rank + 1) * nc].contiguous()
    return cols, (cov_rfft2[:, wf - 1].contiguous() if rank == 0 else None)


def exchange_chunks(rows, world, chunks=None):
    """Row chunks per exchange: the requested number, or 4 when there are peers to overlap with; reduced until every
    chunk is an even number of rows (the row kernels transform rows in pairs)."""
    c = (4 if world > 1 else 1) if chunks is None else int(chunks)
    c = max(1, min(c, rows // 2))
    while c > 1 and (rows % c or (rows // c) % 2):
        c -= 1
    return c


def gp_rfft2_lpdf_vjp_slab_owned(y_rows, loc_rows, cov_cols, cov_nyquist, height, width, comm=None, stages=None,
                                 need_cov_grad=True, chunks=None):
    """Value and gradients of `gp_rfft2_lpdf` (fft2.stan:189-192) for one lattice split over `comm.world` ranks, every
    array in the layout its owner needs and nothing replicated.

    y_rows, loc_rows: this rank's rows, (H / G, W). cov_cols: this rank's packed columns of the spectrum,
    (H, W / (2 G)) (`own_columns`); cov_nyquist: column W/2, (H,), on rank 0, None elsewhere. Returns (lpdf, d/dy of this
    rank's rows, min over the lattice of n c / 2, d/dcov of this rank's columns, d/dcov of column W/2 on rank 0);
    d/dloc = -d/dy; lpdf and the minimum are identical on all ranks. The row passes write and read the all-to-all's own
    (G, rows, columns) buffers: no layout copies on either side of the two exchanges.

    The rank's rows are cut into `chunks` groups (default 4 with peers, `exchange_chunks`), each with its own
    asynchronous all-to-all: the exchange of chunk k travels while the row transforms of chunk k + 1 run (forward), and
    the inverse row transforms of chunk k run while chunk k + 1 travels (inverse). Only the column pass waits for
    everything: it needs every row of its columns."""
    comm = comm or SingleComm()
    stages = stages or HipStages()
    world, rank = comm.world, comm.rank
    rows, nc = slab_shapes(height, width, world)
    if tuple(y_rows.shape) != (rows, width) or tuple(loc_rows.shape) != (rows, width):
        raise ValueError(f"rank {rank} must pass its {rows} x {width} rows of y and loc")
    if tuple(cov_cols.shape) != (height, nc) or (rank == 0) != (cov_nyquist is not None):
        raise ValueError(f"rank {rank} must pass its {height} x {nc} columns of cov (rank 0: and column W/2)")
    n = height * width
    nchunk = exchange_chunks(rows, world, chunks)
    y_rows, loc_rows = y_rows.contiguous(), loc_rows.contiguous()
    if nchunk == 1:
        send = stages.rows_fwd_blocked(y_rows, loc_rows, height, width, world)            # (G, rows, nc)
        cols = comm.all_to_all(send).reshape(height, nc)                                  # all rows, my columns
    else:
        rc = rows // nchunk
        cols = _empty(stages, "cols", (height, nc), torch.complex128, y_rows.device)
        cols3 = cols.view(world, rows, nc)                                                # block p: rank p's rows
        pending = []
        tagged = isinstance(stages, HipStages)  # (the tests' own stages keep the three-argument form)
        for k in range(nchunk):
            r0 = k * rc
            send = (stages.rows_fwd_blocked(y_rows[r0:r0 + rc], loc_rows[r0:r0 + rc], height, width, world, tag=k)
                    if tagged else
                    stages.rows_fwd_blocked(y_rows[r0:r0 + rc], loc_rows[r0:r0 + rc], height, width, world))  # (G, rc, nc)
            work = comm.all_to_all_start(list(send.unbind(0)), [cols3[p, r0:r0 + rc] for p in range(world)])
            pending.append((work, send))                                                  # (send stays alive until waited for)
        for work, _ in pending:
            work.wait()
    lp, mn, dcov_cols, dcov_nyq = stages.cols_lpdf_vjp_owned(cols, cov_cols.contiguous(), cov_nyquist, height, width,
                                                             rank * nc, need_cov_grad)
    if nchunk == 1:
        back = comm.all_to_all(cols.view(world, rows, nc))                                # (G, rows, nc): my rows
        grad_rows = stages.rows_inv_blocked(back, height, width, -1.0 / n)
    else:
        cols3 = cols.view(world, rows, nc)
        grad_rows = _empty(stages, "grad_rows", (rows, width), torch.float64, y_rows.device)
        pending = []
        for k in range(nchunk):
            r0 = k * rc
            back = _empty(stages, f"back{k}", (world, rc, nc), torch.complex128, y_rows.device)
            work = comm.all_to_all_start([cols3[p, r0:r0 + rc] for p in range(world)], list(back.unbind(0)))
            pending.append((work, back))
        for k, (work, back) in enumerate(pending):
            work.wait()
            stages.rows_inv_blocked(back, height, width, -1.0 / n, out=grad_rows[k * rc:(k + 1) * rc])
    lp = comm.all_reduce_sum(lp.clone()) - 0.5 * n * math.log(2.0 * math.pi)
    mn = comm.all_reduce_min(mn.clone())
    return lp.reshape(()), grad_rows, mn.reshape(()), dcov_cols, dcov_nyq


def gp_rfft2_lpdf_vjp_slab(y_rows, loc_rows, cov_rfft2, height, width, comm=None, stages=None, out_cov=None, chunks=None):
    """`gp_rfft2_lpdf_vjp_slab_owned` for callers that hold the whole (H, W/2 + 1) spectrum on every rank: the rank's
    columns are cut out of `cov_rfft2` and, if `out_cov` (H, W/2+1) is given, d/dcov is written back into this rank's
    columns of it. Returns (lpdf, d/dy of this rank's rows, min over the lattice of n c / 2)."""
    comm = comm or SingleComm()
    stages = stages or HipStages()
    world, rank = comm.world, comm.rank
    rows, nc = slab_shapes(height, width, world)
    if tuple(cov_rfft2.shape) != (height, width // 2 + 1):
        raise ValueError("cov_rfft2 keeps the lattice's own (height, width / 2 + 1) layout on every rank")
    n = height * width
    if world == 1 and isinstance(stages, HipStages):
        lp, grad_rows, mn = stages.whole_lattice(y_rows.contiguous(), loc_rows.contiguous(), cov_rfft2.contiguous(),
                                                 out_cov, height, width, -1.0 / n)
        return (lp - 0.5 * n * math.log(2.0 * math.pi)).reshape(()), grad_rows, mn.reshape(())
    cov_cols, cov_nyq = own_columns(cov_rfft2, rank, world)
    lp, grad_rows, mn, dcols, dnyq = gp_rfft2_lpdf_vjp_slab_owned(y_rows, loc_rows, cov_cols, cov_nyq, height, width,
                                                                  comm=comm, stages=stages,
                                                                  need_cov_grad=out_cov is not None, chunks=chunks)
    if out_cov is not None:
        out_cov[:, rank * nc:(rank + 1) * nc] = dcols
        if rank == 0:
            out_cov[:, width // 2] = dnyq
    return lp, grad_rows, mn
