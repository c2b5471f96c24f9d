#!/usr/bin/env python3
"""
Benchmark of the hot path: batched gp_rfft2_lpdf + gradients (fp64) on synthetic 4096 x 4096 lattices.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--size S]

A "step" is one call of the fused value+gradient entry point (gpt_gp_rfft2_lpdf_vjp through the C ABI)
over a batch of B lattices that are already resident in HBM. With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank evaluates its own B lattices -- independent chains, no
data-path collective -- so the scaling is weak and `value` is the whole-job rate.

Prints ONE JSON line (rank 0). `roofline` prices the three-pass pipeline against the algorithmic bytes of
the operator (24 n + 16 H (W/2+1) bytes per evaluation) using pass durations
measured with HIP events inside the library (gpt_profile_begin/end), and lists every pass with its own
compulsory bytes so the numbers can be held against the rocprofv3 summary under profiles/.
`cpu_baseline` times the CPU oracle (oracle/gp_oracle.py, NumPy restatement of the reference) on the host
cores of the same box (one core, and one process per core of the box's CPU share), rank 0 and N == 1 only, on a
bounded sample. `python bench.py --gpus N` without a launcher starts the N ranks itself (torch.distributed.run as a
child process); under a launcher (WORLD_SIZE set) it is one of the ranks. `n_gpus` is the size of the process group.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)


def synthetic(fft, size, batch, device, seed, per_lattice_cov=False):
    """SURVEY 8(d) recipe: Matern-3/2 spectrum + nugget, y = gp_inv_rfft2(z, loc, c) with z ~ N(0, 1). The spectrum
    is shared by the batch, or (the recipe's second run) one per lattice from per-lattice sigma_b, length_scale_b."""
    h = w = size
    if per_lattice_cov:
        b = torch.arange(batch, dtype=torch.float64, device=device)
        sigma = 1.3 + 0.05 * b
        ls = torch.stack([8.0 + 0.25 * b, 12.0 + 0.25 * b], dim=-1)
        cov = fft.gp_periodic_matern_cov_rfft2(1.5, h, w, sigma, ls, [float(h), float(w)]) + 1e-3
    else:
        cov = fft.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.3, [8.0, 12.0], [float(h), float(w)]) + 1e-3
    g = torch.Generator(device=device)
    g.manual_seed(7)
    loc = torch.randn(h, w, dtype=torch.float64, device=device, generator=g)
    g.manual_seed(20250321 + seed)
    z = torch.randn(batch, h, w, dtype=torch.float64, device=device, generator=g)
    y = fft.gp_inv_rfft2(z, loc, cov)
    del z
    return y, loc, cov


def pmc_traffic(size):
    """Bytes per evaluation measured with rocprofv3 --pmc (scripts/pmc_passes.sh) and committed under profiles/."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            rec = json.load(fh)
        return rec["per_eval_bytes"] if rec.get("shape") == [size, size] else None
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(size, budget_s=10.0):
    """The CPU oracle (oracle/gp_oracle.py, NumPy restatement of the reference) on this box's host cores: one core, and
    one process per core of the box's CPU share. Runs as a child process (oracle/cpu_baseline.py: no torch, no GPU)."""
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--size", str(size), "--budget", str(budget_s)]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240, check=True)
        return json.loads(out.stdout.strip().splitlines()[-1])
    except (subprocess.SubprocessError, ValueError, IndexError) as exc:
        return {"value": None, "unit": "evals/s", "cores": 0, "kind": "port", "sample": f"cpu baseline failed: {exc}"}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU, RCCL) as a child job of
    torch.distributed.run -- before this process has touched the GPU -- relay rank 0's JSON line, and exit with the
    job's status."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {n}-rank job failed (exit status {proc.returncode})", file=sys.stderr)
        sys.exit(proc.returncode or 1)
    print(line)
    sys.exit(0)


def matern_cov_columns(nu, h, w, sigma, length_scale, col0, ncols, device):
    """Synthetic input of the slab bench: columns col0 .. col0 + ncols - 1 of gp_periodic_matern_cov_rfft2(nu, h, w, sigma,
    length_scale, period = (h, w)) (fft2.stan:228-247; closed form SURVEY 8(a)) as an (h, ncols) array, without the rest of
    the spectrum. tests/test_gpu_slab.py holds it to the library's kernel."""
    import math

    ki = torch.arange(h, dtype=torch.float64, device=device)
    ki = torch.minimum(ki, h - ki)
    j = torch.arange(col0, col0 + ncols, dtype=torch.float64, device=device)
    r1, r2 = length_scale[0] / h, length_scale[1] / w
    arg = 1.0 + (2.0 / nu) * math.pi ** 2 * ((ki * r1)[:, None] ** 2 + (j * r2)[None, :] ** 2)
    amp = sigma ** 2 * h * w * 4.0 * (math.pi / (2.0 * nu)) * math.gamma(nu + 1.0) / math.gamma(nu) * r1 * r2
    return amp * arg ** (-(nu + 1.0))


def slab_bench(args, fft, device, world, rank, distributed):
    """One lattice over all ranks; a step is one evaluation of gp_rfft2_lpdf + gradients."""
    import torch.distributed as dist

    from gptools_amd import sharding, slab

    h = w = args.size
    rows, nc = slab.slab_shapes(h, w, world)
    comm = slab.DistComm() if distributed else slab.SingleComm()
    stages = slab.HipStages(buffers=slab.SlabBuffers())  # exchange and output buffers allocated once, not per step
    g = torch.Generator(device=device)
    g.manual_seed(11 + rank)
    loc = torch.randn(rows, w, dtype=torch.float64, device=device, generator=g)
    y = loc + torch.randn(rows, w, dtype=torch.float64, device=device, generator=g)
    if world > 1 or os.environ.get("GPT_BENCH_FORCE_NCCL"):
        # a rank holds only its own columns of the spectrum (and of d/dcov), generated in that layout:
        # the whole H x (W/2 + 1) array (4 GiB at 32768^2) never exists on any rank
        cov_cols = matern_cov_columns(1.5, h, w, 1.3, (8.0, 12.0), rank * nc, nc, device) + 1e-3
        cov_nyq = (matern_cov_columns(1.5, h, w, 1.3, (8.0, 12.0), w // 2, 1, device)[:, 0] + 1e-3) if rank == 0 else None

        def step():
            lp_, gy_, mn_, _, _ = slab.gp_rfft2_lpdf_vjp_slab_owned(y, loc, cov_cols, cov_nyq, h, w, comm=comm,
                                                                    stages=stages, chunks=args.chunks or None)
            return lp_, gy_, mn_
    else:
        cov = fft.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.3, [8.0, 12.0], [float(h), float(w)]) + 1e-3
        out_cov = torch.empty(h, w // 2 + 1, dtype=torch.float64, device=device)

        def step():
            return slab.gp_rfft2_lpdf_vjp_slab(y, loc, cov, h, w, comm=comm, stages=stages, out_cov=out_cov)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lp, _, mn = step()
    torch.cuda.synchronize()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0, device=device)
    barrier()
    assert bool(torch.isfinite(lp)) and float(mn) > 0.0
    if rank == 0:
        n = h * w
        b_alg = 24 * n + 16 * h * (w // 2 + 1)
        print(json.dumps({
            "metric": f"gp_rfft2_lpdf+grad evals/sec (fp64, {h}^2, one lattice slab-decomposed)",
            "value": args.steps / elapsed, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"one synthetic {h}x{w} lattice, {rows} rows and {nc} packed columns per rank, "
                                   "two all-to-all exchanges per evaluation",
                       "parallelism": f"slab decomposition over {world} GPU(s)",
                       "exchange": (f"torch.distributed {dist.get_backend()} process group" if distributed
                                    else "none (one rank, no process group)")},
            "roofline": {"bound": "hbm", "achieved": b_alg * args.steps / elapsed / 1e9 / world,
                         "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": b_alg * args.steps / elapsed / world / HBM_PEAK, "traffic": None,
                         "note": "per GPU, wall clock, exchanges and layout copies included"}}))
    if distributed:
        dist.destroy_process_group()


def dry_run(args, world, rank):
    """CPU rehearsal of the multi-rank plumbing (tests/test_bench_launch.py): launcher, rendezvous (gloo), barrier,
    max-over-ranks clock and the JSON line -- no GPU, no kernels, and flagged so that it cannot pass for a measurement."""
    import torch.distributed as dist

    from gptools_amd import sharding

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    if os.environ.get("GPT_BENCH_FAIL_RANK") == str(rank):  # (test hook: a rank that dies takes the job down)
        sys.exit(3)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0)
    n_ranks = dist.get_world_size() if world > 1 else 1
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "gp_rfft2_lpdf+grad evals/sec (fp64, 4096^2)", "value": None, "unit": "evals/s",
                          "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f64", "data": "none", "dry_run": True,
                          "config": {"workload": "DRY RUN: no kernels were launched", "global_batch": n_ranks * args.batch}}))
    if world > 1:
        dist.destroy_process_group()


def self_check(fft, y, loc, cov, out, per_lattice_cov):
    """Outside the timed region: the last lattice of the timed batch against a single-lattice call of the same entry
    point (a different launch geometry, tests/test_gpu_headline_batch.py holds both to the oracle)."""
    lp, gy, gc, _ = out
    i = y.shape[0] - 1
    lp1, gy1, gc1 = fft.gp_rfft2_lpdf_vjp(y[i], loc, cov[i] if per_lattice_cov else cov, check=False)

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max())

    errs = {"lpdf": rel(lp[i:i + 1], lp1.reshape(1)), "grad_y": rel(gy[i], gy1), "grad_cov": rel(gc[i], gc1)}
    assert max(errs.values()) < 1e-12, f"timed batch disagrees with the single-lattice call: {errs}"
    return errs


def profiled_passes(ctx, step, B, prof_steps):
    """Average duration of passes A / B / C per lattice over `prof_steps` extra, untimed steps (the library's per-pass HIP
    events, gpt_profile_begin / gpt_profile_end) -> ([sec per evaluation] * 4, lattices per launch)."""
    pass_ms = (ctypes.c_double * 4)()
    launches = ctypes.c_int64(0)
    ctx.lib.gpt_profile_begin(ctx.handle)
    for _ in range(prof_steps):
        step()
    ctx.lib.gpt_profile_end(ctx.handle, pass_ms, ctypes.byref(launches))
    if launches.value <= 0:
        return None, None
    return [1e-3 * pass_ms[k] / (B * prof_steps) for k in range(4)], B * prof_steps / launches.value


def event_timed(step, steps):
    """Device seconds per step between two HIP events on the launch stream."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return 1e-3 * e0.elapsed_time(e1) / steps


def fused_kernel_extra(fft, ctx, y, loc, steps):
    """NOT the headline: the kernel-fused entry point (gpt_gp_rfft2_lpdf_kernel_vjp) on the same
    lattices, every lattice its own kernel parameters; its own algorithmic bytes: read y, loc, write d/dy = 24 n (no
    spectrum, no d/dcov in memory). Outputs preallocated, as in the headline loop."""
    B, h, w = y.shape
    sigma = torch.linspace(1.25, 1.35, B, dtype=torch.float64, device=y.device)
    ls = torch.tensor([8.0, 12.0], dtype=torch.float64, device=y.device).expand(B, 2)
    per = torch.tensor([float(h), float(w)], dtype=torch.float64, device=y.device).expand(B, 2)
    out = (torch.empty(B, dtype=torch.float64, device=y.device), torch.empty_like(y),
           torch.empty(B, 5, dtype=torch.float64, device=y.device), torch.empty(B, dtype=torch.float64, device=y.device))

    def step():
        return fft.gp_rfft2_lpdf_kernel_vjp(y, loc, "matern", 1.5, sigma, ls, per, nugget=1e-3, check=False, out=out)

    for _ in range(2):
        step()
    sec = event_timed(step, steps) / B
    per_eval, lpl = profiled_passes(ctx, step, B, min(steps, 3))
    b_alg = 24 * h * w
    rec = {"entry_point": "gpt_gp_rfft2_lpdf_kernel_vjp (Matern-3/2 + nugget evaluated inside the column pass, "
                          "d/d(sigma, length_scale, period) reduced there; per-lattice parameters)",
           "evals_per_s": 1.0 / sec, "algorithmic_bytes_per_eval": b_alg, "achieved_GBps": b_alg / sec / 1e9,
           "frac": b_alg / sec / HBM_PEAK, "frac_on_headline_bytes": (24 * h * w + 16 * h * (w // 2 + 1)) / sec / HBM_PEAK,
           "finite": bool(torch.isfinite(out[0]).all() and torch.isfinite(out[2]).all())}
    if per_eval:
        rec["pass_us_per_eval"] = {"row_fwd": 1e6 * per_eval[0], "col": 1e6 * per_eval[1], "row_inv": 1e6 * per_eval[3]}
    return rec


def generic_path_extra(fft, device):
    """NOT the headline: shapes that are neither small nor powers of two (the generic three-pass path, DESIGN 5.3: in-place
    LDS transforms, lean kernels for 5-smooth sizes) and a launch of many large one-workgroup lattices, which is re-routed
    to it. gp_rfft2_lpdf + both gradients, device-resident, outputs preallocated, HIP events around the launches."""
    rec = {}
    g = torch.Generator(device=device)
    g.manual_seed(11)
    for h, w, B in ((1000, 1000, 8), (3000, 3000, 2), (192, 192, 64), (72, 72, 4096), (256, 256, 1)):
        y = torch.randn(B, h, w, dtype=torch.float64, device=device, generator=g)
        loc = torch.zeros(h, w, dtype=torch.float64, device=device)
        cov = torch.exp(0.3 * torch.randn(B, h, w // 2 + 1, dtype=torch.float64, device=device, generator=g))
        out = (torch.empty(B, dtype=torch.float64, device=device), torch.empty_like(y), torch.empty_like(cov),
               torch.empty(B, dtype=torch.float64, device=device))

        def lpdf():
            fft.gp_rfft2_lpdf_vjp(y, loc, cov, check=False, out=out)

        for _ in range(3):
            lpdf()
        sec = event_timed(lpdf, 10)
        b_alg = 24 * h * w + 16 * h * (w // 2 + 1)
        rec[f"{h}x{w}"] = {"lattices_per_launch": B, "path": fft.path_for_shape(h, w), "us_per_launch": 1e6 * sec,
                           "us_per_lattice": 1e6 * sec / B, "evals_per_s": B / sec, "achieved_GBps": B * b_alg / sec / 1e9,
                           "frac": B * b_alg / sec / HBM_PEAK, "finite": bool(torch.isfinite(out[0]).all())}
        del y, cov, out
    rec["72x72"]["note"] = "path 0 by shape; launches of >= 384 such lattices run on the generic path (gpt_api.hip)"
    rec["256x256"]["note"] = "path 1 by shape; launches of <= 2^23 points at 256 rows run on the generic path's kernels (latency)"
    torch.cuda.empty_cache()
    return rec


def small_lattice_extra(fft, device):
    """NOT the headline: BASELINE configs[0] / [2], the shapes of the reference's own docs models, as a sampler farm
    would run them -- 4096 chains of the 35 x 60 trees lattice per launch (one workgroup per lattice), value +
    both gradients, every chain its own spectrum; and the trees model density itself (field from z, masked
    negative-binomial counts on the 25 x 50 corner, white-noise prior, reverse mode) from kernel parameters.
    Device-resident, outputs preallocated where the shim allows, HIP events around the launches."""
    h, w, B = 35, 60, 4096
    g = torch.Generator(device=device)
    g.manual_seed(7)
    y = torch.randn(B, h, w, dtype=torch.float64, device=device, generator=g)
    loc = torch.zeros(h, w, dtype=torch.float64, device=device)
    cov = torch.exp(0.3 * torch.randn(B, h, w // 2 + 1, dtype=torch.float64, device=device, generator=g))
    out = (torch.empty(B, dtype=torch.float64, device=device), torch.empty_like(y), torch.empty_like(cov),
           torch.empty(B, dtype=torch.float64, device=device))

    def lpdf():
        fft.gp_rfft2_lpdf_vjp(y, loc, cov, check=False, out=out)

    for _ in range(3):
        lpdf()
    sec = event_timed(lpdf, 20)
    b_alg = 24 * h * w + 16 * h * (w // 2 + 1)
    rec = {"gp_rfft2_lpdf_vjp": {"what": f"{B} chains of {h} x {w} per launch, one spectrum per chain", "us_per_launch": 1e6 * sec,
                                 "evals_per_s": B / sec, "achieved_GBps": B * b_alg / sec / 1e9, "frac": B * b_alg / sec / HBM_PEAK,
                                 "finite": bool(torch.isfinite(out[0]).all())}}
    obs = torch.poisson(torch.full((25, 50), 2.0, dtype=torch.float64, device=device), generator=g)
    obs[torch.rand(25, 50, device=device, generator=g) < 0.1] = -1.0
    theta = torch.full((B,), 1.7, dtype=torch.float64, device=device)
    sigma = torch.linspace(0.9, 1.3, B, dtype=torch.float64, device=device)
    ls = torch.tensor([2.5, 3.0], dtype=torch.float64, device=device).expand(B, 2)
    per = torch.tensor([float(h), float(w)], dtype=torch.float64, device=device).expand(B, 2)
    half = torch.full((h, w), 0.5, dtype=torch.float64, device=device)
    res = {}

    def dens():
        res["r"] = fft.gp_inv_rfft2_obs_lpdf_kernel_vjp("neg_binomial_2_log", y, half, "matern", 1.5, sigma, ls, per, obs, theta,
                                                        nugget=1e-9, check=False)

    for _ in range(3):
        dens()
    sec = event_timed(dens, 20)
    rec["gp_inv_rfft2_obs_lpdf_kernel_vjp"] = {"what": f"trees model density, {B} chains per launch, Matern-3/2 parameters per chain",
                                               "us_per_launch": 1e6 * sec, "evals_per_s": B / sec,
                                               "finite": bool(torch.isfinite(res["r"]["lp"]).all())}
    return rec


def unshared_leg(fft, ctx, size, B, device, steps):
    """The same step with NOTHING shared: every lattice its own loc and its own cov, so that every byte of B_alg is a
    distinct byte that has to come from HBM (in the headline configuration loc and cov are read about once per launch of
    16 lattices). Same B_alg, same entry point."""
    h = w = size
    y, loc, cov = synthetic(fft, size, B, device, seed=101, per_lattice_cov=True)
    g = torch.Generator(device=device)
    g.manual_seed(13)
    dl = torch.randn(B, h, w, dtype=torch.float64, device=device, generator=g)
    y += dl  # y_b - loc_b is the same draw as before
    locb = loc.unsqueeze(0) + dl
    del dl
    out = (torch.empty(B, dtype=torch.float64, device=device), torch.empty_like(y), torch.empty_like(cov),
           torch.empty(B, dtype=torch.float64, device=device))

    def step():
        fft.gp_rfft2_lpdf_vjp(y, locb, cov, check=False, out=out)

    for _ in range(2):
        step()
    sec = event_timed(step, steps) / B
    per_eval, _ = profiled_passes(ctx, step, B, min(steps, 3))
    b_alg = 24 * h * w + 16 * h * (w // 2 + 1)
    ok = bool(torch.isfinite(out[0]).all()) and float(out[3].min()) >= 0.0
    if not os.environ.get("GPT_BENCH_NOCHECK"):  # last lattice of the batch against a single-lattice call
        lp1, gy1, gc1 = fft.gp_rfft2_lpdf_vjp(y[B - 1], locb[B - 1], cov[B - 1], check=False)
        err = max(float((out[1][B - 1] - gy1).abs().max() / gy1.abs().max()),
                  float((out[2][B - 1] - gc1).abs().max() / gc1.abs().max()),
                  float((out[0][B - 1] - lp1).abs() / lp1.abs()))
        ok = ok and err < 1e-12
    rec = {"what": f"{B} lattices per step, one loc AND one cov per lattice (nothing shared)", "evals_per_s": 1.0 / sec,
           "achieved": b_alg / sec / 1e9, "frac": b_alg / sec / HBM_PEAK, "valid": ok}
    if per_eval:
        rec["pass_us_per_eval"] = {"row_fwd": 1e6 * per_eval[0], "col": 1e6 * per_eval[1], "row_inv": 1e6 * per_eval[3]}
    return rec


def three_pass_ceiling(args):
    """The same harness on gptools_amd/libvar_traffic.so (make traffic): passes A / B / C with every load and store of
    the real kernels and no arithmetic or LDS traffic between them -- what the three-pass structure could reach on this
    box if the transforms were free. Runs as a child process (another build of the library cannot be loaded beside the
    real one); its results are garbage by construction and never checked."""
    lib = os.path.join(ROOT, "gptools_amd", "libvar_traffic.so")
    if not os.path.exists(lib):
        return {"value": None, "note": "gptools_amd/libvar_traffic.so not built (make -C gptools_amd/csrc traffic)"}
    env = dict(os.environ, GPT_LIB=lib, GPT_BENCH_NOCHECK="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch",
           str(args.batch), "--size", str(args.size), "--no-cpu-baseline", "--no-extra", "--no-ceiling"]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, check=True, env=env)
        r = json.loads(out.stdout.strip().splitlines()[-1])
    except (subprocess.SubprocessError, ValueError, IndexError) as exc:
        return {"value": None, "note": f"traffic-only run failed: {exc}"}
    rec = {"what": "traffic-only build of passes A/B/C (loads and stores of the real kernels, no butterflies, no LDS "
                   "exchange, no elementwise arithmetic), same harness, same workload",
           "evals_per_s": r["value"], "frac": r["roofline"]["frac"], "achieved": r["roofline"]["achieved"]}
    if "kernels" in r["roofline"]:
        k = {x["name"].split()[0]: 1e3 * x["avg_ms_per_eval"] for x in r["roofline"]["kernels"]}
        rec["pass_us_per_eval"] = {"row_fwd": k.get("row_fwd"), "col": k.get("col"), "row_inv": k.get("row_inv")}
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="lattices per GPU per step")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--slab", action="store_true",
                    help="not the headline: ONE lattice of --size split over all ranks (row / column slabs, "
                         "all-to-all over RCCL), strong scaling")
    ap.add_argument("--chunks", type=int, default=0, help="--slab: row chunks per exchange (0: 4 with peers, else 1)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-pass HIP-event timing")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra (non-headline) kernel-fused measurement")
    ap.add_argument("--no-ceiling", action="store_true",
                    help="skip roofline.unshared and roofline.three_pass_ceiling (A/B runs of tuning variants)")
    ap.add_argument("--per-lattice-cov", action="store_true",
                    help="not the headline: one spectrum per lattice (per-lattice sigma, length_scale) instead of a "
                         "spectrum shared by the batch")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank plumbing only (gloo, no GPU, no kernels, value null)")
    args = ap.parse_args()

    # `python bench.py --gpus N` (no launcher): this process only starts the N ranks and relays their line. Nothing
    # here has touched the GPU yet (importing torch does not).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.dry_run:
        return dry_run(args, world, rank)
    # GPT_BENCH_FORCE_NCCL=1: a one-rank job still goes through an RCCL process group (the only way to put the nccl
    # branch -- group set-up, the max-over-ranks all-reduce, the slab exchanges -- on hardware with one GPU)
    distributed = world > 1 or bool(os.environ.get("GPT_BENCH_FORCE_NCCL"))
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # rehearsal on a one-GPU box: GPT_BENCH_REHEARSAL=1 puts every rank on device 0 and uses gloo
        if os.environ.get("GPT_BENCH_REHEARSAL"):
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()  # what the job really is
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", torch.cuda.current_device())

    from gptools_amd import fft, sharding

    if args.slab:
        return slab_bench(args, fft, device, world, rank, distributed)
    size, B = args.size, args.batch
    h = w = size
    n = h * w
    hwf = h * (w // 2 + 1)
    y, loc, cov = synthetic(fft, size, B, device, seed=rank, per_lattice_cov=args.per_lattice_cov)
    lp = torch.empty(B, dtype=torch.float64, device=device)
    mn = torch.empty(B, dtype=torch.float64, device=device)
    gy = torch.empty_like(y)
    gc = torch.empty(B, h, w // 2 + 1, dtype=torch.float64, device=device)
    out = (lp, gy, gc, mn)

    def step():
        fft.gp_rfft2_lpdf_vjp(y, loc, cov, check=False, out=out)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    ctx = fft.get_context(device)
    for _ in range(args.warmup):
        step()
    barrier()
    # Timed region: K steps, bracketed by two HIP events on the stream the kernels are launched on (the library
    # follows torch's current stream) and by the host clock. No per-pass events in here: they cost 9 % (a marker
    # packet between every two kernels of the pipeline).
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    device_s = 1e-3 * ev0.elapsed_time(ev1)
    # Per-pass breakdown: extra, untimed steps with the library's per-pass HIP events.
    profile = not args.no_profile
    prof_steps = min(args.steps, 5)
    per_eval = lpl = None
    if profile:
        per_eval, lpl = profiled_passes(ctx, step, B, prof_steps)
    barrier()
    elapsed = sharding.max_over_ranks(elapsed, device=device)  # the job's clock is its slowest rank

    # what was computed in the timed region (not timed): finite values, PD check passed, and the last lattice of the
    # batch equal to a single-lattice evaluation
    check_errs = None
    if not os.environ.get("GPT_BENCH_NOCHECK"):  # timing-only experiment builds compute garbage on purpose
        assert bool(torch.isfinite(lp).all()) and float(mn.min()) >= 0.0, "benchmark produced invalid results"
        check_errs = self_check(fft, y, loc, cov, out, args.per_lattice_cov)

    evals = world * B * args.steps
    value = evals / elapsed
    b_alg = 24 * n + 16 * hwf  # bytes per evaluation, BASELINE.md section 3
    result = {
        "metric": "gp_rfft2_lpdf+grad evals/sec (fp64, 4096^2)" if size == 4096
        else f"gp_rfft2_lpdf+grad evals/sec (fp64, {size}^2)",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "checked": check_errs is not None,
        "check": {"what": "last lattice of the timed batch vs a single-lattice call, max relative error",
                  "errors": check_errs},
        "config": {
            "workload": f"synthetic {size}x{size} lattice gp_rfft2_lpdf + gradients (d/dy, d/dcov), "
                        f"batch of {B} lattices per GPU per step, "
                        + ("loc shared by the batch, one cov per lattice" if args.per_lattice_cov
                           else "cov/loc shared by the batch"),
            "global_batch": world * B,
            "parallelism": f"independent lattices sharded over {world} GPU(s), no collective",
            "path": fft.path_for_shape(h, w),
        },
    }
    if rank == 0:
        pipeline_s = device_s / (B * args.steps)  # device seconds per evaluation over the timed region
        achieved = b_alg / pipeline_s / 1e9
        result["roofline"] = {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": achieved * 1e9 / HBM_PEAK,
            "traffic": pmc_traffic(size),
            "traffic_note": "HBM-side bytes per evaluation from committed rocprofv3 PMC passes "
                            "(profiles/pmc_traffic.json), not collected in this run",
            "kernel": "row_fwd + col + row_inv (three-pass pipeline of one evaluation, launch gaps included)",
            "algorithmic_bytes_per_eval": b_alg,
            "device_s_per_eval": pipeline_s,
            "timing": "two HIP events around the timed region on the launch stream",
            "wall_frac": value / world * b_alg / HBM_PEAK,
        }
        if per_eval:  # seconds per evaluation; lpl = lattices per launch: the shared loc is read once per launch
            pass_bytes = {
                "row_fwd (pass A)": 16 * n + int(8 * n / lpl),  # read y, loc (shared); write packed spectrum
                "col (pass B)": 8 * n + 8 * hwf + 8 * hwf + 8 * n,  # spectrum r/w in place, read c, write dcov
                "finalize": 0,
                "row_inv (pass C)": 8 * n + 8 * n,         # read spectrum, write d/dy
            }
            kernels = []
            for (name, nbytes), sec in zip(pass_bytes.items(), per_eval):
                kernels.append({"name": name, "avg_ms_per_eval": 1e3 * sec, "compulsory_bytes": nbytes,
                                "GBps": (nbytes / sec / 1e9) if sec > 0 and nbytes else None})
            result["roofline"]["kernels"] = kernels
            result["roofline"]["lattices_per_launch"] = lpl
            result["roofline"]["kernels_note"] = (f"average launch durations over {prof_steps} extra untimed steps "
                                                  "with per-pass HIP events inside the library, divided by the "
                                                  f"{lpl:g} lattices a launch processes (rocprofv3's per-launch "
                                                  "averages are that many times larger)")
            dom = max(kernels, key=lambda k: k["avg_ms_per_eval"])
            result["roofline"]["dominant_kernel"] = {"name": dom["name"], "GBps": dom["GBps"],
                                                     "frac": dom["GBps"] * 1e9 / HBM_PEAK}
        if world == 1:
            # what this box's HBM delivers to a plain streaming copy (read + write bytes), beside the nominal peak
            src = torch.empty(1 << 27, dtype=torch.float64, device=device)  # 1 GiB
            dst = torch.empty_like(src)
            dst.copy_(src)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(10):
                dst.copy_(src)
            c1.record()
            torch.cuda.synchronize()
            result["roofline"]["measured_copy_GBps"] = 10 * 2 * src.numel() * 8 / (1e-3 * c0.elapsed_time(c1)) / 1e9
            del src, dst
        if world == 1 and not args.no_extra:
            result["extra"] = {"kernel_fused_lpdf": fused_kernel_extra(fft, ctx, y, loc, max(2, min(args.steps, 5))),
                               "docs_shape_35x60": small_lattice_extra(fft, device)}
            result["extra"]["generic_path"] = generic_path_extra(fft, device)
            # what the roofline fraction means: the same B_alg with nothing shared, and the
            # traffic-only ceiling of the three-pass structure on this box
        if world == 1 and not args.no_extra and not args.no_ceiling:
            del y, gy, gc, out
            torch.cuda.empty_cache()
            if not args.per_lattice_cov:
                result["roofline"]["unshared"] = unshared_leg(fft, ctx, size, B, device, max(2, min(args.steps, 5)))
            torch.cuda.empty_cache()
            result["roofline"]["three_pass_ceiling"] = three_pass_ceiling(args)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(size)
        print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
