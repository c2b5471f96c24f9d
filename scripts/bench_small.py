#!/usr/bin/env python3
"""
Secondary measurements: the launch-bound configurations of BASELINE.json.

    config 1  1-D lattice n = 128 (and the docs' n = 100), one chain
    config 2  35 x 60 lattice (docs/trees: 25 x 50 + padding 10), batches of 1 / 64 / 4096 chains
    config 3  1-D padded sweep n in {128, 144, 160, 176, 192}, 64 chains fused per launch

For each: microseconds per launch (HIP events over `iters` back-to-back launches of the fused value + gradient
entry point, inputs resident in HBM), evaluations per second, and the CPU oracle on the same host (one core).
Prints one JSON object per line; not the headline benchmark (bench.py).
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def time_gpu(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e-3 * e0.elapsed_time(e1) / iters


def time_cpu(fn, budget=2.0):
    fn()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        fn()
        n += 1
    return (time.perf_counter() - t0) / n


def main():
    import gp_oracle as oracle
    from gptools_amd import fft

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1)
    cases = [("config 1: 1-D n=128, one chain", (128,), 1), ("config 1 (docs): 1-D n=100, one chain", (100,), 1)]
    cases += [(f"config 2: 35x60, {b} chain(s) per launch", (35, 60), b) for b in (1, 64, 4096)]
    cases += [(f"config 3: 1-D n={n}, 64 chains per launch", (n,), 64) for n in (128, 144, 160, 176, 192)]
    for label, shape, batch in cases:
        if len(shape) == 1:
            n = shape[0]
            cov = oracle.gp_periodic_matern_cov_rfft(1.5, n, 1.2, n / 10.0, float(n)) + 1e-3
            lp_fn, o_fn = fft.gp_rfft_lpdf_vjp, oracle.gp_rfft_lpdf_vjp
        else:
            h, w = shape
            cov = oracle.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.2, [h / 9.0, w / 7.0], [float(h), float(w)]) + 1e-3
            lp_fn, o_fn = fft.gp_rfft2_lpdf_vjp, oracle.gp_rfft2_lpdf_vjp
        y = rng.normal(size=(batch,) + shape)
        loc = rng.normal(size=shape)
        # every chain its own spectrum (its own hyperparameters), as in a sampler
        covb = cov * np.exp(0.05 * rng.normal(size=(batch,) + (1,) * cov.ndim))
        yd, ld, cd = (torch.as_tensor(x, device=dev) for x in (y, loc, covb))
        sec = time_gpu(lambda: lp_fn(yd, ld, cd, check=False), 200)
        cpu = time_cpu(lambda: o_fn(y[0], loc, covb[0]))
        numel = int(np.prod(shape))
        b_alg = 24 * numel + 16 * int(np.prod(cov.shape))
        print(json.dumps({"case": label, "us_per_launch": 1e6 * sec, "evals_per_s": batch / sec,
                          "GBps_algorithmic": batch * b_alg / sec / 1e9,
                          "cpu_oracle_evals_per_s_1core": 1.0 / cpu, "speedup_vs_1core": batch / sec * cpu}))


if __name__ == "__main__":
    main()
