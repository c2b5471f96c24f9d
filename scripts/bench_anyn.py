#!/usr/bin/env python3
"""
Secondary measurement: the any-length path (gpt_anyn.hip) -- long 1-D series and lattices with a long
non-power-of-two axis. Fused value + gradients (gp_rfft(2)_lpdf_vjp), inputs resident in HBM, HIP events over
back-to-back calls; the CPU oracle (NumPy pocketfft, one core) on the same host beside it. One JSON object per line;
not the headline benchmark (bench.py).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_small import time_cpu, time_gpu  # noqa: E402


def main():
    import gp_oracle as oracle
    from gptools_amd import fft

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2)
    cases = [("1-D n=8192 (four-step 2 x 4096), 64 chains", (8192,), 64),
             ("1-D n=65536 (four-step 16 x 4096), 64 chains", (65536,), 64),
             ("1-D n=10007 (prime: Bluestein on 32768), 64 chains", (10007,), 64),
             ("1-D n=1048576 (four-step 256 x 4096), 8 chains", (1048576,), 8),
             ("1-D n=1000003 (prime: Bluestein on 2^21), 8 chains", (1000003,), 8),
             ("2-D 6000 x 4500 (four-step rows and columns), 1 lattice", (6000, 4500), 1),
             ("2-D 5003 x 5003 (prime: Bluestein on 16384 both ways), 1 lattice", (5003, 5003), 1),
             ("2-D 16384 x 16384 (four-step 4 x 4096 both ways), 1 lattice", (16384, 16384), 1)]
    only = sys.argv[1] if len(sys.argv) > 1 else ""  # substring of the case label
    for label, shape, batch in cases:
        if only not in label:
            continue
        if len(shape) == 1:
            n = shape[0]
            cov = oracle.gp_periodic_matern_cov_rfft(1.5, n, 1.2, n / 10.0, float(n)) + 1e-3
            lp_fn, o_fn = fft.gp_rfft_lpdf_vjp, oracle.gp_rfft_lpdf_vjp
            assert fft.path_for_shape(1, n) == 3
        else:
            h, w = shape
            cov = oracle.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.2, [h / 9.0, w / 7.0], [float(h), float(w)]) + 1e-3
            lp_fn, o_fn = fft.gp_rfft2_lpdf_vjp, oracle.gp_rfft2_lpdf_vjp
            assert fft.path_for_shape(h, w) == 3
        y = rng.normal(size=(batch,) + shape)
        loc = rng.normal(size=shape)
        yd, ld, cd = (torch.as_tensor(x, device=dev) for x in (y, loc, cov))
        sec = time_gpu(lambda: lp_fn(yd, ld, cd, check=False), 10)
        cpu = time_cpu(lambda: o_fn(y[0], loc, cov), budget=3.0)
        numel = int(np.prod(shape))
        b_alg = 24 * numel + 16 * int(np.prod(cov.shape))
        print(json.dumps({"case": label, "ms_per_call": 1e3 * sec, "evals_per_s": batch / sec,
                          "GBps_algorithmic": batch * b_alg / sec / 1e9,
                          "cpu_oracle_evals_per_s_1core": 1.0 / cpu, "speedup_vs_1core": batch / sec * cpu}), flush=True)
        del yd, ld, cd
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
