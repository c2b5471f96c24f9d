#!/usr/bin/env python3
"""Kernel-fused log density (gp_rfft2_lpdf_kernel_vjp: Matern-3/2 parameters in, gradients down to them out) on shapes of
the generic three-pass path, against the three calls of the array interface that it replaces (spectrum, gp_rfft2_lpdf_vjp,
spectrum VJP). Device-resident, per-lattice parameters, microseconds per launch."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_small import time_gpu  # noqa: E402
from gptools_amd import fft  # noqa: E402

dev = torch.device("cuda", 0)
cases = [(128, 128, 1), (128, 128, 64), (192, 192, 64), (300, 210, 64), (1000, 1000, 8), (3000, 3000, 2)]
if len(sys.argv) > 1:  # "HxWxB,HxWxB,..."
    cases = [tuple(int(v) for v in c.split("x")) for c in sys.argv[1].split(",")]
for h, w, b in cases:
    y = torch.randn(b, h, w, dtype=torch.float64, device=dev)
    loc = torch.zeros(h, w, dtype=torch.float64, device=dev)
    sigma = torch.linspace(1.2, 1.4, b, dtype=torch.float64, device=dev)
    ls = torch.tensor([h / 30.0, w / 25.0], dtype=torch.float64, device=dev).expand(b, 2).contiguous()
    per = torch.tensor([float(h), float(w)], dtype=torch.float64, device=dev).expand(b, 2).contiguous()

    def fused():
        return fft.gp_rfft2_lpdf_kernel_vjp(y, loc, "matern", 1.5, sigma, ls, per, nugget=1e-3, check=False)

    def three_calls():
        cov = fft.gp_periodic_matern_cov_rfft2(1.5, h, w, sigma, ls, per) + 1e-3
        lp, gy, gc = fft.gp_rfft2_lpdf_vjp(y, loc, cov, check=False)
        return lp, gy, fft.gp_periodic_matern_cov_rfft2_vjp(1.5, h, w, sigma, ls, per, gc)

    t_f, t_3 = time_gpu(fused, 20), time_gpu(three_calls, 20)
    print(json.dumps({"shape": [h, w], "path": fft.path_for_shape(h, w), "lattices": b, "us_kernel_fused": round(1e6 * t_f, 1),
                      "us_three_calls": round(1e6 * t_3, 1)}), flush=True)
