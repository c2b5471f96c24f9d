#!/usr/bin/env python3
"""
Kernel-fused log density against the composition of the three reference functions, 4096 x 4096,
batch of lattices resident in HBM, every lattice its own kernel parameters:
    composed  cov = gp_periodic_matern_cov_rfft2(...) + nugget; lpdf, d/dy, d/dcov = gp_rfft2_lpdf_vjp(y, loc, cov);
              d/d(sigma, length_scale, period) = kernel reverse mode(d/dcov)         (three library calls)
    fused     gp_rfft2_lpdf_kernel_vjp(...)                                          (one call, no spectrum in memory)
Prints one JSON line.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptools_amd import fft  # noqa: E402


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e-3 * e0.elapsed_time(e1) / iters


def main():
    # usage: bench_fused.py [height [width [batch]]]
    h = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    w = int(sys.argv[2]) if len(sys.argv) > 2 else h
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    loc = torch.randn(h, w, dtype=torch.float64, device=dev, generator=g)
    y = loc + torch.randn(B, h, w, dtype=torch.float64, device=dev, generator=g)
    sigma = torch.linspace(1.0, 1.5, B, dtype=torch.float64, device=dev)
    ls = torch.stack([torch.linspace(h / 12.0, h / 8.0, B), torch.linspace(w / 10.0, w / 7.0, B)], dim=-1).to(dev, torch.float64)
    per = torch.tensor([float(h), float(w)], dtype=torch.float64, device=dev).expand(B, 2)
    nu, eps = 1.5, 1e-3

    def composed():
        cov = fft.gp_periodic_matern_cov_rfft2(nu, h, w, sigma, ls, per) + eps
        lp, gy, gc = fft.gp_rfft2_lpdf_vjp(y, loc, cov, check=False)
        return lp, gy, fft.gp_periodic_matern_cov_rfft2_vjp(nu, h, w, sigma, ls, per, gc)

    def fused():
        return fft.gp_rfft2_lpdf_kernel_vjp(y, loc, "matern", nu, sigma, ls, per, nugget=eps, check=False)

    a, b = composed(), fused()
    rel = float(((a[0] - b[0]).abs() / a[0].abs()).max())
    relg = float((a[1] - b[1]).abs().max() / a[1].abs().max())
    tc, tf = timed(composed, 20), timed(fused, 20)
    n = h * w
    print(json.dumps({"case": f"{h}x{w}, batch {B}, Matern nu={nu}, per-lattice parameters",
                      "composed_us_per_eval": 1e6 * tc / B, "fused_us_per_eval": 1e6 * tf / B,
                      "composed_evals_per_s": B / tc, "fused_evals_per_s": B / tf, "speedup": tc / tf,
                      "fused_algorithmic_bytes": 24 * n, "fused_GBps": 24 * n * B / tf / 1e9,
                      "max_rel_diff_lpdf": rel, "max_rel_diff_grad_y": relg}))


if __name__ == "__main__":
    main()
