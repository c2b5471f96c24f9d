#!/usr/bin/env python3
"""Around the one-workgroup kernel's size limit: gp_rfft2_lpdf_vjp (value + both gradients, device-resident) at
100 x 100 (one workgroup per lattice, unpadded LDS stride) and 102 x 100 (generic three-pass path), 1 / 64 / 1024 chains."""
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
for h, w in [(96, 100), (100, 100), (102, 100), (128, 128), (140, 140)]:
    for b in (1, 64, 1024):
        y = torch.randn(b, h, w, dtype=torch.float64, device=dev)
        loc = torch.randn(h, w, dtype=torch.float64, device=dev)
        cov = torch.rand(b, h, w // 2 + 1, dtype=torch.float64, device=dev) + 0.5
        out = (torch.empty(b, dtype=torch.float64, device=dev), torch.empty_like(y), torch.empty_like(cov),
               torch.empty(b, dtype=torch.float64, device=dev))
        sec = time_gpu(lambda: fft.gp_rfft2_lpdf_vjp(y, loc, cov, check=False, out=out), 30)
        b_alg = 24 * h * w + 16 * h * (w // 2 + 1)
        print(json.dumps({"shape": [h, w], "path": fft.path_for_shape(h, w), "chains": b, "us_per_launch": round(1e6 * sec, 2),
                          "GBps_algorithmic": round(b * b_alg / sec / 1e9, 1)}), flush=True)
