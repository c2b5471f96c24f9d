#!/usr/bin/env python3
"""Lattices that fill more than half a CU's LDS on the one-workgroup path, many per launch: that path against the generic
three-pass path (GPT_SMALL_TO_GENERIC_BATCH=1 in the environment sends them there; =0 keeps them). gp_rfft2_lpdf_vjp,
device-resident.   usage: ab_small_vs_generic.py [batch,batch,...]"""
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
label = "generic" if os.environ.get("GPT_SMALL_TO_GENERIC_BATCH", "0") not in ("0", "") else "one_workgroup"
shapes = [(60, 60), (72, 72), (80, 80), (64, 128), (96, 100), (100, 100)]
if len(sys.argv) > 2:
    shapes = [tuple(int(v) for v in x.split("x")) for x in sys.argv[2].split(",")]
for h, w in shapes:
    for b in ([int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (16, 64, 256, 1024, 4096)):
        y = torch.randn(b, h, w, dtype=torch.float64, device=dev)
        loc = torch.randn(h, w, dtype=torch.float64, device=dev)
        cov = torch.rand(b, h, w // 2 + 1, dtype=torch.float64, device=dev) + 0.5
        out = (torch.empty(b, dtype=torch.float64, device=dev), torch.empty_like(y), torch.empty_like(cov),
               torch.empty(b, dtype=torch.float64, device=dev))
        sec = time_gpu(lambda: fft.gp_rfft2_lpdf_vjp(y, loc, cov, check=False, out=out), 20)
        print(json.dumps({"route": label, "shape": [h, w], "chains": b, "us_per_launch": round(1e6 * sec, 2)}), flush=True)
