#!/usr/bin/env python3
"""Where the kernel-fused log density's time goes between its kernels: run under
rocprofv3 --kernel-trace --output-format csv, then pass the *_kernel_trace.csv to --report.
usage: trace_fused_gaps.py            (the workload: 16 x 4096^2, 4 steps)
       trace_fused_gaps.py --report <kernel_trace.csv>"""
import csv
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-40:]
    prev_end = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print("%-60s dur %9.1f us  gap before %8.1f us" % (r["Kernel_Name"][:60], (e - s) / 1e3, gap))
        prev_end = e
    sys.exit(0)

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptools_amd import fft  # noqa: E402

B, h, w = 16, 4096, 4096
dev = torch.device("cuda", 0)
y = torch.randn(B, h, w, dtype=torch.float64, device=dev)
loc = torch.zeros(h, w, dtype=torch.float64, device=dev)
sigma = torch.linspace(1.25, 1.35, B, dtype=torch.float64, device=dev)
ls = torch.tensor([8.0, 12.0], dtype=torch.float64, device=dev).expand(B, 2)
per = torch.tensor([float(h), float(w)], dtype=torch.float64, device=dev).expand(B, 2)
out = (torch.empty(B, dtype=torch.float64, device=dev), torch.empty_like(y),
       torch.empty(B, 5, dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.float64, device=dev))
for _ in range(4):
    fft.gp_rfft2_lpdf_kernel_vjp(y, loc, "matern", 1.5, sigma, ls, per, nugget=1e-3, check=False, out=out)
torch.cuda.synchronize()
print("ok", float(out[0][0]))
