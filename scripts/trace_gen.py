#!/usr/bin/env python3
"""Phase timeline of the generic path's column kernel (debug build of gpt_generic.hip with -DGPT_TRACE=1, loaded through
GPT_LIB): microseconds per phase of column workgroup 1 of lattice 0.   usage: trace_gen.py H W [batch]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptools_amd import _lib, fft  # noqa: E402

h, w = int(sys.argv[1]), int(sys.argv[2])
b = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda", 0)
y = torch.randn(b, h, w, dtype=torch.float64, device=dev)
loc = torch.randn(h, w, dtype=torch.float64, device=dev)
cov = torch.rand(h, w // 2 + 1, dtype=torch.float64, device=dev) + 0.5
for _ in range(3):
    fft.gp_rfft2_lpdf_vjp(y, loc, cov, check=False)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
lib.gpt_debug_gen_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.gpt_debug_gen_trace(buf, 8) == 0
t = np.array(list(buf), dtype=np.float64) / 100.0  # us
names = ["load", "forward transform", "elementwise", "inverse transform", "store"]
print(f"{h}x{w} x {b}: " + ", ".join(f"{n} {t[i + 1] - t[i]:.1f} us" for i, n in enumerate(names)) + f"; total {t[5] - t[0]:.1f} us")
