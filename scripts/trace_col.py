#!/usr/bin/env python3
"""Phase timeline of the persistent column kernel (debug build with -DGPT_TRACE=1 swapped in as
libgptools_hip.so): prints, per tile index, the mean duration of every phase over the workgroups."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptools_amd import fft  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
y, loc, cov = bench.synthetic(fft, 4096, 2, dev, 0)
for _ in range(3):
    out = fft.gp_rfft2_lpdf_vjp(y, loc, cov, check=False)
torch.cuda.synchronize()
ctx = fft.get_context(dev)
n = 1024 * 8 * 8
buf = (ctypes.c_ulonglong * n)()
ctx.lib.gpt_debug_col_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = ctx.lib.gpt_debug_col_trace(buf, n)
assert rc == 0, rc
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8, 8).astype(np.int64)
nwg = int((t[:, 0, 0] != 0).sum())
t = t[:nwg]
start = t[:, 0, 0].min()
names = ["issue loads", "wait loads", "fwd fft", "packed0 + cov[0]", "elementwise", "inv fft", "issue stores"]
print(f"{nwg} workgroups; times in us (100 MHz clock); kernel span {(t.max() - start) / 100:.1f} us")
for r in range(8):
    if not (t[:, r, 0] != 0).any():
        break
    d = np.diff(t[:, r, :], axis=1) / 100.0
    print(f"tile {r}: start {np.mean(t[:, r, 0] - start) / 100:7.1f} (min {(t[:, r, 0].min() - start) / 100:6.1f} max "
          f"{(t[:, r, 0].max() - start) / 100:6.1f})  " + "  ".join(f"{nm} {np.mean(d[:, k]):5.1f}" for k, nm in enumerate(names)))
# slowest workgroups
last = np.zeros(nwg, dtype=np.int64)
for w in range(nwg):
    rr = [r for r in range(8) if t[w, r, 0] != 0]
    last[w] = t[w, rr[-1], 7] - start
order = np.argsort(last)[::-1]
print("end of last tile per workgroup (us): median %.1f  p90 %.1f  max %.1f" % (np.median(last) / 100, np.percentile(last, 90) / 100, last.max() / 100))
for w in order[:6]:
    print(f"  wg {w:4d}: " + " | ".join(f"tile {r}: start {(t[w, r, 0] - start) / 100:6.1f} phases " + " ".join(f"{x / 100:5.1f}" for x in np.diff(t[w, r, :])) for r in range(8) if t[w, r, 0] != 0))
if t[0, 7, 0] != 0:
    print("packed column 0 (wg 0): lds in %.1f | iterations " % ((t[0, 7, 1] - t[0, 7, 0]) / 100)
          + " ".join("%.1f" % ((t[0, 7, k] - t[0, 7, k - 1]) / 100) for k in range(2, 7) if t[0, 7, k] != 0)
          + " | total %.1f us" % ((t[0, 7, 7] - t[0, 7, 0]) / 100))
if t[:, 6, 0].any():
    d = np.diff(t[:nwg, 6, :6], axis=1) / 100.0
    print("forward column FFT of the last tile, mean over workgroups (us): "
          + "  ".join(f"{nm} {np.mean(d[:, k]):.2f}" for k, nm in enumerate(["stage1", "exchange1", "stage2", "exchange2", "stage3"])))
