#!/usr/bin/env python3
"""Phase timeline of the sub-line column kernel (debug build of sub_2048 with -DGPT_TRACE=1, loaded through
GPT_LIB): mean duration of every phase per tile index over the workgroups."""
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
ctx.lib.gpt_debug_sub_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = ctx.lib.gpt_debug_sub_trace(buf, n)
assert rc == 0, rc
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8, 8).astype(np.int64)
nwg = int((t[:, 0, 0] != 0).sum())
t = t[:nwg]
start = t[:, 0, 0].min()
names = ["issue loads", "wait loads", "fwd fft", "elementwise", "drain stores", "inv fft", "issue stores"]
print(f"{nwg} workgroups; times in us (100 MHz clock); kernel span {(t.max() - start) / 100:.1f} us")
for r in range(8):
    if not (t[:, r, 0] != 0).any():
        break
    d = np.diff(t[:, r, :], axis=1) / 100.0
    print(f"tile {r}: start {np.mean(t[:, r, 0] - start) / 100:7.1f} (min {(t[:, r, 0].min() - start) / 100:6.1f} max "
          f"{(t[:, r, 0].max() - start) / 100:6.1f})  " + "  ".join(f"{nm} {np.mean(d[:, k]):5.1f}" for k, nm in enumerate(names)))
half = nwg // 2
for lbl, sl in (("first half of the grid", slice(0, half)), ("second half", slice(half, nwg))):
    print(lbl, "tile 0 start %.1f, tile 1 end %.1f" % (np.mean(t[sl, 0, 0] - start) / 100, np.mean(t[sl, 1, 7] - start) / 100))
end = (t[:, 1, 7] - start) / 100.0
order = np.argsort(end)[::-1]
print("end of tile 1 per workgroup (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(end, [50, 90, 99, 100])))
print("slowest:", " ".join(f"wg{w}({w % 8}):{end[w]:.0f}" for w in order[:16]))
print("per XCD mean end:", " ".join(f"{np.mean(end[x::8]):.1f}" for x in range(8)))
print("per XCD max end:", " ".join(f"{np.max(end[x::8]):.1f}" for x in range(8)))
