"""Where a single small lattice spends its time: phase-boundary cycle stamps of workgroup 0 of the small-lattice kernel.

Needs the diagnostic build (`make -C gptools_amd/csrc stamps` -> gptools_amd/libvar_stamps.so); run as
    GPT_LIB=gptools_amd/libvar_stamps.so python scripts/small_stamps.py
One JSON object per case: microseconds per phase (cycle counter scaled by the 100 MHz wall clock read at entry and exit).
"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from gptools_amd import _lib, fft  # noqa: E402

PHASES = ["prologue+tw", "load", "rows fwd", "split", "cols fwd", "elementwise", "packed col 0", "reduce", "cols inv",
          "merge", "rows inv", "store"]


HOST = "--host" in sys.argv  # one-lattice cases through gpt_gp_rfft2_lpdf_host (row-major) instead of device tensors


def main():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.gpt_debug_small_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.gpt_debug_small_stage_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    for H, W, batch in [(1, 128, 1), (1, 100, 1), (35, 60, 1), (35, 60, 4096), (64, 64, 1), (100, 100, 1)]:
        g = torch.Generator(device="cpu").manual_seed(1)
        shape = (batch, H, W) if H > 1 else (batch, W)
        y = torch.randn(shape, dtype=torch.float64, generator=g).to(dev)
        loc = torch.zeros_like(y)
        kshape = (H, W // 2 + 1) if H > 1 else (W // 2 + 1,)
        cov = (torch.rand(kshape, dtype=torch.float64, generator=g) + 0.5).to(dev)
        f = fft.gp_rfft2_lpdf_vjp if H > 1 else fft.gp_rfft_lpdf_vjp
        if HOST and batch == 1:
            # the host-pointer call: the kernel reads and writes the caller's arrays in pinned host memory
            import numpy as np
            yh, lh, ch = (np.ascontiguousarray(t.cpu().numpy().reshape(H, -1)) for t in (y, loc, cov))
            lp, gy, gc = np.zeros(1), np.zeros_like(yh), np.zeros_like(ch)
            P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
            glib = _lib.load()
            ctx = fft.get_context()
            for _ in range(5):
                assert glib.gpt_gp_rfft2_lpdf_host(ctx.handle, _lib.GPT_ROW_MAJOR, P(yh), P(lh), P(ch), H, W, int(H == 1), P(lp), P(gy),
                                                   P(gc)) == 0
        else:
            for _ in range(5):
                f(y, loc, cov)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 32)()
        rc = lib.gpt_debug_small_stamps(buf, 32)
        assert rc == 0, rc
        st = list(buf)
        cyc = st[12] - st[0]
        wall_us = (st[17] - st[16]) / 100.0
        scale = wall_us / cyc if cyc else 0.0
        rec = {"H": H, "W": W, "batch": batch, "host_pointers": bool(HOST and batch == 1), "total_us": round(wall_us, 2), "cycles": cyc,
               "phases_us": {n: round((st[i + 1] - st[i]) * scale, 2) for i, n in enumerate(PHASES)}}
        rec["last_thread_us"] = {"interior": round((st[21] - st[20]) * scale, 2), "packed col 0": round((st[22] - st[21]) * scale, 2),
                                 "wave sums + barrier": round((st[23] - st[22]) * scale, 2)}
        sb = (ctypes.c_longlong * 64)()
        assert lib.gpt_debug_small_stage_stamps(sb, 64) == 0
        ss = list(sb)
        # a stage's time = its stamp minus the previous stamp (phase boundary or stage), whichever is later
        marks = sorted([(st[i], "phase") for i in range(13)] + [(ss[2 + 2 * i], int(ss[1 + 2 * i])) for i in range(int(ss[0]))])
        rec["stages_us (radix, us)"] = [(tag, round((t - marks[i - 1][0]) * scale, 2)) for i, (t, tag) in enumerate(marks)
                                        if tag != "phase" and i > 0]
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
