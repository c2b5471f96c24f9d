"""PCIe-inclusive rates of the host-pointer ABI (what the Stan user header calls) at 4096 x 4096: never the bench
value, recorded for DESIGN.md. Column-major (Stan) pageable NumPy arrays in and out, one lattice per call."""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptools_amd import _lib, fft  # noqa: E402


def main():
    H = W = int(os.environ.get("HOST_N", 4096))
    wf = W // 2 + 1
    lib = _lib.load()
    ctx = fft.get_context()
    rng = np.random.default_rng(0)
    F = np.asfortranarray
    P = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    cov = F(fft.gp_periodic_matern_cov_rfft2(1.5, H, W, 1.3, [8.0, 12.0], [float(H), float(W)]).cpu().numpy() + 1e-3)
    y, loc = F(rng.normal(size=(H, W))), F(rng.normal(size=(H, W)))
    lp = np.zeros(1)
    gy, gc = F(np.zeros((H, W))), F(np.zeros((H, wf)))
    oh, ow = int(H * 0.73), int(W * 0.85)
    obs = F(rng.poisson(2.0, size=(oh, ow)).astype(float))
    gls, gth = np.zeros(1), np.zeros(1)

    def lpdf():
        rc = lib.gpt_gp_rfft2_lpdf_host(ctx.handle, _lib.GPT_COL_MAJOR, P(y), P(loc), P(cov), H, W, 0, P(lp), P(gy), P(gc))
        assert rc == 0, ctx.error()

    def obs_model():
        rc = lib.gpt_gp_inv_rfft2_obs_lpdf_host(ctx.handle, _lib.GPT_COL_MAJOR, 2, P(y), None, 0.5, P(cov), P(obs), oh, ow, 1.7,
                                                1, H, W, 0, P(lp), P(gy), P(gls), P(gc), P(gth))
        assert rc == 0, ctx.error()

    n, hwf = H * W, H * wf
    for name, fn, up, down in (("gpt_gp_rfft2_lpdf_host", lpdf, 8 * (2 * n + hwf), 8 * (n + hwf)),
                               ("gpt_gp_inv_rfft2_obs_lpdf_host", obs_model, 8 * (n + hwf + oh * ow), 8 * (n + hwf))):
        for _ in range(3):
            fn()
        t0 = time.perf_counter()
        k = 8 if H * W >= 1 << 22 else 200
        for _ in range(k):
            fn()
        dt = (time.perf_counter() - t0) / k
        print(json.dumps({"entry": name, "H": H, "W": W, "layout": "col_major", "ms_per_call": 1e3 * dt,
                          "evals_per_s": 1 / dt, "host_to_device_MB": up / 1e6, "device_to_host_MB": down / 1e6,
                          "pcie_GBps": (up + down) / dt / 1e9}), flush=True)


if __name__ == "__main__":
    main()
