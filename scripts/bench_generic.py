#!/usr/bin/env python3
"""Secondary measurement: the generic three-pass path (gpt_generic.hip) on BASELINE config 3's 2-D analogue
(128 + 16 q)^2 and a few other non-power-of-two shapes, 64 chains per launch, value + gradients, device-resident.
One JSON object per line; not the headline benchmark."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_small import time_cpu, time_gpu  # noqa: E402


def main():
    import gp_oracle as oracle
    from gptools_amd import fft

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    shapes = [(144, 144), (160, 160), (176, 176), (192, 192), (301, 200), (1000, 1000), (3000, 3000)]
    only = sys.argv[1] if len(sys.argv) > 1 else ""  # "HxW": that shape only (any shape, not only the list's)
    if only and tuple(int(v) for v in only.split("x")) not in shapes:
        shapes = [tuple(int(v) for v in only.split("x"))]
    for h, w in shapes:
        if only and only != f"{h}x{w}":
            continue
        batch = 64 if h * w < 500_000 else (8 if h * w < 5_000_000 else 2)
        cov = oracle.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.2, [h / 9.0, w / 7.0], [float(h), float(w)]) + 1e-3
        y = rng.normal(size=(batch, h, w))
        loc = rng.normal(size=(h, w))
        yd, ld, cd = (torch.as_tensor(x, device=dev) for x in (y, loc, cov))
        sec = time_gpu(lambda: fft.gp_rfft2_lpdf_vjp(yd, ld, cd, check=False), 20)
        cpu = time_cpu(lambda: oracle.gp_rfft2_lpdf_vjp(y[0], loc, cov), budget=2.0)
        b_alg = 24 * h * w + 16 * h * (w // 2 + 1)
        print(json.dumps({"shape": [h, w], "path": fft.path_for_shape(h, w), "chains_per_launch": batch,
                          "us_per_launch": 1e6 * sec, "evals_per_s": batch / sec,
                          "GBps_algorithmic": batch * b_alg / sec / 1e9,
                          "cpu_oracle_evals_per_s_1core": 1.0 / cpu}), flush=True)


if __name__ == "__main__":
    main()
