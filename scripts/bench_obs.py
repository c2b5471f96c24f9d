"""Non-centred model density on the device: gpt_gp_inv_rfft2_obs_lpdf_vjp at the trees-model
shape scaled to BASELINE config 4 (4096 x 4096 padded lattice, 3500 x 3000 observed corner, negative-binomial counts,
10 % masked). One JSON line per configuration; GPT_OBS_FUSED=1 in the environment times the single-row-kernel route, OBS_NEED_F=0 the
call that returns neither f nor d lp / d f."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptools_amd import fft  # noqa: E402


def main():
    H = W = int(os.environ.get("OBS_N", 4096))
    oh, ow = int(H * 0.73), int(W * 0.85)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    cov = fft.gp_periodic_matern_cov_rfft2(1.5, H, W, 0.8, [8.0, 12.0], [H, W]) + 1e-3
    loc = torch.full((H, W), 0.5, dtype=torch.float64, device=dev)
    need_f = os.environ.get("OBS_NEED_F", "1") != "0"
    for family in os.environ.get("OBS_FAMILIES", "neg_binomial_2_log,poisson_log,normal").split(","):
        for B in [int(v) for v in os.environ.get("OBS_BATCHES", "1,4").split(",")]:
            z = torch.randn(B, H, W, dtype=torch.float64, device=dev, generator=g)
            obs = torch.poisson(torch.full((oh, ow), 2.0, dtype=torch.float64, device=dev), generator=g)
            obs[torch.rand(oh, ow, device=dev, generator=g) < 0.1] = -1.0 if family != "normal" else float("nan")
            theta = None if family == "poisson_log" else torch.full((B,), 1.7, dtype=torch.float64, device=dev)
            for _ in range(3):
                fft.gp_inv_rfft2_obs_lpdf_vjp(family, z, loc, cov, obs, theta, True, check=False, need_f=need_f)
            torch.cuda.synchronize()
            steps = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                fft.gp_inv_rfft2_obs_lpdf_vjp(family, z, loc, cov, obs, theta, True, check=False, need_f=need_f)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            n, hwf = H * W, H * (W // 2 + 1)
            # compulsory traffic per lattice: read z, loc, cov, obs; write f, d/df (returned by the shim), grad_z, grad_cov
            alg = 8 * (n + n + hwf + oh * ow + (2 * n if need_f else 0) + n + hwf)
            print(json.dumps({"op": "gp_inv_rfft2_obs_lpdf_vjp", "family": family, "H": H, "W": W, "window": [oh, ow],
                              "batch": B, "need_f": need_f, "fused": os.environ.get("GPT_OBS_FUSED", "0") == "1", "ms_per_call": ms,
                              "evals_per_s": B / ms * 1e3, "algorithmic_GBps": B * alg / ms / 1e6}), flush=True)


if __name__ == "__main__":
    main()
