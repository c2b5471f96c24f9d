#!/usr/bin/env python3
"""
Cross-chain batching measurement (BASELINE config 3 "64 chains fused per launch"): NCHAINS
threads, each a chain evaluating gp_rfft(2)_lpdf + gradients on its own lattice through HOST pointers (what a Stan
chain sees), `rounds` times:
    batched    gpt_batcher_lpdf_host: one fused launch per round for all chains
    unbatched  gpt_gp_rfft2_lpdf_host: every chain its own context, launch, copies and synchronisation
    cpu        the oracle on one core (one chain)
Prints one JSON line per case. Python threads: ctypes releases the GIL inside the library, the per-call Python
overhead (a few microseconds) is in both GPU numbers.
"""
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def run_threads(n, fn):
    ts = [threading.Thread(target=fn, args=(c,)) for c in range(n)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return time.perf_counter() - t0


def main():
    import gp_oracle as o
    from gptools_amd import _lib, batch

    lib = _lib.load()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(2)
    for label, shape, nchains, rounds in (("35x60", (35, 60), 64, 200), ("35x60", (35, 60), 8, 200),
                                          ("1-D n=176", (176,), 64, 200), ("256x256", (256, 256), 16, 50)):
        dim1 = len(shape) == 1
        h, w = (1, shape[0]) if dim1 else shape
        if dim1:
            cov = o.gp_periodic_matern_cov_rfft(1.5, w, 1.2, w / 10.0, float(w)) + 1e-3
        else:
            cov = o.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.2, [h / 9.0, w / 7.0], [float(h), float(w)]) + 1e-3
        y = rng.normal(size=(nchains,) + shape)
        loc = rng.normal(size=shape)
        b = batch.Batcher(nchains, None if dim1 else h, w)

        def batched(c):
            for _ in range(rounds):
                b.lpdf(c, y[c], loc, cov)

        run_threads(nchains, lambda c: b.lpdf(c, y[c], loc, cov))  # warm up
        tb = run_threads(nchains, batched)
        b.close()

        def unbatched(c):
            h_ = ctypes.c_void_p()
            assert lib.gpt_create(0, None, ctypes.byref(h_)) == 0
            lp = ctypes.c_double()
            gy, gc = np.empty(shape), np.empty(cov.shape)
            for _ in range(rounds + 1):
                rc = lib.gpt_gp_rfft2_lpdf_host(h_, _lib.GPT_ROW_MAJOR, P(y[c]), P(loc), P(cov), h, w, int(dim1),
                                                ctypes.byref(lp), P(gy), P(gc))
                assert rc == 0
            lib.gpt_destroy(h_)

        tu = run_threads(nchains, unbatched)
        fn = o.gp_rfft_lpdf_vjp if dim1 else o.gp_rfft2_lpdf_vjp
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 1.0:
            fn(y[0], loc, cov)
            k += 1
        cpu = k / (time.perf_counter() - t0)
        print(json.dumps({"case": f"{label}, {nchains} chains x {rounds} rounds (host pointers)",
                          "batched_evals_per_s": nchains * rounds / tb, "us_per_round_batched": 1e6 * tb / rounds,
                          "unbatched_evals_per_s": nchains * (rounds + 1) / tu,
                          "cpu_oracle_evals_per_s_per_core": cpu}))


if __name__ == "__main__":
    main()
