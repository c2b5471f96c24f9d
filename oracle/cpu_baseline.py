#!/usr/bin/env python3
"""
CPU baseline of bench.py (TEST / MEASUREMENT INFRASTRUCTURE, never part of the product path): times the NumPy
restatement of the reference (oracle/gp_oracle.py: gp_rfft2_lpdf + closed-form gradients, fft2.stan:189-192) on the
host cores of the box, once on one core and once with one process per core, on a bounded sample of the bench
workload. Prints one JSON object.

    python oracle/cpu_baseline.py --size 4096 [--procs P] [--budget SECONDS]

"build restatement -- cmdstan unavailable": the reference's own path is Stan source that needs stanc + Stan Math to
run; this is the same arithmetic through numpy.fft (pocketfft, one thread per process).
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gp_oracle as oracle  # noqa: E402


def cpu_model():
    """Model name as /proc/cpuinfo gives it (the GPU boxes are VMs that report a bare "AMD EPYC"), with the family /
    model numbers that identify the part behind it."""
    name, fam, mod = "unknown", None, None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                key = line.split(":", 1)[0].strip().lower()
                if key == "model name" and name == "unknown":
                    name = line.split(":", 1)[1].strip()
                elif key == "cpu family" and fam is None:
                    fam = line.split(":", 1)[1].strip()
                elif key == "model" and mod is None:
                    mod = line.split(":", 1)[1].strip()
                if name != "unknown" and fam is not None and mod is not None:
                    break
    except OSError:
        pass
    return f"{name} (family {fam}, model {mod})" if fam is not None else name


def usable_cores():
    """The box's CPU share: the affinity mask, capped by the job's thread limit (the GPU boxes export
    OMP_NUM_THREADS = their share; os.cpu_count() shows the whole machine there)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        v = os.environ.get(var)
        if v and v.isdigit() and int(v) > 0:
            n = min(n, int(v))
    return max(1, n)


def _inputs(size, seed):
    rng = np.random.default_rng(seed)
    h = w = size
    cov = oracle.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.3, [8.0, 12.0], [float(h), float(w)]) + 1e-3
    return rng.normal(size=(h, w)), rng.normal(size=(h, w)), cov


def _worker(args):
    size, seed, evals = args
    y, loc, cov = _inputs(size, seed)
    t0 = time.perf_counter()
    for _ in range(evals):
        oracle.gp_rfft2_lpdf_vjp(y, loc, cov)
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--procs", type=int, default=0, help="processes of the all-cores leg (0: the box's CPU share)")
    ap.add_argument("--budget", type=float, default=10.0, help="seconds per leg (approximate)")
    a = ap.parse_args()
    size = a.size
    y, loc, cov = _inputs(size, 0)
    # one core
    n1, t0 = 0, time.perf_counter()
    while True:
        oracle.gp_rfft2_lpdf_vjp(y, loc, cov)
        n1 += 1
        el1 = time.perf_counter() - t0
        if el1 > a.budget or n1 >= 8:
            break
    per_eval = el1 / n1
    # all cores: one process per core, each its own lattice, the same number of evaluations each
    procs = a.procs if a.procs > 0 else usable_cores()
    evals = max(1, min(4, int(a.budget / (1.5 * per_eval))))
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:
        worker_s = pool.map(_worker, [(size, 100 + p, evals) for p in range(procs)])
    el_all = time.perf_counter() - t0  # includes start-up and input generation: a job's view
    el_eval = max(worker_s)            # the evaluations alone, as the GPU figure counts them (slowest process)
    print(json.dumps({
        "value": procs * evals / el_eval,
        "unit": "evals/s",
        "cores": procs,
        "kind": "port",
        "kind_note": "build restatement (oracle/gp_oracle.py, NumPy pocketfft) -- cmdstan unavailable",
        "sample": f"{procs} processes x {evals} evaluations of gp_rfft2_lpdf+grad at {size}x{size}; the slowest process "
                  f"spent {el_eval:.1f} s evaluating (inputs resident, as for the GPU figure); wall clock with process "
                  f"start-up and input generation {el_all:.1f} s",
        "value_wall": procs * evals / el_all,
        "one_core": {"value": n1 / el1, "cores": 1,
                     "sample": f"{n1} evaluations in {el1:.1f} s, single thread"},
        "nproc": os.cpu_count(),
        "cpu_share": usable_cores(),
        "cpu_model": cpu_model(),
    }))


if __name__ == "__main__":
    main()
