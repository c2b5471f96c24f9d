"""Host-pointer calls (gpt_gp_rfft2_lpdf_host, value + both gradients) at the mid sizes between the docs models and the
headline: 60 x 60 ... 256 x 256, row- and column-major, microseconds per call."""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gptools_amd import _lib, fft
lib = _lib.load(); ctx = fft.get_context()
P = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
rng = np.random.default_rng(0)
for H, W in [(60, 60), (64, 64), (100, 100), (128, 128), (200, 200), (256, 256)]:
    wf = W // 2 + 1
    for layout, name in ((_lib.GPT_ROW_MAJOR, "row"), (_lib.GPT_COL_MAJOR, "col")):
        mk = np.ascontiguousarray if name == "row" else np.asfortranarray
        cov = mk(np.exp(0.3 * rng.normal(size=(H, wf))))
        y, loc = mk(rng.normal(size=(H, W))), mk(rng.normal(size=(H, W)))
        lp = np.zeros(1); gy, gc = mk(np.zeros((H, W))), mk(np.zeros((H, wf)))
        def f():
            rc = lib.gpt_gp_rfft2_lpdf_host(ctx.handle, layout, P(y), P(loc), P(cov), H, W, 0, P(lp), P(gy), P(gc))
            assert rc == 0
        for _ in range(5): f()
        t0 = time.perf_counter()
        for _ in range(300): f()
        dt = (time.perf_counter() - t0) / 300
        print(H, W, name, round(1e6 * dt, 1), "us", flush=True)
