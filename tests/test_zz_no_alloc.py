"""
The allocation-free entry point gpt_gp_rfft2_lpdf_vjp_noalloc (include/gptools_hip.h): called through the C ABI and
held to the existing entry point gpt_gp_rfft2_lpdf_vjp on the same inputs at the path's parity bar (relative 1e-10;
the two differ only in where the twiddle tables come from: device sincospi against host long-double tables), on two
calls in a row, the second with a smaller input -- the calls share one built-in scratch array per GPU, which keeps the
previous call's contents. Collected last (test_zz_*): the round's end runs pytest -x.
"""
import numpy as np
import pytest

import gp_oracle as o

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def _close(actual, desired, what):
    a, d = actual.cpu().numpy(), desired.cpu().numpy()
    s = float(np.max(np.abs(d)))
    np.testing.assert_allclose(a, d, rtol=RTOL, atol=RTOL * max(s, 1e-300), err_msg=what)


def _both(fft, shape, batch, seed, need_cov_grad=True, with_loc=True):
    """The same lattices through both entry points; returns ((lp, gy, gc, mn) x 2, inputs)."""
    import torch

    dev = torch.device("cuda:0")
    dim1 = len(shape) == 1
    h, w = (1, shape[0]) if dim1 else shape
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    y = torch.randn(batch, h, w, dtype=torch.float64, device=dev, generator=g)
    loc = torch.randn(h, w, dtype=torch.float64, device=dev, generator=g) if with_loc else None
    cov = torch.exp(0.3 * torch.randn(batch, h, w // 2 + 1, dtype=torch.float64, device=dev, generator=g))
    ctx = fft.get_context(dev)
    outs = []
    for name in ("gpt_gp_rfft2_lpdf_vjp", "gpt_gp_rfft2_lpdf_vjp_noalloc"):
        lp = torch.full((batch,), float("nan"), dtype=torch.float64, device=dev)
        mn = torch.full((batch,), float("nan"), dtype=torch.float64, device=dev)
        gy = torch.full_like(y, float("nan"))
        gc = torch.full_like(cov, float("nan")) if need_cov_grad else None
        ctx.call(name, y.data_ptr(), h * w, loc.data_ptr() if with_loc else None, 0, cov.data_ptr(), h * (w // 2 + 1), h, w,
                 batch, int(dim1), lp.data_ptr(), gy.data_ptr(), gc.data_ptr() if need_cov_grad else None, mn.data_ptr())
        torch.cuda.synchronize()
        outs.append((lp, gy, gc, mn))
    return outs, (y, loc, cov)


def _compare(outs, what):
    (lp0, gy0, gc0, mn0), (lp1, gy1, gc1, mn1) = outs
    _close(lp1, lp0, f"{what}: lpdf")
    _close(gy1, gy0, f"{what}: d/dy")
    if gc0 is not None:
        _close(gc1, gc0, f"{what}: d/dcov")
    _close(mn1, mn0, f"{what}: min_eig")


@pytest.fixture(scope="module")
def fft(gpu_lib):
    from gptools_amd import fft as f

    return f


# (first call, second call with a SMALLER input): power-of-two three-pass (sub-line and whole-column kernels), generic
# three-pass, the one-workgroup path, 1-D
PAIRS = [(((2048, 2048), 3), ((1024, 512), 5)),
         (((1024, 2048), 2), ((256, 256), 7)),
         (((301, 200), 4), ((110, 110), 3)),
         (((35, 60), 64), ((5, 8), 3)),
         (((176,), 64), ((128,), 5))]


@pytest.mark.parametrize("first,second", PAIRS, ids=lambda p: "x".join(map(str, p[0])) + f"x{p[1]}")
def test_noalloc_matches_existing_entry_point(fft, first, second):
    outs, (y, loc, cov) = _both(fft, first[0], first[1], seed=1)
    _compare(outs, f"first call {first}")
    # ... and against the oracle on one lattice, so that "equal" cannot mean "equally wrong"
    shp = first[0]
    yy, ll, cc = y[0].cpu().numpy(), loc.cpu().numpy(), cov[0].cpu().numpy()
    if len(shp) == 1:
        lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(yy[0], ll[0], cc[0])
        gy_o, gc_o = gy_o[None], gc_o[None]
    else:
        lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(yy, ll, cc)
    lp1, gy1, gc1, _ = outs[1]
    np.testing.assert_allclose(float(lp1[0]), lp_o, rtol=RTOL)
    np.testing.assert_allclose(gy1[0].cpu().numpy(), gy_o, rtol=RTOL, atol=RTOL * np.abs(gy_o).max())
    np.testing.assert_allclose(gc1[0].cpu().numpy(), gc_o, rtol=RTOL, atol=RTOL * np.abs(gc_o).max())
    outs2, _ = _both(fft, second[0], second[1], seed=2)
    _compare(outs2, f"second, smaller call {second}")
    outs3, _ = _both(fft, second[0], second[1], seed=3, need_cov_grad=False, with_loc=False)
    _compare(outs3, f"third call {second} without d/dcov and loc")


def test_noalloc_allocates_nothing_on_a_fresh_context(fft):
    """On a context that has run nothing else (no cached workspace, partials or twiddle tables), the entry point is
    called FIRST and repeatedly: the device's free memory does not move (hipMemGetInfo through torch), the context's
    cached buffers stay empty, and the results equal the allocating entry point's (called afterwards)."""
    import torch

    dev = torch.device("cuda:0")
    h, w, batch = 1024, 512, 6
    # (in a fresh process the very first launch loads the library's code object, which moves the device's free memory:
    # one throw-away call on a throw-away context takes that out of the measurement)
    fft.close_contexts()
    warm = fft.get_context(dev)
    t = torch.ones(1, h, w, dtype=torch.float64, device=dev)
    tc = torch.ones(1, h, w // 2 + 1, dtype=torch.float64, device=dev)
    tl = torch.empty(1, dtype=torch.float64, device=dev)
    warm.call("gpt_gp_rfft2_lpdf_vjp_noalloc", t.data_ptr(), h * w, None, 0, tc.data_ptr(), h * (w // 2 + 1), h, w, 1, 0,
              tl.data_ptr(), torch.empty_like(t).data_ptr(), None, None)
    torch.cuda.synchronize()
    fft.close_contexts()
    ctx = fft.get_context(dev)  # gpt_create: resolves the built-in array; allocates one pinned page
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    y = torch.randn(batch, h, w, dtype=torch.float64, device=dev, generator=g)
    loc = torch.randn(h, w, dtype=torch.float64, device=dev, generator=g)
    cov = torch.exp(0.3 * torch.randn(batch, h, w // 2 + 1, dtype=torch.float64, device=dev, generator=g))
    outs = []
    for _ in range(2):
        outs.append((torch.empty(batch, dtype=torch.float64, device=dev), torch.empty_like(y), torch.empty_like(cov),
                     torch.empty(batch, dtype=torch.float64, device=dev)))
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)

    def call(name, o_):
        ctx.call(name, y.data_ptr(), h * w, loc.data_ptr(), 0, cov.data_ptr(), h * (w // 2 + 1), h, w, batch, 0,
                 o_[0].data_ptr(), o_[1].data_ptr(), o_[2].data_ptr(), o_[3].data_ptr())
        torch.cuda.synchronize()

    for _ in range(3):
        call("gpt_gp_rfft2_lpdf_vjp_noalloc", outs[0])
        free1, _ = torch.cuda.mem_get_info(dev)
        assert free1 == free0, f"device free memory moved by {free0 - free1} bytes across an allocation-free call"
    call("gpt_gp_rfft2_lpdf_vjp", outs[1])  # this one allocates workspace, partials and tables
    free2, _ = torch.cuda.mem_get_info(dev)
    assert free2 < free0
    _compare([outs[1], outs[0]], "fresh context")
    with pytest.raises(NotImplementedError, match="any-length"):
        yy = torch.zeros(1, 6, 5000, dtype=torch.float64, device=dev)
        cc = torch.ones(1, 6, 2501, dtype=torch.float64, device=dev)
        ctx.call("gpt_gp_rfft2_lpdf_vjp_noalloc", yy.data_ptr(), 30000, None, 0, cc.data_ptr(), 6 * 2501, 6, 5000, 1, 0,
                 outs[0][0].data_ptr(), torch.empty_like(yy).data_ptr(), None, None)


def test_noalloc_rejects_what_does_not_fit(fft):
    import torch

    dev = torch.device("cuda:0")
    ctx = fft.get_context(dev)
    h = w = 4096  # 128 MiB of packed spectrum per lattice: beyond the 40 MiB array
    y = torch.zeros(1, h, w, dtype=torch.float64, device=dev)
    cov = torch.ones(1, h, w // 2 + 1, dtype=torch.float64, device=dev)
    lp = torch.zeros(1, dtype=torch.float64, device=dev)
    gy = torch.empty_like(y)
    with pytest.raises(NotImplementedError, match="built-in scratch"):
        ctx.call("gpt_gp_rfft2_lpdf_vjp_noalloc", y.data_ptr(), h * w, None, 0, cov.data_ptr(), h * (w // 2 + 1), h, w, 1,
                 0, lp.data_ptr(), gy.data_ptr(), None, None)
    with pytest.raises(ValueError):
        ctx.call("gpt_gp_rfft2_lpdf_vjp_noalloc", y.data_ptr(), h * w, None, 0, cov.data_ptr(), h * (w // 2 + 1), h, w, 1,
                 0, lp.data_ptr(), None, None, None)
