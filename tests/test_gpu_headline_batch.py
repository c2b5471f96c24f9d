"""
GPU parity of the code path `bench.py` times: gp_rfft2_lpdf + gradients (fft2.stan:189-192) over a batch of
many large lattices in ONE call, so that the library takes its many-lattice launches (the workgroup numbering that
shares `loc` / `cov` between the lattices of a launch, the streaming workspace accesses, the wide column tiles)
instead of the one-lattice-at-a-time launches every other test reaches.

Each case checks, through the C ABI,
  * lattices 0, middle, last against the CPU oracle at the north-star tolerance (relative 1e-10), and
  * every lattice against a single-lattice call of the same entry point (another launch geometry, same arithmetic
    up to the order of the reductions) at 1e-13,
with a spectrum shared by the batch, a spectrum per lattice, and without d/dcov.
"""
import numpy as np
import pytest

import gp_oracle as o

pytestmark = pytest.mark.gpu

RTOL = 1e-10
RTOL_SELF = 1e-13

# (H, W, lattices): 16 x 4096^2 is bench.py's default workload; the others reach the same kind of launch through
# other line lengths (sub-lines of 2048 points with rows of 8192; sub-lines of 4096 points)
CASES = [(4096, 4096, 16), (4096, 8192, 6), (8192, 2048, 12)]


def _close(actual, desired, rtol, what):
    actual = actual.cpu().numpy() if hasattr(actual, "cpu") else np.asarray(actual)
    desired = desired.cpu().numpy() if hasattr(desired, "cpu") else np.asarray(desired)
    s = float(np.max(np.abs(desired)))
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=rtol * max(s, 1e-300), err_msg=what)


@pytest.fixture(scope="module")
def fft(gpu_lib):
    from gptools_amd import fft as f

    return f


def _inputs(fft, h, w, batch, per_lattice_cov):
    import torch

    dev = torch.device("cuda:0")
    if per_lattice_cov:
        b = torch.arange(batch, dtype=torch.float64, device=dev)
        cov = fft.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.3 + 0.05 * b,
                                               torch.stack([8.0 + 0.25 * b, 12.0 + 0.25 * b], dim=-1),
                                               [float(h), float(w)]) + 1e-3
    else:
        cov = fft.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.3, [8.0, 12.0], [float(h), float(w)]) + 1e-3
    # break the row symmetry of the spectrum (the edge columns' mirror rows then matter)
    g = torch.Generator(device=dev)
    g.manual_seed(5 + h + w)
    cov = cov * torch.exp(0.05 * torch.randn(cov.shape[-2:], dtype=torch.float64, device=dev, generator=g))
    loc = torch.randn(h, w, dtype=torch.float64, device=dev, generator=g)
    y = torch.empty(batch, h, w, dtype=torch.float64, device=dev)
    for i in range(batch):  # every lattice its own draw
        z = torch.randn(h, w, dtype=torch.float64, device=dev, generator=g)
        y[i] = fft.gp_inv_rfft2(z, loc, cov[i] if per_lattice_cov else cov)
        del z
    return y, loc, cov


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}")
@pytest.mark.parametrize("per_lattice_cov", [False, True], ids=["shared_cov", "cov_per_lattice"])
def test_many_lattice_launch_vs_oracle(fft, case, per_lattice_cov):
    import torch

    h, w, batch = case
    y, loc, cov = _inputs(fft, h, w, batch, per_lattice_cov)
    lp, gy, gc = fft.gp_rfft2_lpdf_vjp(y, loc, cov)
    assert lp.shape == (batch,) and gy.shape == y.shape and gc.shape == (batch, h, w // 2 + 1)
    loc_h = loc.cpu().numpy()
    for i in sorted({0, batch // 2 - 1, batch - 1}):
        cov_i = cov[i] if per_lattice_cov else cov
        lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y[i].cpu().numpy(), loc_h, cov_i.cpu().numpy())
        _close(lp[i], lp_o, RTOL, f"lpdf of lattice {i}")
        _close(gy[i], gy_o, RTOL, f"d/dy of lattice {i}")
        _close(gc[i], gc_o, RTOL, f"d/dcov of lattice {i}")
    # the same lattices one call at a time (cache-sized launch geometry)
    for i in range(batch):
        cov_i = cov[i] if per_lattice_cov else cov
        lp1, gy1, gc1 = fft.gp_rfft2_lpdf_vjp(y[i], loc, cov_i)
        _close(lp[i], lp1, RTOL_SELF, f"lpdf of lattice {i} vs single call")
        _close(gy[i], gy1, RTOL_SELF, f"d/dy of lattice {i} vs single call")
        _close(gc[i], gc1, RTOL_SELF, f"d/dcov of lattice {i} vs single call")
        del lp1, gy1, gc1
    # without d/dcov (another instantiation of the column kernel) and value only
    lp2, gy2, gc2 = fft.gp_rfft2_lpdf_vjp(y, loc, cov, need_cov_grad=False)
    assert gc2 is None
    _close(lp2, lp, RTOL_SELF, "lpdf without d/dcov")
    _close(gy2, gy, RTOL_SELF, "d/dy without d/dcov")
    _close(fft.gp_rfft2_lpdf(y, loc, cov), lp, RTOL_SELF, "value-only entry point")
    del lp2, gy2
    torch.cuda.empty_cache()


def test_many_lattice_launch_rejects(fft):
    """The positive-definiteness check (fft2.stan:20-23) of a many-lattice launch is per lattice."""
    import torch

    h = w = 4096
    batch = 16
    dev = torch.device("cuda:0")
    cov = fft.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.3, [8.0, 12.0], [float(h), float(w)]) + 1e-3
    cov = cov.expand(batch, h, w // 2 + 1).contiguous()
    cov[11, h - 3, 0] = -2.0  # a row of the edge column the transform never reads (fft2.stan:65) still rejects
    y = torch.zeros(batch, h, w, dtype=torch.float64, device=dev)
    lp, gy, gc = fft.gp_rfft2_lpdf_vjp(y, None, cov, check=False)
    ctx = fft.get_context(dev)
    mn = torch.empty(batch, dtype=torch.float64, device=dev)
    ctx.call("gpt_gp_rfft2_lpdf", y.data_ptr(), h * w, None, 0, cov.data_ptr(), h * (w // 2 + 1), h, w, batch, 0,
             lp.data_ptr(), mn.data_ptr())
    mn = mn.cpu().numpy()
    assert mn[11] == -0.5 * h * w * 2.0 and (np.delete(mn, 11) > 0).all()
    with pytest.raises(ValueError, match="not positive-definite"):
        fft.gp_rfft2_lpdf_vjp(y, None, cov)
