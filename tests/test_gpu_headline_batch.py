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


@pytest.mark.parametrize("kind,nu", [("matern", 1.5), ("expquad", 0.0)])
def test_kernel_fused_many_lattice_launch(fft, kind, nu):
    """The kernel-fused entry point (gpt_gp_rfft2_lpdf_kernel_vjp: fft2.stan:189-192 over :206-247 + nugget) on the
    launch `bench.py` times for it: 16 lattices of 4096 x 4096 in one call, every lattice its own kernel parameters.
    Lattices 0, 7 and 15 against the oracle's composition of the three reference functions and the chain rule at 1e-10;
    every lattice against the library's own composition (spectrum kernel -> gp_rfft2_lpdf_vjp -> kernel reverse mode);
    and with caller-owned outputs (`out=`), the form a sampler loop uses."""
    import torch

    h = w = 4096
    batch = 16
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    loc = torch.randn(h, w, dtype=torch.float64, device=dev, generator=g)
    y = loc + 0.7 * torch.randn(batch, h, w, dtype=torch.float64, device=dev, generator=g)
    b = torch.arange(batch, dtype=torch.float64, device=dev)
    sigma = 1.0 + 0.04 * b
    ls = torch.stack([6.0 + 0.5 * b, 14.0 - 0.3 * b], dim=-1)
    per = torch.stack([float(h) * (1.0 + 0.01 * b), float(w) * (1.1 - 0.01 * b)], dim=-1)
    eps = 1e-3
    lp, gy, (sb, lb, pb) = fft.gp_rfft2_lpdf_kernel_vjp(y, loc, kind, nu, sigma, ls, per, nugget=eps)
    assert lp.shape == (batch,) and gy.shape == y.shape and lb.shape == (batch, 2) and pb.shape == (batch, 2)
    loc_h = loc.cpu().numpy()
    for i in (0, 7, 15):
        s_i, l_i, p_i = float(sigma[i]), ls[i].cpu().numpy(), per[i].cpu().numpy()
        cov = (o.gp_periodic_exp_quad_cov_rfft2(h, w, s_i, l_i, p_i) if kind == "expquad"
               else o.gp_periodic_matern_cov_rfft2(nu, h, w, s_i, l_i, p_i)) + eps
        lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y[i].cpu().numpy(), loc_h, cov)
        bars = (o.gp_periodic_exp_quad_cov_rfft2_vjp(h, w, s_i, l_i, p_i, gc_o) if kind == "expquad"
                else o.gp_periodic_matern_cov_rfft2_vjp(nu, h, w, s_i, l_i, p_i, gc_o))
        _close(lp[i], lp_o, RTOL, f"lpdf of lattice {i}")
        _close(gy[i], gy_o, RTOL, f"d/dy of lattice {i}")
        # sums of ~n terms of both signs: tolerance relative to the sum of magnitudes
        scale = float(np.abs(gc_o * cov).sum())
        for got, want, p, what in zip((sb[i], lb[i], pb[i]), bars, (s_i, l_i, p_i), ("sigma", "length_scale", "period")):
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-9, atol=1e-10 * scale / float(np.min(np.abs(p))),
                                       err_msg=f"d/d{what} of lattice {i}")
    # every lattice against the library's own composition, four lattices at a time
    for i0 in range(0, batch, 4):
        sl = slice(i0, i0 + 4)
        cov = (fft.gp_periodic_exp_quad_cov_rfft2(h, w, sigma[sl], ls[sl], per[sl]) if kind == "expquad"
               else fft.gp_periodic_matern_cov_rfft2(nu, h, w, sigma[sl], ls[sl], per[sl])) + eps
        lp_c, gy_c, gc_c = fft.gp_rfft2_lpdf_vjp(y[sl], loc, cov)
        _close(lp[sl], lp_c, 1e-12, f"lpdf of lattices {i0}.. vs composition")
        _close(gy[sl], gy_c, 1e-12, f"d/dy of lattices {i0}.. vs composition")
        del cov, lp_c, gy_c, gc_c
    # caller-owned outputs
    out = (torch.empty_like(lp), torch.empty_like(gy), torch.empty(batch, 5, dtype=torch.float64, device=dev),
           torch.empty_like(lp))
    lp2, gy2, (sb2, lb2, pb2) = fft.gp_rfft2_lpdf_kernel_vjp(y, loc, kind, nu, sigma, ls, per, nugget=eps, out=out)
    assert lp2.data_ptr() == out[0].data_ptr() and gy2.data_ptr() == out[1].data_ptr()
    assert torch.equal(lp2, lp) and torch.equal(gy2, gy) and torch.equal(sb2, sb) and torch.equal(lb2, lb)
    assert torch.equal(pb2, pb)
    del out, gy2, gy
    torch.cuda.empty_cache()
