"""
The Stan-free part of the cmdstan binding (include/gptools_hip_bind.hpp), exercised from C++ exactly as the Stan-typed
veneer (include/gptools_hip.hpp) calls it: tests/cpp/bind_driver (built in-tree by `make -C tests/cpp`, called by
__graft_entry__.build()) links libgptools_hip.so, runs every value function and every reverse mode on column-major
arrays and writes the results; here they are held to the CPU oracle at relative 1e-10. The reverse modes of the linear
maps (rfft2, inv_rfft2, expand_rfft, pack / unpack) are checked against the transpose of the oracle's forward map.

CPU part (not marked gpu): the header and the driver pass a C++17 front end.
"""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import gp_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "bind_driver")
RTOL = 1e-10
MASK = (1 << 64) - 1


def test_bind_header_compiles_standalone(tmp_path):
    """g++ -fsyntax-only over the Stan-free header (alone) and over the driver that calls all of it."""
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    one = tmp_path / "one.cpp"
    one.write_text('#include <gptools_hip_bind.hpp>\nint main() { return 0; }\n')
    for src in (str(one), os.path.join(ROOT, "tests", "cpp", "bind_driver.cpp")):
        subprocess.run([cxx, "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        src], check=True)


class Lcg:
    """The driver's generator (64-bit LCG, uniform in (-1, 1))."""

    def __init__(self, seed):
        self.s = seed

    def draw(self, n):
        out = np.empty(n)
        s = self.s
        for i in range(n):
            s = (s * 6364136223846793005 + 1442695040888963407) & MASK
            out[i] = (s >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0
        self.s = s
        return out


def _read_records(path):
    rec = {}
    with open(path, "rb") as fh:
        data = fh.read()
    pos = 0
    while pos < len(data):
        (ln,) = struct.unpack_from("<q", data, pos)
        pos += 8
        name = data[pos:pos + ln].decode()
        pos += ln
        (cnt,) = struct.unpack_from("<q", data, pos)
        pos += 8
        rec[name] = np.frombuffer(data, dtype="<f8", count=cnt, offset=pos).copy()
        pos += 8 * cnt
    return rec


def close(actual, desired, what):
    desired = np.asarray(desired)
    s = max(float(np.max(np.abs(desired))), 1e-300)
    np.testing.assert_allclose(actual, desired, rtol=RTOL, atol=RTOL * s, err_msg=what)


@pytest.fixture(scope="module")
def records(gpu_lib, tmp_path_factory):
    if not os.path.exists(DRIVER):
        pytest.fail(f"{DRIVER} is missing: build it with `make -C tests/cpp` (done by __graft_entry__.build())")
    out = tmp_path_factory.mktemp("bind") / "records.bin"
    subprocess.run([DRIVER, str(out)], check=True, timeout=300)
    return _read_records(out)


CASES = [(1, 8, 1, 11), (1, 7, 1, 12), (1, 128, 1, 13), (5, 8, 0, 21), (6, 7, 0, 22), (35, 60, 0, 23), (256, 512, 0, 24)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[1]}")
def test_bind_functions_vs_oracle(records, case):
    H, W, dim1, seed = case
    tag = (f"n{W}" if dim1 else f"{H}x{W}") + "/"
    wf = W // 2 + 1
    n, nf = H * W, H * wf
    g = Lcg(seed)
    F = lambda v, cols: v.reshape((H, cols), order="F")  # noqa: E731  column-major -> (H, cols)
    C = lambda v: v[0::2] + 1j * v[1::2]  # noqa: E731  interleaved -> complex
    y, loc, z, bar = (F(g.draw(n), W) for _ in range(4))
    cov = F(1.5 + g.draw(nf), wf)
    barf = F(g.draw(nf), wf)
    zc = F(C(g.draw(2 * nf)), wf)
    barc = F(C(g.draw(2 * nf)), wf)
    R = lambda name, cols, cplx=False: F(C(records[tag + name]) if cplx else records[tag + name], cols)  # noqa: E731
    sq = (lambda a: a[0]) if dim1 else (lambda a: a)  # the 1-D oracle functions take vectors

    def up(a):  # oracle result -> (H, cols)
        return np.asarray(a)[None] if dim1 else np.asarray(a)

    # real-FFT layer
    close(R("rfft2", wf, True), up(o.rfft(sq(y)) if dim1 else o.rfft2(y)), "rfft2")
    close(R("inv_rfft2", W), up(o.inv_rfft(sq(zc), W) if dim1 else o.inv_rfft2(zc, W)), "inv_rfft2")
    # reverse mode of rfft2: ybar = Re sum_{k in half} zbar_k exp(+i theta) = n * Re ifft2(half spectrum, zero elsewhere)
    full = np.zeros((H, W), dtype=complex)
    full[:, :wf] = barc
    close(R("rfft2_vjp", W), n * np.fft.ifft2(full).real, "rfft2_vjp")
    # reverse mode of inv_rfft2 by the transpose identity <inv(z), xbar> = <z, zbar> (real inner product) on a basis
    zb = R("inv_rfft2_vjp", wf, True)
    inv = (lambda a: up(o.inv_rfft(sq(a), W))) if dim1 else (lambda a: o.inv_rfft2(a, W))
    lhs = float(np.sum(inv(zc) * bar))
    rhs = float(np.sum(zc.real * zb.real + zc.imag * zb.imag))
    np.testing.assert_allclose(rhs, lhs, rtol=1e-9, atol=1e-9 * np.abs(bar).sum() * np.abs(zc).max())
    probe = np.zeros((H, wf), dtype=complex)
    for (i, j, v) in [(0, 0, 1.0), (H - 1, 0, 1j), (H // 2, wf - 1, 1j), (H - 1, wf - 1, 1.0), (H // 3, 1 % wf, 1j)]:
        probe[:] = 0
        probe[i, j] = v
        want = float(np.sum(inv(probe) * bar))
        got = zb[i, j].real if v == 1.0 else zb[i, j].imag
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12)
    if dim1:
        close(C(records[tag + "expand_rfft"]), o.expand_rfft(zc[0], W), "expand_rfft")
        fullbar = C(g.draw(2 * W))
        # transpose identity on random directions
        d = C(np.random.default_rng(1).normal(size=2 * wf))
        lhs = np.sum((o.expand_rfft(d, W) * np.conj(fullbar)).real)
        hb = C(records[tag + "expand_rfft_vjp"])
        np.testing.assert_allclose(np.sum((d * np.conj(hb)).real), lhs, rtol=1e-10)
    # scales and layout helpers
    scale = up(o.gp_evaluate_rfft_scale(sq(cov), W) if dim1 else o.gp_evaluate_rfft2_scale(cov, W))
    close(R("scale", wf), scale, "scale")
    close(R("scale_vjp", wf), barf * scale / (2 * cov), "scale_vjp")
    unpack = (lambda a: up(o.gp_unpack_rfft(sq(a), W))) if dim1 else (lambda a: o.gp_unpack_rfft2(a, W))
    pack = (lambda a: up(o.gp_pack_rfft(sq(a)))) if dim1 else o.gp_pack_rfft2
    assert np.array_equal(R("unpack", W), unpack(zc))
    assert np.array_equal(R("pack", wf, True), pack(z))
    rng = np.random.default_rng(2)
    dz = rng.normal(size=(H, wf)) + 1j * rng.normal(size=(H, wf))
    ub = R("unpack_vjp", wf, True)
    np.testing.assert_allclose(np.sum(dz.real * ub.real + dz.imag * ub.imag), np.sum(unpack(dz) * bar), rtol=1e-10)
    dr = rng.normal(size=(H, W))
    pb = R("pack_vjp", W)
    pk = pack(dr)
    np.testing.assert_allclose(np.sum(dr * pb), np.sum(pk.real * barc.real + pk.imag * barc.imag), rtol=1e-10)
    # GP functions and reverse modes
    if dim1:
        close(R("gp_rfft", W), up(o.gp_rfft(y[0], loc[0], cov[0])), "gp_rfft")
        close(R("gp_inv_rfft", W), up(o.gp_inv_rfft(z[0], loc[0], cov[0])), "gp_inv_rfft")
        zb_o, _, cb_o = o.gp_inv_rfft_vjp(z[0], loc[0], cov[0], bar[0])
        close(R("gp_inv_rfft_vjp_z", W), up(zb_o), "gp_inv_rfft_vjp_z")
        close(R("gp_inv_rfft_vjp_cov", wf), up(cb_o), "gp_inv_rfft_vjp_cov")
        lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(y[0], loc[0], cov[0])
        ladj_o, ladj_g = o.gp_rfft_log_abs_det_jac(cov[0], W), o.gp_rfft_log_abs_det_jac_vjp(cov[0], W)
    else:
        close(R("gp_rfft", W), o.gp_rfft2(y, loc, cov), "gp_rfft2")
        yb_o, _, cb2_o = o.gp_rfft2_vjp(y, loc, cov, bar)
        close(R("gp_rfft_vjp_y", W), yb_o, "gp_rfft2_vjp_y")
        close(R("gp_rfft_vjp_cov", wf), cb2_o, "gp_rfft2_vjp_cov")
        close(R("gp_inv_rfft", W), o.gp_inv_rfft2(z, loc, cov), "gp_inv_rfft2")
        zb_o, _, cb_o = o.gp_inv_rfft2_vjp(z, loc, cov, bar)
        close(R("gp_inv_rfft_vjp_z", W), zb_o, "gp_inv_rfft2_vjp_z")
        close(R("gp_inv_rfft_vjp_cov", wf), cb_o, "gp_inv_rfft2_vjp_cov")
        lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y, loc, cov)
        ladj_o, ladj_g = o.gp_rfft2_log_abs_det_jac(cov, W), o.gp_rfft2_log_abs_det_jac_vjp(cov, W)
    close(records[tag + "lpdf_value_only"][0], lp_o, "lpdf value")
    close(records[tag + "lpdf"][0], lp_o, "lpdf")
    close(R("lpdf_grad_y", W), up(gy_o), "lpdf d/dy")
    close(R("lpdf_grad_cov", wf), up(gc_o), "lpdf d/dcov")
    close(records[tag + "ladj"][0], ladj_o, "ladj")
    close(R("ladj_grad_cov", wf), up(ladj_g), "ladj d/dcov")
    # spectral kernels
    if dim1:
        par = (1.3, 0.11 * W, 1.0 * W)
        eq, mt = o.gp_periodic_exp_quad_cov_rfft(W, *par), o.gp_periodic_matern_cov_rfft(1.5, W, *par)
        eqb = o.gp_periodic_exp_quad_cov_rfft_vjp(W, *par, barf[0])
        mtb = o.gp_periodic_matern_cov_rfft_vjp(1.5, W, *par, barf[0])
    else:
        sigma, ls, per = 1.3, np.array([0.09 * H, 0.12 * W]), np.array([1.0 * H, 1.1 * W])
        eq, mt = o.gp_periodic_exp_quad_cov_rfft2(H, W, sigma, ls, per), o.gp_periodic_matern_cov_rfft2(1.5, H, W, sigma, ls, per)
        eqb = o.gp_periodic_exp_quad_cov_rfft2_vjp(H, W, sigma, ls, per, barf)
        mtb = o.gp_periodic_matern_cov_rfft2_vjp(1.5, H, W, sigma, ls, per, barf)
    close(R("expquad", wf), up(eq), "exp-quad spectrum")
    close(R("matern32", wf), up(mt), "Matern spectrum")
    for name, want, spec in (("expquad_vjp", eqb, eq), ("matern32_vjp", mtb, mt)):
        flat = np.concatenate([np.ravel(x) for x in want])
        scale_ = float(np.abs(barf * up(spec)).sum())
        np.testing.assert_allclose(records[tag + name], flat, rtol=1e-9, atol=1e-10 * scale_, err_msg=name)
    # the log density from kernel parameters (one launch where the lattice fits a workgroup or the column pass evaluates the
    # kernel; composed otherwise): Matern 3/2 + nugget 1e-4
    covk = mt + 1e-4
    if dim1:
        lpk, gyk, gck = o.gp_rfft_lpdf_vjp(y[0], loc[0], covk)
        want = o.gp_periodic_matern_cov_rfft_vjp(1.5, W, *par, gck)
        gyk, covk_, gck_ = gyk[None], covk[None], gck[None]
    else:
        lpk, gyk, gck = o.gp_rfft2_lpdf_vjp(y, loc, covk)
        want = o.gp_periodic_matern_cov_rfft2_vjp(1.5, H, W, sigma, ls, per, gck)
        covk_, gck_ = covk, gck
    close(records[tag + "klpdf"][0], lpk, "kernel lpdf")
    close(R("klpdf_grad_y", W), gyk, "kernel lpdf d/dy")
    np.testing.assert_allclose(records[tag + "klpdf_grad_par"], np.concatenate([np.ravel(x) for x in want]), rtol=1e-9,
                               atol=1e-10 * float(np.abs(gck_ * covk_).sum()), err_msg="kernel lpdf d/dparams")
    if not dim1:
        oh, ow = H - H // 4, W - W // 4
        obs = records[tag + "obs"].reshape((oh, ow), order="F")
        assert obs[1, 0] == -1.0
        lp, f, zbar, locbar, lsum, cbar, thbar = o.gp_inv_rfft2_obs_lpdf_vjp(o.OBS_NEG_BINOMIAL_2_LOG, z, np.full((H, W), 0.35),
                                                                              cov, obs, 2.5, True)
        close(records[tag + "obs_lpdf"][0], lp, "obs lpdf")
        close(R("obs_grad_z", W), zbar, "obs d/dz")
        close(R("obs_grad_cov", wf), cbar, "obs d/dcov")
        close(records[tag + "obs_grad_loc_theta"], [lsum, thbar], "obs d/dloc, d/dtheta")
        # the kernel-parameter form (one-launch kernel for the lattices that fit a workgroup, composed otherwise)
        lp, zbar, lsum, bars, thbar = o.gp_inv_rfft2_obs_lpdf_kernel_vjp(o.OBS_NEG_BINOMIAL_2_LOG, z, np.full((H, W), 0.35), "matern",
                                                                          1.5, sigma, ls, per, obs, 2.5, nugget=1e-4)
        close(records[tag + "obsk_lpdf"][0], lp, "obs (kernel) lpdf")
        close(R("obsk_grad_z", W), zbar, "obs (kernel) d/dz")
        close(records[tag + "obsk_grad_loc_theta"], [lsum, thbar], "obs (kernel) d/dloc, d/dtheta")
        covk = mt + 1e-4
        cbark = o.gp_inv_rfft2_obs_lpdf_vjp(o.OBS_NEG_BINOMIAL_2_LOG, z, np.full((H, W), 0.35), covk, obs, 2.5, True)[5]
        np.testing.assert_allclose(records[tag + "obsk_grad_par"], np.concatenate([np.ravel(x) for x in bars]), rtol=1e-9,
                                   atol=1e-10 * float(np.abs(cbark * covk).sum()), err_msg="obs (kernel) d/dparams")


@pytest.mark.gpu
def test_bind_rejects_map_to_the_reference_messages(records):
    assert records["rejects"].tolist() == [1.0, 1.0, 1.0]
