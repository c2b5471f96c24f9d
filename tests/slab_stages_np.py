"""NumPy stand-ins for the three local passes of gptools_amd.slab (CPU tests of the exchange protocol).
Test infrastructure: built on the oracle's entry classification, never imported by the package."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import gp_oracle as oracle  # noqa: E402


class NumpyStages:
    # the layouts of the exchange, built on the plain stand-ins below (copies are fine in a test stand-in)
    def rows_fwd_blocked(self, y_rows, loc_rows, height, width, world):
        spec = self.rows_fwd(y_rows, loc_rows, height, width)
        rows, nc = spec.shape[0], width // 2 // world
        return spec.view(rows, world, nc).permute(1, 0, 2).contiguous()

    def cols_lpdf_vjp_owned(self, spec_cols, cov_cols, cov_nyquist, height, width, col0, need_cov_grad=True):
        nc = spec_cols.shape[1]
        cov = torch.full((height, width // 2 + 1), float("nan"), dtype=torch.float64)  # only the owned columns are valid
        cov[:, col0:col0 + nc] = cov_cols
        if col0 == 0:
            cov[:, width // 2] = cov_nyquist
        out = torch.full_like(cov, float("nan")) if need_cov_grad else None
        lp, mn = self.cols_lpdf_vjp(spec_cols, cov, out, height, width, col0)
        if not need_cov_grad:
            return lp, mn, None, None
        return lp, mn, out[:, col0:col0 + nc].contiguous(), (out[:, width // 2].contiguous() if col0 == 0 else None)

    def rows_inv_blocked(self, recv, height, width, alpha, out=None):
        world, rows, nc = recv.shape
        res = self.rows_inv(recv.permute(1, 0, 2).contiguous().view(rows, world * nc), height, width, alpha)
        if out is None:
            return res
        out.copy_(res)
        return out

    def rows_fwd(self, y_rows, loc_rows, height, width):
        f = np.fft.rfft(y_rows.numpy() - loc_rows.numpy(), axis=1)
        spec = f[:, : width // 2].copy()
        spec[:, 0] = f[:, 0].real + 1j * f[:, width // 2].real  # DC + i Nyquist (both real)
        return torch.from_numpy(spec)

    @staticmethod
    def _edge(f, c, kind, n):
        """One edge column (all H rows): log-density terms, d/dcov, and F/c completed Hermitian."""
        height = f.shape[0]
        g = np.zeros(height, dtype=np.complex128)
        gc = np.zeros(height)
        lp = 0.0
        for k in range(height):
            if kind[k] == 1:
                q = f[k].real ** 2 / (n * c[k])
                lp -= 0.5 * (q + np.log(c[k]))
                gc[k] = 0.5 * (q - 1.0) / c[k]
                g[k] = f[k].real / c[k]
            elif kind[k] == 2:
                q = abs(f[k]) ** 2 / (n * c[k])
                lp -= q + np.log(c[k])
                gc[k] = (q - 1.0) / c[k]
                g[k] = f[k] / c[k]
                g[height - k] = np.conj(g[k])
        return g, gc, lp

    def cols_lpdf_vjp(self, spec_cols, cov, out_cov, height, width, col0):
        c = cov.numpy()
        z = spec_cols.numpy()
        n = height * width
        fhat = np.fft.fft(z, axis=0)
        kinds = oracle._entry_kinds2(height, width)
        g = np.empty_like(fhat)
        lp, mn = 0.0, np.inf
        oc = out_cov.numpy() if out_cov is not None else None
        for jl in range(z.shape[1]):
            j = col0 + jl
            if j == 0:
                zm = np.conj(np.roll(fhat[::-1, 0], 1))  # conj(Z[(H - k) % H])
                fa, fb = 0.5 * (fhat[:, 0] + zm), (fhat[:, 0] - zm) / 2j
                ga, gca, lpa = self._edge(fa, c[:, 0], kinds[:, 0], n)
                gb, gcb, lpb = self._edge(fb, c[:, width // 2], kinds[:, width // 2], n)
                g[:, 0] = ga + 1j * gb
                lp += lpa + lpb
                mn = min(mn, 0.5 * n * c[:, 0].min(), 0.5 * n * c[:, width // 2].min())
                if oc is not None:
                    oc[:, 0], oc[:, width // 2] = gca, gcb
            else:
                q = np.abs(fhat[:, jl]) ** 2 / (n * c[:, j])
                lp -= np.sum(q + np.log(c[:, j]))
                g[:, jl] = fhat[:, jl] / c[:, j]
                mn = min(mn, 0.5 * n * c[:, j].min())
                if oc is not None:
                    oc[:, j] = (q - 1.0) / c[:, j]
        z[...] = height * np.fft.ifft(g, axis=0)  # unnormalised inverse, in place
        return torch.tensor([lp], dtype=torch.float64), torch.tensor([mn], dtype=torch.float64)

    def rows_inv(self, spec_rows, height, width, alpha):
        s = spec_rows.numpy()
        f = np.zeros((s.shape[0], width // 2 + 1), dtype=np.complex128)
        f[:, : width // 2] = s
        f[:, 0] = s[:, 0].real
        f[:, width // 2] = s[:, 0].imag
        return torch.from_numpy(alpha * width * np.fft.irfft(f, n=width, axis=1))
