// gptools_hip_bind.hpp -- the Stan-free part of the cmdstan binding: everything between Stan's types and the C ABI
// (include/gptools_hip.h) that does not need a Stan header. Plain C++17, raw COLUMN-MAJOR pointers (Stan / Eigen
// storage order), sizes, and exceptions that mirror what the reference's Stan code throws:
//     reject(...)                  -> std::domain_error       (fft1.stan:19-21, fft2.stan:21-23, util.stan:274-276)
//     argument sizes do not match  -> std::invalid_argument
//     HIP failure                  -> std::runtime_error
// One function per reference function (value) and one per reverse mode (`*_vjp`, the vector-Jacobian product Stan's
// autodiff would have produced: the reference has no gradient code). include/gptools_hip.hpp is the thin Stan-typed
// veneer over this file. This file is compiled in the build's tests (g++ -fsyntax-only on CPU; tests/cpp/bind_driver.cpp
// runs every function on the GPU and tests/test_gpu_bind_cpp.py holds the results to the oracle).
//
// Complex cotangents follow Stan's convention for complex<var>: the adjoint of a complex value is the pair
// (d/dRe, d/dIm) stored as a complex number.
#ifndef GPTOOLS_HIP_BIND_HPP
#define GPTOOLS_HIP_BIND_HPP

#include <gptools_hip.h>

#include <complex>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace gptools_hip {

using cplx = std::complex<double>;

// One library context per calling thread (cmdstan chains are processes, or threads under STAN_THREADS); the device is
// GPTOOLS_HIP_DEVICE (default 0).
inline gpt_ctx* ctx() {
    struct Holder {
        gpt_ctx* c = nullptr;
        Holder() {
            int dev = 0;
            if (const char* e = std::getenv("GPTOOLS_HIP_DEVICE")) dev = std::atoi(e);
            if (gpt_create(dev, nullptr, &c) != GPT_OK) throw std::runtime_error("gptools_hip: no usable GPU");
        }
        ~Holder() { gpt_destroy(c); }
    };
    thread_local Holder h;
    return h.c;
}

// Map the library's status codes onto what the reference's Stan code does.
inline void check(int rc, const char* fn) {
    if (rc == GPT_OK) return;
    const std::string text = gpt_last_error(ctx());
    if (rc == GPT_ERR_NOT_PD) throw std::domain_error(text); // Stan `reject`, the reference's own message
    if (rc == GPT_ERR_SHAPE) throw std::invalid_argument(std::string(fn) + ": " + text);
    throw std::runtime_error(std::string(fn) + ": " + text);
}

// Lattice shape: the 1-D functions are height == 1 with the 1-D packing layout (fft1.stan:44-51).
struct Shape {
    int64_t H, W;
    int dim1;
    int64_t n() const { return H * W; }
    int64_t wf() const { return W / 2 + 1; }
    int64_t nf() const { return H * wf(); } // entries of the half spectrum
};
inline Shape shape1(int64_t n) { return Shape{1, n, 1}; }
inline Shape shape2(int64_t H, int64_t W) { return Shape{H, W, 0}; }
inline void require_sizes(bool ok, const char* fn) {
    if (!ok) throw std::invalid_argument(std::string(fn) + ": argument sizes do not match");
}

inline const double* re(const cplx* p) { return reinterpret_cast<const double*>(p); }
inline double* re(cplx* p) { return reinterpret_cast<double*>(p); }

// ===================================================================================================
// Real-FFT layer: stan/gptools/util.stan:255-330
// rfft (util.stan:263-265) / rfft2 (util.stan:302-304): y[H x W] -> out[H x (W/2+1)]
inline void rfft2(const double* y, Shape s, cplx* out) {
    check(gpt_rfft2_host(ctx(), GPT_COL_MAJOR, y, s.H, s.W, re(out)), s.dim1 ? "rfft" : "rfft2");
}
// inv_rfft (util.stan:291-293) / inv_rfft2 (util.stan:315-330): z[H x (W/2+1)] -> out[H x W]
inline void inv_rfft2(const cplx* z, Shape s, double* out) {
    check(gpt_inv_rfft2_host(ctx(), GPT_COL_MAJOR, re(z), s.H, s.W, out), s.dim1 ? "inv_rfft" : "inv_rfft2");
}
// expand_rfft (util.stan:271-281); `ysize` is the size of the caller's vector: the reference rejects a mismatch
inline void expand_rfft(const cplx* y, int64_t ysize, int64_t n, cplx* out) {
    if (ysize != n / 2 + 1) // util.stan:274-276
        throw std::domain_error("expected complex vector with " + std::to_string(n / 2 + 1) + " elements but got " +
                                std::to_string(ysize));
    check(gpt_expand_rfft_host(ctx(), re(y), n, re(out)), "expand_rfft");
}

// Column weights of the half spectrum inside a real inverse transform: 1 on the columns that are their own mirror image
// (j = 0 and, W even, j = W/2), 2 on the others (util.stan:322-328 fills the mirrored half with conjugates).
inline double half_weight(Shape s, int64_t j) { return (j == 0 || 2 * j == s.W) ? 1.0 : 2.0; }

// Reverse mode of rfft(2): cotangent of the half spectrum -> cotangent of y. y enters Z_k = sum_n y_n e^{-i theta_kn}
// linearly, so ybar_n = Re sum_{k in half} zbar_k e^{+i theta_kn} = n * inv_rfft2(zbar / weight).
inline void rfft2_vjp(const cplx* zbar, Shape s, double* ybar) {
    std::vector<cplx> t((size_t)s.nf());
    for (int64_t j = 0; j < s.wf(); ++j)
        for (int64_t i = 0; i < s.H; ++i) t[(size_t)(i + j * s.H)] = zbar[i + j * s.H] / half_weight(s, j);
    inv_rfft2(t.data(), s, ybar);
    const double n = (double)s.n();
    for (int64_t k = 0; k < s.n(); ++k) ybar[k] *= n;
}
// Reverse mode of inv_rfft(2): cotangent of the real lattice -> cotangent of the half spectrum,
// zbar_k = weight_k / n * rfft2(xbar)_k (every row of the self-mirrored columns is an input of its own: util.stan:320).
inline void inv_rfft2_vjp(const double* xbar, Shape s, cplx* zbar) {
    rfft2(xbar, s, zbar);
    const double inv_n = 1.0 / (double)s.n();
    for (int64_t j = 0; j < s.wf(); ++j)
        for (int64_t i = 0; i < s.H; ++i) zbar[i + j * s.H] *= half_weight(s, j) * inv_n;
}
// Reverse mode of expand_rfft: out[k] = y[k] (k <= n/2), out[n-k] = conj(y[k]) (k = 1..(n-1)/2)
inline void expand_rfft_vjp(const cplx* outbar, int64_t n, cplx* ybar) {
    const int64_t nrfft = n / 2 + 1, nc = (n - 1) / 2;
    for (int64_t k = 0; k < nrfft; ++k) ybar[k] = outbar[k];
    for (int64_t k = 1; k <= nc; ++k) ybar[k] += std::conj(outbar[n - k]);
}

// ===================================================================================================
// Scales and layout helpers: fft1.stan:14-51,108-118; fft2.stan:13-100
// gp_evaluate_rfft_scale (fft1.stan:14-30) / gp_evaluate_rfft2_scale (fft2.stan:13-40); rejects min(n c / 2) < 0
inline void gp_evaluate_scale(const double* cov, Shape s, double* out) {
    check(gpt_gp_evaluate_rfft2_scale_host(ctx(), GPT_COL_MAJOR, cov, s.H, s.W, s.dim1, out),
          s.dim1 ? "gp_evaluate_rfft_scale" : "gp_evaluate_rfft2_scale");
}
// scale = sqrt(m n c / 2), m in {1, 2}: d scale / d c = scale / (2 c)
inline void gp_evaluate_scale_vjp(const double* cov, const double* scale, const double* bar, Shape s, double* covbar) {
    for (int64_t k = 0; k < s.nf(); ++k) covbar[k] = bar[k] * scale[k] / (2.0 * cov[k]);
}
// gp_unpack_rfft (fft1.stan:44-51) / gp_unpack_rfft2 (fft2.stan:55-74)
inline void gp_unpack(const cplx* z, Shape s, double* out) {
    check(gpt_gp_unpack_rfft2_host(ctx(), GPT_COL_MAJOR, re(z), s.H, s.W, s.dim1, out),
          s.dim1 ? "gp_unpack_rfft" : "gp_unpack_rfft2");
}
// gp_pack_rfft (fft1.stan:108-118) / gp_pack_rfft2 (fft2.stan:85-100)
inline void gp_pack(const double* z, Shape s, cplx* out) {
    check(gpt_gp_pack_rfft2_host(ctx(), GPT_COL_MAJOR, z, s.H, s.W, s.dim1, re(out)), s.dim1 ? "gp_pack_rfft" : "gp_pack_rfft2");
}

namespace detail {
// 1-D building blocks on strided vectors (stride in elements), fft1.stan:44-51 and :108-118, as adjoints:
// unpack: z[k] = Re x[k] (k <= n/2), z[n/2 + k] = Im x[k] (k = 1..(n-1)/2)   =>   xbar[k] = zbar[k] + i zbar[n/2 + k]
inline void unpack1_vjp(const double* zbar, int64_t zs, int64_t n, cplx* xbar, int64_t xs) {
    const int64_t nrfft = n / 2 + 1, nc = (n - 1) / 2;
    for (int64_t k = 0; k < nrfft; ++k)
        xbar[k * xs] = cplx(zbar[k * zs], (k >= 1 && k <= nc) ? zbar[(nrfft + k - 1) * zs] : 0.0);
}
// pack: x[k] = z[k] + i z[n/2 + k] (imaginary part for k = 1..(n-1)/2 only)   =>   zbar[k] = Re xbar[k], zbar[n/2+k] = Im xbar[k]
inline void pack1_vjp(const cplx* xbar, int64_t xs, int64_t n, double* zbar, int64_t zs) {
    const int64_t nrfft = n / 2 + 1, nc = (n - 1) / 2;
    for (int64_t k = 0; k < nrfft; ++k) zbar[k * zs] = xbar[k * xs].real();
    for (int64_t k = 1; k <= nc; ++k) zbar[(nrfft + k - 1) * zs] = xbar[k * xs].imag();
}
} // namespace detail

// Reverse mode of gp_unpack_rfft(2): cotangent of the real layout -> cotangent of the half spectrum (zero where the
// reference never reads: rows > H/2 of the self-mirrored columns, fft2.stan:65,71)
inline void gp_unpack_vjp(const double* bar, Shape s, cplx* zbar) {
    if (s.dim1) {
        detail::unpack1_vjp(bar, 1, s.W, zbar, 1);
        return;
    }
    const int64_t H = s.H, W = s.W, wc = (W - 1) / 2;
    for (int64_t k = 0; k < s.nf(); ++k) zbar[k] = cplx(0.0, 0.0);
    detail::unpack1_vjp(bar, 1, H, zbar, 1); // column 0 <- column 0 of the real layout, along the height
    for (int64_t j = 1; j <= wc; ++j)
        for (int64_t i = 0; i < H; ++i) zbar[i + j * H] = cplx(bar[i + j * H], bar[i + (j + wc) * H]);
    if (W % 2 == 0) detail::unpack1_vjp(bar + (W - 1) * H, 1, H, zbar + (W / 2) * H, 1);
}
// Reverse mode of gp_pack_rfft(2): cotangent of the half spectrum -> cotangent of the real layout
inline void gp_pack_vjp(const cplx* bar, Shape s, double* zbar) {
    if (s.dim1) {
        detail::pack1_vjp(bar, 1, s.W, zbar, 1);
        return;
    }
    const int64_t H = s.H, W = s.W, wc = (W - 1) / 2;
    std::vector<cplx> col((size_t)(H / 2 + 1));
    expand_rfft_vjp(bar, H, col.data()); // column 0 = expand_rfft(pack_rfft(z[:, 0]), H), fft2.stan:91
    detail::pack1_vjp(col.data(), 1, H, zbar, 1);
    for (int64_t j = 1; j <= wc; ++j)
        for (int64_t i = 0; i < H; ++i) {
            zbar[i + j * H] = bar[i + j * H].real();
            zbar[i + (j + wc) * H] = bar[i + j * H].imag();
        }
    if (W % 2 == 0) {
        expand_rfft_vjp(bar + (W / 2) * H, H, col.data());
        detail::pack1_vjp(col.data(), 1, H, zbar + (W - 1) * H, 1);
    }
}

// ===================================================================================================
// The differentiable GP functions
// gp_rfft (fft1.stan:64-66) / gp_rfft2 (fft2.stan:112-114) when `forward`, gp_inv_rfft (fft1.stan:139-143) /
// gp_inv_rfft2 (fft2.stan:126-130) otherwise. a = y or z.
inline void gp_transform(bool forward, const double* a, const double* loc, const double* cov, Shape s, double* out) {
    check(forward ? gpt_gp_rfft2_host(ctx(), GPT_COL_MAJOR, a, loc, cov, s.H, s.W, s.dim1, out)
                  : gpt_gp_inv_rfft2_host(ctx(), GPT_COL_MAJOR, a, loc, cov, s.H, s.W, s.dim1, out),
          forward ? (s.dim1 ? "gp_rfft" : "gp_rfft2") : (s.dim1 ? "gp_inv_rfft" : "gp_inv_rfft2"));
}
// Reverse mode: bar = cotangent of the output -> abar (of y or z), covbar (may be null). The cotangent of loc is -abar
// (forward) or bar itself (inverse).
inline void gp_transform_vjp(bool forward, const double* a, const double* loc, const double* cov, const double* bar, Shape s,
                             double* abar, double* covbar) {
    check(forward ? gpt_gp_rfft2_vjp_host(ctx(), GPT_COL_MAJOR, a, loc, cov, bar, s.H, s.W, s.dim1, abar, covbar)
                  : gpt_gp_inv_rfft2_vjp_host(ctx(), GPT_COL_MAJOR, a, cov, bar, s.H, s.W, s.dim1, abar, covbar),
          forward ? "gp_rfft(2) reverse mode" : "gp_inv_rfft(2) reverse mode");
}
// gp_rfft_lpdf (fft1.stan:96-101) / gp_rfft2_lpdf (fft2.stan:189-192), all constants kept. grad_y null: value only;
// otherwise ONE fused trip to the GPU returns the value, d/dy (d/dloc = -d/dy) and, if asked, d/dcov.
inline double gp_lpdf(const double* y, const double* loc, const double* cov, Shape s, double* grad_y, double* grad_cov) {
    double value = 0.0;
    check(gpt_gp_rfft2_lpdf_host(ctx(), GPT_COL_MAJOR, y, loc, cov, s.H, s.W, s.dim1, &value, grad_y, grad_cov),
          s.dim1 ? "gp_rfft_lpdf" : "gp_rfft2_lpdf");
    return value;
}
// Extension: the same log density over a spectrum given by its kernel (gpt_gp_rfft2_lpdf_kernel_host): par / pbar as for
// spectrum() below; grad_y and pbar may be null.
inline double gp_kernel_lpdf(bool matern, double nu, const double* y, const double* loc, const double* par, double nugget,
                             Shape s, double* grad_y, double* pbar) {
    double value = 0.0;
    check(gpt_gp_rfft2_lpdf_kernel_host(ctx(), GPT_COL_MAJOR, matern ? 1 : 0, nu, y, loc, par, nugget, s.H, s.W, s.dim1, &value,
                                        grad_y, pbar),
          s.dim1 ? "gp_rfft_kernel_lpdf" : "gp_rfft2_kernel_lpdf");
    return value;
}
// gp_rfft_log_abs_det_jac (fft1.stan:78-82) / gp_rfft2_log_abs_det_jac (fft2.stan:143-173); grad_cov may be null
inline double gp_ladj(const double* cov, Shape s, double* grad_cov) {
    double value = 0.0;
    check(gpt_gp_rfft2_log_abs_det_jac_host(ctx(), GPT_COL_MAJOR, cov, s.H, s.W, s.dim1, &value, grad_cov),
          s.dim1 ? "gp_rfft_log_abs_det_jac" : "gp_rfft2_log_abs_det_jac");
    return value;
}
// Spectral kernels: gp_periodic_exp_quad_cov_rfft(2) (fft1.stan:154-158, fft2.stan:206-213) and
// gp_periodic_matern_cov_rfft(2) (fft1.stan:170-176, fft2.stan:228-247). par = (sigma, l, p) for 1-D,
// (sigma, l1, l2, p1, p2) for 2-D.
inline void spectrum(bool matern, double nu, Shape s, const double* par, double* out) {
    check(matern ? gpt_gp_periodic_matern_cov_rfft2_host(ctx(), GPT_COL_MAJOR, nu, s.H, s.W, par, s.dim1, out)
                 : gpt_gp_periodic_exp_quad_cov_rfft2_host(ctx(), GPT_COL_MAJOR, s.H, s.W, par, s.dim1, out),
          matern ? "gp_periodic_matern_cov_rfft(2)" : "gp_periodic_exp_quad_cov_rfft(2)");
}
// Reverse mode: covbar -> pbar (same layout as par); nu is data (the reference's models pass literals)
inline void spectrum_vjp(bool matern, double nu, Shape s, const double* par, const double* covbar, double* pbar) {
    check(matern ? gpt_gp_periodic_matern_cov_rfft2_vjp_host(ctx(), GPT_COL_MAJOR, nu, s.H, s.W, par, covbar, s.dim1, pbar)
                 : gpt_gp_periodic_exp_quad_cov_rfft2_vjp_host(ctx(), GPT_COL_MAJOR, s.H, s.W, par, covbar, s.dim1, pbar),
          "spectral kernel reverse mode");
}
// Extension (no counterpart in the reference): the non-centred model density of docs/trees/trees.stan:31-49 in one call
// (gpt_gp_inv_rfft2_obs_lpdf_host): value + d/dz, d/d(scalar loc), d/dcov (may be null), d/dtheta.
struct ObsGrad {
    double loc_sum, theta;
};
inline double obs_lpdf(int family, const double* z, double loc, const double* cov, const double* obs, int64_t oh, int64_t ow,
                       double theta, int prior, Shape s, double* grad_z, double* grad_cov, ObsGrad* g) {
    double value = 0.0, gls = 0.0, gth = 0.0;
    check(gpt_gp_inv_rfft2_obs_lpdf_host(ctx(), GPT_COL_MAJOR, family, z, nullptr, loc, cov, obs, oh, ow, theta, prior, s.H,
                                         s.W, s.dim1, &value, grad_z, &gls, grad_cov, &gth),
          s.dim1 ? "gp_inv_rfft_obs_lpdf" : "gp_inv_rfft2_obs_lpdf");
    if (g) *g = ObsGrad{gls, gth};
    return value;
}

// ... and with the spectrum given by its kernel parameters (gpt_gp_inv_rfft2_obs_lpdf_kernel_host): par / pbar as for
// spectrum(); pbar may be null.
inline double obs_kernel_lpdf(int family, bool matern, double nu, const double* z, double loc, const double* par, double nugget,
                              const double* obs, int64_t oh, int64_t ow, double theta, int prior, Shape s, double* grad_z,
                              double* pbar, ObsGrad* g) {
    double value = 0.0, gls = 0.0, gth = 0.0;
    check(gpt_gp_inv_rfft2_obs_lpdf_kernel_host(ctx(), GPT_COL_MAJOR, family, matern ? 1 : 0, nu, z, nullptr, loc, par, nugget, obs,
                                                oh, ow, theta, prior, s.H, s.W, s.dim1, &value, grad_z, &gls, pbar, &gth),
          s.dim1 ? "gp_inv_rfft_obs_kernel_lpdf" : "gp_inv_rfft2_obs_kernel_lpdf");
    if (g) *g = ObsGrad{gls, gth};
    return value;
}

} // namespace gptools_hip
#endif // GPTOOLS_HIP_BIND_HPP
