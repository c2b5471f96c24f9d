// gptools_hip.hpp -- C++ user header for cmdstan (USER_HEADER=...) that binds the Fourier-GP functions of
// onnela-lab/gptools, declared without bodies in gptools_amd/stan/gptools/*.stan, to libgptools_hip.so.
//
//   stanc --allow-undefined --include-paths=<gptools_amd/stan> model.stan
//   make  USER_HEADER=<repo>/include/gptools_hip.hpp  CXXFLAGS+=-I<repo>/include
//         LDFLAGS+="-L<repo>/gptools_amd -Wl,-rpath,<repo>/gptools_amd" LDLIBS+=-lgptools_hip
// (gptools_amd.stan.compile_model sets all of this).
//
// cmdstan includes a user header INSIDE the model's namespace, after the forward declarations stanc
// generated for the undefined functions, so nothing here names a namespace of its own. Every function is
// a template over its argument types (double or stan::math::var elements), takes the trailing
// `std::ostream* pstream__` stanc adds, and `_lpdf` functions carry the leading `bool propto__`.
//
// Values come from the host-pointer C ABI (include/gptools_hip.h, GPT_COL_MAJOR: Stan matrices are Eigen
// column-major); reverse mode registers ONE callback per call that asks the library for the vector-
// Jacobian product (the fused gp_rfft2_lpdf + gradient kernel for the log density).
//
// STATUS: written against the Stan Math 4.x API (cmdstan 2.36, the version the reference pins in
// .github/workflows/python.yml:19). The build image has neither stanc nor Stan Math, so this header has
// NOT been compiled here; the C ABI underneath it is what the GPU parity tests exercise (the same entry
// points, column-major, in tests/test_gpu_parity.py::test_host_entry_points_col_major).
#ifndef GPTOOLS_HIP_HPP
#define GPTOOLS_HIP_HPP

#include <gptools_hip.h>
#include <stan/math.hpp>

#include <ostream>
#include <stdexcept>
#include <string>

namespace gptools_hip_detail {

// One library context per calling thread (cmdstan chains are processes, or threads under STAN_THREADS).
inline gpt_ctx* ctx() {
    struct Holder {
        gpt_ctx* c = nullptr;
        Holder() {
            if (gpt_create(0, nullptr, &c) != GPT_OK) throw std::runtime_error("gptools_hip: no usable GPU");
        }
        ~Holder() { gpt_destroy(c); }
    };
    thread_local Holder h;
    return h.c;
}

// Map the library's status codes onto what the reference's Stan code does.
inline void check(int rc, const char* fn) {
    if (rc == GPT_OK) return;
    const std::string msg = std::string(fn) + ": " + gpt_last_error(ctx());
    if (rc == GPT_ERR_NOT_PD) throw std::domain_error(gpt_last_error(ctx()));  // Stan `reject`
    if (rc == GPT_ERR_SHAPE) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

template <typename T> using value_mat_t = Eigen::Matrix<double, T::RowsAtCompileTime, T::ColsAtCompileTime>;

template <typename T> inline auto val(const T& x) { return stan::math::value_of(x).eval(); }

template <typename... Ts> using any_var = stan::math::disjunction<stan::is_var<stan::scalar_type_t<Ts>>...>;

inline void require_no_var(bool has_var, const char* fn) {
    if (has_var)
        throw std::domain_error(std::string(fn) +
                                " is not differentiable through the HIP path; differentiate gp_rfft(2), "
                                "gp_inv_rfft(2), gp_rfft(2)_lpdf or the spectral kernels instead");
}

}  // namespace gptools_hip_detail

// ===================================================================================================
// Real-FFT layer (stan/gptools/util.stan:255-330): values only.
template <typename T0__>
inline Eigen::Matrix<std::complex<double>, -1, 1> rfft(const T0__& y, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    d::require_no_var(d::any_var<T0__>::value, "rfft");
    const Eigen::VectorXd yv = d::val(y);
    Eigen::Matrix<std::complex<double>, -1, 1> out(yv.size() / 2 + 1);
    d::check(gpt_rfft2_host(d::ctx(), GPT_COL_MAJOR, yv.data(), 1, yv.size(), reinterpret_cast<double*>(out.data())),
             "rfft");
    return out;
}

template <typename T0__>
inline Eigen::Matrix<std::complex<double>, -1, 1> expand_rfft(const T0__& y, const int& n, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    if (y.size() != n / 2 + 1)  // util.stan:274-276
        throw std::domain_error("expected complex vector with " + std::to_string(n / 2 + 1) + " elements but got " +
                                std::to_string(y.size()));
    const Eigen::Matrix<std::complex<double>, -1, 1> yv = y;
    Eigen::Matrix<std::complex<double>, -1, 1> out(n);
    d::check(gpt_expand_rfft_host(d::ctx(), reinterpret_cast<const double*>(yv.data()), n,
                                  reinterpret_cast<double*>(out.data())),
             "expand_rfft");
    return out;
}

template <typename T0__>
inline Eigen::VectorXd inv_rfft(const T0__& z, const int& n, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    const Eigen::Matrix<std::complex<double>, -1, 1> zv = z;
    Eigen::VectorXd out(n);
    d::check(gpt_inv_rfft2_host(d::ctx(), GPT_COL_MAJOR, reinterpret_cast<const double*>(zv.data()), 1, n, out.data()),
             "inv_rfft");
    return out;
}

template <typename T0__>
inline Eigen::Matrix<std::complex<double>, -1, -1> rfft2(const T0__& y, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    d::require_no_var(d::any_var<T0__>::value, "rfft2");
    const Eigen::MatrixXd yv = d::val(y);
    Eigen::Matrix<std::complex<double>, -1, -1> out(yv.rows(), yv.cols() / 2 + 1);
    d::check(gpt_rfft2_host(d::ctx(), GPT_COL_MAJOR, yv.data(), yv.rows(), yv.cols(),
                            reinterpret_cast<double*>(out.data())),
             "rfft2");
    return out;
}

template <typename T0__>
inline Eigen::MatrixXd inv_rfft2(const T0__& z, const int& m, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    const Eigen::Matrix<std::complex<double>, -1, -1> zv = z;
    Eigen::MatrixXd out(zv.rows(), m);
    d::check(gpt_inv_rfft2_host(d::ctx(), GPT_COL_MAJOR, reinterpret_cast<const double*>(zv.data()), zv.rows(), m,
                                out.data()),
             "inv_rfft2");
    return out;
}

// ===================================================================================================
// Layout helpers and scales (fft1.stan:14-51,108-118; fft2.stan:13-100): values only.
#define GPTOOLS_HIP_SCALE(NAME, MAT, DIM1)                                                                       \
    template <typename T0__> inline MAT NAME(const T0__& cov, const int& n, std::ostream* pstream__) {          \
        namespace d = gptools_hip_detail;                                                                        \
        d::require_no_var(d::any_var<T0__>::value, #NAME);                                                       \
        const MAT cv = d::val(cov);                                                                              \
        MAT out(cv.rows(), cv.cols());                                                                           \
        d::check(gpt_gp_evaluate_rfft2_scale_host(d::ctx(), GPT_COL_MAJOR, cv.data(), DIM1 ? 1 : cv.rows(), n,   \
                                                  DIM1, out.data()),                                             \
                 #NAME);                                                                                         \
        return out;                                                                                              \
    }
GPTOOLS_HIP_SCALE(gp_evaluate_rfft_scale, Eigen::VectorXd, 1)
GPTOOLS_HIP_SCALE(gp_evaluate_rfft2_scale, Eigen::MatrixXd, 0)
#undef GPTOOLS_HIP_SCALE

template <typename T0__>
inline Eigen::VectorXd gp_unpack_rfft(const T0__& x, const int& n, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    const Eigen::Matrix<std::complex<double>, -1, 1> xv = x;
    Eigen::VectorXd out(n);
    d::check(gpt_gp_unpack_rfft2_host(d::ctx(), GPT_COL_MAJOR, reinterpret_cast<const double*>(xv.data()), 1, n, 1,
                                      out.data()),
             "gp_unpack_rfft");
    return out;
}

template <typename T0__>
inline Eigen::MatrixXd gp_unpack_rfft2(const T0__& z, const int& m, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    const Eigen::Matrix<std::complex<double>, -1, -1> zv = z;
    Eigen::MatrixXd out(zv.rows(), m);
    d::check(gpt_gp_unpack_rfft2_host(d::ctx(), GPT_COL_MAJOR, reinterpret_cast<const double*>(zv.data()), zv.rows(),
                                      m, 0, out.data()),
             "gp_unpack_rfft2");
    return out;
}

template <typename T0__>
inline Eigen::Matrix<std::complex<double>, -1, 1> gp_pack_rfft(const T0__& z, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    d::require_no_var(d::any_var<T0__>::value, "gp_pack_rfft");
    const Eigen::VectorXd zv = d::val(z);
    Eigen::Matrix<std::complex<double>, -1, 1> out(zv.size() / 2 + 1);
    d::check(gpt_gp_pack_rfft2_host(d::ctx(), GPT_COL_MAJOR, zv.data(), 1, zv.size(), 1,
                                    reinterpret_cast<double*>(out.data())),
             "gp_pack_rfft");
    return out;
}

template <typename T0__>
inline Eigen::Matrix<std::complex<double>, -1, -1> gp_pack_rfft2(const T0__& z, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    d::require_no_var(d::any_var<T0__>::value, "gp_pack_rfft2");
    const Eigen::MatrixXd zv = d::val(z);
    Eigen::Matrix<std::complex<double>, -1, -1> out(zv.rows(), zv.cols() / 2 + 1);
    d::check(gpt_gp_pack_rfft2_host(d::ctx(), GPT_COL_MAJOR, zv.data(), zv.rows(), zv.cols(), 0,
                                    reinterpret_cast<double*>(out.data())),
             "gp_pack_rfft2");
    return out;
}

// ===================================================================================================
// The differentiable functions. MAT is Eigen::VectorXd (1-D, dim1 = 1, "height" 1) or Eigen::MatrixXd.
namespace gptools_hip_detail {

template <typename MAT> inline int64_t height_of(const MAT& m, int dim1) { return dim1 ? 1 : m.rows(); }
template <typename MAT> inline int64_t width_of(const MAT& m, int dim1) { return dim1 ? m.size() : m.cols(); }

// result matrix type: var elements as soon as one argument has them
template <typename MAT, typename... Ts>
using result_mat_t = std::conditional_t<any_var<Ts...>::value,
                                        Eigen::Matrix<stan::math::var, MAT::RowsAtCompileTime, MAT::ColsAtCompileTime>,
                                        MAT>;

// z = gp_rfft(2)(y, loc, cov) or f = gp_inv_rfft(2)(z, loc, cov); FORWARD selects which.
template <bool FORWARD, typename MAT, typename T0, typename T1, typename T2>
inline result_mat_t<MAT, T0, T1, T2> transform(const T0& a0, const T1& loc, const T2& cov, int dim1, const char* fn) {
    using stan::math::var;
    const MAT av = val(a0), lv = val(loc), cv = val(cov);
    const int64_t H = height_of(av, dim1), W = width_of(av, dim1);
    if (lv.size() != av.size() || cv.size() != H * (W / 2 + 1))
        throw std::invalid_argument(std::string(fn) + ": argument sizes do not match");
    MAT out(av.rows(), av.cols());
    check(FORWARD ? gpt_gp_rfft2_host(ctx(), GPT_COL_MAJOR, av.data(), lv.data(), cv.data(), H, W, dim1, out.data())
                  : gpt_gp_inv_rfft2_host(ctx(), GPT_COL_MAJOR, av.data(), lv.data(), cv.data(), H, W, dim1, out.data()),
          fn);
    if constexpr (!any_var<T0, T1, T2>::value) {
        return out;
    } else {
        using stan::math::arena_t;
        arena_t<Eigen::Matrix<var, MAT::RowsAtCompileTime, MAT::ColsAtCompileTime>> res = out;
        auto arena_a = stan::math::to_arena(a0);
        auto arena_l = stan::math::to_arena(loc);
        auto arena_c = stan::math::to_arena(cov);
        stan::math::reverse_pass_callback([res, arena_a, arena_l, arena_c, dim1, H, W, fn]() mutable {
            const MAT bar = res.adj();
            const MAT av2 = stan::math::value_of(arena_a), lv2 = stan::math::value_of(arena_l),
                      cv2 = stan::math::value_of(arena_c);
            MAT abar(av2.rows(), av2.cols()), cbar(cv2.rows(), cv2.cols());
            if (FORWARD)  // zbar -> ybar, covbar ; locbar = -ybar
                check(gpt_gp_rfft2_vjp_host(ctx(), GPT_COL_MAJOR, av2.data(), lv2.data(), cv2.data(), bar.data(), H, W,
                                            dim1, abar.data(), cbar.data()),
                      fn);
            else  // fbar -> zbar, covbar ; locbar = fbar
                check(gpt_gp_inv_rfft2_vjp_host(ctx(), GPT_COL_MAJOR, av2.data(), cv2.data(), bar.data(), H, W, dim1,
                                                abar.data(), cbar.data()),
                      fn);
            if constexpr (!stan::is_constant<T0>::value) arena_a.adj() += abar;
            if constexpr (!stan::is_constant<T1>::value) {
                if (FORWARD) arena_l.adj() -= abar;
                else arena_l.adj() += bar;
            }
            if constexpr (!stan::is_constant<T2>::value) arena_c.adj() += cbar;
        });
        return Eigen::Matrix<var, MAT::RowsAtCompileTime, MAT::ColsAtCompileTime>(res);
    }
}

template <typename MAT, typename T0, typename T1, typename T2>
inline stan::return_type_t<T0, T1, T2> lpdf(const T0& y, const T1& loc, const T2& cov, int dim1, const char* fn) {
    using stan::math::var;
    const MAT yv = val(y), lv = val(loc), cv = val(cov);
    const int64_t H = height_of(yv, dim1), W = width_of(yv, dim1);
    if (lv.size() != yv.size() || cv.size() != H * (W / 2 + 1))
        throw std::invalid_argument(std::string(fn) + ": argument sizes do not match");
    double value = 0.0;
    if constexpr (!any_var<T0, T1, T2>::value) {
        check(gpt_gp_rfft2_lpdf_host(ctx(), GPT_COL_MAJOR, yv.data(), lv.data(), cv.data(), H, W, dim1, &value, nullptr,
                                     nullptr),
              fn);
        return value;
    } else {
        // fused value + gradient: one trip to the GPU per log-density evaluation
        stan::math::arena_t<MAT> gy(yv.rows(), yv.cols()), gc(cv.rows(), cv.cols());
        check(gpt_gp_rfft2_lpdf_host(ctx(), GPT_COL_MAJOR, yv.data(), lv.data(), cv.data(), H, W, dim1, &value,
                                     gy.data(), stan::is_constant<T2>::value ? nullptr : gc.data()),
              fn);
        auto arena_y = stan::math::to_arena(y);
        auto arena_l = stan::math::to_arena(loc);
        auto arena_c = stan::math::to_arena(cov);
        return stan::math::make_callback_var(value, [arena_y, arena_l, arena_c, gy, gc](auto& vi) mutable {
            if constexpr (!stan::is_constant<T0>::value) arena_y.adj() += vi.adj() * gy;
            if constexpr (!stan::is_constant<T1>::value) arena_l.adj() -= vi.adj() * gy;
            if constexpr (!stan::is_constant<T2>::value) arena_c.adj() += vi.adj() * gc;
        });
    }
}

template <typename MAT, typename T0>
inline stan::return_type_t<T0> ladj(const T0& cov, int width, int dim1, const char* fn) {
    const MAT cv = val(cov);
    const int64_t H = dim1 ? 1 : cv.rows();
    double value = 0.0;
    if constexpr (!any_var<T0>::value) {
        check(gpt_gp_rfft2_log_abs_det_jac_host(ctx(), GPT_COL_MAJOR, cv.data(), H, width, dim1, &value, nullptr), fn);
        return value;
    } else {
        stan::math::arena_t<MAT> gc(cv.rows(), cv.cols());
        check(gpt_gp_rfft2_log_abs_det_jac_host(ctx(), GPT_COL_MAJOR, cv.data(), H, width, dim1, &value, gc.data()), fn);
        auto arena_c = stan::math::to_arena(cov);
        return stan::math::make_callback_var(value,
                                             [arena_c, gc](auto& vi) mutable { arena_c.adj() += vi.adj() * gc; });
    }
}

// Spectral kernels: params = (sigma, l1, l2, p1, p2) [2-D] or (sigma, l, p) [1-D]
template <typename MAT, bool MATERN, typename... Ts>
inline result_mat_t<MAT, Ts...> spectrum(double nu, int64_t H, int64_t W, int dim1, const double* par, const char* fn,
                                         const std::function<void(const double*)>& scatter) {
    using stan::math::var;
    MAT out;
    if (dim1) out.resize(W / 2 + 1, 1);
    else out.resize(H, W / 2 + 1);
    check(MATERN ? gpt_gp_periodic_matern_cov_rfft2_host(ctx(), GPT_COL_MAJOR, nu, H, W, par, dim1, out.data())
                 : gpt_gp_periodic_exp_quad_cov_rfft2_host(ctx(), GPT_COL_MAJOR, H, W, par, dim1, out.data()),
          fn);
    if constexpr (!any_var<Ts...>::value) {
        return out;
    } else {
        stan::math::arena_t<Eigen::Matrix<var, MAT::RowsAtCompileTime, MAT::ColsAtCompileTime>> res = out;
        const int npar = dim1 ? 3 : 5;
        stan::math::arena_t<Eigen::VectorXd> p(npar);
        for (int i = 0; i < npar; ++i) p[i] = par[i];
        stan::math::reverse_pass_callback([res, p, nu, H, W, dim1, fn, scatter]() mutable {
            const MAT cbar = res.adj();
            double pbar[5] = {0, 0, 0, 0, 0};
            check(MATERN ? gpt_gp_periodic_matern_cov_rfft2_vjp_host(ctx(), GPT_COL_MAJOR, nu, H, W, p.data(),
                                                                     cbar.data(), dim1, pbar)
                         : gpt_gp_periodic_exp_quad_cov_rfft2_vjp_host(ctx(), GPT_COL_MAJOR, H, W, p.data(),
                                                                       cbar.data(), dim1, pbar),
                  fn);
            scatter(pbar);
        });
        return Eigen::Matrix<var, MAT::RowsAtCompileTime, MAT::ColsAtCompileTime>(res);
    }
}

template <typename T> inline void add_adj(const T& x, double g) {
    if constexpr (stan::is_var<T>::value) x.adj() += g;
}

}  // namespace gptools_hip_detail

// ---- fft1.stan ---------------------------------------------------------------------------------------
template <typename T0__, typename T1__, typename T2__>
inline auto gp_rfft(const T0__& y, const T1__& loc, const T2__& cov_rfft, std::ostream* pstream__) {
    return gptools_hip_detail::transform<true, Eigen::VectorXd>(y, loc, cov_rfft, 1, "gp_rfft");
}
template <typename T0__, typename T1__, typename T2__>
inline auto gp_inv_rfft(const T0__& z, const T1__& loc, const T2__& cov_rfft, std::ostream* pstream__) {
    return gptools_hip_detail::transform<false, Eigen::VectorXd>(z, loc, cov_rfft, 1, "gp_inv_rfft");
}
template <typename T0__>
inline auto gp_rfft_log_abs_det_jac(const T0__& cov_rfft, const int& n, std::ostream* pstream__) {
    return gptools_hip_detail::ladj<Eigen::VectorXd>(cov_rfft, n, 1, "gp_rfft_log_abs_det_jac");
}
// called as a function in the reference (`y ~ gp_rfft(...)` drops nothing: fft1.stan:100 keeps constants)
template <bool propto__, typename T0__, typename T1__, typename T2__>
inline auto gp_rfft_lpdf(const T0__& y, const T1__& loc, const T2__& cov_rfft, std::ostream* pstream__) {
    return gptools_hip_detail::lpdf<Eigen::VectorXd>(y, loc, cov_rfft, 1, "gp_rfft_lpdf");
}

template <typename T1__, typename T2__, typename T3__>
inline auto gp_periodic_exp_quad_cov_rfft(const int& n, const T1__& sigma, const T2__& length_scale,
                                          const T3__& period, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    using stan::math::value_of;
    const double par[3] = {value_of(sigma), value_of(length_scale), value_of(period)};
    return d::spectrum<Eigen::VectorXd, false, T1__, T2__, T3__>(0.0, 1, n, 1, par, "gp_periodic_exp_quad_cov_rfft",
                                                                 [sigma, length_scale, period](const double* g) {
                                                                     d::add_adj(sigma, g[0]);
                                                                     d::add_adj(length_scale, g[1]);
                                                                     d::add_adj(period, g[2]);
                                                                 });
}
template <typename T0__, typename T2__, typename T3__, typename T4__>
inline auto gp_periodic_matern_cov_rfft(const T0__& nu, const int& n, const T2__& sigma, const T3__& length_scale,
                                        const T4__& period, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    using stan::math::value_of;
    d::require_no_var(stan::is_var<T0__>::value, "gp_periodic_matern_cov_rfft (d/dnu)");
    const double par[3] = {value_of(sigma), value_of(length_scale), value_of(period)};
    return d::spectrum<Eigen::VectorXd, true, T2__, T3__, T4__>(value_of(nu), 1, n, 1, par,
                                                                "gp_periodic_matern_cov_rfft",
                                                                [sigma, length_scale, period](const double* g) {
                                                                    d::add_adj(sigma, g[0]);
                                                                    d::add_adj(length_scale, g[1]);
                                                                    d::add_adj(period, g[2]);
                                                                });
}

// ---- fft2.stan ---------------------------------------------------------------------------------------
template <typename T0__, typename T1__, typename T2__>
inline auto gp_rfft2(const T0__& y, const T1__& loc, const T2__& cov_rfft2, std::ostream* pstream__) {
    return gptools_hip_detail::transform<true, Eigen::MatrixXd>(y, loc, cov_rfft2, 0, "gp_rfft2");
}
template <typename T0__, typename T1__, typename T2__>
inline auto gp_inv_rfft2(const T0__& z, const T1__& loc, const T2__& cov_rfft2, std::ostream* pstream__) {
    return gptools_hip_detail::transform<false, Eigen::MatrixXd>(z, loc, cov_rfft2, 0, "gp_inv_rfft2");
}
template <typename T0__>
inline auto gp_rfft2_log_abs_det_jac(const T0__& cov_rfft2, const int& width, std::ostream* pstream__) {
    return gptools_hip_detail::ladj<Eigen::MatrixXd>(cov_rfft2, width, 0, "gp_rfft2_log_abs_det_jac");
}
template <bool propto__, typename T0__, typename T1__, typename T2__>
inline auto gp_rfft2_lpdf(const T0__& y, const T1__& loc, const T2__& cov_rfft2, std::ostream* pstream__) {
    return gptools_hip_detail::lpdf<Eigen::MatrixXd>(y, loc, cov_rfft2, 0, "gp_rfft2_lpdf");
}

template <typename T2__, typename T3__, typename T4__>
inline auto gp_periodic_exp_quad_cov_rfft2(const int& height, const int& width, const T2__& sigma,
                                           const T3__& length_scale, const T4__& period, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    using stan::math::value_of;
    auto ls = stan::math::to_arena(length_scale);
    auto pd = stan::math::to_arena(period);
    const double par[5] = {value_of(sigma), value_of(ls[0]), value_of(ls[1]), value_of(pd[0]), value_of(pd[1])};
    return d::spectrum<Eigen::MatrixXd, false, T2__, T3__, T4__>(0.0, height, width, 0, par,
                                                                 "gp_periodic_exp_quad_cov_rfft2",
                                                                 [sigma, ls, pd](const double* g) mutable {
                                                                     d::add_adj(sigma, g[0]);
                                                                     d::add_adj(ls.coeffRef(0), g[1]);
                                                                     d::add_adj(ls.coeffRef(1), g[2]);
                                                                     d::add_adj(pd.coeffRef(0), g[3]);
                                                                     d::add_adj(pd.coeffRef(1), g[4]);
                                                                 });
}
template <typename T0__, typename T3__, typename T4__, typename T5__>
inline auto gp_periodic_matern_cov_rfft2(const T0__& nu, const int& height, const int& width, const T3__& sigma,
                                         const T4__& length_scale, const T5__& period, std::ostream* pstream__) {
    namespace d = gptools_hip_detail;
    using stan::math::value_of;
    d::require_no_var(stan::is_var<T0__>::value, "gp_periodic_matern_cov_rfft2 (d/dnu)");
    auto ls = stan::math::to_arena(length_scale);
    auto pd = stan::math::to_arena(period);
    const double par[5] = {value_of(sigma), value_of(ls[0]), value_of(ls[1]), value_of(pd[0]), value_of(pd[1])};
    return d::spectrum<Eigen::MatrixXd, true, T3__, T4__, T5__>(value_of(nu), height, width, 0, par,
                                                                "gp_periodic_matern_cov_rfft2",
                                                                [sigma, ls, pd](const double* g) mutable {
                                                                    d::add_adj(sigma, g[0]);
                                                                    d::add_adj(ls.coeffRef(0), g[1]);
                                                                    d::add_adj(ls.coeffRef(1), g[2]);
                                                                    d::add_adj(pd.coeffRef(0), g[3]);
                                                                    d::add_adj(pd.coeffRef(1), g[4]);
                                                                });
}

// ---- extension: non-centred model density in one call (gptools/fft_obs.stan; include/gptools_hip.h,
// gpt_gp_inv_rfft2_obs_lpdf_host). No counterpart in the reference, where docs/trees/trees.stan:31-49 spells the same
// density as gp_inv_rfft2 + std_normal + a masked neg_binomial_2 loop. Only z, cov and the data travel to the GPU and
// only the adjoints of z and cov (and four scalars) come back.
namespace gptools_hip_detail {
template <typename MAT, typename T0, typename TOBS, typename TLOC, typename TCOV, typename TTH>
inline stan::return_type_t<T0, TLOC, TCOV, TTH> obs_lpdf(const T0& z, int family, const TOBS& obs, const TLOC& loc,
                                                         const TCOV& cov, const TTH& theta, int prior, int dim1,
                                                         const char* fn) {
    const MAT zv = val(z), ov = val(obs), cv = val(cov);
    const int64_t H = height_of(zv, dim1), W = width_of(zv, dim1);
    const int64_t oh = height_of(ov, dim1), ow = width_of(ov, dim1);
    if (cv.size() != H * (W / 2 + 1) || oh > H || ow > W || oh < 1 || ow < 1)
        throw std::invalid_argument(std::string(fn) + ": argument sizes do not match");
    const double lv = stan::math::value_of(loc), tv = stan::math::value_of(theta);
    double value = 0.0, gls = 0.0, gth = 0.0;
    stan::math::arena_t<MAT> gz(zv.rows(), zv.cols()), gc(cv.rows(), cv.cols());
    check(gpt_gp_inv_rfft2_obs_lpdf_host(ctx(), GPT_COL_MAJOR, family, zv.data(), nullptr, lv, cv.data(), ov.data(), oh, ow,
                                         tv, prior, H, W, dim1, &value, gz.data(), &gls,
                                         stan::is_constant<TCOV>::value ? nullptr : gc.data(), &gth),
          fn);
    if constexpr (!any_var<T0, TLOC, TCOV, TTH>::value) {
        return value;
    } else {
        auto arena_z = stan::math::to_arena(z);
        auto arena_c = stan::math::to_arena(cov);
        return stan::math::make_callback_var(value, [arena_z, arena_c, loc, theta, gz, gc, gls, gth](auto& vi) mutable {
            if constexpr (!stan::is_constant<T0>::value) arena_z.adj() += vi.adj() * gz;
            if constexpr (!stan::is_constant<TCOV>::value) arena_c.adj() += vi.adj() * gc;
            add_adj(loc, vi.adj() * gls);
            add_adj(theta, vi.adj() * gth);
        });
    }
}
}  // namespace gptools_hip_detail

template <bool propto__, typename T0__, typename T2__, typename T3__, typename T4__, typename T5__>
inline auto gp_inv_rfft2_obs_lpdf(const T0__& z, const int& family, const T2__& obs, const T3__& loc,
                                  const T4__& cov_rfft2, const T5__& theta, const int& prior, std::ostream* pstream__) {
    return gptools_hip_detail::obs_lpdf<Eigen::MatrixXd>(z, family, obs, loc, cov_rfft2, theta, prior, 0,
                                                         "gp_inv_rfft2_obs_lpdf");
}
template <bool propto__, typename T0__, typename T2__, typename T3__, typename T4__, typename T5__>
inline auto gp_inv_rfft_obs_lpdf(const T0__& z, const int& family, const T2__& obs, const T3__& loc,
                                 const T4__& cov_rfft, const T5__& theta, const int& prior, std::ostream* pstream__) {
    return gptools_hip_detail::obs_lpdf<Eigen::VectorXd>(z, family, obs, loc, cov_rfft, theta, prior, 1,
                                                         "gp_inv_rfft_obs_lpdf");
}

#endif  // GPTOOLS_HIP_HPP
