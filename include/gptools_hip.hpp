// gptools_hip.hpp -- C++ user header for cmdstan (USER_HEADER=...): the Stan-typed veneer that binds the Fourier-GP
// functions of onnela-lab/gptools, declared without bodies in gptools_amd/stan/gptools/*.stan, to libgptools_hip.so.
//
//   stanc --allow-undefined --include-paths=<gptools_amd/stan> model.stan
//   make  USER_HEADER=<repo>/include/gptools_hip.hpp  CXXFLAGS+=-I<repo>/include
//         LDFLAGS+="-L<repo>/gptools_amd -Wl,-rpath,<repo>/gptools_amd" LDLIBS+=-lgptools_hip
// (gptools_amd.stan.compile_model sets all of this).
//
// Everything that does not name a Stan type -- context, error mapping, every value and reverse-mode call sequence on
// column-major pointers -- lives in gptools_hip_bind.hpp, which IS compiled and run by this repository's tests
// (tests/test_gpu_bind_cpp.py). What is left here is mechanical: unwrap Stan's arguments into plain column-major
// matrices, call one bind function, and register ONE reverse-pass callback that calls the matching `*_vjp`.
//
// cmdstan includes a user header INSIDE the model's namespace, after the forward declarations stanc generated for the
// undefined functions. A definition only binds to such a declaration if its template head is the same, so every
// function below repeats the form stanc 2.36 emits: one type parameter per non-int argument, a trailing
// `stan::require_all_t<...>* = nullptr` listing `stan::is_col_vector` / `stan::is_eigen_matrix_dynamic` +
// `stan::is_vt_not_complex` / `stan::is_vt_complex` per container and `stan::is_stan_scalar` per real, a return type
// spelled with `stan::promote_args_t<stan::base_type_t<...>>`, `const int&` for ints, the trailing
// `std::ostream* pstream__`, and a leading `bool propto__` on `_lpdf` functions.
//
// STATUS -- EXPERIMENTAL, UNCOMPILED: the build image has neither stanc nor Stan Math (cmdstan 2.36 is what the
// reference pins, .github/workflows/python.yml:19), so THIS file has never met a compiler and the template heads are
// written from the stanc3 code generator's documented output, not checked against a generated .hpp. If a model fails
// to link with "undefined reference to <function>" or "call is ambiguous", diff the forward declaration in the
// generated <model>.hpp against the head here. All functions accept `var` arguments,
// including the real-FFT layer and the layout helpers (linear maps: their reverse mode is the adjoint transform).
#ifndef GPTOOLS_HIP_HPP
#define GPTOOLS_HIP_HPP

#include <gptools_hip_bind.hpp>
#include <stan/math.hpp>

#include <ostream>

namespace gptools_hip_stan {
namespace b = gptools_hip;
using stan::math::var;
using b::cplx;
using VecD = Eigen::Matrix<double, -1, 1>;
using MatD = Eigen::Matrix<double, -1, -1>;
using VecC = Eigen::Matrix<cplx, -1, 1>;
using MatC = Eigen::Matrix<cplx, -1, -1>;
template <typename S, int C> using Mat = Eigen::Matrix<S, -1, C>;

template <typename T> using scalar_of = stan::promote_args_t<stan::base_type_t<T>>;
template <typename... Ts> inline constexpr bool any_var_v = stan::is_var<stan::return_type_t<Ts...>>::value;
template <typename T> inline constexpr bool is_var_v = stan::is_var<stan::base_type_t<T>>::value;

// plain column-major copies of the values of (possibly var, possibly expression) arguments
template <int C, typename T> inline Mat<double, C> real_val(const T& x) { return stan::math::value_of(x); }
template <int C, typename T> inline Mat<cplx, C> cplx_val(const T& x) { return stan::math::value_of(x); }

// a complex<var> container over two arena matrices of var (real and imaginary parts keep their own varis)
template <int C> struct CplxVarOut {
    stan::math::arena_t<Mat<var, C>> re, im;
    explicit CplxVarOut(const Mat<cplx, C>& v) : re(v.real()), im(v.imag()) {}
    Mat<std::complex<var>, C> result() const {
        Mat<std::complex<var>, C> r(re.rows(), re.cols());
        for (Eigen::Index k = 0; k < r.size(); ++k) r.coeffRef(k) = std::complex<var>(re.coeff(k), im.coeff(k));
        return r;
    }
    Mat<cplx, C> adj() const {
        Mat<cplx, C> a(re.rows(), re.cols());
        for (Eigen::Index k = 0; k < a.size(); ++k) a.coeffRef(k) = cplx(re.coeff(k).adj(), im.coeff(k).adj());
        return a;
    }
};
// add a complex cotangent to a container of complex<var> held in the arena
template <typename A, int C> inline void add_cplx_adj(A& z, const Mat<cplx, C>& g) {
    for (Eigen::Index k = 0; k < g.size(); ++k) {
        z.coeffRef(k).real().adj() += g.coeff(k).real();
        z.coeffRef(k).imag().adj() += g.coeff(k).imag();
    }
}

// ---- real input -> complex output (rfft, rfft2, gp_pack_rfft, gp_pack_rfft2) ---------------------------------------
template <int C, typename T, typename Fwd, typename Vjp>
inline Mat<std::complex<scalar_of<T>>, C> real_to_cplx(const T& x, b::Shape s, Eigen::Index orows, Eigen::Index ocols,
                                                        Fwd fwd, Vjp vjp) {
    const Mat<double, C> xv = real_val<C>(x);
    Mat<cplx, C> out(orows, ocols);
    fwd(xv.data(), s, out.data());
    if constexpr (!is_var_v<T>) {
        return out;
    } else {
        CplxVarOut<C> res(out);
        auto ax = stan::math::to_arena(x);
        stan::math::reverse_pass_callback([res, ax, s, vjp]() mutable {
            const Mat<cplx, C> bar = res.adj();
            Mat<double, C> xbar(ax.rows(), ax.cols());
            vjp(bar.data(), s, xbar.data());
            ax.adj() += xbar;
        });
        return res.result();
    }
}
// ---- complex input -> real output (inv_rfft, inv_rfft2, gp_unpack_rfft, gp_unpack_rfft2) ---------------------------
template <int C, typename T, typename Fwd, typename Vjp>
inline Mat<scalar_of<T>, C> cplx_to_real(const T& z, b::Shape s, Eigen::Index orows, Eigen::Index ocols, Fwd fwd, Vjp vjp) {
    const Mat<cplx, C> zv = cplx_val<C>(z);
    Mat<double, C> out(orows, ocols);
    fwd(zv.data(), s, out.data());
    if constexpr (!is_var_v<T>) {
        return out;
    } else {
        stan::math::arena_t<Mat<var, C>> res = out;
        stan::math::arena_t<Mat<std::complex<var>, C>> az = z;
        stan::math::reverse_pass_callback([res, az, s, vjp]() mutable {
            const Mat<double, C> bar = res.adj();
            Mat<cplx, C> zbar(az.rows(), az.cols());
            vjp(bar.data(), s, zbar.data());
            add_cplx_adj<decltype(az), C>(az, zbar);
        });
        return Mat<var, C>(res);
    }
}
// ---- gp_evaluate_rfft_scale / gp_evaluate_rfft2_scale -------------------------------------------------------------
template <int C, typename T> inline Mat<scalar_of<T>, C> scale(const T& cov, b::Shape s) {
    const Mat<double, C> cv = real_val<C>(cov);
    b::require_sizes(cv.size() == s.nf(), "gp_evaluate_rfft(2)_scale");
    Mat<double, C> out(cv.rows(), cv.cols());
    b::gp_evaluate_scale(cv.data(), s, out.data());
    if constexpr (!is_var_v<T>) {
        return out;
    } else {
        stan::math::arena_t<Mat<var, C>> res = out;
        auto ac = stan::math::to_arena(cov);
        stan::math::reverse_pass_callback([res, ac, s]() mutable {
            const Mat<double, C> bar = res.adj(), cv2 = stan::math::value_of(ac), sc = stan::math::value_of(res);
            Mat<double, C> cbar(cv2.rows(), cv2.cols());
            b::gp_evaluate_scale_vjp(cv2.data(), sc.data(), bar.data(), s, cbar.data());
            ac.adj() += cbar;
        });
        return Mat<var, C>(res);
    }
}
// ---- gp_rfft(2) / gp_inv_rfft(2) -----------------------------------------------------------------------------------
template <bool FORWARD, int C, typename T0, typename T1, typename T2>
inline Mat<stan::return_type_t<T0, T1, T2>, C> transform(const T0& a0, const T1& loc, const T2& cov, int dim1, const char* fn) {
    const Mat<double, C> av = real_val<C>(a0), lv = real_val<C>(loc), cv = real_val<C>(cov);
    const b::Shape s{dim1 ? 1 : av.rows(), dim1 ? av.size() : av.cols(), dim1};
    b::require_sizes(lv.size() == av.size() && cv.size() == s.nf(), fn);
    Mat<double, C> out(av.rows(), av.cols());
    b::gp_transform(FORWARD, av.data(), lv.data(), cv.data(), s, out.data());
    if constexpr (!any_var_v<T0, T1, T2>) {
        return out;
    } else {
        stan::math::arena_t<Mat<var, C>> res = out;
        auto aa = stan::math::to_arena(a0);
        auto al = stan::math::to_arena(loc);
        auto ac = stan::math::to_arena(cov);
        stan::math::reverse_pass_callback([res, aa, al, ac, s]() mutable {
            const Mat<double, C> bar = res.adj(), av2 = stan::math::value_of(aa), lv2 = stan::math::value_of(al),
                                 cv2 = stan::math::value_of(ac);
            Mat<double, C> abar(av2.rows(), av2.cols()), cbar(cv2.rows(), cv2.cols());
            b::gp_transform_vjp(FORWARD, av2.data(), lv2.data(), cv2.data(), bar.data(), s, abar.data(), cbar.data());
            if constexpr (is_var_v<T0>) aa.adj() += abar;
            if constexpr (is_var_v<T1>) {
                if (FORWARD) al.adj() -= abar; // z depends on y - loc
                else al.adj() += bar;          // f = ... + loc
            }
            if constexpr (is_var_v<T2>) ac.adj() += cbar;
        });
        return Mat<var, C>(res);
    }
}
// ---- gp_rfft_lpdf / gp_rfft2_lpdf: fused value + gradient, ONE trip to the GPU per log-density evaluation -----------
template <int C, typename T0, typename T1, typename T2>
inline stan::return_type_t<T0, T1, T2> lpdf(const T0& y, const T1& loc, const T2& cov, int dim1, const char* fn) {
    const Mat<double, C> yv = real_val<C>(y), lv = real_val<C>(loc), cv = real_val<C>(cov);
    const b::Shape s{dim1 ? 1 : yv.rows(), dim1 ? yv.size() : yv.cols(), dim1};
    b::require_sizes(lv.size() == yv.size() && cv.size() == s.nf(), fn);
    if constexpr (!any_var_v<T0, T1, T2>) {
        return b::gp_lpdf(yv.data(), lv.data(), cv.data(), s, nullptr, nullptr);
    } else {
        stan::math::arena_t<Mat<double, C>> gy(yv.rows(), yv.cols()), gc(cv.rows(), cv.cols());
        const double value = b::gp_lpdf(yv.data(), lv.data(), cv.data(), s, gy.data(), is_var_v<T2> ? gc.data() : nullptr);
        auto ay = stan::math::to_arena(y);
        auto al = stan::math::to_arena(loc);
        auto ac = stan::math::to_arena(cov);
        return stan::math::make_callback_var(value, [ay, al, ac, gy, gc](auto& vi) mutable {
            if constexpr (is_var_v<T0>) ay.adj() += vi.adj() * gy;
            if constexpr (is_var_v<T1>) al.adj() -= vi.adj() * gy;
            if constexpr (is_var_v<T2>) ac.adj() += vi.adj() * gc;
        });
    }
}
template <int C, typename T0> inline stan::return_type_t<T0> ladj(const T0& cov, b::Shape s) {
    const Mat<double, C> cv = real_val<C>(cov);
    b::require_sizes(cv.size() == s.nf(), "gp_rfft(2)_log_abs_det_jac");
    if constexpr (!is_var_v<T0>) {
        return b::gp_ladj(cv.data(), s, nullptr);
    } else {
        stan::math::arena_t<Mat<double, C>> gc(cv.rows(), cv.cols());
        const double value = b::gp_ladj(cv.data(), s, gc.data());
        auto ac = stan::math::to_arena(cov);
        return stan::math::make_callback_var(value, [ac, gc](auto& vi) mutable { ac.adj() += vi.adj() * gc; });
    }
}
// ---- spectral kernels: NP parameters (sigma, length_scale[, ...], period[, ...]) held as vars or doubles -----------
// `Scatter` is a small functor type (no std::function: the autodiff arena never runs destructors)
template <int C, bool MATERN, int NP, bool ANYVAR, typename Scatter>
inline Mat<std::conditional_t<ANYVAR, var, double>, C> spectrum(double nu, b::Shape s, const double (&par)[NP], Scatter scatter) {
    Mat<double, C> out(s.dim1 ? s.wf() : s.H, s.dim1 ? 1 : s.wf());
    b::spectrum(MATERN, nu, s, par, out.data());
    if constexpr (!ANYVAR) {
        return out;
    } else {
        stan::math::arena_t<Mat<var, C>> res = out;
        stan::math::arena_t<VecD> p(NP);
        for (int i = 0; i < NP; ++i) p.coeffRef(i) = par[i];
        stan::math::reverse_pass_callback([res, p, nu, s, scatter]() mutable {
            const Mat<double, C> cbar = res.adj();
            double pbar[5] = {0, 0, 0, 0, 0};
            b::spectrum_vjp(MATERN, nu, s, p.data(), cbar.data(), pbar);
            scatter(pbar);
        });
        return Mat<var, C>(res);
    }
}
template <typename T> inline void add_adj(const T& x, double g) {
    if constexpr (stan::is_var<T>::value) x.adj() += g;
}
template <typename TS, typename TL, typename TP> struct Scatter1 { // 1-D kernels: three scalars
    TS sigma;
    TL ls;
    TP per;
    void operator()(const double* g) const {
        add_adj(sigma, g[0]);
        add_adj(ls, g[1]);
        add_adj(per, g[2]);
    }
};
template <typename TS, typename AL, typename AP> struct Scatter2 { // 2-D kernels: a scalar and two arena 2-vectors
    TS sigma;
    AL ls;
    AP per;
    void operator()(const double* g) {
        add_adj(sigma, g[0]);
        if constexpr (is_var_v<AL>) {
            ls.coeffRef(0).adj() += g[1];
            ls.coeffRef(1).adj() += g[2];
        }
        if constexpr (is_var_v<AP>) {
            per.coeffRef(0).adj() += g[3];
            per.coeffRef(1).adj() += g[4];
        }
    }
};
}  // namespace gptools_hip_stan

#define GPT_REQ_VEC(T) stan::is_col_vector<T>, stan::is_vt_not_complex<T>
#define GPT_REQ_CVEC(T) stan::is_col_vector<T>, stan::is_vt_complex<T>
#define GPT_REQ_MAT(T) stan::is_eigen_matrix_dynamic<T>, stan::is_vt_not_complex<T>
#define GPT_REQ_CMAT(T) stan::is_eigen_matrix_dynamic<T>, stan::is_vt_complex<T>
#define GPT_BASE(T) stan::base_type_t<T>

// ===================================================================================================
// util.stan:255-330 -- real-FFT layer (differentiable: linear maps)
template <typename T0__, stan::require_all_t<GPT_REQ_VEC(T0__)>* = nullptr>
Eigen::Matrix<std::complex<stan::promote_args_t<GPT_BASE(T0__)>>, -1, 1> rfft(const T0__& y, std::ostream* pstream__) {
    namespace g = gptools_hip_stan;
    const auto s = g::b::shape1(y.size());
    return g::real_to_cplx<1>(y, s, s.wf(), 1, g::b::rfft2, g::b::rfft2_vjp);
}
template <typename T0__, stan::require_all_t<GPT_REQ_CVEC(T0__)>* = nullptr>
Eigen::Matrix<std::complex<stan::promote_args_t<GPT_BASE(T0__)>>, -1, 1> expand_rfft(const T0__& y, const int& n,
                                                                                     std::ostream* pstream__) {
    // util.stan:271-281 on Stan's own containers (a copy and a conjugated, reversed copy: nothing for a GPU to do,
    // and complex<var> entries keep their varis); the size check is the reference's reject (util.stan:274-276)
    if (y.size() != n / 2 + 1)
        throw std::domain_error("expected complex vector with " + std::to_string(n / 2 + 1) + " elements but got " +
                                std::to_string(y.size()));
    Eigen::Matrix<std::complex<stan::promote_args_t<GPT_BASE(T0__)>>, -1, 1> out(n);
    const int nrfft = n / 2 + 1, nc = (n - 1) / 2;
    for (int k = 0; k < nrfft; ++k) out.coeffRef(k) = y.coeff(k);
    for (int k = 1; k <= nc; ++k) out.coeffRef(n - k) = stan::math::conj(y.coeff(k));
    return out;
}
template <typename T0__, stan::require_all_t<GPT_REQ_CVEC(T0__)>* = nullptr>
Eigen::Matrix<stan::promote_args_t<GPT_BASE(T0__)>, -1, 1> inv_rfft(const T0__& z, const int& n, std::ostream* pstream__) {
    namespace g = gptools_hip_stan;
    const auto s = g::b::shape1(n);
    g::b::require_sizes(z.size() == s.wf(), "inv_rfft");
    return g::cplx_to_real<1>(z, s, n, 1, g::b::inv_rfft2, g::b::inv_rfft2_vjp);
}
template <typename T0__, stan::require_all_t<GPT_REQ_MAT(T0__)>* = nullptr>
Eigen::Matrix<std::complex<stan::promote_args_t<GPT_BASE(T0__)>>, -1, -1> rfft2(const T0__& y, std::ostream* pstream__) {
    namespace g = gptools_hip_stan;
    const auto s = g::b::shape2(y.rows(), y.cols());
    return g::real_to_cplx<-1>(y, s, s.H, s.wf(), g::b::rfft2, g::b::rfft2_vjp);
}
template <typename T0__, stan::require_all_t<GPT_REQ_CMAT(T0__)>* = nullptr>
Eigen::Matrix<stan::promote_args_t<GPT_BASE(T0__)>, -1, -1> inv_rfft2(const T0__& z, const int& m, std::ostream* pstream__) {
    namespace g = gptools_hip_stan;
    const auto s = g::b::shape2(z.rows(), m);
    g::b::require_sizes(z.cols() == s.wf(), "inv_rfft2");
    return g::cplx_to_real<-1>(z, s, s.H, m, g::b::inv_rfft2, g::b::inv_rfft2_vjp);
}

// ===================================================================================================
// fft1.stan
template <typename T0__, stan::require_all_t<GPT_REQ_VEC(T0__)>* = nullptr>
Eigen::Matrix<stan::promote_args_t<GPT_BASE(T0__)>, -1, 1> gp_evaluate_rfft_scale(const T0__& cov_rfft, const int& n,
                                                                                  std::ostream* pstream__) {
    return gptools_hip_stan::scale<1>(cov_rfft, gptools_hip::shape1(n));
}
template <typename T0__, stan::require_all_t<GPT_REQ_CVEC(T0__)>* = nullptr>
Eigen::Matrix<stan::promote_args_t<GPT_BASE(T0__)>, -1, 1> gp_unpack_rfft(const T0__& x, const int& n, std::ostream* pstream__) {
    namespace g = gptools_hip_stan;
    const auto s = g::b::shape1(n);
    g::b::require_sizes(x.size() == s.wf(), "gp_unpack_rfft");
    return g::cplx_to_real<1>(x, s, n, 1, g::b::gp_unpack, g::b::gp_unpack_vjp);
}
template <typename T0__, stan::require_all_t<GPT_REQ_VEC(T0__)>* = nullptr>
Eigen::Matrix<std::complex<stan::promote_args_t<GPT_BASE(T0__)>>, -1, 1> gp_pack_rfft(const T0__& z, std::ostream* pstream__) {
    namespace g = gptools_hip_stan;
    const auto s = g::b::shape1(z.size());
    return g::real_to_cplx<1>(z, s, s.wf(), 1, g::b::gp_pack, g::b::gp_pack_vjp);
}
template <typename T0__, typename T1__, typename T2__,
          stan::require_all_t<GPT_REQ_VEC(T0__), GPT_REQ_VEC(T1__), GPT_REQ_VEC(T2__)>* = nullptr>
Eigen::Matrix<stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T1__), GPT_BASE(T2__)>, -1, 1> gp_rfft(
    const T0__& y, const T1__& loc, const T2__& cov_rfft, std::ostream* pstream__) {
    return gptools_hip_stan::transform<true, 1>(y, loc, cov_rfft, 1, "gp_rfft");
}
template <typename T0__, typename T1__, typename T2__,
          stan::require_all_t<GPT_REQ_VEC(T0__), GPT_REQ_VEC(T1__), GPT_REQ_VEC(T2__)>* = nullptr>
Eigen::Matrix<stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T1__), GPT_BASE(T2__)>, -1, 1> gp_inv_rfft(
    const T0__& z, const T1__& loc, const T2__& cov_rfft, std::ostream* pstream__) {
    return gptools_hip_stan::transform<false, 1>(z, loc, cov_rfft, 1, "gp_inv_rfft");
}
template <typename T0__, stan::require_all_t<GPT_REQ_VEC(T0__)>* = nullptr>
stan::promote_args_t<GPT_BASE(T0__)> gp_rfft_log_abs_det_jac(const T0__& cov_rfft, const int& n, std::ostream* pstream__) {
    return gptools_hip_stan::ladj<1>(cov_rfft, gptools_hip::shape1(n));
}
// (`y ~ gp_rfft(...)` and the function-call form both keep every constant: fft1.stan:100 calls std_normal_lpdf as a
// function, so propto__ changes nothing)
template <bool propto__, typename T0__, typename T1__, typename T2__,
          stan::require_all_t<GPT_REQ_VEC(T0__), GPT_REQ_VEC(T1__), GPT_REQ_VEC(T2__)>* = nullptr>
stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T1__), GPT_BASE(T2__)> gp_rfft_lpdf(const T0__& y, const T1__& loc,
                                                                                    const T2__& cov_rfft,
                                                                                    std::ostream* pstream__) {
    return gptools_hip_stan::lpdf<1>(y, loc, cov_rfft, 1, "gp_rfft_lpdf");
}
template <typename T1__, typename T2__, typename T3__,
          stan::require_all_t<stan::is_stan_scalar<T1__>, stan::is_stan_scalar<T2__>, stan::is_stan_scalar<T3__>>* = nullptr>
Eigen::Matrix<stan::promote_args_t<T1__, T2__, T3__>, -1, 1> gp_periodic_exp_quad_cov_rfft(const int& n, const T1__& sigma,
                                                                                             const T2__& length_scale,
                                                                                             const T3__& period,
                                                                                             std::ostream* pstream__) {
    namespace g = gptools_hip_stan;
    using stan::math::value_of;
    const double par[3] = {value_of(sigma), value_of(length_scale), value_of(period)};
    return g::spectrum<1, false, 3, g::any_var_v<T1__, T2__, T3__>>(0.0, g::b::shape1(n), par,
                                                                   g::Scatter1<T1__, T2__, T3__>{sigma, length_scale, period});
}
template <typename T0__, typename T2__, typename T3__, typename T4__,
          stan::require_all_t<stan::is_stan_scalar<T0__>, stan::is_stan_scalar<T2__>, stan::is_stan_scalar<T3__>,
                              stan::is_stan_scalar<T4__>>* = nullptr>
Eigen::Matrix<stan::promote_args_t<T0__, T2__, T3__, T4__>, -1, 1> gp_periodic_matern_cov_rfft(
    const T0__& nu, const int& n, const T2__& sigma, const T3__& length_scale, const T4__& period, std::ostream* pstream__) {
    namespace g = gptools_hip_stan;
    using stan::math::value_of;
    // nu is a literal in every model of the reference (trees.stan:28, padded.stan:19): data, no d/dnu
    if (stan::is_var<T0__>::value) throw std::domain_error("gp_periodic_matern_cov_rfft: nu must be data");
    const double par[3] = {value_of(sigma), value_of(length_scale), value_of(period)};
    return g::spectrum<1, true, 3, g::any_var_v<T0__, T2__, T3__, T4__>>(
        value_of(nu), g::b::shape1(n), par, g::Scatter1<T2__, T3__, T4__>{sigma, length_scale, period});
}

// ===================================================================================================
// fft2.stan
template <typename T0__, stan::require_all_t<GPT_REQ_MAT(T0__)>* = nullptr>
Eigen::Matrix<stan::promote_args_t<GPT_BASE(T0__)>, -1, -1> gp_evaluate_rfft2_scale(const T0__& cov_rfft2, const int& width,
                                                                                    std::ostream* pstream__) {
    return gptools_hip_stan::scale<-1>(cov_rfft2, gptools_hip::shape2(cov_rfft2.rows(), width));
}
template <typename T0__, stan::require_all_t<GPT_REQ_CMAT(T0__)>* = nullptr>
Eigen::Matrix<stan::promote_args_t<GPT_BASE(T0__)>, -1, -1> gp_unpack_rfft2(const T0__& z, const int& m, std::ostream* pstream__) {
    namespace g = gptools_hip_stan;
    const auto s = g::b::shape2(z.rows(), m);
    g::b::require_sizes(z.cols() == s.wf(), "gp_unpack_rfft2");
    return g::cplx_to_real<-1>(z, s, s.H, m, g::b::gp_unpack, g::b::gp_unpack_vjp);
}
template <typename T0__, stan::require_all_t<GPT_REQ_MAT(T0__)>* = nullptr>
Eigen::Matrix<std::complex<stan::promote_args_t<GPT_BASE(T0__)>>, -1, -1> gp_pack_rfft2(const T0__& z, std::ostream* pstream__) {
    namespace g = gptools_hip_stan;
    const auto s = g::b::shape2(z.rows(), z.cols());
    return g::real_to_cplx<-1>(z, s, s.H, s.wf(), g::b::gp_pack, g::b::gp_pack_vjp);
}
template <typename T0__, typename T1__, typename T2__,
          stan::require_all_t<GPT_REQ_MAT(T0__), GPT_REQ_MAT(T1__), GPT_REQ_MAT(T2__)>* = nullptr>
Eigen::Matrix<stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T1__), GPT_BASE(T2__)>, -1, -1> gp_rfft2(
    const T0__& y, const T1__& loc, const T2__& cov_rfft2, std::ostream* pstream__) {
    return gptools_hip_stan::transform<true, -1>(y, loc, cov_rfft2, 0, "gp_rfft2");
}
template <typename T0__, typename T1__, typename T2__,
          stan::require_all_t<GPT_REQ_MAT(T0__), GPT_REQ_MAT(T1__), GPT_REQ_MAT(T2__)>* = nullptr>
Eigen::Matrix<stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T1__), GPT_BASE(T2__)>, -1, -1> gp_inv_rfft2(
    const T0__& z, const T1__& loc, const T2__& cov_rfft2, std::ostream* pstream__) {
    return gptools_hip_stan::transform<false, -1>(z, loc, cov_rfft2, 0, "gp_inv_rfft2");
}
template <typename T0__, stan::require_all_t<GPT_REQ_MAT(T0__)>* = nullptr>
stan::promote_args_t<GPT_BASE(T0__)> gp_rfft2_log_abs_det_jac(const T0__& cov_rfft2, const int& width, std::ostream* pstream__) {
    return gptools_hip_stan::ladj<-1>(cov_rfft2, gptools_hip::shape2(cov_rfft2.rows(), width));
}
template <bool propto__, typename T0__, typename T1__, typename T2__,
          stan::require_all_t<GPT_REQ_MAT(T0__), GPT_REQ_MAT(T1__), GPT_REQ_MAT(T2__)>* = nullptr>
stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T1__), GPT_BASE(T2__)> gp_rfft2_lpdf(const T0__& y, const T1__& loc,
                                                                                     const T2__& cov_rfft2,
                                                                                     std::ostream* pstream__) {
    return gptools_hip_stan::lpdf<-1>(y, loc, cov_rfft2, 0, "gp_rfft2_lpdf");
}
template <typename T2__, typename T3__, typename T4__,
          stan::require_all_t<stan::is_stan_scalar<T2__>, GPT_REQ_VEC(T3__), GPT_REQ_VEC(T4__)>* = nullptr>
Eigen::Matrix<stan::promote_args_t<T2__, GPT_BASE(T3__), GPT_BASE(T4__)>, -1, -1> gp_periodic_exp_quad_cov_rfft2(
    const int& height, const int& width, const T2__& sigma, const T3__& length_scale, const T4__& period,
    std::ostream* pstream__) {
    namespace g = gptools_hip_stan;
    using stan::math::value_of;
    auto ls = stan::math::to_arena(length_scale);
    auto pd = stan::math::to_arena(period);
    const double par[5] = {value_of(sigma), value_of(ls.coeff(0)), value_of(ls.coeff(1)), value_of(pd.coeff(0)),
                           value_of(pd.coeff(1))};
    return g::spectrum<-1, false, 5, g::any_var_v<T2__, T3__, T4__>>(
        0.0, g::b::shape2(height, width), par, g::Scatter2<T2__, decltype(ls), decltype(pd)>{sigma, ls, pd});
}
template <typename T0__, typename T3__, typename T4__, typename T5__,
          stan::require_all_t<stan::is_stan_scalar<T0__>, stan::is_stan_scalar<T3__>, GPT_REQ_VEC(T4__), GPT_REQ_VEC(T5__)>* =
              nullptr>
Eigen::Matrix<stan::promote_args_t<T0__, T3__, GPT_BASE(T4__), GPT_BASE(T5__)>, -1, -1> gp_periodic_matern_cov_rfft2(
    const T0__& nu, const int& height, const int& width, const T3__& sigma, const T4__& length_scale, const T5__& period,
    std::ostream* pstream__) {
    namespace g = gptools_hip_stan;
    using stan::math::value_of;
    if (stan::is_var<T0__>::value) throw std::domain_error("gp_periodic_matern_cov_rfft2: nu must be data");
    auto ls = stan::math::to_arena(length_scale);
    auto pd = stan::math::to_arena(period);
    const double par[5] = {value_of(sigma), value_of(ls.coeff(0)), value_of(ls.coeff(1)), value_of(pd.coeff(0)),
                           value_of(pd.coeff(1))};
    return g::spectrum<-1, true, 5, g::any_var_v<T0__, T3__, T4__, T5__>>(
        value_of(nu), g::b::shape2(height, width), par, g::Scatter2<T3__, decltype(ls), decltype(pd)>{sigma, ls, pd});
}

// ===================================================================================================
// Extension (gptools/fft_obs.stan; no counterpart in the reference, where docs/trees/trees.stan:31-49 spells the same
// density as gp_inv_rfft2 + std_normal + a masked neg_binomial_2 loop): non-centred model density in one call. Only z,
// cov and the data travel to the GPU and only the adjoints of z and cov (and four scalars) come back.
namespace gptools_hip_stan {
template <int C, typename T0, typename TOBS, typename TLOC, typename TCOV, typename TTH>
inline stan::return_type_t<T0, TLOC, TCOV, TTH> obs_lpdf(const T0& z, int family, const TOBS& obs, const TLOC& loc,
                                                         const TCOV& cov, const TTH& theta, int prior, int dim1,
                                                         const char* fn) {
    const Mat<double, C> zv = real_val<C>(z), ov = real_val<C>(obs), cv = real_val<C>(cov);
    const b::Shape s{dim1 ? 1 : zv.rows(), dim1 ? zv.size() : zv.cols(), dim1};
    const int64_t oh = dim1 ? 1 : ov.rows(), ow = dim1 ? ov.size() : ov.cols();
    b::require_sizes(cv.size() == s.nf() && oh >= 1 && ow >= 1 && oh <= s.H && ow <= s.W, fn);
    b::ObsGrad og{0.0, 0.0};
    stan::math::arena_t<Mat<double, C>> gz(zv.rows(), zv.cols()), gc(cv.rows(), cv.cols());
    const double value = b::obs_lpdf(family, zv.data(), stan::math::value_of(loc), cv.data(), ov.data(), oh, ow,
                                     stan::math::value_of(theta), prior, s, gz.data(), is_var_v<TCOV> ? gc.data() : nullptr, &og);
    if constexpr (!any_var_v<T0, TLOC, TCOV, TTH>) {
        return value;
    } else {
        auto az = stan::math::to_arena(z);
        auto ac = stan::math::to_arena(cov);
        return stan::math::make_callback_var(value, [az, ac, loc, theta, gz, gc, og](auto& vi) mutable {
            if constexpr (is_var_v<T0>) az.adj() += vi.adj() * gz;
            if constexpr (is_var_v<TCOV>) ac.adj() += vi.adj() * gc;
            add_adj(loc, vi.adj() * og.loc_sum);
            add_adj(theta, vi.adj() * og.theta);
        });
    }
}
}  // namespace gptools_hip_stan
template <bool propto__, typename T0__, typename T2__, typename T3__, typename T4__, typename T5__,
          stan::require_all_t<GPT_REQ_MAT(T0__), GPT_REQ_MAT(T2__), stan::is_stan_scalar<T3__>, GPT_REQ_MAT(T4__),
                              stan::is_stan_scalar<T5__>>* = nullptr>
stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T2__), T3__, GPT_BASE(T4__), T5__> gp_inv_rfft2_obs_lpdf(
    const T0__& z, const int& family, const T2__& obs, const T3__& loc, const T4__& cov_rfft2, const T5__& theta,
    const int& prior, std::ostream* pstream__) {
    return gptools_hip_stan::obs_lpdf<-1>(z, family, obs, loc, cov_rfft2, theta, prior, 0, "gp_inv_rfft2_obs_lpdf");
}
template <bool propto__, typename T0__, typename T2__, typename T3__, typename T4__, typename T5__,
          stan::require_all_t<GPT_REQ_VEC(T0__), GPT_REQ_VEC(T2__), stan::is_stan_scalar<T3__>, GPT_REQ_VEC(T4__),
                              stan::is_stan_scalar<T5__>>* = nullptr>
stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T2__), T3__, GPT_BASE(T4__), T5__> gp_inv_rfft_obs_lpdf(
    const T0__& z, const int& family, const T2__& obs, const T3__& loc, const T4__& cov_rfft, const T5__& theta,
    const int& prior, std::ostream* pstream__) {
    return gptools_hip_stan::obs_lpdf<1>(z, family, obs, loc, cov_rfft, theta, prior, 1, "gp_inv_rfft_obs_lpdf");
}

// ===================================================================================================
// Extension (gptools/fft_kernel.stan): gp_rfft(2)_lpdf over a spectrum given by its kernel, one call
// (gpt_gp_rfft2_lpdf_kernel_host): y, loc and the kernel parameters up, the value and their adjoints back.
namespace gptools_hip_stan {
// NP = 5: (sigma, length_scale[2], period[2]) for the 2-D functions; NP = 3: (sigma, length_scale, period) for 1-D.
// `scatter(adj, pbar)` adds adj * pbar[i] to the adjoints of whichever parameters are vars.
template <int C, bool MATERN, int NP, typename T0, typename T1, typename Scatter>
inline auto kernel_lpdf(const T0& y, const T1& loc, double nu, const double (&par)[NP], double nugget, int dim1, bool any_par_var,
                        Scatter scatter, const char* fn) {
    const Mat<double, C> yv = real_val<C>(y), lv = real_val<C>(loc);
    const b::Shape s{dim1 ? 1 : yv.rows(), dim1 ? yv.size() : yv.cols(), dim1};
    b::require_sizes(lv.size() == yv.size(), fn);
    using ret_t = std::conditional_t<any_var_v<T0, T1> || Scatter::any_var, var, double>;
    if constexpr (std::is_same_v<ret_t, double>) {
        (void)any_par_var;
        return b::gp_kernel_lpdf(MATERN, nu, yv.data(), lv.data(), par, nugget, s, nullptr, nullptr);
    } else {
        stan::math::arena_t<Mat<double, C>> gy(yv.rows(), yv.cols());
        stan::math::arena_t<Eigen::Matrix<double, NP, 1>> pbar;
        pbar.setZero();
        const double value = b::gp_kernel_lpdf(MATERN, nu, yv.data(), lv.data(), par, nugget, s, gy.data(),
                                               any_par_var ? pbar.data() : nullptr);
        auto ay = stan::math::to_arena(y);
        auto al = stan::math::to_arena(loc);
        return ret_t(stan::math::make_callback_var(value, [ay, al, gy, pbar, scatter](auto& vi) mutable {
            if constexpr (is_var_v<T0>) ay.adj() += vi.adj() * gy;
            if constexpr (is_var_v<T1>) al.adj() -= vi.adj() * gy;
            scatter(vi.adj(), pbar);
        }));
    }
}
// adjoint scatter of (sigma, length_scale[2], period[2]) / (sigma, length_scale, period)
template <typename TS, typename TL, typename TP> struct KScatter2 {
    static constexpr bool any_var = any_var_v<TS, TL, TP>;
    TS sigma;
    stan::math::arena_t<TL> ls;
    stan::math::arena_t<TP> pd;
    template <typename V> void operator()(double adj, const V& pbar) {
        add_adj(sigma, adj * pbar.coeff(0));
        if constexpr (is_var_v<TL>) {
            ls.adj().coeffRef(0) += adj * pbar.coeff(1);
            ls.adj().coeffRef(1) += adj * pbar.coeff(2);
        }
        if constexpr (is_var_v<TP>) {
            pd.adj().coeffRef(0) += adj * pbar.coeff(3);
            pd.adj().coeffRef(1) += adj * pbar.coeff(4);
        }
    }
};
template <typename TS, typename TL, typename TP> struct KScatter1 {
    static constexpr bool any_var = any_var_v<TS, TL, TP>;
    TS sigma;
    TL ls;
    TP pd;
    template <typename V> void operator()(double adj, const V& pbar) {
        add_adj(sigma, adj * pbar.coeff(0));
        add_adj(ls, adj * pbar.coeff(1));
        add_adj(pd, adj * pbar.coeff(2));
    }
};
template <bool MATERN, typename T0, typename T1, typename TS, typename TL, typename TP>
inline auto kernel_lpdf2(const T0& y, const T1& loc, double nu, const TS& sigma, const TL& length_scale, const TP& period,
                         double nugget, const char* fn) {
    using stan::math::value_of;
    b::require_sizes(length_scale.size() == 2 && period.size() == 2, fn);
    const double par[5] = {value_of(sigma), value_of(length_scale.coeff(0)), value_of(length_scale.coeff(1)),
                           value_of(period.coeff(0)), value_of(period.coeff(1))};
    return kernel_lpdf<-1, MATERN, 5>(y, loc, nu, par, nugget, 0, any_var_v<TS, TL, TP>,
                                      KScatter2<TS, TL, TP>{sigma, stan::math::to_arena(length_scale), stan::math::to_arena(period)}, fn);
}
template <bool MATERN, typename T0, typename T1, typename TS, typename TL, typename TP>
inline auto kernel_lpdf1(const T0& y, const T1& loc, double nu, const TS& sigma, const TL& length_scale, const TP& period,
                         double nugget, const char* fn) {
    using stan::math::value_of;
    const double par[3] = {value_of(sigma), value_of(length_scale), value_of(period)};
    return kernel_lpdf<1, MATERN, 3>(y, loc, nu, par, nugget, 1, any_var_v<TS, TL, TP>, KScatter1<TS, TL, TP>{sigma, length_scale, period},
                                     fn);
}
}  // namespace gptools_hip_stan
template <bool propto__, typename T0__, typename T1__, typename T2__, typename T3__, typename T4__, typename T5__, typename T6__,
          stan::require_all_t<GPT_REQ_MAT(T0__), GPT_REQ_MAT(T1__), stan::is_stan_scalar<T2__>, stan::is_stan_scalar<T3__>,
                              GPT_REQ_VEC(T4__), GPT_REQ_VEC(T5__), stan::is_stan_scalar<T6__>>* = nullptr>
stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T1__), T2__, T3__, GPT_BASE(T4__), stan::promote_args_t<GPT_BASE(T5__), T6__>>
gp_rfft2_matern_lpdf(const T0__& y, const T1__& loc, const T2__& nu, const T3__& sigma, const T4__& length_scale,
                     const T5__& period, const T6__& nugget, std::ostream* pstream__) {
    return gptools_hip_stan::kernel_lpdf2<true>(y, loc, stan::math::value_of(nu), sigma, length_scale, period,
                                                stan::math::value_of(nugget), "gp_rfft2_matern_lpdf");
}
template <bool propto__, typename T0__, typename T1__, typename T2__, typename T3__, typename T4__, typename T5__,
          stan::require_all_t<GPT_REQ_MAT(T0__), GPT_REQ_MAT(T1__), stan::is_stan_scalar<T2__>, GPT_REQ_VEC(T3__), GPT_REQ_VEC(T4__),
                              stan::is_stan_scalar<T5__>>* = nullptr>
stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T1__), T2__, GPT_BASE(T3__), GPT_BASE(T4__), T5__>
gp_rfft2_exp_quad_lpdf(const T0__& y, const T1__& loc, const T2__& sigma, const T3__& length_scale, const T4__& period,
                       const T5__& nugget, std::ostream* pstream__) {
    return gptools_hip_stan::kernel_lpdf2<false>(y, loc, 0.0, sigma, length_scale, period, stan::math::value_of(nugget),
                                                 "gp_rfft2_exp_quad_lpdf");
}
template <bool propto__, typename T0__, typename T1__, typename T2__, typename T3__, typename T4__, typename T5__, typename T6__,
          stan::require_all_t<GPT_REQ_VEC(T0__), GPT_REQ_VEC(T1__), stan::is_stan_scalar<T2__>, stan::is_stan_scalar<T3__>,
                              stan::is_stan_scalar<T4__>, stan::is_stan_scalar<T5__>, stan::is_stan_scalar<T6__>>* = nullptr>
stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T1__), T2__, T3__, T4__, stan::promote_args_t<T5__, T6__>>
gp_rfft_matern_lpdf(const T0__& y, const T1__& loc, const T2__& nu, const T3__& sigma, const T4__& length_scale,
                    const T5__& period, const T6__& nugget, std::ostream* pstream__) {
    return gptools_hip_stan::kernel_lpdf1<true>(y, loc, stan::math::value_of(nu), sigma, length_scale, period,
                                                stan::math::value_of(nugget), "gp_rfft_matern_lpdf");
}
template <bool propto__, typename T0__, typename T1__, typename T2__, typename T3__, typename T4__, typename T5__,
          stan::require_all_t<GPT_REQ_VEC(T0__), GPT_REQ_VEC(T1__), stan::is_stan_scalar<T2__>, stan::is_stan_scalar<T3__>,
                              stan::is_stan_scalar<T4__>, stan::is_stan_scalar<T5__>>* = nullptr>
stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T1__), T2__, T3__, T4__, T5__>
gp_rfft_exp_quad_lpdf(const T0__& y, const T1__& loc, const T2__& sigma, const T3__& length_scale, const T4__& period,
                      const T5__& nugget, std::ostream* pstream__) {
    return gptools_hip_stan::kernel_lpdf1<false>(y, loc, 0.0, sigma, length_scale, period, stan::math::value_of(nugget),
                                                 "gp_rfft_exp_quad_lpdf");
}

// ... and with the spectrum given by its kernel (gptools/fft_obs.stan: gp_inv_rfft2_obs_matern_lpdf,
// gp_inv_rfft2_obs_exp_quad_lpdf): gpt_gp_inv_rfft2_obs_lpdf_kernel_host, five parameters up and their five adjoints back.
namespace gptools_hip_stan {
template <bool MATERN, typename T0, typename TOBS, typename TLOC, typename TS, typename TL, typename TP, typename TTH>
inline stan::return_type_t<T0, TLOC, TS, TL, TP, TTH> obs_kernel_lpdf(const T0& z, int family, const TOBS& obs, const TLOC& loc,
                                                                      double nu, const TS& sigma, const TL& length_scale,
                                                                      const TP& period, double nugget, const TTH& theta,
                                                                      int prior, const char* fn) {
    using stan::math::value_of;
    const Mat<double, -1> zv = real_val<-1>(z), ov = real_val<-1>(obs);
    const b::Shape s{zv.rows(), zv.cols(), 0};
    const int64_t oh = ov.rows(), ow = ov.cols();
    b::require_sizes(length_scale.size() == 2 && period.size() == 2 && oh >= 1 && ow >= 1 && oh <= s.H && ow <= s.W, fn);
    auto ls = stan::math::to_arena(length_scale);
    auto pd = stan::math::to_arena(period);
    const double par[5] = {value_of(sigma), value_of(ls.coeff(0)), value_of(ls.coeff(1)), value_of(pd.coeff(0)),
                           value_of(pd.coeff(1))};
    b::ObsGrad og{0.0, 0.0};
    stan::math::arena_t<Mat<double, -1>> gz(zv.rows(), zv.cols());
    stan::math::arena_t<Eigen::Matrix<double, 5, 1>> pbar;
    pbar.setZero();
    const double value = b::obs_kernel_lpdf(family, MATERN, nu, zv.data(), value_of(loc), par, nugget, ov.data(), oh, ow,
                                            value_of(theta), prior, s, gz.data(), pbar.data(), &og);
    if constexpr (!any_var_v<T0, TLOC, TS, TL, TP, TTH>) {
        return value;
    } else {
        auto az = stan::math::to_arena(z);
        return stan::math::make_callback_var(value, [az, loc, sigma, ls, pd, theta, gz, pbar, og](auto& vi) mutable {
            if constexpr (is_var_v<T0>) az.adj() += vi.adj() * gz;
            add_adj(loc, vi.adj() * og.loc_sum);
            add_adj(theta, vi.adj() * og.theta);
            add_adj(sigma, vi.adj() * pbar.coeff(0));
            if constexpr (is_var_v<TL>) {
                ls.adj().coeffRef(0) += vi.adj() * pbar.coeff(1);
                ls.adj().coeffRef(1) += vi.adj() * pbar.coeff(2);
            }
            if constexpr (is_var_v<TP>) {
                pd.adj().coeffRef(0) += vi.adj() * pbar.coeff(3);
                pd.adj().coeffRef(1) += vi.adj() * pbar.coeff(4);
            }
        });
    }
}
}  // namespace gptools_hip_stan
template <bool propto__, typename T0__, typename T2__, typename T3__, typename T4__, typename T5__, typename T6__, typename T7__,
          typename T8__, typename T9__,
          stan::require_all_t<GPT_REQ_MAT(T0__), GPT_REQ_MAT(T2__), stan::is_stan_scalar<T3__>, stan::is_stan_scalar<T4__>,
                              stan::is_stan_scalar<T5__>, GPT_REQ_VEC(T6__), GPT_REQ_VEC(T7__), stan::is_stan_scalar<T8__>,
                              stan::is_stan_scalar<T9__>>* = nullptr>
stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T2__), T3__, T4__, T5__, stan::promote_args_t<GPT_BASE(T6__), GPT_BASE(T7__), T8__, T9__>>
gp_inv_rfft2_obs_matern_lpdf(const T0__& z, const int& family, const T2__& obs, const T3__& loc, const T4__& nu, const T5__& sigma,
                             const T6__& length_scale, const T7__& period, const T8__& nugget, const T9__& theta,
                             const int& prior, std::ostream* pstream__) {
    return gptools_hip_stan::obs_kernel_lpdf<true>(z, family, obs, loc, stan::math::value_of(nu), sigma, length_scale, period,
                                                   stan::math::value_of(nugget), theta, prior, "gp_inv_rfft2_obs_matern_lpdf");
}
template <bool propto__, typename T0__, typename T2__, typename T3__, typename T4__, typename T5__, typename T6__, typename T7__,
          typename T8__,
          stan::require_all_t<GPT_REQ_MAT(T0__), GPT_REQ_MAT(T2__), stan::is_stan_scalar<T3__>, stan::is_stan_scalar<T4__>,
                              GPT_REQ_VEC(T5__), GPT_REQ_VEC(T6__), stan::is_stan_scalar<T7__>, stan::is_stan_scalar<T8__>>* = nullptr>
stan::promote_args_t<GPT_BASE(T0__), GPT_BASE(T2__), T3__, T4__, GPT_BASE(T5__), stan::promote_args_t<GPT_BASE(T6__), T7__, T8__>>
gp_inv_rfft2_obs_exp_quad_lpdf(const T0__& z, const int& family, const T2__& obs, const T3__& loc, const T4__& sigma,
                               const T5__& length_scale, const T6__& period, const T7__& nugget, const T8__& theta,
                               const int& prior, std::ostream* pstream__) {
    return gptools_hip_stan::obs_kernel_lpdf<false>(z, family, obs, loc, 0.0, sigma, length_scale, period,
                                                    stan::math::value_of(nugget), theta, prior, "gp_inv_rfft2_obs_exp_quad_lpdf");
}

#undef GPT_REQ_VEC
#undef GPT_REQ_CVEC
#undef GPT_REQ_MAT
#undef GPT_REQ_CMAT
#undef GPT_BASE
#endif  // GPTOOLS_HIP_HPP
