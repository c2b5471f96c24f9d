// Device functions of the observation models (gpt_obs.hip; the fused row kernel of gpt_pow2_impl.h).
#pragma once
#include "../../include/gptools_hip.h"
#include "gpt_common.h"

namespace {

constexpr double HALF_LOG_2PI = 0.91893853320467274178;

// lgamma(x) and digamma(x), x > 0: recurrence up to x >= 10, then Stirling's series (next terms < 4e-17). One
// logarithm is shared; far cheaper than the general-purpose lgamma, which matters when every point of a lattice
// evaluates it (negative-binomial counts).
__device__ __forceinline__ void lgamma_digamma_pos(double x, double& lg, double& dg) {
    double r = 0.0, p = 1.0;
    while (x < 10.0) {
        r -= 1.0 / x;
        p *= x;
        x += 1.0;
    }
    const double lx = log(x), i = 1.0 / x, i2 = i * i;
    const double ser =
        i2 * (1.0 / 12 - i2 * (1.0 / 120 - i2 * (1.0 / 252 - i2 * (1.0 / 240 - i2 * (1.0 / 132 - i2 * (691.0 / 32760 - i2 / 12))))));
    dg = r + lx - 0.5 * i - ser;
    const double st =
        i * (1.0 / 12 - i2 * (1.0 / 360 - i2 * (1.0 / 1260 - i2 * (1.0 / 1680 - i2 * (1.0 / 1188 - i2 * (691.0 / 360360 - i2 / 156))))));
    lg = (x - 0.5) * lx - x + HALF_LOG_2PI + st - (p == 1.0 ? 0.0 : log(p));
}
__device__ __forceinline__ double digamma_pos(double x) {
    double lg, dg;
    lgamma_digamma_pos(x, lg, dg);
    return dg;
}
__device__ __forceinline__ double lgamma_pos(double x) {
    double lg, dg;
    lgamma_digamma_pos(x, lg, dg);
    return lg;
}

// log(exp(a) + exp(b))
__device__ __forceinline__ double log_sum_exp2(double a, double b) {
    const double m = fmax(a, b);
    return m + log1p(exp(-fabs(a - b)));
}

struct ObsTerm {
    double lp, df, dth;
};

// one observed point. FAM: GPT_OBS_NORMAL (theta = standard deviation), GPT_OBS_POISSON_LOG (no parameter),
// GPT_OBS_NEG_BINOMIAL_2_LOG (theta = precision phi; mean exp(f)). All normalising constants are kept, as in
// gp_rfft2_lpdf (fft2.stan:189-192).
template <int FAM> __device__ __forceinline__ ObsTerm obs_term(double y, double f, double th, const double* k) {
    ObsTerm t;
    if (FAM == GPT_OBS_NORMAL) {
        // k[0] = 1 / theta, k[1] = log(theta)
        const double r = (y - f) * k[0];
        t.lp = -0.5 * r * r - k[1] - HALF_LOG_2PI;
        t.df = r * k[0];
        t.dth = (r * r - 1.0) * k[0];
    } else if (FAM == GPT_OBS_POISSON_LOG) {
        const double mu = exp(f);
        t.lp = y * f - mu - lgamma_pos(y + 1.0);
        t.df = y - mu;
        t.dth = 0.0;
    } else {
        // k[0] = log(phi), k[1] = lgamma(phi), k[2] = digamma(phi)
        const double lse = log_sum_exp2(f, k[0]); // log(mu + phi)
        const double npf = y + th;
        double lg, dg;
        lgamma_digamma_pos(npf, lg, dg);
        t.lp = lg - lgamma_pos(y + 1.0) - k[1] + y * (f - lse) + th * (k[0] - lse);
        t.df = y - npf * exp(f - lse);
        t.dth = dg - k[2] + k[0] - lse + 1.0 - npf * exp(-lse);
    }
    return t;
}

// family chosen at run time (uniform branch): one instance of each density in the calling loop
__device__ __forceinline__ ObsTerm obs_term_rt(int family, double y, double f, double th, const double* k) {
    if (family == GPT_OBS_NORMAL) return obs_term<GPT_OBS_NORMAL>(y, f, th, k);
    if (family == GPT_OBS_POISSON_LOG) return obs_term<GPT_OBS_POISSON_LOG>(y, f, th, k);
    return obs_term<GPT_OBS_NEG_BINOMIAL_2_LOG>(y, f, th, k);
}
// per-lattice constants of a family: see obs_term
__device__ __forceinline__ void obs_consts(int family, double th, double (&k)[3]) {
    k[0] = k[1] = k[2] = 0.0;
    if (family == GPT_OBS_NORMAL) {
        k[0] = 1.0 / th;
        k[1] = log(th);
    } else if (family == GPT_OBS_NEG_BINOMIAL_2_LOG) {
        k[0] = log(th);
        lgamma_digamma_pos(th, k[1], k[2]);
    }
}

// A run of n points `stride` apart of one lattice row, staged in LDS by the calling thread (f at lf[], the data at
// ly[], NaN = not observed): f -> d lp / d f in place, optional copies of both to global memory, the four sums
// returned. Deliberately NOT inlined: the fused row kernel calls it four times and its code (three densities with
// their special functions) would otherwise push that kernel past the instruction cache.
struct ObsSums {
    double lp, dth, df, n;
};
#ifndef GPT_OBS_CALL
#define GPT_OBS_CALL 0
#endif
#if GPT_OBS_CALL
__device__ __noinline__
#else
__device__ __forceinline__
#endif
ObsSums obs_points(double* lf, const double* ly, int n, int stride, int family, double th,
                                           double k0, double k1, double k2, double* fo, double* go) {
    const double k[3] = {k0, k1, k2};
    ObsSums s = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int m = 0; m < n; ++m) {
        const int j = m * stride;
        const double f = lf[j], y = ly[j];
        double g = 0.0;
        if (y == y && (family == GPT_OBS_NORMAL || y >= 0.0)) {
            const ObsTerm tm = obs_term_rt(family, y, f, th, k);
            g = tm.df;
            s.lp += tm.lp;
            s.dth += tm.dth;
            s.df += tm.df;
            s.n += 1.0;
        }
        lf[j] = g;
        if (fo) __builtin_nontemporal_store(f, fo + j);
        if (go) __builtin_nontemporal_store(g, go + j);
    }
    return s;
}

} // namespace
