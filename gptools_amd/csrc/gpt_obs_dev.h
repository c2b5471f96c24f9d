// Device functions of the observation models (gpt_obs.hip; the fused row kernel of gpt_pow2_impl.h).
#pragma once
#include "../../include/gptools_hip.h"
#include "gpt_common.h"

namespace {

constexpr double HALF_LOG_2PI = 0.91893853320467274178;

// lgamma(x) and digamma(x), x > 0: recurrence (as a running product) up to x >= 10, then Stirling's series (next terms < 4e-17). One
// logarithm is shared; far cheaper than the general-purpose lgamma, which matters when every point of a lattice
// evaluates it (negative-binomial counts).
__device__ __forceinline__ void lgamma_digamma_pos(double x, double& lg, double& dg) {
    // p = x (x + 1) ... (x + k - 1) and its derivative dp: sum of 1 / (x + i) = dp / p, one division for the lot
    // (x > 0 needs at most ten steps; anything else -- zero, negative, NaN -- gives NaN and, above all, terminates:
    // for x <= -2^53, x + 1 == x)
    if (!(x > 0.0)) {
        lg = dg = __builtin_nan("");
        return;
    }
    double p = 1.0, dp = 0.0;
#pragma unroll 1
    for (int it = 0; it < 10 && x < 10.0; ++it) {
        dp = fma(dp, x, p);
        p *= x;
        x += 1.0;
    }
    const double r = -dp / p;
    const double lx = log(x), i = 1.0 / x, i2 = i * i;
    const double ser =
        i2 * (1.0 / 12 - i2 * (1.0 / 120 - i2 * (1.0 / 252 - i2 * (1.0 / 240 - i2 * (1.0 / 132 - i2 * (691.0 / 32760 - i2 / 12))))));
    dg = r + lx - 0.5 * i - ser;
    const double st =
        i * (1.0 / 12 - i2 * (1.0 / 360 - i2 * (1.0 / 1260 - i2 * (1.0 / 1680 - i2 * (1.0 / 1188 - i2 * (691.0 / 360360 - i2 / 156))))));
    lg = (x - 0.5) * lx - x + HALF_LOG_2PI + st - (p == 1.0 ? 0.0 : log(p));
}
// log(k!) for k < 64: counts are small integers nearly everywhere (trees.stan: trees per quadrat)
__device__ const double GPT_LOG_FACT[64] = {
    0.0, 0.0, 0.6931471805599453, 1.791759469228055,
    3.1780538303479458, 4.787491742782046, 6.579251212010101, 8.525161361065415,
    10.60460290274525, 12.801827480081469, 15.104412573075516, 17.502307845873887,
    19.987214495661885, 22.552163853123425, 25.19122118273868, 27.89927138384089,
    30.671860106080672, 33.50507345013689, 36.39544520803305, 39.339884187199495,
    42.335616460753485, 45.38013889847691, 48.47118135183523, 51.60667556776438,
    54.78472939811232, 58.00360522298052, 61.261701761002, 64.55753862700634,
    67.88974313718154, 71.25703896716801, 74.65823634883016, 78.0922235533153,
    81.55795945611504, 85.05446701758152, 88.58082754219768, 92.1361756036871,
    95.7196945421432, 99.33061245478743, 102.96819861451381, 106.63176026064346,
    110.32063971475739, 114.0342117814617, 117.77188139974507, 121.53308151543864,
    125.3172711493569, 129.12393363912722, 132.95257503561632, 136.80272263732635,
    140.67392364823425, 144.5657439463449, 148.47776695177302, 152.40959258449735,
    156.3608363030788, 160.3311282166309, 164.32011226319517, 168.32744544842765,
    172.3527971391628, 176.39584840699735, 180.45629141754378, 184.53382886144948,
    188.6281734236716, 192.7390472878449, 196.86618167289, 201.00931639928152,
};
// lgamma(y + 1) for a count y >= 0
__device__ __forceinline__ double log_factorial(double y) {
    if (y < 64.0 && y == floor(y)) return GPT_LOG_FACT[(int)y];
    double lg, dg;
    lgamma_digamma_pos(y + 1.0, lg, dg);
    return lg;
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
        t.lp = y * f - mu - log_factorial(y);
        t.df = y - mu;
        t.dth = 0.0;
    } else {
        // k[0] = log(phi), k[1] = lgamma(phi), k[2] = digamma(phi)
        const double lse = log_sum_exp2(f, k[0]); // log(mu + phi)
        const double npf = y + th;
        // lgamma(y + phi) - lgamma(phi) and digamma(y + phi) - digamma(phi)
        double lgd, dgd;
        if (y <= 32.0 && y == floor(y)) {
            // small counts (nearly all of such data): the rising product phi (phi + 1) ... (phi + y - 1) and its
            // derivative -- one logarithm and one division, and no cancellation between two large values
            double p = 1.0, dp = 0.0, x = th;
            for (int i = (int)y; i > 0; --i) {
                dp = fma(dp, x, p);
                p *= x;
                x += 1.0;
            }
            lgd = log(p);
            dgd = dp / p;
        } else {
            double lg, dg;
            lgamma_digamma_pos(npf, lg, dg);
            lgd = lg - k[1];
            dgd = dg - k[2];
        }
        t.lp = lgd - log_factorial(y) + y * (f - lse) + th * (k[0] - lse);
        t.df = y - npf * exp(f - lse);
        t.dth = dgd + k[0] - lse + 1.0 - npf * exp(-lse);
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
    // Stan Math rejects a scale / precision that is not positive and finite; here the lattice's density becomes NaN
    // (the host entry point and the Python shim reject before launching)
    if (family != GPT_OBS_POISSON_LOG && !(th > 0.0 && th < INFINITY)) {
        k[0] = k[1] = k[2] = __builtin_nan("");
        return;
    }
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
