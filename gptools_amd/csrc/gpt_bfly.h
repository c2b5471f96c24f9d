// Radix-2/4/8/16 butterflies on registers (natural order in and out), shared by the power-of-two kernels
// (gpt_pow2_impl.h) and the LDS Stockham routine of the generic paths (gpt_lds_fft.h).
#pragma once
#include "gpt_common.h"

// Scheduling fence: hipcc's scheduler otherwise interleaves the whole (fully unrolled) transform for ILP and
// stretches live ranges until the 16-element register tile spills; program order is in-place.
#ifndef GPT_FENCE
#define GPT_FENCE 1
#endif
#if GPT_FENCE
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SCHED_FENCE() ((void)0)
#endif
namespace pow2 {

template <bool INV> __device__ __forceinline__ cplx tw_dir(cplx w) {
    if (INV) w.y = -w.y;
    return w;
}

// multiply by -i (forward) / +i (inverse)
template <bool INV> __device__ __forceinline__ cplx mul_mi(cplx a) {
    return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
}

template <bool INV> __device__ __forceinline__ void bfly2(cplx& a, cplx& b) {
    const cplx t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

// 4-point DFT, natural order in and out: X[p] = sum_q x[q] w4^{qp}
template <bool INV> __device__ __forceinline__ void bfly4(cplx& x0, cplx& x1, cplx& x2, cplx& x3) {
    const cplx a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = mul_mi<INV>(csub(x1, x3));
    x0 = cadd(a, c);
    x1 = cadd(b, d);
    x2 = csub(a, c);
    x3 = csub(b, d);
}

template <int R, bool INV> struct Bfly;

template <bool INV> struct Bfly<2, INV> {
    static __device__ __forceinline__ void run(cplx* x) { bfly2<INV>(x[0], x[1]); }
};
template <bool INV> struct Bfly<4, INV> {
    static __device__ __forceinline__ void run(cplx* x) { bfly4<INV>(x[0], x[1], x[2], x[3]); }
};
template <bool INV> struct Bfly<8, INV> {
    // 8 = 2 (q1) x 4 (q2): X[p1 + 2 p2] = sum_q2 w8^{q2 p1} (sum_q1 x[4 q1 + q2] w2^{q1 p1}) w4^{q2 p2}
    static __device__ __forceinline__ void run(cplx* x) {
        constexpr double h = 0.70710678118654752440;
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) bfly2<INV>(x[q2], x[4 + q2]);
        // rows p1 = 0: x[0..3]; p1 = 1: x[4..7] times w8^{q2}
        x[5] = INV ? make_double2(h * (x[5].x - x[5].y), h * (x[5].x + x[5].y))
                   : make_double2(h * (x[5].x + x[5].y), h * (x[5].y - x[5].x));
        x[6] = mul_mi<INV>(x[6]);
        x[7] = INV ? make_double2(-h * (x[7].x + x[7].y), h * (x[7].x - x[7].y))
                   : make_double2(h * (x[7].y - x[7].x), -h * (x[7].x + x[7].y));
        bfly4<INV>(x[0], x[1], x[2], x[3]); // p1 = 0 -> outputs X[0 + 2 p2]
        bfly4<INV>(x[4], x[5], x[6], x[7]); // p1 = 1 -> outputs X[1 + 2 p2]
        // reorder: currently x[p2] = X[2 p2], x[4 + p2] = X[1 + 2 p2]
        const cplx t1 = x[1], t2 = x[2], t3 = x[3], t4 = x[4], t5 = x[5], t6 = x[6];
        x[1] = t4; x[2] = t1; x[3] = t5; x[4] = t2; x[5] = t6; x[6] = t3;
    }
};
template <bool INV> struct Bfly<16, INV> {
    // 16 = 4 (q1) x 4 (q2): X[p1 + 4 p2] = sum_q2 w16^{q2 p1} (sum_q1 x[4 q1 + q2] w4^{q1 p1}) w4^{q2 p2}
    static __device__ __forceinline__ void run(cplx* x) {
        constexpr double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173; // cos, sin(pi/8)
        constexpr double h = 0.70710678118654752440;
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            bfly4<INV>(x[q2], x[4 + q2], x[8 + q2], x[12 + q2]);
            if (q2 & 1) SCHED_FENCE();
        }
        // now x[4 p1 + q2] = y[p1][q2]; twiddle by w16^{q2 p1} = exp(-+ 2 pi i q2 p1 / 16)
        auto tw = [](cplx a, double c, double s) -> cplx { // a * (c - i s) forward, a * (c + i s) inverse
            return INV ? make_double2(a.x * c - a.y * s, a.y * c + a.x * s)
                       : make_double2(a.x * c + a.y * s, a.y * c - a.x * s);
        };
        // p1 = 1: exponents 1, 2, 3
        x[5] = tw(x[5], c1, s1);
        x[6] = tw(x[6], h, h);
        x[7] = tw(x[7], s1, c1);
        // p1 = 2: exponents 2, 4, 6
        x[9] = tw(x[9], h, h);
        x[10] = mul_mi<INV>(x[10]);
        x[11] = tw(x[11], -h, h);
        // p1 = 3: exponents 3, 6, 9
        x[13] = tw(x[13], s1, c1);
        x[14] = tw(x[14], -h, h);
        x[15] = tw(x[15], -c1, -s1);
        SCHED_FENCE();
#pragma unroll
        for (int p1 = 0; p1 < 4; ++p1) {
            bfly4<INV>(x[4 * p1], x[4 * p1 + 1], x[4 * p1 + 2], x[4 * p1 + 3]);
            if (p1 & 1) SCHED_FENCE();
        }
        // x[4 p1 + p2] = X[p1 + 4 p2] -> transpose the 4 x 4 register tile
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = a + 1; b < 4; ++b) {
                const cplx t = x[4 * a + b];
                x[4 * a + b] = x[4 * b + a];
                x[4 * b + a] = t;
            }
    }
};

} // namespace pow2
