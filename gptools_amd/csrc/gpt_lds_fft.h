// Mixed-radix Stockham line FFT on lines that live in LDS (any length: arbitrary prime factors). Shared by
// the small-lattice kernel and the generic three-pass kernels.
#pragma once
#include "gpt_common.h"
#include "gpt_bfly.h"

// One Stockham stage of radix R (compile-time) over `nlines` lines living in LDS: one work item = one BUTTERFLY
// (R inputs M = N/R apart, R outputs Ns apart), inputs and twiddles loaded once per butterfly, the R-point DFT
// done on registers against the R roots w_R^r (R^2 complex MACs: fine for the small primes; 2 and 4 use
// additions only).
template <int R, bool INVERSE>
__device__ __forceinline__ void stage_lds(const cplx* src, cplx* dst, const cplx* __restrict__ tw, int N, int Ns,
                                          int nlines, int es, int ls) {
    const int M = N / R, tstep = N / (Ns * R), total = nlines * M;
    cplx wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        wr[r] = tw[r * M];
        if (INVERSE) wr[r].y = -wr[r].y;
    }
    for (int w = threadIdx.x; w < total; w += blockDim.x) {
        const int line = w / M, j = w - line * M;
        const int hi = j / Ns, k = j - hi * Ns;
        const cplx* in = src + (int64_t)line * ls + (int64_t)j * es;
        cplx x[R];
#pragma unroll
        for (int q = 0; q < R; ++q) x[q] = in[(int64_t)q * M * es];
        if (Ns > 1) {
            const int e = k * tstep; // q e < N for q < R
#pragma unroll
            for (int q = 1; q < R; ++q) {
                cplx t = tw[q * e];
                if (INVERSE) t.y = -t.y;
                x[q] = cmul(x[q], t);
            }
        }
        cplx* out = dst + (int64_t)line * ls + (int64_t)(hi * Ns * R + k) * es;
        const int64_t os = (int64_t)Ns * es;
        if constexpr (R == 2) {
            out[0] = cadd(x[0], x[1]);
            out[os] = csub(x[0], x[1]);
        } else if constexpr (R == 4) {
            const cplx a = cadd(x[0], x[2]), b = csub(x[0], x[2]), c = cadd(x[1], x[3]), d0 = csub(x[1], x[3]);
            const cplx d = INVERSE ? make_double2(-d0.y, d0.x) : make_double2(d0.y, -d0.x); // -+ i (x1 - x3)
            out[0] = cadd(a, c);
            out[os] = cadd(b, d);
            out[2 * os] = csub(a, c);
            out[3 * os] = csub(b, d);
        } else if constexpr (R == 8 || R == 16) { // register butterflies of the power-of-two kernels: one LDS pass for 3 / 4 bits
            pow2::Bfly<R, INVERSE>::run(x);
#pragma unroll
            for (int p = 0; p < R; ++p) out[p * os] = x[p];
        } else {
#pragma unroll
            for (int p = 0; p < R; ++p) {
                cplx acc = x[0];
#pragma unroll
                for (int q = 1; q < R; ++q) {
                    const cplx t = wr[(p * q) % R];
                    acc.x = fma(x[q].x, t.x, fma(-x[q].y, t.y, acc.x));
                    acc.y = fma(x[q].x, t.y, fma(x[q].y, t.x, acc.y));
                }
                out[p * os] = acc;
            }
        }
    }
}

// Stockham autosort stage loop over `nlines` lines living in LDS. Element e of line l sits at
// src[l * ls + e * es]. Radices 2, 3, 4, 5, 7, 8, 11, 13, 16 take stage_lds (8 and 16: only plans that ask for them); any other prime (e.g. 67: n = 134 = 2 * 67)
// the generic loop below: one work item = one OUTPUT element (R complex MACs against the w_N table).
// Returns the buffer that holds the result.
template <bool INVERSE>
__device__ cplx* fft_lines_lds(cplx* src, cplx* dst, const LinePlan& p, int nlines, int es, int ls) {
    const int N = p.n;
    int Ns = 1;
    for (int s = 0; s < p.ns; ++s) {
        const int R = p.r[s];
        bool done = true;
        switch (R) {
        case 2: stage_lds<2, INVERSE>(src, dst, p.tw, N, Ns, nlines, es, ls); break;
        case 3: stage_lds<3, INVERSE>(src, dst, p.tw, N, Ns, nlines, es, ls); break;
        case 4: stage_lds<4, INVERSE>(src, dst, p.tw, N, Ns, nlines, es, ls); break;
        case 5: stage_lds<5, INVERSE>(src, dst, p.tw, N, Ns, nlines, es, ls); break;
        case 7: stage_lds<7, INVERSE>(src, dst, p.tw, N, Ns, nlines, es, ls); break;
        case 8: stage_lds<8, INVERSE>(src, dst, p.tw, N, Ns, nlines, es, ls); break;
        case 16: stage_lds<16, INVERSE>(src, dst, p.tw, N, Ns, nlines, es, ls); break;
        case 11: stage_lds<11, INVERSE>(src, dst, p.tw, N, Ns, nlines, es, ls); break;
        case 13: stage_lds<13, INVERSE>(src, dst, p.tw, N, Ns, nlines, es, ls); break;
        default: done = false;
        }
        if (done) {
            __syncthreads();
            cplx* tmp = src;
            src = dst;
            dst = tmp;
            Ns *= R;
            continue;
        }
        const int M = N / R;
        const int tstep = N / (Ns * R);
        const int total = nlines * N;
        for (int w = threadIdx.x; w < total; w += blockDim.x) {
            const int line = w / N;
            const int o = w - line * N;
            const int k = o % Ns;
            const int t = o / Ns;
            const int pp = t % R;
            const int j = (t / R) * Ns + k;
            const int base = (int)(((int64_t)k * tstep + (int64_t)pp * M) % N);
            const cplx* in = src + (int64_t)line * ls + (int64_t)j * es;
            double ar = 0.0, ai = 0.0;
            int e = 0;
            for (int q = 0; q < R; ++q) {
                const cplx v = in[(int64_t)q * M * es];
                cplx tw = p.tw[e];
                if (INVERSE) tw.y = -tw.y;
                ar = fma(v.x, tw.x, ar);
                ar = fma(-v.y, tw.y, ar);
                ai = fma(v.x, tw.y, ai);
                ai = fma(v.y, tw.x, ai);
                e += base;
                if (e >= N) e -= N;
            }
            dst[(int64_t)line * ls + (int64_t)o * es] = make_double2(ar, ai);
        }
        __syncthreads();
        cplx* tmp = src;
        src = dst;
        dst = tmp;
        Ns *= R;
    }
    return src;
}

