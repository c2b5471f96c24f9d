// Mixed-radix Stockham line FFT on lines that live in LDS (any length: arbitrary prime factors). Shared by
// the small-lattice kernel and the generic three-pass kernels.
#pragma once
#include "gpt_common.h"

// Stockham autosort stage loop over `nlines` lines living in LDS. Element e of line l sits at
// src[l * ls + e * es]. One work item = one OUTPUT element (R complex MACs against the w_N table), which
// keeps every radix - including large primes such as 67 (n = 134 = 2 * 67) - on the same code path.
// Returns the buffer that holds the result.
template <bool INVERSE>
__device__ cplx* fft_lines_lds(cplx* src, cplx* dst, const LinePlan& p, int nlines, int es, int ls) {
    const int N = p.n;
    int Ns = 1;
    for (int s = 0; s < p.ns; ++s) {
        const int R = p.r[s];
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

