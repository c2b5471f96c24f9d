// Mixed-radix Stockham line FFT on lines that live in LDS (any length: arbitrary prime factors). Shared by
// the small-lattice kernel and the generic three-pass kernels.
#pragma once
#include "gpt_common.h"
#include "gpt_bfly.h"

// One Stockham stage of radix R (compile-time) over `nlines` lines living in LDS: one work item = one BUTTERFLY
// (R inputs M = N/R apart, R outputs Ns apart), inputs and twiddles loaded once per butterfly, the R-point DFT
// done on registers against the R roots w_R^r (R^2 complex MACs: fine for the small primes; 2 and 4 use
// additions only).
// exact w / d for w * d < 2^32 (LDS offsets and work-item numbers: w < 2^17, d <= 4096): one multiply-high instead of
// the ~40-instruction integer division; `m` = fastdiv_magic(d), computed once per stage
__device__ __forceinline__ unsigned fastdiv_magic(unsigned d) { return d == 1 ? 0u : 0xFFFFFFFFu / d + 1u; }
__device__ __forceinline__ unsigned fastdiv(unsigned w, unsigned d, unsigned m) { return d == 1 ? w : __umulhi(w, m); }

// cos / sin of 2 pi m / R for the odd prime radices with a butterfly of their own (m any integer; folded at compile time)
template <int R> struct constexpr_root {
    double c, s;
    __host__ __device__ constexpr constexpr_root(int m) : c(0.0), s(0.0) {
        constexpr double C7[3] = {0.62348980185873359439, -0.22252093395631433737, -0.90096886790241903498};
        constexpr double S7[3] = {0.78183148246802980363, 0.97492791218182361934, 0.43388373911755823142};
        constexpr double C11[5] = {0.84125353283118120551, 0.41541501300188643508, -0.14231483827328500480,
                                   -0.65486073394528498959, -0.95949297361449736865};
        constexpr double S11[5] = {0.54064081745559755543, 0.90963199535451833011, 0.98982144188093279524,
                                   0.75574957435425826890, 0.28173255684142967104};
        constexpr double C13[6] = {0.88545602565320991051, 0.56806474673115592289, 0.12053668025532300601,
                                   -0.35460488704253545489, -0.74851074817110119231, -0.97094181742605201180};
        constexpr double S13[6] = {0.46472317204376850652, 0.82298386589365635224, 0.99270887409805397272,
                                   0.93501624268541483342, 0.66312265824079519305, 0.23931566428755768339};
        static_assert(R == 7 || R == 11 || R == 13, "radix without a root table");
        m %= R;
        const bool refl = m > (R - 1) / 2; // cos(2 pi (R - m) / R) = cos(2 pi m / R), sin changes sign
        const int k = refl ? R - m : m;
        if (k == 0) {
            c = 1.0;
            s = 0.0;
        } else {
            c = R == 7 ? C7[(k - 1) % 3] : R == 11 ? C11[(k - 1) % 5] : C13[(k - 1) % 6];
            s = R == 7 ? S7[(k - 1) % 3] : R == 11 ? S11[(k - 1) % 5] : S13[(k - 1) % 6];
            if (refl) s = -s;
        }
    }
};

// TWP: pointer type of the twiddle table w_N^m -- `const cplx*` (global memory) or LdsTw (a table the kernel copied into
// LDS: the address space is part of the type, so that the look-ups compile to ds_read and not to flat loads).
// The lines themselves are always in LDS and are addressed through LDS-typed pointers as well (LdsLine): through plain
// `cplx*` the compiler could not prove it across the call and emitted flat loads / stores for every element.
#ifndef GPT_STAGE_STAMP
#define GPT_STAGE_STAMP(R) // (diagnostic build of gpt_small.hip: cycle counter after every stage)
#endif
#ifndef GPT_LDS_FFT_INLINE
#define GPT_LDS_FFT_INLINE // (gpt_small.hip: __forceinline__ -- a single small lattice is latency, and a call spills)
#endif
// (a small handle class rather than a bare `__attribute__((address_space(3))) cplx*`: cplx is a class type whose
// assignment operator wants a generic `this`, which the host pass of the compiler rejects for an LDS object)
typedef double gpt_d2 __attribute__((ext_vector_type(2)));
struct LdsRef {
    __attribute__((address_space(3))) gpt_d2* p;
    __device__ __forceinline__ operator cplx() const {
        const gpt_d2 v = *p;
        return make_double2(v.x, v.y);
    }
    __device__ __forceinline__ void operator=(const cplx& v) const { *p = gpt_d2{v.x, v.y}; }
};
struct LdsLine {
    __attribute__((address_space(3))) gpt_d2* p;
    __device__ __forceinline__ LdsLine() : p(nullptr) {}
    __device__ __forceinline__ explicit LdsLine(void* generic) : p((__attribute__((address_space(3))) gpt_d2*)generic) {}
    __device__ __forceinline__ LdsRef operator[](int i) const { return LdsRef{p + i}; }
    __device__ __forceinline__ LdsLine operator+(int i) const {
        LdsLine r;
        r.p = p + i;
        return r;
    }
    __device__ __forceinline__ bool operator==(const LdsLine& o) const { return p == o.p; }
    __device__ __forceinline__ bool operator!=(const LdsLine& o) const { return p != o.p; }
};
typedef LdsLine LdsTw; // a twiddle table the kernel copied into LDS

// Per-stage constants of a Stockham stage: radix R over sub-transforms of length Ns done so far, M = N / R butterflies per
// line, twiddle step N / (Ns R), and the multiply-high magics of M and Ns. Either worked out in the kernel (stage_consts:
// one integer division each -- ~25 dependent instructions, and the optimiser hoists them for EVERY radix of the switch
// below in front of the stage loop), or on the host once per launch (gpt_small.hip: a single small lattice is latency,
// and these divisions were half of every stage's instructions).
struct StageK {
    int R, Ns, M, tstep;
    unsigned mM, mNs;
};
__host__ __device__ __forceinline__ unsigned fastdiv_magic_hd(unsigned d) { return d == 1 ? 0u : 0xFFFFFFFFu / d + 1u; }
__host__ __device__ __forceinline__ StageK stage_consts(int N, int R, int Ns) {
    StageK k;
    k.R = R;
    k.Ns = Ns;
    k.M = N / R;
    k.tstep = N / (Ns * R);
    k.mM = fastdiv_magic_hd((unsigned)k.M);
    k.mNs = fastdiv_magic_hd((unsigned)Ns);
    return k;
}
// The R-point butterfly of one work item: inputs x[0..R) (already multiplied by their twiddles), outputs to out[p * os].
// OUT: anything indexable that yields an assignable reference (an LDS line, or a register array with os = 1).
template <int R, bool INVERSE, typename OUT> __device__ __forceinline__ void bfly_store(cplx (&x)[R], const OUT out, const int os) {
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
    } else if constexpr (R == 3) {
        constexpr double s3 = 0.86602540378443864676; // sin(2 pi / 3)
        const cplx t = cadd(x[1], x[2]), u = csub(x[1], x[2]);
        const cplx m1 = make_double2(fma(-0.5, t.x, x[0].x), fma(-0.5, t.y, x[0].y));
        // -+ i s3 u
        const cplx m2 = INVERSE ? make_double2(-s3 * u.y, s3 * u.x) : make_double2(s3 * u.y, -s3 * u.x);
        out[0] = cadd(x[0], t);
        out[os] = cadd(m1, m2);
        out[2 * os] = csub(m1, m2);
    } else if constexpr (R == 5) {
        constexpr double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410; // cos(2 pi / 5), cos(4 pi / 5)
        constexpr double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917; // sin(2 pi / 5), sin(4 pi / 5)
        const cplx t1 = cadd(x[1], x[4]), t2 = cadd(x[2], x[3]), t3 = csub(x[1], x[4]), t4 = csub(x[2], x[3]);
        const cplx a1 = make_double2(fma(c1, t1.x, fma(c2, t2.x, x[0].x)), fma(c1, t1.y, fma(c2, t2.y, x[0].y)));
        const cplx a2 = make_double2(fma(c2, t1.x, fma(c1, t2.x, x[0].x)), fma(c2, t1.y, fma(c1, t2.y, x[0].y)));
        const cplx b1 = make_double2(fma(s1, t3.x, s2 * t4.x), fma(s1, t3.y, s2 * t4.y));
        const cplx b2 = make_double2(fma(s2, t3.x, -s1 * t4.x), fma(s2, t3.y, -s1 * t4.y));
        // forward: X1 = a1 - i b1, X4 = a1 + i b1, X2 = a2 - i b2, X3 = a2 + i b2 (inverse: the signs swap)
        const cplx ib1 = INVERSE ? make_double2(-b1.y, b1.x) : make_double2(b1.y, -b1.x);
        const cplx ib2 = INVERSE ? make_double2(-b2.y, b2.x) : make_double2(b2.y, -b2.x);
        out[0] = make_double2(x[0].x + t1.x + t2.x, x[0].y + t1.y + t2.y);
        out[os] = cadd(a1, ib1);
        out[2 * os] = cadd(a2, ib2);
        out[3 * os] = csub(a2, ib2);
        out[4 * os] = csub(a1, ib1);
    } else if constexpr (R == 8 || R == 16) { // register butterflies of the power-of-two kernels: one LDS pass for 3 / 4 bits
        pow2::Bfly<R, INVERSE>::run(x);
#pragma unroll
        for (int p = 0; p < R; ++p) out[p * os] = x[p];
    } else {
        // odd prime radix (7, 11, 13), inputs paired about the middle: with s_q = x_q + x_{R-q}, d_q = x_q - x_{R-q},
        //   X_p = a_p -+ i b_p,  X_{R-p} = a_p +- i b_p,  a_p = x_0 + sum_q s_q cos(2 pi p q / R),  b_p = sum_q d_q sin(...)
        // -- (R - 1)^2 real multiply-adds instead of the 4 (R - 1)^2 of the plain R-point sum
        constexpr int h = (R - 1) / 2;
        cplx sq[h], dq[h];
#pragma unroll
        for (int q = 1; q <= h; ++q) {
            sq[q - 1] = cadd(x[q], x[R - q]);
            dq[q - 1] = csub(x[q], x[R - q]);
        }
        cplx o0 = x[0];
#pragma unroll
        for (int q = 0; q < h; ++q) o0 = cadd(o0, sq[q]);
        out[0] = o0;
#pragma unroll
        for (int p = 1; p <= h; ++p) {
            cplx av = x[0], bv = make_double2(0.0, 0.0);
#pragma unroll
            for (int q = 1; q <= h; ++q) {
                constexpr_root<R> rt(p * q);
                av.x = fma(sq[q - 1].x, rt.c, av.x);
                av.y = fma(sq[q - 1].y, rt.c, av.y);
                bv.x = fma(dq[q - 1].x, rt.s, bv.x);
                bv.y = fma(dq[q - 1].y, rt.s, bv.y);
            }
            // forward: X_p = a - i b, X_{R-p} = a + i b (inverse: the other way round)
            const cplx lo = make_double2(av.x + bv.y, av.y - bv.x), hi2 = make_double2(av.x - bv.y, av.y + bv.x);
            out[p * os] = INVERSE ? hi2 : lo;
            out[(R - p) * os] = INVERSE ? lo : hi2;
        }
    }
}

template <int R, bool INVERSE, typename TWP>
__device__ __forceinline__ void stage_lds(const LdsLine src, const LdsLine dst, TWP tw, const StageK& sk, int nlines,
                                          unsigned mL, int es, int ls) {
    // (M = N / R, tstep = N / (Ns R) and the division magics of M, Ns, nlines arrive ready-made: see StageK)
    const int Ns = sk.Ns, M = sk.M, tstep = sk.tstep, total = nlines * M;
    const unsigned mM = sk.mM, mNs = sk.mNs;
    // (unrolling this loop 2-4 times so that the next butterfly's loads are issued early was measured: 10-20 % slower)
    // Work items are numbered so that neighbouring lanes touch neighbouring LDS addresses: along a line when its
    // elements are contiguous (es == 1: row tiles), across the lines when the lines are interleaved (ls == 1: column
    // tiles, element stride = tile width -- numbered along the line, every access of a wave was a 2- to 4-way bank
    // conflict).
    const bool across = (ls == 1 && es > 1);
#pragma unroll 1
    for (int w = threadIdx.x; w < total; w += blockDim.x) {
        int line, j;
        if (across) {
            j = (int)fastdiv((unsigned)w, (unsigned)nlines, mL);
            line = w - j * nlines;
        } else {
            line = (int)fastdiv((unsigned)w, (unsigned)M, mM);
            j = w - line * M;
        }
        const int hi = (int)fastdiv((unsigned)j, (unsigned)Ns, mNs), k = j - hi * Ns;
        const LdsLine in = src + (line * ls + j * es);
        cplx x[R];
#pragma unroll
        for (int q = 0; q < R; ++q) x[q] = in[q * M * es];
        if (Ns > 1) {
            const int e = k * tstep; // q e < N for q < R
            // (computing w^e with sincospi instead of loading it, for stages with only a butterfly or two per thread, was
            // measured: no faster per workgroup, 15-25 % slower per launch)
#pragma unroll
            for (int q = 1; q < R; ++q) {
                cplx t = tw[q * e];
                if (INVERSE) t.y = -t.y;
                x[q] = cmul(x[q], t);
            }
        }
        const LdsLine out = dst + (line * ls + (hi * Ns * R + k) * es);
        const int os = Ns * es;
        bfly_store<R, INVERSE>(x, out, os);
    }
}

// Stockham autosort stage loop over `nlines` lines living in LDS. Element e of line l sits at
// src[l * ls + e * es]. Radices 2, 3, 4, 5, 7, 8, 11, 13, 16 take stage_lds (8 and 16: only plans that ask for them); any other prime (e.g. 67: n = 134 = 2 * 67)
// the generic loop below: one work item = one OUTPUT element (R complex MACs against the w_N table).
// Returns the buffer that holds the result.
//
// stage_lds_any: one stage with ready-made constants; false: the radix has no butterfly of its own (generic loop of the caller).
template <bool INVERSE, typename TWP>
__device__ __forceinline__ bool stage_lds_any(LdsLine src, LdsLine dst, TWP ptw, const StageK& sk, int nlines, unsigned mL,
                                              int es, int ls) {
    switch (sk.R) {
    case 2: stage_lds<2, INVERSE, TWP>(src, dst, ptw, sk, nlines, mL, es, ls); return true;
    case 3: stage_lds<3, INVERSE, TWP>(src, dst, ptw, sk, nlines, mL, es, ls); return true;
    case 4: stage_lds<4, INVERSE, TWP>(src, dst, ptw, sk, nlines, mL, es, ls); return true;
    case 5: stage_lds<5, INVERSE, TWP>(src, dst, ptw, sk, nlines, mL, es, ls); return true;
    case 7: stage_lds<7, INVERSE, TWP>(src, dst, ptw, sk, nlines, mL, es, ls); return true;
    case 8: stage_lds<8, INVERSE, TWP>(src, dst, ptw, sk, nlines, mL, es, ls); return true;
    case 16: stage_lds<16, INVERSE, TWP>(src, dst, ptw, sk, nlines, mL, es, ls); return true;
    case 11: stage_lds<11, INVERSE, TWP>(src, dst, ptw, sk, nlines, mL, es, ls); return true;
    case 13: stage_lds<13, INVERSE, TWP>(src, dst, ptw, sk, nlines, mL, es, ls); return true;
    default: return false;
    }
}

// `stages`: null (constants worked out here, stage by stage) or the host-made table of the plan's p.ns stages.
template <bool INVERSE, typename TWP>
__device__ GPT_LDS_FFT_INLINE LdsLine fft_lines_lds_tw(LdsLine src, LdsLine dst, const LinePlan& p, TWP ptw, int nlines, int es,
                                                       int ls, const StageK* stages = nullptr, unsigned mL_ready = 0) {
    const int N = p.n;
    const unsigned mL = stages ? mL_ready : fastdiv_magic((unsigned)nlines);
    int Ns = 1;
    for (int s = 0; s < p.ns; ++s) {
        const StageK sk = stages ? stages[s] : stage_consts(N, p.r[s], Ns);
        const int R = sk.R;
        const bool done = stage_lds_any<INVERSE, TWP>(src, dst, ptw, sk, nlines, mL, es, ls);
        if (done) {
            __syncthreads();
            GPT_STAGE_STAMP(R);
            LdsLine tmp = src;
            src = dst;
            dst = tmp;
            Ns *= R;
            continue;
        }
        const int M = sk.M, tstep = sk.tstep, total = nlines * N;
        const unsigned mN = fastdiv_magic((unsigned)N), mNs = sk.mNs, mR = fastdiv_magic((unsigned)R);
        const bool across = (ls == 1 && es > 1); // (as in stage_lds)
        for (int w = threadIdx.x; w < total; w += blockDim.x) {
            int line, o;
            if (across) {
                o = (int)fastdiv((unsigned)w, (unsigned)nlines, mL);
                line = w - o * nlines;
            } else {
                line = (int)fastdiv((unsigned)w, (unsigned)N, mN);
                o = w - line * N;
            }
            const int t = (int)fastdiv((unsigned)o, (unsigned)Ns, mNs);
            const int k = o - t * Ns;
            const int th = (int)fastdiv((unsigned)t, (unsigned)R, mR);
            const int pp = t - th * R;
            const int j = th * Ns + k;
            const int base = (int)(((int64_t)k * tstep + (int64_t)pp * M) % N);
            const LdsLine in = src + (line * ls + j * es);
            double ar = 0.0, ai = 0.0;
            int e = 0;
            for (int q = 0; q < R; ++q) {
                const cplx v = in[q * M * es];
                cplx tw = ptw[e];
                if (INVERSE) tw.y = -tw.y;
                ar = fma(v.x, tw.x, ar);
                ar = fma(-v.y, tw.y, ar);
                ai = fma(v.x, tw.y, ai);
                ai = fma(v.y, tw.x, ai);
                e += base;
                if (e >= N) e -= N;
            }
            dst[line * ls + o * es] = make_double2(ar, ai);
        }
        __syncthreads();
        LdsLine tmp = src;
        src = dst;
        dst = tmp;
        Ns *= R;
    }
    return src;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same stages IN PLACE: one LDS buffer instead of two. Every thread reads the inputs of all its butterflies into
// registers, the workgroup meets at a barrier, then the butterflies are done and written to their Stockham positions of
// the SAME buffer (a second barrier ends the stage). The registers are the second buffer: EPT = elements a thread may
// hold (the caller keeps nlines * N <= EPT * blockDim.x), so a stage of radix R runs ceil(EPT / R) butterflies per
// thread. Twice the barriers of the ping-pong form and half its LDS: for the three-pass kernels of gpt_generic.hip, whose
// tiles are memory latency and want two or three workgroups on a CU (the one-workgroup kernels keep the ping-pong form).
template <int R, bool INVERSE, int EPT, typename TWP>
__device__ __forceinline__ void stage_lds_ip(const LdsLine buf, TWP tw, const StageK& sk, int nlines, unsigned mL, int es,
                                             int ls) {
    constexpr int IT = (EPT + R - 1) / R;
    const int Ns = sk.Ns, M = sk.M, tstep = sk.tstep, total = nlines * M;
    const unsigned mM = sk.mM, mNs = sk.mNs;
    const bool across = (ls == 1 && es > 1); // (as in stage_lds)
    cplx x[IT][R];
    int oo[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int w = (int)threadIdx.x + it * (int)blockDim.x;
        oo[it] = 0;
        if (w < total) {
            int line, j;
            if (across) {
                j = (int)fastdiv((unsigned)w, (unsigned)nlines, mL);
                line = w - j * nlines;
            } else {
                line = (int)fastdiv((unsigned)w, (unsigned)M, mM);
                j = w - line * M;
            }
            const int hi = (int)fastdiv((unsigned)j, (unsigned)Ns, mNs), k = j - hi * Ns;
            const LdsLine in = buf + (line * ls + j * es);
#pragma unroll
            for (int q = 0; q < R; ++q) x[it][q] = in[q * M * es];
            if (Ns > 1) {
                const int e = k * tstep;
#pragma unroll
                for (int q = 1; q < R; ++q) {
                    // (twiddles from a table in LDS instead: bounded by a build that computes a stand-in and loads nothing --
                    // 13 % of a 1000 x 4 column tile, 14 % of 3000 x 1, nothing at 192 x 8; not worth the LDS)
                    cplx t = tw[q * e];
                    if (INVERSE) t.y = -t.y;
                    x[it][q] = cmul(x[it][q], t);
                }
            }
            oo[it] = line * ls + (hi * Ns * R + k) * es;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int w = (int)threadIdx.x + it * (int)blockDim.x;
        if (w < total) bfly_store<R, INVERSE>(x[it], buf + oo[it], Ns * es);
    }
    __syncthreads();
}

// any other prime radix, in place: one work item = one OUTPUT element, a thread's outputs wait in registers for the barrier
template <bool INVERSE, int EPT, typename TWP>
__device__ __forceinline__ void stage_lds_ip_prime(const LdsLine buf, TWP ptw, const StageK& sk, int N, int nlines, unsigned mL,
                                                   int es, int ls) {
    const int R = sk.R, Ns = sk.Ns, M = sk.M, tstep = sk.tstep, total = nlines * N;
    const unsigned mN = fastdiv_magic((unsigned)N), mNs = sk.mNs, mR = fastdiv_magic((unsigned)R);
    const bool across = (ls == 1 && es > 1);
    cplx y[EPT];
    int oo[EPT];
#pragma unroll
    for (int it = 0; it < EPT; ++it) {
        const int w = (int)threadIdx.x + it * (int)blockDim.x;
        oo[it] = 0;
        y[it] = make_double2(0.0, 0.0);
        if (w < total) {
            int line, o;
            if (across) {
                o = (int)fastdiv((unsigned)w, (unsigned)nlines, mL);
                line = w - o * nlines;
            } else {
                line = (int)fastdiv((unsigned)w, (unsigned)N, mN);
                o = w - line * N;
            }
            const int t = (int)fastdiv((unsigned)o, (unsigned)Ns, mNs);
            const int k = o - t * Ns;
            const int th = (int)fastdiv((unsigned)t, (unsigned)R, mR);
            const int pp = t - th * R;
            const int j = th * Ns + k;
            const int base = (int)(((int64_t)k * tstep + (int64_t)pp * M) % N);
            const LdsLine in = buf + (line * ls + j * es);
            double ar = 0.0, ai = 0.0;
            int e = 0;
            for (int q = 0; q < R; ++q) {
                const cplx v = in[q * M * es];
                cplx tw = ptw[e];
                if (INVERSE) tw.y = -tw.y;
                ar = fma(v.x, tw.x, ar);
                ar = fma(-v.y, tw.y, ar);
                ai = fma(v.x, tw.y, ai);
                ai = fma(v.y, tw.x, ai);
                e += base;
                if (e >= N) e -= N;
            }
            y[it] = make_double2(ar, ai);
            oo[it] = line * ls + o * es;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < EPT; ++it) {
        const int w = (int)threadIdx.x + it * (int)blockDim.x;
        if (w < total) buf[oo[it]] = y[it];
    }
    __syncthreads();
}

// all the stages of plan `p` over `nlines` lines of `buf`, in place (stage constants and the magic of nlines from the host);
// the caller's data must be visible (a barrier) before the call, the result is visible to the whole workgroup after it.
// LEAN: the instantiation for plans of radices 2, 3, 4, 5 and 8 only (lean_plan below: what a 5-smooth length gets) --
// a kernel's registers are the maximum over every stage it COULD run, and the radix-7 / 11 / 13 / 16 / any-prime stages
// are what that maximum is: 117 - 127 registers for the generic row kernels with them, 64 - 68 without (8 waves per SIMD
// instead of 4), 128 with 10 - 22 spilled against 86 - 96 for the column kernel. (Radix 7 in the lean set was compiled:
// +24 registers everywhere -- 88 for the row kernels, 106 - 117 for the column kernel -- so 7-smooth lengths stay full.)
template <bool INVERSE, int EPT, typename TWP, bool LEAN = false>
__device__ __forceinline__ void fft_lines_lds_ip(const LdsLine buf, const LinePlan& p, TWP ptw, int nlines, int es, int ls,
                                                 const StageK* stages, unsigned mL) {
    int Ns = 1;
    for (int s = 0; s < p.ns; ++s) {
        const StageK sk = stages ? stages[s] : stage_consts(p.n, p.r[s], Ns); // (null: worked out here, gpt_anyn.hip)
        Ns *= sk.R;
        switch (sk.R) {
        case 2: stage_lds_ip<2, INVERSE, EPT, TWP>(buf, ptw, sk, nlines, mL, es, ls); break;
        case 3: stage_lds_ip<3, INVERSE, EPT, TWP>(buf, ptw, sk, nlines, mL, es, ls); break;
        case 4: stage_lds_ip<4, INVERSE, EPT, TWP>(buf, ptw, sk, nlines, mL, es, ls); break;
        case 5: stage_lds_ip<5, INVERSE, EPT, TWP>(buf, ptw, sk, nlines, mL, es, ls); break;
        case 8: stage_lds_ip<8, INVERSE, EPT, TWP>(buf, ptw, sk, nlines, mL, es, ls); break;
        default:
            if constexpr (!LEAN) {
                switch (sk.R) {
                case 7: stage_lds_ip<7, INVERSE, EPT, TWP>(buf, ptw, sk, nlines, mL, es, ls); break;
                case 11: stage_lds_ip<11, INVERSE, EPT, TWP>(buf, ptw, sk, nlines, mL, es, ls); break;
                case 13: stage_lds_ip<13, INVERSE, EPT, TWP>(buf, ptw, sk, nlines, mL, es, ls); break;
                case 16: stage_lds_ip<16, INVERSE, EPT, TWP>(buf, ptw, sk, nlines, mL, es, ls); break;
                default: stage_lds_ip_prime<INVERSE, EPT, TWP>(buf, ptw, sk, p.n, nlines, mL, es, ls); break;
                }
            }
            break;
        }
    }
}

__host__ __device__ inline bool lean_plan(const LinePlan& p) {
    for (int s = 0; s < p.ns; ++s) {
        const int r = p.r[s];
        if (!(r == 2 || r == 3 || r == 4 || r == 5 || r == 8)) return false;
    }
    return true;
}

// twiddles where the plan has them (global memory)
template <bool INVERSE>
__device__ cplx* fft_lines_lds(cplx* src, cplx* dst, const LinePlan& p, int nlines, int es, int ls) {
    const LdsLine r = fft_lines_lds_tw<INVERSE, const cplx*>(LdsLine(src), LdsLine(dst), p, p.tw, nlines, es, ls);
    return r == LdsLine(src) ? src : dst;
}
