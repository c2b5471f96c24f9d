// Power-of-two three-pass path (the 4096 x 4096 headline shape): templates, instantiated once per line
// length N in gpt_pow2_n.hip.
//
//   pass A  row_fwd   one workgroup per ROW PAIR: (y - loc) of rows 2r, 2r+1 -> one complex line of length
//                     W (two-for-one real transform), register-resident radix-16 Stockham FFT, split, write
//                     the packed half spectrum S[H][W/2] (DC + i*Nyquist in column 0).
//   pass B  col       one workgroup per TC adjacent packed columns: column FFT (length H), the op's
//                     elementwise stage (log-density terms, F/c, d/dcov, ...) on registers, inverse column
//                     FFT, write back IN PLACE. Wavefront shuffle + LDS reduction -> per-workgroup partials.
//   pass C  row_inv   one workgroup per row pair: merge the two half spectra into one complex line,
//                     inverse FFT, write the two real rows (scaled, +loc / negated as the op needs).
//
// Line FFT: a "register tile" is E = 16 complex fp64 elements (v[m] <-> element t + T*m, T = N/16 tiles per
// line). Every Stockham stage is a radix-16 (last stage 2/4/8/16) butterfly on registers followed by an
// exchange through LDS; LDS is addressed through an XOR swizzle so that both the strided writes and the
// contiguous reads of an exchange are (nearly) bank-conflict free without padding; re and im travel
// separately, so a line needs N * 8 bytes. The row kernels hold one tile per thread; the column kernel
// holds V = 2 tiles per thread, which makes a workgroup TC = 4 columns wide at H = 4096.
//
// Why the column kernel wants wide tiles (measured on MI355X with scripts/trace_col.py, profiles/r01_d_*):
// the pass is bound by the NUMBER of line requests the memory pipeline takes, not by bytes -- a request
// that uses 32 of a line's 128 bytes costs as much as one that uses all of them, loads and stores alike
// (with 2-column tiles 13 of 38 us per tile went into just issuing the spectrum loads). TC adjacent
// columns are TC*16 contiguous bytes of a spectrum row and TC*8 bytes of a cov / d/dcov row.
#pragma once
#include "gpt_common.h"
#include "gpt_host.h"
#include "gpt_obs_dev.h"
#include "gpt_bfly.h"

#ifndef GPT_TRACE
#define GPT_TRACE 0
#endif
#ifndef GPT_ROWS_SAME
#define GPT_ROWS_SAME 0
#endif
// Timing-only build (make traffic -> libvar_traffic.so, bench.py's roofline.three_pass_ceiling): the three passes with
// every load and store of the real kernels and none of the arithmetic or LDS traffic between them. Results are garbage.
#ifndef GPT_TRAFFIC_ONLY
#define GPT_TRAFFIC_ONLY 0
#endif
#if GPT_TRACE
// debug builds only (scripts/trace_col.py): phase timestamps (100 MHz clock) of wave 0 of every column workgroup
__device__ unsigned long long g_col_trace[1024 * 8 * 8];
#define TRACE_MARK(r_, k_)                                                                         \
    do {                                                                                           \
        if (threadIdx.x == 0 && (r_) < 8)                                                          \
            g_col_trace[((blockIdx.x & 1023) * 8 + (r_)) * 8 + (k_)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define TRACE_MARK(r_, k_) ((void)0)
#endif

namespace pow2 {

#ifndef GPT_E
#define GPT_E 16
#endif
constexpr int E = GPT_E; // elements of a register tile (= the largest butterfly radix)
#ifndef GPT_COL_WGS_PER_CU
#define GPT_COL_WGS_PER_CU 1
#endif

// register tiles per thread in the column kernel
#ifndef GPT_COL_V
#define GPT_COL_V 2
#endif
#ifndef GPT_COL_THREADS
#define GPT_COL_THREADS 512
#endif

// LDS address swizzle of a line (index in doubles). Mode 0: XOR the low four bits with the next four. The
// LDS serves a ds_read_b64 in two groups of 32 lanes over 64 four-byte banks and a ds_write_b64 in four
// groups of 16 lanes over 32 banks (MI355X_MICROARCH.md, LDS): one line per 16/32 consecutive lanes is
// conflict free with mode 0. Mode 1 (column kernel, 4 interleaved columns: a read group is 8 positions x 4
// columns, a write group 4 x 4) additionally swaps address bits 2 and 4, so that 8 consecutive positions
// land in two 4-element runs 16 elements apart; with the columns' regions skewed by 4 elements both the
// read and the write groups then touch every bank once.
template <int MODE = 0> __device__ __forceinline__ int phys(int o) {
    int q = o ^ ((o >> 4) & 15);
    if (MODE == 1) {
        const int d = ((q >> 2) ^ (q >> 4)) & 1;
        q ^= (d << 2) | (d << 4);
    }
    return q;
}

// Twiddle bases. Stage s (radix R, NS = product of the earlier radices, NS > 1) multiplies input q of
// butterfly u by w^q with w = w_N^{((t + T u) mod NS) * N/(NS R)}: these bases depend on the thread only,
// so a kernel loads them ONCE (NPRE table look-ups) and no transform issues a global load afterwards --
// which also keeps the transforms from waiting on outstanding prefetches (vmcnt retires in order).
// Radix of the stage that starts with NS transformed points: the odd-sized radix (2, 4 or 8, when log2 N is not
// a multiple of 4) comes FIRST, where NS = 1 means no twiddles at all; every later stage is a radix-16 stage
// with one twiddle base per register tile.
constexpr int stage_radix(int n, int ns) {
    if (ns > 1) return E;
    int r = n;
    while (r > E) r /= E;
    return r;
}
template <int N> constexpr int tw_npre() {
    int cnt = 0;
    for (int ns = 1; ns < N;) {
        const int r = stage_radix(N, ns);
        if (ns > 1) cnt += E / r;
        ns *= r;
    }
    return cnt;
}
template <int N, int NS, int IDX>
__device__ __forceinline__ void tw_load_rec(cplx* w, const cplx* __restrict__ tw, int t) {
    constexpr int R = stage_radix(N, NS);
    constexpr int T = N / E, U = E / R;
    if constexpr (NS > 1) {
        constexpr int tstep = N / (NS * R);
#pragma unroll
        for (int u = 0; u < U; ++u) w[IDX + u] = tw[((t + T * u) & (NS - 1)) * tstep];
    }
    if constexpr (NS * R < N) tw_load_rec<N, NS * R, IDX + (NS > 1 ? U : 0)>(w, tw, t);
}
template <int N> __device__ __forceinline__ void tw_load(cplx* w, const cplx* __restrict__ tw, int t) {
    tw_load_rec<N, 1, 0>(w, tw, t);
}

// One Stockham stage on registers: U = E/R butterflies of radix R per thread. The R-1 twiddles of a
// butterfly are powers of its base (a shallow multiplication tree, relative error of a few ulp).
template <int N, int R, int NS, bool INV>
__device__ __forceinline__ void stage_bfly(cplx (&v)[E], const cplx* w1) {
    constexpr int U = E / R;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        cplx x[R];
#pragma unroll
        for (int q = 0; q < R; ++q) x[q] = v[u + q * U];
        if (NS > 1) {
            // powers w^1 .. w^(R-1) as a shallow tree: w2 = w1^2, w3 = w2 w1, w4 = w2^2, then w^(4g + r) =
            // (w4)^g * w^r: depth <= 5 multiplications, relative error of a few ulp
            const cplx wa = tw_dir<INV>(w1[u]);
            x[1] = cmul(x[1], wa);
            if (R > 2) {
                const cplx wb = cmul(wa, wa), wc = cmul(wb, wa);
                x[2] = cmul(x[2], wb);
                x[3] = cmul(x[3], wc);
                if (R > 4) {
                    const cplx w4 = cmul(wb, wb);
                    cplx base = w4;
#pragma unroll
                    for (int g = 1; g < R / 4; ++g) {
                        SCHED_FENCE();
                        x[4 * g] = cmul(x[4 * g], base);
                        x[4 * g + 1] = cmul(x[4 * g + 1], cmul(base, wa));
                        x[4 * g + 2] = cmul(x[4 * g + 2], cmul(base, wb));
                        x[4 * g + 3] = cmul(x[4 * g + 3], cmul(base, wc));
                        if (g + 1 < R / 4) base = cmul(base, w4);
                    }
                }
            }
        }
        SCHED_FENCE();
        Bfly<R, INV>::run(x);
#pragma unroll
        for (int p = 0; p < R; ++p) v[u + p * U] = x[p];
    }
}

// Swizzled LDS position of element t + T*m given rb = phys(t), t < T. Bits 4..7 of t + T*m are (t >> 4) +
// (T/16) m with no carries between the two terms (T/16 is a power of two above t >> 4), so the XOR mask of
// position t + T*m is that of t combined with a compile-time constant: 256/T distinct bases per thread and
// immediate offsets instead of one address register per element.
template <int T, int SWZ> __device__ __forceinline__ int rd_addr(int rb, int t, int m) {
    if (SWZ == 0 && T % 16 == 0) return (rb ^ (((T / 16) * m) & 15)) + T * m;
    if (T % 256 == 0) return rb + T * m;
    return phys<SWZ>(t + T * m);
}

// Exchange after a (non-final) stage: scatter the butterfly outputs to their Stockham positions, gather
// back the elements t + T*m the next stage needs. Real and imaginary parts travel separately, so a line
// needs N * 8 bytes of LDS (32 KB for N = 4096). A thread carries V register tiles (tile k belongs to
// position t[k] of the line); all of them cross in the same four barrier intervals.
template <int N, int R, int NS, int V, int SWZ>
__device__ __forceinline__ void exchange(cplx (&v)[V][E], const int (&t)[V], double* lds) {
    constexpr int T = N / E, U = E / R;
    int o0[V][U];
#pragma unroll
    for (int k = 0; k < V; ++k)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = t[k] + T * u;
            o0[k][u] = (j / NS) * NS * R + (j & (NS - 1));
        }
    __syncthreads(); // everyone is done reading the previous contents of this LDS region
#pragma unroll
    for (int k = 0; k < V; ++k)
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int p = 0; p < R; ++p) lds[phys<SWZ>(o0[k][u] + p * NS)] = v[k][u + p * U].x;
    __syncthreads();
    // (t + T m) >> 4 & 15 does not depend on m when T is a multiple of 256: one swizzled base + offsets
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int rb = phys<SWZ>(t[k]);
#pragma unroll
        for (int m = 0; m < E; ++m) v[k][m].x = lds[rd_addr<T, SWZ>(rb, t[k], m)];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k)
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int p = 0; p < R; ++p) lds[phys<SWZ>(o0[k][u] + p * NS)] = v[k][u + p * U].y;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int rb = phys<SWZ>(t[k]);
#pragma unroll
        for (int m = 0; m < E; ++m) v[k][m].y = lds[rd_addr<T, SWZ>(rb, t[k], m)];
    }
}

template <int N, int NS, int IDX, bool INV, int V, int SWZ>
__device__ __forceinline__ void fft_rec(cplx (&v)[V][E], const int (&t)[V], double* lds,
                                        const cplx (&wpre)[V][tw_npre<N>() > 0 ? tw_npre<N>() : 1]) {
    constexpr int R = stage_radix(N, NS);
    constexpr int U = E / R;
#if GPT_TRAFFIC_ONLY
    return;
#endif
#if GPT_TRACE
    constexpr int SIDX = NS == 1 ? 0 : (NS == 16 ? 1 : 2); // stage number (radix-16 plans)
    if (!INV && V > 1) TRACE_MARK(6, 2 * SIDX);
#endif
#pragma unroll
    for (int k = 0; k < V; ++k) {
        stage_bfly<N, R, NS, INV>(v[k], wpre[k] + IDX);
        SCHED_FENCE();
    }
#if GPT_TRACE
    if (!INV && V > 1) TRACE_MARK(6, 2 * SIDX + 1);
#endif
    if constexpr (NS * R < N) {
        exchange<N, R, NS, V, SWZ>(v, t, lds);
        fft_rec<N, NS * R, IDX + (NS > 1 ? U : 0), INV, V, SWZ>(v, t, lds, wpre);
    }
}

// In: v[k][m] = x[t[k] + T m]. Out: v[k][m] = X[t[k] + T m] (natural order). `lds` is this line's region of
// N doubles, `wpre[k]` the twiddle bases of line position t[k] (tw_load).
template <int N, bool INV, int V, int SWZ>
__device__ __forceinline__ void fft_regs(cplx (&v)[V][E], const int (&t)[V], double* lds,
                                         const cplx (&wpre)[V][tw_npre<N>() > 0 ? tw_npre<N>() : 1]) {
    fft_rec<N, 1, 0, INV, V, SWZ>(v, t, lds, wpre);
}

// one tile per thread (row kernels)
template <int N, bool INV>
__device__ __forceinline__ void fft_regs(cplx (&v)[E], int t, double* lds, const cplx* wpre) {
    cplx(&vv)[1][E] = reinterpret_cast<cplx(&)[1][E]>(v);
    const int tt[1] = {t};
    cplx w[1][tw_npre<N>() > 0 ? tw_npre<N>() : 1];
#pragma unroll
    for (int i = 0; i < tw_npre<N>(); ++i) w[0][i] = wpre[i];
    fft_rec<N, 1, 0, INV, 1, 0>(vv, tt, lds, w);
}

// ---------------------------------------------------------------------------------------------------
// Geometry of the workgroups.
template <int N> struct Geo {
    static constexpr int T = N / E;                      // register tiles per line
    static constexpr int ROW_THREADS = T < 256 ? 256 : T; // row kernels: one tile per thread, >= 256 threads
    static constexpr int ROW_LINES = ROW_THREADS / T;     // row pairs per workgroup
    static constexpr int V = GPT_COL_V;                   // column kernel: tiles per thread
    static constexpr int TP = T / V;                      // ... threads per column
    static constexpr int COL_THREADS = TP < GPT_COL_THREADS ? GPT_COL_THREADS : TP;
    static constexpr int TC = COL_THREADS / TP;           // adjacent packed columns per workgroup
    static constexpr int SWZ = TC == 4 ? 1 : 0;           // LDS swizzle mode of the column kernel (phys)
    static constexpr int SKEW = TC == 4 ? 4 : 8;          // ... and the offset between its columns' regions
};

// Workspace stores of pass A. NTS: non-temporal, for launches whose workspace cannot be cache-resident (units of many
// lattices): the streamed lines leave the Infinity Cache to what is reused (4096^2 x 16: +3 %). Non-temporal LOADS of
// the workspace in pass C were measured too and lose (row_inv 53 -> 70 us).
typedef double gpt_rd2v __attribute__((ext_vector_type(2)));
template <bool NTS> __device__ __forceinline__ void rows_st(cplx* p, cplx v) {
    if (NTS) {
        const gpt_rd2v t = {v.x, v.y};
        __builtin_nontemporal_store(t, reinterpret_cast<gpt_rd2v*>(p));
    } else {
        *p = v;
    }
}

// pass A --------------------------------------------------------------------------------------------
// in_mode: 0 = x, 1 = x - loc, 2 = aux
// SUB2 (sub-line variant with R = 2, gpt_pow2_sub.h): the workgroup's two rows are r and r + H/2 instead of
// 2r and 2r + 1, and the first radix-2 step of the column transform is done on them before they are stored:
// u0 = S[r] + S[r + H/2] -> workspace row r, u1 = (S[r] - S[r + H/2]) w_H^r -> workspace row H/2 + r.
template <int W, bool SUB2 = false, bool NTS = false>
__global__ void __launch_bounds__(Geo<W>::ROW_THREADS) row_fwd_kernel(LatArgs a, int in_mode, cplx* __restrict__ ws,
                                                                      int wstride, double* __restrict__ edge) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = Geo<W>::T, LINES = Geo<W>::ROW_LINES;
    const int line = threadIdx.x / T, t = threadIdx.x - line * T;
    double* lds = reinterpret_cast<double*>(smem) + (size_t)line * W;
    // Workgroups are dispatched x-fastest: numbering them lattice-fastest puts the same rows of all the lattices of
    // a launch next to each other in time, so a loc (and the twiddles) shared by the batch is read from HBM once.
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int64_t b = lin % gridDim.y;
    const int H = a.H;
    const int npairs = H / 2;
    const int r = (int)(lin / gridDim.y) * LINES + line; // row pair
    const bool active = r < npairs;
#if GPT_ROWS_SAME // timing-only builds (scripts/mkvar.sh): every workgroup works on row pair 0 -- the pass without its HBM traffic
    const int rr = 0;
#else
    const int rr = active ? r : 0;
#endif
    const double* src = in_mode == 2 ? a.aux + b * a.aux_bs : a.x + b * a.x_bs;
    const int ra = SUB2 ? rr : 2 * rr, rb = SUB2 ? rr + H / 2 : 2 * rr + 1; // the two rows of this line
    const double* r0 = src + (int64_t)ra * W;
    const double* r1 = src + (int64_t)rb * W;
    cplx wpre[tw_npre<W>()];
    tw_load<W>(wpre, a.pw.tw, t);
    cplx v[E];
#pragma unroll
    for (int m = 0; m < E; ++m) // read once: non-temporal, the cache space is worth more to the workspace (-20 us)
        v[m] = make_double2(__builtin_nontemporal_load(r0 + t + T * m), __builtin_nontemporal_load(r1 + t + T * m));
    if (in_mode == 1 && a.loc) {
        const double* l0 = a.loc + b * a.loc_bs + (int64_t)ra * W;
        const double* l1 = a.loc + b * a.loc_bs + (int64_t)rb * W;
#pragma unroll
        for (int m = 0; m < E; ++m) {
            v[m].x -= l0[t + T * m]; // loc may be shared by the whole batch: ordinary loads
            v[m].y -= l1[t + T * m];
        }
    }
    fft_regs<W, false>(v, t, lds, wpre);
    // split the two rows: need Z[j] (own registers) and Z[W - j] (through LDS, re then im)
    cplx zc[E / 2];
#pragma unroll
    for (int m = 0; m < E / 2; ++m) zc[m] = v[E - 1 - m]; // (what the timing-only build keeps)
    if (!GPT_TRAFFIC_ONLY) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < E; ++m) lds[phys(t + T * m)] = v[m].x;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < E / 2; ++m) {
        const int j = t + T * m;
        zc[m].x = lds[phys(j == 0 ? 0 : W - j)];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < E; ++m) lds[phys(t + T * m)] = v[m].y;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < E / 2; ++m) {
        const int j = t + T * m;
        zc[m].y = lds[phys(j == 0 ? 0 : W - j)];
    }
    }
    if (!active) return;
    if (edge && a.cov && t == 0) {
        // stage the two edge columns of cov contiguously for the column pass (its packed column 0 would
        // otherwise gather them, 64 lines per load instruction, while everybody else streams)
        const double* cv = a.cov + b * a.cov_bs;
        const int wf = W / 2 + 1;
        double* e = edge + b * 4 * (int64_t)H;
        e[ra] = cv[(int64_t)ra * wf];
        e[rb] = cv[(int64_t)rb * wf];
        e[H + ra] = cv[(int64_t)ra * wf + W / 2];
        e[H + rb] = cv[(int64_t)rb * wf + W / 2];
    }
    cplx* s0 = ws + ((int64_t)b * H + ra) * wstride;
    cplx* s1 = ws + ((int64_t)b * H + rb) * wstride;
    cplx wr = make_double2(1.0, 0.0);
    if (SUB2) wr = a.ph.tw[rr]; // uniform: w_H^r
#pragma unroll
    for (int m = 0; m < E / 2; ++m) {
        const int j = t + T * m;
        const cplx z = v[m];
        cplx za, zb;
        if (j == 0) {
            const cplx zn = v[E / 2]; // Z[W/2] lives in the same thread (t == 0, m == E/2)
            za = make_double2(z.x, zn.x);
            zb = make_double2(z.y, zn.y);
        } else {
            za = make_double2(0.5 * (z.x + zc[m].x), 0.5 * (z.y - zc[m].y));
            zb = make_double2(0.5 * (z.y + zc[m].y), -0.5 * (z.x - zc[m].x));
        }
        const unsigned jo = spec_col(a, j);
        if (SUB2) {
            rows_st<NTS>(&s0[jo], cadd(za, zb));
            rows_st<NTS>(&s1[jo], cmul(csub(za, zb), wr));
        } else {
            rows_st<NTS>(&s0[jo], za);
            rows_st<NTS>(&s1[jo], zb);
        }
    }
}

// pass C --------------------------------------------------------------------------------------------
// out = alpha * value (+ loc when add_loc)
// SUB2: the mirror image of row_fwd_kernel<W, true>: S[r] = v0 + conj(w_H^r) v1, S[r + H/2] = v0 - conj(w_H^r) v1
// with v0, v1 the workspace rows r and H/2 + r.
template <int W, bool SUB2 = false>
__global__ void __launch_bounds__(Geo<W>::ROW_THREADS) row_inv_kernel(LatArgs a, double alpha, int add_loc,
                                                                      const cplx* __restrict__ ws, int wstride,
                                                                      FinArgs fin, const double* __restrict__ edge) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = Geo<W>::T, LINES = Geo<W>::ROW_LINES;
    const int line = threadIdx.x / T, t = threadIdx.x - line * T;
    double* lds = reinterpret_cast<double*>(smem) + (size_t)line * W;
    const int64_t b = blockIdx.y;
    const int H = a.H;
    const int r = blockIdx.x * LINES + line;
    const bool active = r < H / 2;
#if GPT_ROWS_SAME
    const int rr = 0;
#else
    const int rr = active ? r : 0;
#endif
    const int ra = SUB2 ? rr : 2 * rr, rb = SUB2 ? rr + H / 2 : 2 * rr + 1;
    const cplx* s0 = ws + ((int64_t)b * H + ra) * wstride;
    const cplx* s1 = ws + ((int64_t)b * H + rb) * wstride;
    cplx wr = make_double2(1.0, 0.0);
    if (SUB2) wr = cconj(a.ph.tw[rr]);
    cplx wpre[tw_npre<W>()];
    tw_load<W>(wpre, a.pw.tw, t);
    cplx v[E];
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int j = t + T * m;
        const int q = W - j; // 1 .. W/2 when m >= E/2
        const int col = m < E / 2 ? j : (q == W / 2 ? 0 : q);
        const unsigned co = spec_col(a, col);
        cplx A = s0[co], B = s1[co];
        if (SUB2) {
            const cplx u = cmul(B, wr);
            B = csub(A, u);
            A = cadd(A, u);
        }
        if (m < E / 2) {
            if (j == 0) v[m] = make_double2(A.x, B.x); // DC terms are real
            else v[m] = make_double2(A.x - B.y, A.y + B.x);
        } else {
            if (q == W / 2) v[m] = make_double2(A.y, B.y); // Nyquist terms ride in the imaginary slot of column 0
            else v[m] = make_double2(A.x + B.y, -A.y + B.x); // conj(A) + i conj(B)
        }
        if (Geo<W>::ROW_THREADS >= 1024) // (128 registers: keep the merge from sinking into the transform, see rowreal_inv_kernel)
            asm volatile("" : "+v"(v[m].x), "+v"(v[m].y));
    }
    fft_regs<W, true>(v, t, lds, wpre);
    if (fin.partials && blockIdx.x == 0) {
        // the column pass left per-workgroup partials: this workgroup adds them up (saves a launch)
        __syncthreads();
        finalize_block(fin, b, reinterpret_cast<double*>(smem));
    }
    if (!active) return;
    if (edge && a.out_cov && t == 0) {
        // the column pass left d/dcov of the two edge columns contiguous: put rows 2rr, 2rr+1 in place
        const double* e = edge + b * 4 * (int64_t)H + 2 * (int64_t)H;
        const int wf = W / 2 + 1;
        double* oc = a.out_cov + b * (int64_t)H * wf;
        oc[(int64_t)ra * wf] = e[ra];
        oc[(int64_t)rb * wf] = e[rb];
        oc[(int64_t)ra * wf + W / 2] = e[H + ra];
        oc[(int64_t)rb * wf + W / 2] = e[H + rb];
    }
    double* o0 = a.out + b * (int64_t)H * W + (int64_t)ra * W;
    double* o1 = a.out + b * (int64_t)H * W + (int64_t)rb * W;
    if (add_loc && a.loc) {
        const double* l0 = a.loc + b * a.loc_bs + (int64_t)ra * W;
        const double* l1 = a.loc + b * a.loc_bs + (int64_t)rb * W;
#pragma unroll
        for (int m = 0; m < E; ++m) {
            o0[t + T * m] = fma(alpha, v[m].x, l0[t + T * m]);
            o1[t + T * m] = fma(alpha, v[m].y, l1[t + T * m]);
        }
    } else {
#pragma unroll
        for (int m = 0; m < E; ++m) {
            __builtin_nontemporal_store(alpha * v[m].x, o0 + t + T * m);
            __builtin_nontemporal_store(alpha * v[m].y, o1 + t + T * m);
        }
    }
}

// Long rows ------------------------------------------------------------------------------------------
// Rows of W = 2 N points whose two-for-one complex line (W points) no workgroup can hold (W = 32768: BASELINE config 5,
// the slab passes): ONE real row is one complex line of N = W / 2 points, z[n] = x[2n] + i x[2n+1] -- a contiguous
// 16-byte load per lane -- and with Z = FFT_N(z), E[k] = (Z[k] + conj Z[N-k]) / 2, O[k] = (Z[k] - conj Z[N-k]) / (2i):
//     X[k] = E[k] + w_W^k O[k],  k = 0 .. N      (X[0] = Re Z[0] + Im Z[0], X[N] = Re Z[0] - Im Z[0]: both real)
// which goes to the packed spectrum row as it is (column 0 = X[0] + i X[N]). One workgroup per row, ONE sweep over
// y / loc and the spectrum; round 2 took three (radix-8 step into scratch lines, eight sub-line transforms, split).
// a.H = rows of this call, a.W = W, a.pw.tw = w_W^m, twN = w_N^m.
// Geometry: lines of 16384 points run 512 threads with two register tiles each (256 registers: the 1024-thread,
// one-tile form has 128 and spilled 74 of them in the inverse kernel); shorter lines one tile per thread.
template <int N> struct GeoR {
    static constexpr int T = N / E;
    static constexpr int V = N >= 16384 ? 2 : 1;
    static constexpr int TP = T / V;                  // threads per line
    static constexpr int THREADS = TP < 256 ? 256 : TP;
    static constexpr int LINES = THREADS / TP;
};
template <int N>
__global__ void __launch_bounds__(GeoR<N>::THREADS) rowreal_fwd_kernel(LatArgs a, cplx* __restrict__ spec, int ld,
                                                                       const cplx* __restrict__ twN) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = GeoR<N>::T, V = GeoR<N>::V, TP = GeoR<N>::TP, LINES = GeoR<N>::LINES, W = 2 * N;
    constexpr int NPRE = tw_npre<N>();
    const int line = threadIdx.x / TP, tp = threadIdx.x - line * TP;
    double* lds = reinterpret_cast<double*>(smem) + (size_t)line * N;
    const int r = blockIdx.x * LINES + line;
    const bool active = r < a.H;
    int rr = active ? r : 0;
    if (LINES == 1) rr = __builtin_amdgcn_readfirstlane(rr); // uniform row base: scalar base + 32-bit lane offsets
    const double* x = a.x + (int64_t)rr * W;
    int tl[V];
    cplx wl[V][NPRE > 0 ? NPRE : 1];
    cplx v[V][E];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        tl[k] = tp + k * TP;
        tw_load<N>(wl[k], twN, tl[k]);
#pragma unroll
        for (int m = 0; m < E; ++m) { // read once: non-temporal
            const gpt_rd2v u = __builtin_nontemporal_load(reinterpret_cast<const gpt_rd2v*>(x + 2u * (unsigned)(tl[k] + T * m)));
            v[k][m] = make_double2(u.x, u.y);
        }
    }
    if (a.loc) {
        const double* l = a.loc + (int64_t)rr * W;
#pragma unroll
        for (int k = 0; k < V; ++k)
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const cplx u = *reinterpret_cast<const cplx*>(l + 2u * (unsigned)(tl[k] + T * m));
                v[k][m].x -= u.x;
                v[k][m].y -= u.y;
            }
    }
    fft_regs<N, false, V, 0>(v, tl, lds, wl);
    // A thread finishes the frequencies j = tl + T m, m < E/2, AND their mirrors N - j: both need exactly Z[j] (own
    // register) and Z[N - j] (through LDS, re then im), X[N - j] = conj(E[j] - w^j O[j]).
    cplx zc[V][E / 2];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k)
#pragma unroll
        for (int m = 0; m < E; ++m) lds[phys(tl[k] + T * m)] = v[k][m].x;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k)
#pragma unroll
        for (int m = 0; m < E / 2; ++m) zc[k][m].x = lds[phys((N - (tl[k] + T * m)) & (N - 1))];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k)
#pragma unroll
        for (int m = 0; m < E; ++m) lds[phys(tl[k] + T * m)] = v[k][m].y;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k)
#pragma unroll
        for (int m = 0; m < E / 2; ++m) zc[k][m].y = lds[phys((N - (tl[k] + T * m)) & (N - 1))];
    if (!active) return;
    cplx* s = spec + (int64_t)rr * ld;
#pragma unroll
    for (int k = 0; k < V; ++k)
#pragma unroll
        for (int m = 0; m < E / 2; ++m) {
            const int j = tl[k] + T * m;
            const cplx z = v[k][m];
            if (j == 0) {
                s[spec_col(a, 0)] = make_double2(z.x + z.y, z.x - z.y); // DC + i Nyquist
                s[spec_col(a, N / 2)] = cconj(v[0][E / 2]);             // X[N/2] = conj Z[N/2] (position 0 holds it)
            } else {
                const cplx e = make_double2(0.5 * (z.x + zc[k][m].x), 0.5 * (z.y - zc[k][m].y));
                const cplx wo = cmul(a.pw.tw[(unsigned)j], make_double2(0.5 * (z.y + zc[k][m].y), -0.5 * (z.x - zc[k][m].x)));
                s[spec_col(a, j)] = cadd(e, wo);
                s[spec_col(a, N - j)] = cconj(csub(e, wo));
            }
        }
}

// The way back: Z[k] = E[k] + i O[k] with E[k] = (X[k] + conj X[N-k]) / 2, O[k] = conj(w_W^k) (X[k] - conj X[N-k]) / 2,
// unnormalised inverse transform of N points = N z; out = 2 alpha * (that), so that alpha keeps the meaning it has in
// row_inv_kernel (a factor on the unnormalised inverse transform of W points).
template <int N>
__global__ void __launch_bounds__(GeoR<N>::THREADS) rowreal_inv_kernel(LatArgs a, double alpha,
                                                                       const cplx* __restrict__ spec, int ld,
                                                                       const cplx* __restrict__ twN) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = GeoR<N>::T, V = GeoR<N>::V, TP = GeoR<N>::TP, LINES = GeoR<N>::LINES, W = 2 * N;
    constexpr int NPRE = tw_npre<N>();
    const int line = threadIdx.x / TP, tp = threadIdx.x - line * TP;
    double* lds = reinterpret_cast<double*>(smem) + (size_t)line * N;
    const int r = blockIdx.x * LINES + line;
    const bool active = r < a.H;
    int rr = active ? r : 0;
    if (LINES == 1) rr = __builtin_amdgcn_readfirstlane(rr); // uniform row base: scalar base + 32-bit lane offsets
    const cplx* s = spec + (int64_t)rr * ld;
    int tl[V];
    cplx wl[V][NPRE > 0 ? NPRE : 1];
    cplx v[V][E];
    constexpr int CH = 4; // entries whose A, B and twiddle loads are in flight together
#pragma unroll
    for (int k = 0; k < V; ++k) {
        tl[k] = tp + k * TP;
        tw_load<N>(wl[k], twN, tl[k]);
#pragma unroll
        for (int m0 = 0; m0 < E; m0 += CH) {
            cplx A[CH], B[CH], w[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int j = tl[k] + T * (m0 + i);
                A[i] = s[spec_col(a, j)];
                B[i] = s[spec_col(a, (N - j) & (N - 1))]; // (the mirrored half of the same row: served by the caches)
                w[i] = a.pw.tw[(unsigned)j];
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int j = tl[k] + T * (m0 + i);
                if (j == 0) {
                    v[k][m0 + i] = make_double2(0.5 * (A[i].x + A[i].y), 0.5 * (A[i].x - A[i].y));
                } else {
                    const cplx e = make_double2(0.5 * (A[i].x + B[i].x), 0.5 * (A[i].y - B[i].y));
                    const cplx d = make_double2(0.5 * (A[i].x - B[i].x), 0.5 * (A[i].y + B[i].y));
                    const cplx o = cmulc(d, w[i]);
                    v[k][m0 + i] = make_double2(e.x - o.y, e.y + o.x);
                }
                // pin the value here: otherwise the arithmetic sinks down to the transform and the raw A, B and w of
                // all 32 entries (12 registers each instead of 4) stay alive and spill
                asm volatile("" : "+v"(v[k][m0 + i].x), "+v"(v[k][m0 + i].y));
            }
            SCHED_FENCE();
        }
    }
    fft_regs<N, true, V, 0>(v, tl, lds, wl);
    if (!active) return;
    double* out = a.out + (int64_t)rr * W;
    const double al = 2.0 * alpha;
#pragma unroll
    for (int k = 0; k < V; ++k)
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const gpt_rd2v u = {al * v[k][m].x, al * v[k][m].y};
            __builtin_nontemporal_store(u, reinterpret_cast<gpt_rd2v*>(out + 2u * (unsigned)(tl[k] + T * m)));
        }
}

// pass C of gp_inv_rfft2, an observation model on the field, and pass A of the transform's reverse mode in one
// kernel (gpt_gp_inv_rfft2_obs_lpdf_vjp): the workspace rows of this line come in as the spectrum of f and go out,
// in place, as the spectrum of d lp / d f. f and d/df exist only in LDS unless the caller asks for copies.
// `lds` (W doubles per line) is the transforms' exchange space and, in between, the thread-private staging of the
// rows' points for a rolled loop over them.
template <int W, bool SUB2 = false>
__global__ void __launch_bounds__(Geo<W>::ROW_THREADS) row_obs_kernel(LatArgs a, double alpha, cplx* ws, int wstride,
                                                                      FinArgs fin, double* __restrict__ edge,
                                                                      ObsRowArgs o) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = Geo<W>::T, LINES = Geo<W>::ROW_LINES;
    const int line = threadIdx.x / T, t = threadIdx.x - line * T;
    double* lds = reinterpret_cast<double*>(smem) + (size_t)line * W;
    const int64_t b = blockIdx.y;
    const int H = a.H;
    const int r = blockIdx.x * LINES + line;
    const bool active = r < H / 2;
    const int rr = active ? r : 0;
    const int ra = SUB2 ? rr : 2 * rr, rb = SUB2 ? rr + H / 2 : 2 * rr + 1;
    cplx* s0 = ws + ((int64_t)b * H + ra) * wstride;
    cplx* s1 = ws + ((int64_t)b * H + rb) * wstride;
    cplx wr = make_double2(1.0, 0.0);
    if (SUB2) wr = a.ph.tw[rr];
    cplx wpre[tw_npre<W>()];
    tw_load<W>(wpre, a.pw.tw, t);
    cplx v[E];
    { // as row_inv_kernel
        const cplx wc = cconj(wr);
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int j = t + T * m;
            const int q = W - j;
            const int col = m < E / 2 ? j : (q == W / 2 ? 0 : q);
            cplx A = s0[col], B = s1[col];
            if (SUB2) {
                const cplx u = cmul(B, wc);
                B = csub(A, u);
                A = cadd(A, u);
            }
            if (m < E / 2) {
                if (j == 0) v[m] = make_double2(A.x, B.x);
                else v[m] = make_double2(A.x - B.y, A.y + B.x);
            } else {
                if (q == W / 2) v[m] = make_double2(A.y, B.y);
                else v[m] = make_double2(A.x + B.y, -A.y + B.x);
            }
        }
    }
    fft_regs<W, true>(v, t, lds, wpre);
    __syncthreads();
    if (fin.partials && blockIdx.x == 0) {
        finalize_block(fin, b, reinterpret_cast<double*>(smem));
        __syncthreads();
    }
    // Half a row at a time: f = alpha * value + loc and the data go to the thread's own LDS positions, obs_points
    // (a real call: one copy of the densities' code) turns them into d lp / d f, which comes back into the registers.
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    {
        const double th = o.theta ? o.theta[b] : 0.0;
        double k[3];
        obs_consts(o.family, th, k);
        const double* ob = o.obs + b * o.obs_bs;
        constexpr int EH = E / 2;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int half = ph >> 1, m0 = (ph & 1) * EH;
            const int row = half ? rb : ra;
            const double* l = a.loc ? a.loc + b * a.loc_bs + (int64_t)row * W : nullptr;
            // the data of these points (NaN = nothing observed there) and loc: EH loads each in flight together
            double yv[EH], lv[EH];
            const bool in_rows = active && row < o.oh;
#pragma unroll
            for (int m = 0; m < EH; ++m) {
                const int j = t + T * (m0 + m);
                yv[m] = (in_rows && j < o.ow) ? ob[(int64_t)row * o.ow + j] : __builtin_nan("");
                lv[m] = l ? l[j] : 0.0;
            }
#pragma unroll
            for (int m = 0; m < EH; ++m) {
                lds[t + T * m] = fma(alpha, half ? v[m0 + m].y : v[m0 + m].x, lv[m]);
                lds[W / 2 + t + T * m] = yv[m];
            }
            double* fo = (o.f_out && active) ? o.f_out + (b * H + row) * (int64_t)W + T * m0 : nullptr;
            double* go = (o.fbar_out && active) ? o.fbar_out + (b * H + row) * (int64_t)W + T * m0 : nullptr;
            const ObsSums ps = obs_points(lds + t, lds + W / 2 + t, EH, T, o.family, th, k[0], k[1], k[2], fo ? fo + t : nullptr,
                                          go ? go + t : nullptr);
            s[0] += ps.lp;
            s[1] += ps.dth;
            s[2] += ps.df;
            s[3] += ps.n;
#pragma unroll
            for (int m = 0; m < EH; ++m) {
                if (half) v[m0 + m].y = lds[t + T * m];
                else v[m0 + m].x = lds[t + T * m];
            }
        }
    }
    block_reduce_store<4>(s, INFINITY, o.partials, b); // (synchronises: the staging is free again)
    // as row_fwd_kernel
    fft_regs<W, false>(v, t, lds, wpre);
    cplx zc[E / 2];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < E; ++m) lds[phys(t + T * m)] = v[m].x;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < E / 2; ++m) {
        const int j = t + T * m;
        zc[m].x = lds[phys(j == 0 ? 0 : W - j)];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < E; ++m) lds[phys(t + T * m)] = v[m].y;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < E / 2; ++m) {
        const int j = t + T * m;
        zc[m].y = lds[phys(j == 0 ? 0 : W - j)];
    }
    if (!active) return;
    if (edge && a.cov && t == 0) {
        const double* cv = a.cov + b * a.cov_bs;
        const int wf = W / 2 + 1;
        double* e = edge + b * 4 * (int64_t)H;
        e[ra] = cv[(int64_t)ra * wf];
        e[rb] = cv[(int64_t)rb * wf];
        e[H + ra] = cv[(int64_t)ra * wf + W / 2];
        e[H + rb] = cv[(int64_t)rb * wf + W / 2];
    }
#pragma unroll
    for (int m = 0; m < E / 2; ++m) {
        const int j = t + T * m;
        const cplx z = v[m];
        cplx za, zb;
        if (j == 0) {
            const cplx zn = v[E / 2];
            za = make_double2(z.x, zn.x);
            zb = make_double2(z.y, zn.y);
        } else {
            za = make_double2(0.5 * (z.x + zc[m].x), 0.5 * (z.y - zc[m].y));
            zb = make_double2(0.5 * (z.y + zc[m].y), -0.5 * (z.x - zc[m].x));
        }
        if (SUB2) {
            s0[j] = cadd(za, zb);
            s1[j] = cmul(csub(za, zb), wr);
        } else {
            s0[j] = za;
            s1[j] = zb;
        }
    }
}

// pass B --------------------------------------------------------------------------------------------
// Persistent: a workgroup walks over several column tiles (TC adjacent packed columns each).
// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (blockIdx.x % 8 labels the XCD),
// and 8 / TC neighbouring tiles write pieces of the SAME 128-byte lines of the spectrum and of d/dcov.
// Each XCD gets a contiguous range of tiles, walked in step by its workgroups, so the pieces meet in one
// L2 (speed only, never correctness).
// Thread layout: c = threadIdx % TC (column inside the tile: adjacent lanes -> adjacent columns, so one
// load instruction reads TC*16 contiguous bytes per row), tp = threadIdx / TC (position in the column);
// register tile k of the thread sits at line position t[k] = tp + k*TP, i.e. rows t[k] + T*m.
template <int H, int OP>
__global__ void __launch_bounds__(Geo<H>::COL_THREADS, 2) col_kernel(LatArgs a, cplx* __restrict__ ws, int wstride,
                                                                     double* __restrict__ partials, int ntiles,
                                                                     double* __restrict__ edge, int edge_flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = Geo<H>::T, TC = Geo<H>::TC, V = Geo<H>::V, TP = Geo<H>::TP;
    constexpr int NPRE = tw_npre<H>();
    constexpr bool NEEDS_COV = (OP != OP_RFFT2 && OP != OP_INV_RFFT2);
    const int c = threadIdx.x % TC, tp = threadIdx.x / TC;
    // adjacent columns' regions are skewed by 8 elements (64 B): the lanes of one LDS lane group alternate
    // between columns, and an unskewed region start would put them on the same banks
    double* lds = reinterpret_cast<double*>(smem) + (size_t)c * (H + Geo<H>::SKEW);
    cplx* lds0 = reinterpret_cast<cplx*>(smem);    // whole buffer as one complex column (packed column 0)
    double* red = reinterpret_cast<double*>(smem); // reused after the FFTs are done with LDS
    const int64_t b = blockIdx.y;
    const ZLayout L = make_layout(a);
    EwPtrs P = make_ptrs(a, b);
    if (edge) { // edge_flags: 1 = cov staged by pass A, 2 = d/dcov to be scattered by pass C
        if (edge_flags & 1) P.edge_cov = edge + b * 4 * (int64_t)H;
        if (edge_flags & 2) P.edge_dcov = edge + b * 4 * (int64_t)H + 2 * (int64_t)H;
    }
    cplx* wsb = ws + (int64_t)b * H * wstride; // uniform base; 32-bit element offsets below
    cplx wpre[V][NPRE];
#pragma unroll
    for (int k = 0; k < V; ++k) tw_load<H>(wpre[k], a.ph.tw, tp + k * TP);

    int first, step, count;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        if ((nwg & 7) == 0 && (ntiles & 7) == 0) {
            const int per = ntiles >> 3, i = bid >> 3;
            step = nwg >> 3;
            first = (bid & 7) * per + i;
            count = i < per ? (per - i + step - 1) / step : 0;
        } else {
            step = nwg;
            first = bid;
            count = bid < ntiles ? (ntiles - bid + step - 1) / step : 0;
        }
    }
    EwAcc acc;
    acc_init(acc);
#pragma unroll 1
    for (int r = 0; r < count; ++r) {
        const int tile = first + r * step;
        const int j0 = tile * TC;
        const int jl = j0 + c;      // packed column inside the spectrum buffer
        const int jp = a.col0 + jl; // ... and inside the lattice (cov, d/dcov, z layouts)
        // The thread's line positions, laundered through an empty asm: every address below then depends on a
        // per-iteration opaque value, which keeps loop-invariant code motion from hoisting ~100 address
        // registers out of the tile loop (they are cheap to recompute and expensive to keep) ...
        int tl[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            tl[k] = tp + k * TP;
            asm volatile("" : "+v"(tl[k]));
        }
        // ... and the twiddle bases: their power trees (60 complex values per tile) would otherwise be
        // hoisted out of the loop and parked in scratch memory.
        cplx wl[V][NPRE];
#pragma unroll
        for (int k = 0; k < V; ++k)
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                wl[k][i] = wpre[k][i];
                asm volatile("" : "+v"(wl[k][i].x), "+v"(wl[k][i].y));
            }
        cplx v[V][E];
        TRACE_MARK(r, 0);
#pragma unroll
        for (int k = 0; k < V; ++k)
#pragma unroll
            for (int m = 0; m < E; ++m)
                v[k][m] = op_has_fwd(OP) ? wsb[(unsigned)(tl[k] + T * m) * (unsigned)wstride + (unsigned)jl]
                                         : make_double2(0.0, 0.0);
#if GPT_TRACE
        TRACE_MARK(r, 1); // loads issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        TRACE_MARK(r, 2); // ... and arrived
        const bool own0 = (a.col0 + j0 == 0); // workgroup-uniform: this tile holds the packed column 0
        if (op_has_fwd(OP)) fft_regs<H, false, V, Geo<H>::SWZ>(v, tl, lds, wl);
        TRACE_MARK(r, 3);
        if (own0) {
            // Packed column 0 (DC + i*Nyquist): rows k and H-k must meet, so the column goes through LDS
            // and is processed by a ROLLED loop (few registers, little code; one column out of W/2) that
            // requests the next row's cov values before it works on the current row.
            TRACE_MARK(7, 0);
            __syncthreads();
            if (c == 0) {
#pragma unroll
                for (int k = 0; k < V; ++k)
#pragma unroll
                    for (int m = 0; m < E; ++m) lds0[phys(tl[k] + T * m)] = v[k][m];
            }
            __syncthreads();
            TRACE_MARK(7, 1);
            EdgeCov ec = load_edge_cov<OP>(a, L, P, threadIdx.x <= H / 2 ? threadIdx.x : 0);
#pragma unroll 1
            for (int kk = threadIdx.x; kk <= H / 2; kk += Geo<H>::COL_THREADS) {
                const int kn = kk + Geo<H>::COL_THREADS;
                const EdgeCov en = load_edge_cov<OP>(a, L, P, kn <= H / 2 ? kn : 0);
                const bool self = (kk == 0) || (kk == H / 2);
                cplx Zk = lds0[phys(kk)];
                cplx Zm = self ? Zk : lds0[phys(H - kk)];
                ew_packed0_c<OP>(a, L, P, kk, Zk, Zm, ec, acc);
                if (op_has_inv(OP)) {
                    lds0[phys(kk)] = Zk;
                    if (!self) lds0[phys(H - kk)] = Zm;
                }
                ec = en;
                TRACE_MARK(7, 2 + kk / Geo<H>::COL_THREADS);
            }
            __syncthreads();
            TRACE_MARK(7, 7);
            if (c == 0 && op_has_inv(OP)) {
#pragma unroll
                for (int k = 0; k < V; ++k)
#pragma unroll
                    for (int m = 0; m < E; ++m) v[k][m] = lds0[phys(tl[k] + T * m)];
            }
            __syncthreads();
        }
        if (!(own0 && c == 0)) {
            // one register tile at a time: its 16 cov values are requested together, then consumed
#pragma unroll
            for (int k = 0; k < V; ++k) {
                double cv[E];
#pragma unroll
                for (int m = 0; m < E; ++m)
                    cv[m] = NEEDS_COV ? P.cov[(unsigned)(tl[k] + T * m) * (unsigned)L.Wf + (unsigned)jp] : 1.0;
#if GPT_TRACE
                if (k == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    TRACE_MARK(r, 4);
                }
#endif
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    v[k][m] = ew_interior_c<OP>(a, L, P, tl[k] + T * m, jp, v[k][m], cv[m], acc);
                    if ((m & 3) == 3) SCHED_FENCE();
                }
            }
        }
        acc_fold(acc);
        __builtin_amdgcn_sched_barrier(0);
        TRACE_MARK(r, 5);
        if (op_has_inv(OP)) {
            // Launder the line positions all LDS addresses derive from, tied to a value the elementwise stage
            // produced: otherwise the compiler keeps the forward transform's addresses alive to reuse them
            // here, or schedules parts of the inverse transform above the elementwise stage (spills).
            int t_inv[V];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                t_inv[k] = tl[k];
                asm volatile("" : "+v"(t_inv[k]), "+v"(v[0][0].x));
#pragma unroll
                for (int i = 0; i < NPRE; ++i) asm volatile("" : "+v"(wl[k][i].x), "+v"(wl[k][i].y), "+v"(v[0][0].x));
            }
            fft_regs<H, true, V, Geo<H>::SWZ>(v, t_inv, lds, wl);
            TRACE_MARK(r, 6);
#pragma unroll
            for (int k = 0; k < V; ++k)
#pragma unroll
                for (int m = 0; m < E; ++m)
                    wsb[(unsigned)(t_inv[k] + T * m) * (unsigned)wstride + (unsigned)jl] = v[k][m];
            TRACE_MARK(r, 7);
        }
    }
    if (OP != OP_RFFT2 && OP != OP_INV_RFFT2) {
        double s = acc.sum, mn = acc.mn;
        block_sum_min(s, mn, red);
        if (threadIdx.x == 0) {
            double* p = partials + ((int64_t)b * gridDim.x + blockIdx.x) * GPT_PART;
            p[0] = s;
            p[1] = p[2] = p[3] = p[4] = 0.0;
            p[5] = mn;
        }
    }
}

template <int N> struct Launch {
    // sub2: the R = 2 sub-line variant of the row kernels (rows r and r + H/2 per workgroup)
    static hipError_t row_fwd(const LatArgs& a, int in_mode, cplx* ws, int wstride, double* edge, bool sub2,
                              hipStream_t st);
    static hipError_t row_inv(const LatArgs& a, double alpha, int add_loc, const cplx* ws, int wstride,
                              const FinArgs& fin, const double* edge, bool sub2, hipStream_t st);
    static hipError_t col(int op, const LatArgs& a, cplx* ws, int wstride, double* partials, int nwg, double* edge,
                          int edge_flags, hipStream_t st);
    // inverse rows -> observation model -> forward rows, in place (rows of up to 8192 points); its grid is
    // row_obs_blocks(H) workgroups per lattice = partials per lattice in ObsRowArgs
    static hipError_t row_obs(const LatArgs& a, double alpha, cplx* ws, int wstride, const FinArgs& fin, double* edge,
                              const ObsRowArgs& o, bool sub2, hipStream_t st);
    static int row_obs_blocks(int H) { return (H / 2 + Geo<N>::ROW_LINES - 1) / Geo<N>::ROW_LINES; }
    // rows of 2 N points, one real row per complex line of N points (a.H rows of a.W = 2 N points; twN = w_N^m)
    static hipError_t rowreal_fwd(const LatArgs& a, cplx* spec, int ld, const cplx* twN, hipStream_t st);
    static hipError_t rowreal_inv(const LatArgs& a, double alpha, const cplx* spec, int ld, const cplx* twN,
                                  hipStream_t st);
    // grid of the persistent column kernel = number of per-lattice partials, one workgroup per CU it may use
    static int col_tile() { return Geo<N>::TC; }
    static int col_blocks(int W, int cus) {
        const int ntiles = (W / 2) / Geo<N>::TC;
        int cap = cus * GPT_COL_WGS_PER_CU;
        if (cap >= 8) cap &= ~7; // multiples of 8 keep the XCD-aware tile order
        if (cap < 1) cap = 1;
        return ntiles < cap ? ntiles : cap;
    }
};

} // namespace pow2
