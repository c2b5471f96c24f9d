// Sub-line variant of the power-of-two three-pass path: the column transform of length H = R * HS is split
// into one radix-R step ACROSS rows, done where the rows already sit in registers (epilogue of pass A,
// prologue of pass C), and R independent sub-line transforms of length HS in pass B.
//
//   forward (decimation in frequency), column x[i], i = r + HS p:
//       u_q[r]    = w_H^{r q} * sum_p x[r + HS p] w_R^{p q}            (pass A', stored at workspace row q HS + r)
//       X[R k + q] = FFT_HS(u_q)[k]                                    (pass B', sub-line q; logical row R k + q)
//   inverse (decimation in time), unnormalised:
//       v_q       = IFFT_HS(Y[R . + q])                                (pass B', in place)
//       y[r + HS p] = sum_q w_R^{-p q} w_H^{-r q} v_q[r]               (pass C')
//
// Why: a pass-B tile of TC columns x HS rows has to fit the register file, so a 4x shorter line makes the tile
// 2x wider (8 columns = one full 128-byte line of the spectrum per row instead of 64-byte pieces; the pass is
// bound by line requests, gpt_pow2_impl.h) AND 2x smaller (128 KB), so that two workgroups share a CU and one
// computes while the other waits for memory. The row kernels pay for it with one LDS hop of half their data.
//
// Hermitian pairing on the packed column 0 (rows i and H - i meet in the elementwise stage): i = R k + q pairs
// with R (HS - k) for q = 0 and with R (HS - 1 - k) + (R - q) otherwise, i.e. sub-line q pairs with sub-line
// R - q. The first R/2 + 1 workgroups of pass B' take that one column (col0_role), all tiles skip it.
#pragma once
#include "gpt_pow2_impl.h"

namespace pow2 {

#ifndef GPT_SUB_WPS2
#define GPT_SUB_WPS2 2
#endif
#ifndef GPT_SUB_WPS1
#define GPT_SUB_WPS1 4
#endif
template <int HS, int V_, int THREADS_> struct GeoS {
    static constexpr int T = HS / E;
    static constexpr int V = V_;
    static constexpr int TP = T / V;
    static constexpr int THREADS = THREADS_;
    static constexpr int TC = THREADS / TP; // adjacent packed columns per tile
    static constexpr int SKEW = 4;
    static constexpr int LINES = THREADS / T; // single-tile lines the workgroup can hold (col0_role)
    static_assert(THREADS % TP == 0 && TC >= 1, "tile geometry");
    static_assert(LINES >= 2, "col0_role needs two lines");
    static constexpr size_t lds_bytes() {
        const size_t a = (size_t)TC * (HS + SKEW) * sizeof(double), b = 2 * (size_t)HS * sizeof(cplx),
                     c = (size_t)LINES * HS * sizeof(double);
        return a > b ? (a > c ? a : c) : (b > c ? b : c);
    }
};

// ---------------------------------------------------------------------------------------------------
// Packed column 0 of sub-lines q and R - q (q = blockIdx-supplied, 0 <= q <= R/2). Every thread of the
// workgroup walks through the same barriers; threads beyond the two live lines transform zeros in LDS
// regions of their own.
template <int HS, int THREADS, int OP>
__device__ __forceinline__ void col0_role(const LatArgs& a, const ZLayout& L, const EwPtrs& P, int R, int q,
                                          cplx* __restrict__ wsb, int wstride, const cplx* __restrict__ tws,
                                          unsigned char* smem, EwAcc& acc) {
    constexpr int T = HS / E;
    const int H = a.H;
    const int line = threadIdx.x / T, t = threadIdx.x - line * T;
    const bool paired = (q != 0) && (2 * q != R);
    const bool live = line == 0 || (line == 1 && paired);
    const int qrow = line == 0 ? q : R - q;
    double* lds = reinterpret_cast<double*>(smem) + (size_t)line * HS;
    cplx* lds0 = reinterpret_cast<cplx*>(smem); // [2][HS] complex, once the transforms are done with LDS
    cplx wpre[tw_npre<HS>() > 0 ? tw_npre<HS>() : 1];
    tw_load<HS>(wpre, tws, t);
    cplx v[E];
#pragma unroll
    for (int m = 0; m < E; ++m)
        v[m] = (live && op_has_fwd(OP)) ? wsb[(unsigned)(qrow * HS + t + T * m) * (unsigned)wstride]
                                        : make_double2(0.0, 0.0);
    if (op_has_fwd(OP)) fft_regs<HS, false>(v, t, lds, wpre);
    __syncthreads();
    if (live) {
#pragma unroll
        for (int m = 0; m < E; ++m) lds0[line * HS + t + T * m] = v[m];
    }
    __syncthreads();
    {
        // pairs: q == 0: (k, HS - k), k = 0..HS/2 (two self pairs); q == R/2: (k, HS - 1 - k), k < HS/2;
        // otherwise (line 0 [k], line 1 [HS - 1 - k]) for every k
        const int npairs = q == 0 ? HS / 2 + 1 : (paired ? HS : HS / 2);
#pragma unroll 1
        for (int kk = threadIdx.x; kk < npairs; kk += THREADS) {
            int ia, ib; // LDS slots of the pair; row = logical row (<= H/2) of slot ia... or of slot ib
            bool self = false, swap = false;
            int row;
            if (q == 0) {
                ia = kk;
                ib = (HS - kk) & (HS - 1);
                self = (ia == ib);
                row = R * kk;
            } else if (!paired) {
                ia = kk;
                ib = HS - 1 - kk;
                row = R * kk + q;
            } else {
                ia = kk;
                ib = HS + (HS - 1 - kk);
                row = R * kk + q;
                if (2 * row > H) { // the mirror row is the one below H/2
                    row = H - row;
                    swap = true;
                }
            }
            cplx Zk = lds0[swap ? ib : ia];
            cplx Zm = self ? Zk : lds0[swap ? ia : ib];
            ew_packed0_c<OP>(a, L, P, row, Zk, Zm, load_edge_cov<OP>(a, L, P, row), acc);
            if (op_has_inv(OP)) {
                lds0[swap ? ib : ia] = Zk;
                if (!self) lds0[swap ? ia : ib] = Zm;
            }
        }
    }
    __syncthreads();
    if (op_has_inv(OP)) {
        if (live) {
#pragma unroll
            for (int m = 0; m < E; ++m) v[m] = lds0[line * HS + t + T * m];
        }
        // (the first exchange of the transform starts with a barrier: lds0 is free by then)
        fft_regs<HS, true>(v, t, lds, wpre);
        if (live) {
#pragma unroll
            for (int m = 0; m < E; ++m) wsb[(unsigned)(qrow * HS + t + T * m) * (unsigned)wstride] = v[m];
        }
    }
    __syncthreads();
}

// Workspace accesses of the column kernel. NT: non-temporal (launches whose workspace cannot be cache-resident anyway:
// the streamed lines then leave the Infinity Cache to what is reused, the cov shared by the lattices of a unit;
// 4096^2 x 16: 117.7 -> 107.1 us per lattice).
typedef double gpt_d2v __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ cplx ws_ld(const cplx* p) {
    if (NT) {
        const gpt_d2v v = __builtin_nontemporal_load(reinterpret_cast<const gpt_d2v*>(p));
        return make_double2(v.x, v.y);
    }
    return *p;
}
template <bool NT> __device__ __forceinline__ void ws_st(cplx* p, cplx v) {
    if (NT) {
        const gpt_d2v t = {v.x, v.y};
        __builtin_nontemporal_store(t, reinterpret_cast<gpt_d2v*>(p));
    } else {
        *p = v;
    }
}

// pass B' -------------------------------------------------------------------------------------------
// Tiles: (sub-line q, TC adjacent packed columns), ntq column tiles per sub-line; tile = q * ntq + ct.
// Same persistent walk and XCD-aware order as col_kernel.
template <int HS, int V, int THREADS, int OP, bool NT = false>
__global__ void __launch_bounds__(THREADS, (V == 1 ? GPT_SUB_WPS1 : GPT_SUB_WPS2))  // waves per SIMD the register budget must allow
    colsub_kernel(LatArgs a, int R, cplx* __restrict__ ws, int wstride, const cplx* __restrict__ tws,
                  double* __restrict__ partials, int ntq, double* __restrict__ edge, int edge_flags, int npart) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = GeoS<HS, V, THREADS>;
    constexpr int T = G::T, TC = G::TC, TP = G::TP;
    constexpr int NPRE = tw_npre<HS>();
    constexpr bool NEEDS_COV = (OP != OP_RFFT2 && OP != OP_INV_RFFT2 && !op_is_kvjp(OP)); // cov is LOADED
    constexpr int CVN = V == 1 ? 8 : E; // cov values requested together (register budget of the 128-VGPR build)
    const int c = threadIdx.x % TC, tp = threadIdx.x / TC;
    double* lds = reinterpret_cast<double*>(smem) + (size_t)c * (HS + G::SKEW);
    double* red = reinterpret_cast<double*>(smem);
    const int64_t b = blockIdx.y;
    const int H = a.H;
    const ZLayout L = make_layout(a);
    EwPtrs P = make_ptrs(a, b);
    kern_prepare<OP>(a, P);
    if (edge) {
        if (edge_flags & 1) P.edge_cov = edge + b * 4 * (int64_t)H;
        if (edge_flags & 2) P.edge_dcov = edge + b * 4 * (int64_t)H + 2 * (int64_t)H;
    }
    cplx* wsb = ws + (int64_t)b * H * wstride;
    EwAcc acc;
    acc_init(acc);
    if (a.col0 == 0 && (int)blockIdx.x <= R / 2) {
        col0_role<HS, THREADS, OP>(a, L, P, R, (int)blockIdx.x, wsb, wstride, tws, smem, acc);
        acc_fold(acc);
    }
    const int ntiles = R * ntq;
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
#pragma unroll 1
    for (int r = 0; r < count; ++r) {
        const int tile = first + r * step;
        const int q = tile / ntq, ct = tile - q * ntq;
        const int j0 = ct * TC;
        const int jl = j0 + c;      // packed column inside the spectrum buffer
        const int jp = a.col0 + jl; // ... and inside the lattice
        const bool skip = (jp == 0); // the packed column 0 belongs to col0_role
        const int rowq = q * HS;
        int tl[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            tl[k] = tp + k * TP;
            asm volatile("" : "+v"(tl[k]));
        }
        // twiddle bases: (re)loaded per tile from the cached table rather than kept across the loop
        cplx wl[V][NPRE > 0 ? NPRE : 1];
#pragma unroll
        for (int k = 0; k < V; ++k) tw_load<HS>(wl[k], tws, tl[k]);
        cplx v[V][E];
        TRACE_MARK(r, 0);
#pragma unroll
        for (int k = 0; k < V; ++k)
#pragma unroll
            for (int m = 0; m < E; ++m) // (the lanes of the packed column 0 load it too: no per-load predicate)
                v[k][m] = op_has_fwd(OP)
                              ? ws_ld<NT>(&wsb[(unsigned)(rowq + tl[k]) * (unsigned)wstride + (unsigned)jl +
                                           (unsigned)(T * m) * (unsigned)wstride])
                              : make_double2(0.0, 0.0);
#if GPT_TRACE
        TRACE_MARK(r, 1); // loads issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TRACE_MARK(r, 2); // ... and arrived
#endif
        if (op_has_fwd(OP)) fft_regs<HS, false, V, 0>(v, tl, lds, wl);
        TRACE_MARK(r, 3); // forward transform done
        if (!skip) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                // offsets into cov / d/dcov: one multiplication per register tile, the rows of a tile are a uniform
                // stride apart
                const unsigned icb = (unsigned)(R * tl[k] + q) * (unsigned)L.Wf + (unsigned)jp;
                const unsigned icstep = (unsigned)(R * T) * (unsigned)L.Wf;
#pragma unroll
                for (int m0 = 0; m0 < E; m0 += CVN) {
                    double cv[CVN];
#pragma unroll
                    for (int m = 0; m < CVN; ++m) {
                        const unsigned ic = icb + (unsigned)(m0 + m) * icstep;
                        cv[m] = NEEDS_COV ? P.cov[ic] : 1.0;
                    }
#pragma unroll
                    for (int m = 0; m < CVN; ++m) {
                        int irow = R * (tl[k] + T * (m0 + m)) + q;
                        // (kernel-fused op: pin the row index here, or the 32 kernel evaluations of a thread are
                        // started side by side and spill)
                        if (op_is_kvjp(OP)) asm volatile("" : "+v"(irow));
                        v[k][m0 + m] = ew_interior_ic<OP>(a, L, P, irow, jp, icb + (unsigned)(m0 + m) * icstep,
                                                            v[k][m0 + m], cv[m], acc);
                        if ((m & 3) == 3 || op_is_kvjp(OP)) SCHED_FENCE(); // (the kernel evaluation is long)
                    }
                    SCHED_FENCE();
                }
            }
        }
        acc_fold(acc);
        __builtin_amdgcn_sched_barrier(0);
#if GPT_TRACE
        TRACE_MARK(r, 4); // elementwise stage issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TRACE_MARK(r, 5); // ... and its stores retired
#endif
        if (op_has_inv(OP)) {
            int t_inv[V];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                t_inv[k] = tl[k];
                asm volatile("" : "+v"(t_inv[k]), "+v"(v[0][0].x));
#pragma unroll
                for (int i = 0; i < NPRE; ++i) asm volatile("" : "+v"(wl[k][i].x), "+v"(wl[k][i].y), "+v"(v[0][0].x));
            }
            fft_regs<HS, true, V, 0>(v, t_inv, lds, wl);
            TRACE_MARK(r, 6); // inverse transform done
            if (!skip) {
#pragma unroll
                for (int k = 0; k < V; ++k)
#pragma unroll
                    for (int m = 0; m < E; ++m)
                        ws_st<NT>(&wsb[(unsigned)(rowq + t_inv[k]) * (unsigned)wstride + (unsigned)jl +
                                   (unsigned)(T * m) * (unsigned)wstride], v[k][m]);
            }
            TRACE_MARK(r, 7);
        }
    }
    if (OP != OP_RFFT2 && OP != OP_INV_RFFT2) {
        double s = acc.sum, mn = acc.mn;
        double kk[3] = {acc.k0, acc.k1, acc.k2};
        if (op_is_kvjp(OP)) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { // (block_sum_min below ends with a barrier-protected read by thread 0)
                double t = kk[i], dummy = 0.0;
                block_sum_min(t, dummy, red);
                kk[i] = t;
                __syncthreads();
            }
        }
        block_sum_min(s, mn, red);
        if (threadIdx.x == 0) {
            // a lattice has npart partial slots, whatever the grid of this geometry: the surplus ones get the
            // neutral element
            double* p = partials + ((int64_t)b * npart + blockIdx.x) * GPT_PART;
            p[0] = s;
            p[1] = op_is_kvjp(OP) ? kk[0] : 0.0;
            p[2] = op_is_kvjp(OP) ? kk[1] : 0.0;
            p[3] = op_is_kvjp(OP) ? kk[2] : 0.0;
            p[4] = 0.0;
            p[5] = mn;
            for (int k = blockIdx.x + gridDim.x; k < npart; k += gridDim.x) {
                double* q = partials + ((int64_t)b * npart + k) * GPT_PART;
                q[0] = q[1] = q[2] = q[3] = q[4] = 0.0;
                q[5] = INFINITY;
            }
        }
    }
}

// Plain complex line transform in place (unnormalised): `nlines` contiguous lines of N points. The sub-line
// transforms of the long-row passes (gpt_long.hip).
template <int N, bool INV>
__global__ void __launch_bounds__(Geo<N>::ROW_THREADS) line_fft_kernel(cplx* __restrict__ data, int64_t nlines,
                                                                       const cplx* __restrict__ tw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = Geo<N>::T, LINES = Geo<N>::ROW_LINES;
    const int line = threadIdx.x / T, t = threadIdx.x - line * T;
    double* lds = reinterpret_cast<double*>(smem) + (size_t)line * N;
    const int64_t l = (int64_t)blockIdx.x * LINES + line;
    const bool active = l < nlines;
    cplx* d = data + (active ? l : 0) * N;
    cplx wpre[tw_npre<N>() > 0 ? tw_npre<N>() : 1];
    tw_load<N>(wpre, tw, t);
    cplx v[E];
#pragma unroll
    for (int m = 0; m < E; ++m) v[m] = d[t + T * m];
    fft_regs<N, INV>(v, t, lds, wpre);
    if (!active) return;
#pragma unroll
    for (int m = 0; m < E; ++m) d[t + T * m] = v[m];
}

// Host-side launchers, one specialisation per line length (gpt_sub_n.hip).
template <int N> struct LaunchSub {
    // N = HS, the sub-line length
    static hipError_t colsub(int op, const LatArgs& a, int R, cplx* ws, int wstride, const cplx* tws, double* partials,
                             int nwg, double* edge, int edge_flags, hipStream_t st);
    static int col_tile();
    static int col_blocks(int R, int ncols, int cus);
    static hipError_t line_fft(bool inverse, cplx* data, int64_t nlines, const cplx* tw, hipStream_t st);
};

} // namespace pow2
