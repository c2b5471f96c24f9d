// Generic three-pass path: lattices of ANY shape (mixed radix, arbitrary prime factors) that are too large
// for one workgroup's LDS (gpt_small.hip) and are not power-of-two sized (gpt_pow2_impl.h). Same structure
// and the same packed-spectrum workspace as the power-of-two path, with the transforms done by the in-place LDS
// Stockham routine of gpt_lds_fft.h (one buffer; a thread's butterflies wait in registers for the barrier), so one line
// of up to 4096 points is 64 KB of LDS and two or three workgroups share a CU:
//   pass A  g_row_fwd  LP row pairs per workgroup: two-for-one real transform, split, pack
//   pass B  g_col      TC adjacent packed columns per workgroup: column FFT, elementwise stage, inverse FFT
//   pass C  g_row_inv  LP row pairs per workgroup: merge, inverse transform, store
// Every kernel exists twice: LEAN (line plans of radices 2, 3, 4, 5, 8: half the registers, what a 5-smooth length gets)
// and full (any factorisation). Numbers and what bounds them: DESIGN.md 5.3.
#include <atomic>

#include "gpt_common.h"
#include "gpt_host.h"
#ifndef GPT_LDS_FFT_INLINE
#define GPT_LDS_FFT_INLINE __forceinline__ // (as in gpt_small.hip: no stack copy of the plan, no spills around a call)
#endif
#include "gpt_lds_fft.h"

#ifndef GPT_TRACE
#define GPT_TRACE 0
#endif
#ifndef GPT_GEN_XCD_TILES
#define GPT_GEN_XCD_TILES 1 // (0: tile x of the grid is tile x of the lattice, for A/B builds)
#endif
#if GPT_TRACE
// debug builds only (scripts/trace_gen.py): phase timestamps (100 MHz clock) of thread 0 of column workgroup 1, lattice 0
__device__ unsigned long long g_gen_trace[16];
#define GMARK(k_)                                                                                   \
    do {                                                                                            \
        if (threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 0) g_gen_trace[k_] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define GMARK(k_) ((void)0)
#endif

namespace {

constexpr int G_THREADS = 512;
constexpr int G_MAX_LINE = 4096;      // longest transform (complex points) held in LDS
constexpr int G_BUF_ELEMS = 4096;     // most elements of a workgroup's tile (64 KB)
constexpr int G_EPT = 8;              // elements a thread holds across the barrier of an in-place stage: G_BUF_ELEMS / G_THREADS

struct GenGeo {
    int NC;      // packed columns
    int wstride; // workspace row stride (complex elements)
    int Hp;      // row pairs
    int LP;      // row pairs per workgroup
    int TC;      // columns per workgroup
    int TCs;     // LDS stride between rows of a column tile (= TC)
    // Tile sizes (complex elements; the LDS a launch asks for is the tile itself, not the bucket) and workgroup sizes of
    // the row and the column kernels: short lines take tiles of up to 1024 elements (<= 16 KB per workgroup, 256 threads),
    // so that several workgroups share a CU and one loads while another transforms, and a launch of few lattices still
    // fills the chip; only lines of more than 2048 points need 4096-element tiles (<= 64 KB). The column kernel sizes its tile for
    // eight columns first: tile width (bytes per workspace row) matters more to it than occupancy (301 x 200 x 64
    // lattices: 3-column tiles 204 us, 13-column tiles as before).
    int rbuf, cbuf, rthreads, cthreads;
    // multiply-high magics (gpt_lds_fft.h) of W, NC and of the line counts of a workgroup -- LP row pairs / TC columns, or
    // what is left for the last workgroup of a lattice
    unsigned mW, mNC, mLP, mLPtail, mTC, mTCtail;
};
// Stage constants of the two line plans, made on the host (StageK, gpt_lds_fft.h)
struct GenStages {
    StageK s[GPT_MAX_STAGES];
};
GenStages make_stages(const LinePlan& p) {
    GenStages g;
    int Ns = 1;
    for (int s = 0; s < p.ns && s < GPT_MAX_STAGES; ++s) {
        g.s[s] = stage_consts(p.n, p.r[s], Ns);
        Ns *= p.r[s];
    }
    return g;
}

// Column tiles of up to 8192 elements (128 KB, one workgroup of 1024 threads per CU) instead of 4096 (two workgroups of
// 512): taken for lattices of more than 2048 rows, whose tiles would otherwise be ONE column wide (3000^2: 305 -> 281 us
// per lattice; 1000^2, 8 columns instead of 4: 22.1 -> 22.9, so not there). GPT_GEN_COL_TILE=4096 / 8192 forces either
// for every lattice of more than 512 rows (A/B knob, profiles/r03_s_generic_inplace_ab.jsonl).
constexpr int G_COL_THREADS_MAX = 1024;
int col_tile_max(int H) {
    static const int forced = [] {
        const char* e = getenv("GPT_GEN_COL_TILE");
        const int n = e ? atoi(e) : 0;
        return n == 8192 || n == 4096 ? n : 0;
    }();
    return forced ? forced : (H > 2048 ? 8192 : G_BUF_ELEMS);
}
int buf_for_line(int n) { return n <= 1024 ? 1024 : (n <= 2048 ? 2048 : G_BUF_ELEMS); }

GenGeo make_geo(int H, int W) {
    GenGeo g;
    g.NC = 1 + (W - 1) / 2;
    g.wstride = (g.NC + 7) / 8 * 8 + 8;
    g.Hp = (H + 1) / 2;
    g.rbuf = buf_for_line(W);
    g.cbuf = buf_for_line(8 * H); // at least 8 adjacent columns (whole 128-byte lines of the workspace) where LDS allows
    if (8 * H > G_BUF_ELEMS && col_tile_max(H) > G_BUF_ELEMS) g.cbuf = col_tile_max(H);
    g.rthreads = g.rbuf <= 1024 ? 256 : G_THREADS;
    g.LP = g.rbuf / W;
    if (g.LP < 1) g.LP = 1;
    if (g.LP > g.Hp) g.LP = g.Hp;
    // tile widths that divide or are a multiple of a 128-byte line of the workspace (8 columns): 16, 8, 4, 2, 1 -- a tile
    // of 10 or 14 columns starts and ends inside lines that its neighbours fetch again
    g.TC = g.cbuf / H;
    g.TC = g.TC >= 16 ? 16 : g.TC >= 8 ? 8 : g.TC >= 4 ? 4 : g.TC >= 2 ? 2 : 1;
    if (g.TC > g.NC) g.TC = g.NC;
    // threads for G_EPT elements each (a multiple of a wave, at least two): a small tile takes a small workgroup, several
    // of which then share a CU and a launch of few lattices has no second round of workgroups
    // (tiles of more than 2048 elements: all 512 threads, whatever the count -- 3000 x 1 with 384 threads: 191 us per
    // lattice against 170)
    const int tile = H * g.TC;
    g.cthreads = tile > 4096 ? G_COL_THREADS_MAX : tile > 2048 ? G_THREADS : ((tile + G_EPT - 1) / G_EPT + 63) / 64 * 64;
    if (g.cthreads < 128) g.cthreads = 128;
    // (no padding between the columns of a tile: the stages number their work items ACROSS the columns, so neighbouring
    // lanes touch neighbouring elements at any tile width, and an odd stride would only cost LDS -- 1000 rows x 4
    // columns: 80 KB with the stride of 5 that was here, one workgroup per CU; 64 KB, two, without)
    g.TCs = g.TC;
    g.mW = fastdiv_magic_hd((unsigned)W);
    g.mNC = fastdiv_magic_hd((unsigned)g.NC);
    g.mLP = fastdiv_magic_hd((unsigned)g.LP);
    g.mLPtail = fastdiv_magic_hd((unsigned)(g.Hp % g.LP ? g.Hp % g.LP : g.LP));
    g.mTC = fastdiv_magic_hd((unsigned)g.TC);
    g.mTCtail = fastdiv_magic_hd((unsigned)(g.NC % g.TC ? g.NC % g.TC : g.TC));
    return g;
}

template <int IN_MODE> // 0 = x, 1 = x - loc, 2 = aux
__device__ __forceinline__ double g_in(const LatArgs& a, int64_t b, int i, int j) {
    const int64_t off = (int64_t)i * a.W + j;
    if (IN_MODE == 2) return a.aux[b * a.aux_bs + off];
    const double v = a.x[b * a.x_bs + off];
    if (IN_MODE == 1 && a.loc) return v - a.loc[b * a.loc_bs + off];
    return v;
}

template <int IN_MODE, bool LEAN>
__global__ void __launch_bounds__(G_THREADS) g_row_fwd(LatArgs a, GenGeo g, GenStages st, cplx* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsLine cur(smem); // (LDS-typed handle, gpt_lds_fft.h)
    const int64_t b = blockIdx.y;
    const int H = a.H, W = a.W;
    const int r0 = blockIdx.x * g.LP;
    const int nl = min(g.LP, g.Hp - r0);
    const bool weven = (W % 2) == 0;
#pragma unroll 4
    for (int w = threadIdx.x; w < nl * W; w += blockDim.x) {
        const int l = (int)fastdiv((unsigned)w, (unsigned)W, g.mW), j = w - l * W, r = r0 + l;
        const double re = g_in<IN_MODE>(a, b, 2 * r, j);
        const double im = (2 * r + 1 < H) ? g_in<IN_MODE>(a, b, 2 * r + 1, j) : 0.0;
        cur[w] = make_double2(re, im);
    }
    __syncthreads();
    fft_lines_lds_ip<false, G_EPT, const cplx*, LEAN>(cur, a.pw, a.pw.tw, nl, 1, W, st.s, nl == g.LP ? g.mLP : g.mLPtail);
    const LdsLine res = cur;
    cplx* wsb = ws + (int64_t)b * H * g.wstride;
    for (int w = threadIdx.x; w < 2 * nl * g.NC; w += blockDim.x) {
        const int li = (int)fastdiv((unsigned)w, (unsigned)g.NC, g.mNC), jp = w - li * g.NC;
        const int l = li >> 1, odd = li & 1;
        const int i = 2 * (r0 + l) + odd;
        if (i >= H) continue;
        const LdsLine line = res + l * W;
        auto rowspec = [&](int j) -> cplx {
            const cplx z = line[j];
            const cplx zc = line[j == 0 ? 0 : W - j];
            if (!odd) return make_double2(0.5 * (z.x + zc.x), 0.5 * (z.y - zc.y));
            return make_double2(0.5 * (z.y + zc.y), -0.5 * (z.x - zc.x));
        };
        cplx v;
        if (jp == 0) {
            v.x = rowspec(0).x;
            v.y = weven ? rowspec(W / 2).x : 0.0;
        } else {
            v = rowspec(jp);
        }
        wsb[(int64_t)i * g.wstride + jp] = v;
    }
}

template <int OP, bool LEAN>
__global__ void __launch_bounds__(G_COL_THREADS_MAX) g_col(LatArgs a, GenGeo g, GenStages st, cplx* __restrict__ ws,
                                                    double* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* red = reinterpret_cast<double*>(smem); // 512 bytes
    const LdsLine cur(smem + 512);
    const int64_t b = blockIdx.y;
    const int H = a.H;
    const ZLayout L = make_layout(a);
    EwPtrs P = make_ptrs(a, b);
    kern_prepare<OP>(a, P); // (the kernel-fused ops: this lattice's kernel parameters -> wave-uniform constants)
    // Workgroups go round the eight XCDs (each with an L2 of its own) in dispatch order: the tiles of one XCD are made
    // NEIGHBOURS in memory -- tile x of the grid is tile (x / 8) of the (x % 8)-th eighth of the columns -- so that the
    // workspace and `cov` lines that narrow tiles (tall lattices: 1 - 4 columns, 16 - 64 bytes of every 128-byte line)
    // share are fetched by one L2 and not by all eight.
    const int nfull = (int)gridDim.x >> 3, nrem = (int)gridDim.x & 7, xr = (int)blockIdx.x & 7;
    const int tile = GPT_GEN_XCD_TILES ? xr * nfull + min(xr, nrem) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int j0 = tile * g.TC;
    const int nc = min(g.TC, g.NC - j0);
    const int sS = g.TCs;
    cplx* wsb = ws + (int64_t)b * H * g.wstride;
    const unsigned mnc = nc == g.TC ? g.mTC : g.mTCtail;
    GMARK(0);
    // `cov` of the thread's entries (at most G_EPT: the tile has at most G_EPT * blockDim.x elements), asked for before the
    // workspace loads so that its latency passes behind theirs and the transform's, not in front of the elementwise stage
    constexpr bool uses_cov = OP != OP_RFFT2 && OP != OP_INV_RFFT2 && !op_is_kvjp(OP); // (those evaluate the kernel instead)
    double cpre[G_EPT];
    if (uses_cov) {
#pragma unroll
        for (int it = 0; it < G_EPT; ++it) {
            const int w = (int)threadIdx.x + it * (int)blockDim.x;
            cpre[it] = 1.0;
            if (w < H * nc) {
                const int k = (int)fastdiv((unsigned)w, (unsigned)nc, mnc), c = w - k * nc;
                if (j0 + c != 0) cpre[it] = P.cov[(unsigned)k * (unsigned)L.Wf + (unsigned)(j0 + c)];
            }
        }
    }
    if (op_has_fwd(OP)) {
        // (unrolled: the loads of eight iterations are in flight together; one workgroup per CU has nothing else to hide
        // a memory latency per iteration behind)
#pragma unroll 8
        for (int w = threadIdx.x; w < H * nc; w += blockDim.x) {
            const int k = (int)fastdiv((unsigned)w, (unsigned)nc, mnc), c = w - k * nc;
            cur[k * sS + c] = wsb[(unsigned)k * (unsigned)g.wstride + (unsigned)(j0 + c)];
        }
        __syncthreads();
        GMARK(1);
        fft_lines_lds_ip<false, G_EPT, const cplx*, LEAN>(cur, a.ph, a.ph.tw, nc, sS, 1, st.s, mnc);
    }
    GMARK(2);
    EwAcc acc;
    acc_init(acc);
#pragma unroll
    for (int it = 0; it < G_EPT; ++it) {
        const int w = (int)threadIdx.x + it * (int)blockDim.x;
        if (w >= H * nc) break;
        const int k = (int)fastdiv((unsigned)w, (unsigned)nc, mnc), c = w - k * nc;
        const int jp = j0 + c;
        if (jp == 0) continue;
        const cplx G = ew_interior_c<OP>(a, L, P, k, jp, cur[k * sS + c], uses_cov ? cpre[it] : 1.0, acc);
        if (op_has_inv(OP)) cur[k * sS + c] = G;
        acc_renorm(acc);
    }
    if (j0 == 0) {
        for (int k = threadIdx.x; k <= H / 2; k += blockDim.x) {
            const int km = k ? H - k : 0; // (H - k) % H
            cplx Zk = cur[k * sS], Zm = cur[km * sS];
            ew_packed0<OP>(a, L, P, k, Zk, Zm, acc);
            if (op_has_inv(OP)) {
                cur[k * sS] = Zk;
                if (km != k) cur[km * sS] = Zm;
            }
        }
    }
    acc_fold(acc);
    __syncthreads();
    GMARK(3);
    if (op_has_inv(OP)) {
        fft_lines_lds_ip<true, G_EPT, const cplx*, LEAN>(cur, a.ph, a.ph.tw, nc, sS, 1, st.s, mnc);
        const LdsLine res = cur;
        GMARK(4);
        for (int w = threadIdx.x; w < H * nc; w += blockDim.x) {
            const int k = (int)fastdiv((unsigned)w, (unsigned)nc, mnc), c = w - k * nc;
            wsb[(unsigned)k * (unsigned)g.wstride + (unsigned)(j0 + c)] = res[k * sS + c];
        }
        GMARK(5);
    }
    if (OP != OP_RFFT2 && OP != OP_INV_RFFT2) {
        double s = acc.sum, mn = acc.mn;
        double kk[3] = {acc.k0, acc.k1, acc.k2}; // (kernel-fused ops: the three sums of acc_kern, finalize_block turns them
                                                 // into d / d(sigma, length_scale, period))
        if (op_is_kvjp(OP)) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { // (block_sum_min ends with a barrier-protected read by thread 0)
                double t = kk[i], dummy = 0.0;
                block_sum_min(t, dummy, red);
                kk[i] = t;
                __syncthreads();
            }
        }
        block_sum_min(s, mn, red);
        if (threadIdx.x == 0) {
            double* p = partials + ((int64_t)b * gridDim.x + blockIdx.x) * GPT_PART;
            p[0] = s;
            p[1] = op_is_kvjp(OP) ? kk[0] : 0.0;
            p[2] = op_is_kvjp(OP) ? kk[1] : 0.0;
            p[3] = op_is_kvjp(OP) ? kk[2] : 0.0;
            p[4] = 0.0;
            p[5] = mn;
        }
    }
}

template <bool LEAN>
__global__ void __launch_bounds__(G_THREADS) g_row_inv(LatArgs a, GenGeo g, GenStages st, double alpha, int add_loc,
                                                       const cplx* __restrict__ ws, FinArgs fin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsLine cur(smem);
    const int64_t b = blockIdx.y;
    const int H = a.H, W = a.W;
    const int r0 = blockIdx.x * g.LP;
    const int nl = min(g.LP, g.Hp - r0);
    const bool weven = (W % 2) == 0;
    const cplx* wsb = ws + (int64_t)b * H * g.wstride;
    for (int w = threadIdx.x; w < nl * W; w += blockDim.x) {
        const int l = (int)fastdiv((unsigned)w, (unsigned)W, g.mW), j = w - l * W, r = r0 + l;
        auto rowspec = [&](int i, int jj) -> cplx {
            const bool mir = jj > W / 2;
            const int q = mir ? W - jj : jj;
            const cplx* row = wsb + (int64_t)i * g.wstride;
            cplx v;
            if (q == 0) v = make_double2(row[0].x, 0.0);
            else if (weven && q == W / 2) v = make_double2(row[0].y, 0.0);
            else v = row[q];
            if (mir) v.y = -v.y;
            return v;
        };
        const cplx A = rowspec(2 * r, j);
        cplx B = make_double2(0.0, 0.0);
        if (2 * r + 1 < H) B = rowspec(2 * r + 1, j);
        cur[w] = make_double2(A.x - B.y, A.y + B.x);
    }
    __syncthreads();
    fft_lines_lds_ip<true, G_EPT, const cplx*, LEAN>(cur, a.pw, a.pw.tw, nl, 1, W, st.s, nl == g.LP ? g.mLP : g.mLPtail);
    const LdsLine res = cur;
    double* ob = a.out + b * (int64_t)H * W;
    for (int w = threadIdx.x; w < nl * W; w += blockDim.x) {
        const int l = (int)fastdiv((unsigned)w, (unsigned)W, g.mW), j = w - l * W, r = r0 + l;
        const cplx v = res[w];
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * r + h;
            if (i >= H) break;
            double o = alpha * (h ? v.y : v.x);
            if (add_loc && a.loc) o += a.loc[b * a.loc_bs + (int64_t)i * W + j];
            ob[(int64_t)i * W + j] = o;
        }
    }
    if (fin.partials && blockIdx.x == 0) {
        // the column pass left per-workgroup partials: the first workgroup of every lattice adds them up (saves a launch)
        __syncthreads();
        finalize_block(fin, b, reinterpret_cast<double*>(smem));
    }
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (kernel, device): set once per pair (one bit per device; raising
// the cap twice is harmless, so a lost race costs one extra call), not on every launch.
hipError_t raise_lds_cap(const void* kernel, std::atomic<uint64_t>& done, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = dev < 64 ? (uint64_t)1 << dev : 0;
    if (done.load(std::memory_order_relaxed) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_relaxed);
    return e;
}

template <int OP, bool LEAN>
hipError_t launch_col_t(const LatArgs& a, const GenGeo& g, cplx* ws, double* partials, int nblk, hipStream_t st) {
    const size_t lds = 512 + ((size_t)a.H * g.TCs + 16) * sizeof(cplx);
    static std::atomic<uint64_t> done{0};
    hipError_t e = raise_lds_cap(reinterpret_cast<const void*>(&g_col<OP, LEAN>), done,
                                 (int)(512 + (size_t)(2 * G_BUF_ELEMS + 16) * sizeof(cplx)));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((g_col<OP, LEAN>), dim3(nblk, (unsigned)a.batch), dim3(g.cthreads), lds, st, a, g, make_stages(a.ph),
                       ws, partials);
    return hipGetLastError();
}
// (the lean instantiation where the column plan allows it: see fft_lines_lds_ip)
template <int OP>
hipError_t launch_col(const LatArgs& a, const GenGeo& g, cplx* ws, double* partials, int nblk, hipStream_t st) {
    return lean_plan(a.ph) ? launch_col_t<OP, true>(a, g, ws, partials, nblk, st)
                           : launch_col_t<OP, false>(a, g, ws, partials, nblk, st);
}

template <int IN_MODE, bool LEAN> hipError_t launch_row_fwd_t(const LatArgs& a, const GenGeo& g, cplx* ws, hipStream_t st) {
    const size_t lds = (size_t)g.LP * a.W * sizeof(cplx);
    static std::atomic<uint64_t> done{0};
    hipError_t e = raise_lds_cap(reinterpret_cast<const void*>(&g_row_fwd<IN_MODE, LEAN>), done,
                                 (int)((size_t)G_BUF_ELEMS * sizeof(cplx)));
    if (e != hipSuccess) return e;
    const int nb = (g.Hp + g.LP - 1) / g.LP;
    hipLaunchKernelGGL((g_row_fwd<IN_MODE, LEAN>), dim3(nb, (unsigned)a.batch), dim3(g.rthreads), lds, st, a, g,
                       make_stages(a.pw), ws);
    return hipGetLastError();
}
template <int IN_MODE> hipError_t launch_row_fwd(const LatArgs& a, const GenGeo& g, cplx* ws, hipStream_t st) {
    return lean_plan(a.pw) ? launch_row_fwd_t<IN_MODE, true>(a, g, ws, st) : launch_row_fwd_t<IN_MODE, false>(a, g, ws, st);
}
template <bool LEAN>
hipError_t launch_row_inv_t(const LatArgs& a, const GenGeo& g, double alpha, int add_loc, const cplx* ws, const FinArgs& fin,
                            hipStream_t st) {
    const size_t lds = (size_t)g.LP * a.W * sizeof(cplx);
    static std::atomic<uint64_t> done{0};
    hipError_t e = raise_lds_cap(reinterpret_cast<const void*>(&g_row_inv<LEAN>), done, (int)((size_t)G_BUF_ELEMS * sizeof(cplx)));
    if (e != hipSuccess) return e;
    const int nb = (g.Hp + g.LP - 1) / g.LP;
    hipLaunchKernelGGL(g_row_inv<LEAN>, dim3(nb, (unsigned)a.batch), dim3(g.rthreads), lds, st, a, g, make_stages(a.pw), alpha,
                       add_loc, ws, fin);
    return hipGetLastError();
}

} // namespace

bool gpt_generic_supported(int H, int W) { return H >= 1 && W >= 2 && H <= G_MAX_LINE && W <= G_MAX_LINE; }

size_t gpt_generic_workspace_bytes(int H, int W) {
    const GenGeo g = make_geo(H, W);
    return (size_t)H * (size_t)g.wstride * sizeof(cplx);
}

int gpt_generic_col_blocks(int H, int W) {
    const GenGeo g = make_geo(H, W);
    return (g.NC + g.TC - 1) / g.TC;
}

hipError_t gpt_launch_generic(int op, const LatArgs& a, cplx* ws, double* partials, hipStream_t st) {
    if (!gpt_generic_supported(a.H, a.W)) return hipErrorInvalidValue;
    const GenGeo g = make_geo(a.H, a.W);
    const int nblk = gpt_generic_col_blocks(a.H, a.W);
    const double n = (double)a.H * (double)a.W;
    hipError_t e = hipSuccess;
    if (op_has_fwd(op)) {
        if (op == OP_RFFT2) e = launch_row_fwd<0>(a, g, ws, st);
        else if (op == OP_INV_VJP) e = launch_row_fwd<2>(a, g, ws, st);
        else e = launch_row_fwd<1>(a, g, ws, st);
        if (e != hipSuccess) return e;
    }
    switch (op) {
    case OP_RFFT2: e = launch_col<OP_RFFT2>(a, g, ws, partials, nblk, st); break;
    case OP_INV_RFFT2: e = launch_col<OP_INV_RFFT2>(a, g, ws, partials, nblk, st); break;
    case OP_GP_RFFT2: e = launch_col<OP_GP_RFFT2>(a, g, ws, partials, nblk, st); break;
    case OP_GP_INV_RFFT2: e = launch_col<OP_GP_INV_RFFT2>(a, g, ws, partials, nblk, st); break;
    case OP_LPDF: e = launch_col<OP_LPDF>(a, g, ws, partials, nblk, st); break;
    case OP_LPDF_VJP: e = launch_col<OP_LPDF_VJP>(a, g, ws, partials, nblk, st); break;
    case OP_INV_VJP: e = launch_col<OP_INV_VJP>(a, g, ws, partials, nblk, st); break;
    case OP_FWD_VJP: e = launch_col<OP_FWD_VJP>(a, g, ws, partials, nblk, st); break;
    case OP_LPDF_VJP_NOCOV: e = launch_col<OP_LPDF_VJP_NOCOV>(a, g, ws, partials, nblk, st); break;
    case OP_LPDF_KVJP: e = launch_col<OP_LPDF_KVJP>(a, g, ws, partials, nblk, st); break;
    case OP_LPDF_KVJP_EQ: e = launch_col<OP_LPDF_KVJP_EQ>(a, g, ws, partials, nblk, st); break;
    default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    FinArgs fin = {nullptr, 0, 0.0, nullptr, nullptr, nullptr, nullptr};
    if (op != OP_RFFT2 && op != OP_INV_RFFT2) {
        const bool lp = (op == OP_LPDF || op_is_lpdf_vjp(op) || op == OP_INV_VJP); // (OP_INV_VJP: the z prior)
        const double add = -0.5 * n * 1.8378770664093454836;
        if (op_has_inv(op)) { // pass C adds the partials up (and finishes the kernel-parameter gradient)
            fin = FinArgs{partials, nblk, add, lp ? a.out_lpdf : nullptr, a.min_eig, op_is_kvjp(op) ? a.kpar : nullptr,
                          op_is_kvjp(op) ? a.kbar : nullptr};
        } else {
            e = gpt_launch_finalize(partials, nblk, a.batch, 1, add, lp ? a.out_lpdf : nullptr, a.min_eig, st);
            if (e != hipSuccess) return e;
        }
    }
    if (op_has_inv(op)) {
        double alpha = 1.0;
        int add_loc = 0;
        if (op == OP_INV_RFFT2) alpha = 1.0 / n;
        else if (op == OP_GP_INV_RFFT2) { alpha = 1.0 / n; add_loc = 1; }
        else if (op_is_lpdf_vjp(op)) alpha = -1.0 / n;
        e = lean_plan(a.pw) ? launch_row_inv_t<true>(a, g, alpha, add_loc, ws, fin, st)
                            : launch_row_inv_t<false>(a, g, alpha, add_loc, ws, fin, st);
    }
    return e;
}

#if GPT_TRACE
extern "C" int gpt_debug_gen_trace(unsigned long long* dst, size_t n) {
    if (n > 16) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gen_trace), n * sizeof(unsigned long long));
}
#endif
