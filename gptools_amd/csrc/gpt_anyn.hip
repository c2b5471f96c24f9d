// Any-length path: lattices with an axis that no other path takes -- 1-D series of more than ~5000 points, axes longer
// than 4096 points that are not powers of two, power-of-two shapes outside the tuned kernels' range. The reference takes
// any n (util.stan:263-265 rfft, fft1.stan:64-66); so does this path, up to lines of 2^23 points.
//
// Same mathematics and the same elementwise stage (gpt_common.h) as the three-pass paths, but every piece is a sweep
// over global memory, and the line transforms are built from the LDS Stockham routine (gpt_lds_fft.h, <= 4096 points):
//   * n <= 4096, smooth (largest prime factor <= 127):   one workgroup per few lines;
//   * n = n1 * n2, both <= 4096, smooth:                  four-step -- n1-point transforms over elements n2 apart, twiddle
//     w_n^(k1 j2), n2-point transforms of contiguous pieces. The result stays in the order the last step leaves it
//     (X[k1 + n1 k2] at position k1 n2 + k2); the consumers address it through pos(), the inverse transform takes it
//     in that order and returns natural order, so no reordering sweep exists;
//   * anything else (a large prime factor, e.g. n = 10007): Bluestein's chirp-z on a power-of-two length
//     M >= 2n - 1, itself direct or four-step; chirps from exact integer arithmetic (j^2 mod 2n) and sincospi.
// One evaluation: rows (load two real rows per complex line -> transform -> split into the packed spectrum), transpose,
// columns (transform -> elementwise stage on 32 x 32 tiles turned through LDS so that cov / d/dcov / z are touched along
// their rows -> inverse transform), transpose, rows (merge -> inverse transform -> store). Correctness first: ~12 sweeps
// where the tuned path has three.
#include "gpt_common.h"
#include "gpt_host.h"
#include "gpt_lds_fft.h"

#include <map>

namespace {

constexpr int A_THREADS = 512;
constexpr int A_BUF = 4096;                          // most complex elements of a workgroup's LDS tile (transformed in place)
constexpr int A_EPT = A_BUF / A_THREADS;             // elements a thread holds across the barrier of an in-place stage
constexpr int A_SMOOTH = 127;                        // largest prime factor the Stockham stages are given
constexpr int64_t A_CHUNK_ELEMS = (int64_t)1 << 25;  // complex elements per scratch buffer (512 MiB)
constexpr int A_MAX_LOG2_M = 24;                     // longest Bluestein / four-step length: 4096 * 4096

struct Fft { // direct (n1 == 1, p2 = plan of n) or four-step (p1 = plan of n1, p2 = plan of n2)
    int n, n1, n2;
    LinePlan p1, p2;
};
struct Axis {
    int n;
    bool blue;    // Bluestein: f is the transform of length M = f.n
    Fft f;
    const cplx* chirp; // exp(-i pi j^2 / n), j < n
    const cplx* bhat;  // transform (f, forward, in f's output order) of the wrapped conjugate chirp
    int pn1, pn2;      // output order of the forward transform: X[k] at (k % pn1) * pn2 + k / pn1
};

__device__ __forceinline__ cplx unit_root(int64_t e, int64_t n, bool inv) { // exp(-+ 2 pi i e / n), 0 <= e < n
    double s, c;
    sincospi(2.0 * (double)e / (double)n, &s, &c);
    return make_double2(c, inv ? s : -s);
}
__device__ __forceinline__ int pos_of(int k, int n1, int n2) { return (k % n1) * n2 + k / n1; }

__global__ void anyn_table_kernel(cplx* __restrict__ tw, int n) { // w_n^m (exact at the octants)
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < n) tw[m] = unit_root(m, n, false);
}

// ---- line transforms ----------------------------------------------------------------------------------
// contiguous lines of n <= 4096 points, LP lines per workgroup
template <bool INV, bool LEAN> // LEAN: plans of radices 2, 3, 4, 5, 8 only (gpt_lds_fft.h: half the registers)
__global__ void __launch_bounds__(A_THREADS) anyn_lines_fft(cplx* __restrict__ data, int64_t nlines, LinePlan p, int LP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsLine cur(smem);
    const int n = p.n;
    const int64_t l0 = (int64_t)blockIdx.x * LP;
    const int nl = (int)(nlines - l0 < LP ? nlines - l0 : LP);
    cplx* d = data + l0 * n;
    const int tot = nl * n;
#pragma unroll 4
    for (int w = threadIdx.x; w < tot; w += blockDim.x) cur[w] = d[w];
    __syncthreads();
    fft_lines_lds_ip<INV, A_EPT, const cplx*, LEAN>(cur, p, p.tw, nl, 1, n, nullptr, fastdiv_magic((unsigned)nl));
    for (int w = threadIdx.x; w < tot; w += blockDim.x) d[w] = cur[w];
}

// four-step, the strided step: every line is an n1 x n2 matrix; n1-point transforms down TC adjacent columns per
// workgroup, with the twiddle w_n^(k1 j2) behind (forward) or its conjugate in front (inverse)
template <bool INV, bool LEAN>
__global__ void __launch_bounds__(A_THREADS) anyn_cols_fft(cplx* __restrict__ data, LinePlan p1, int n2, int TC, int TCs,
                                                           int ntile) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsLine cur(smem);
    const int n1 = p1.n;
    const int64_t n = (int64_t)n1 * n2;
    const int64_t line = blockIdx.x / ntile;
    const int tile = (int)(blockIdx.x - line * ntile);
    const int j0 = tile * TC;
    const int nc = min(TC, n2 - j0);
    cplx* d = data + line * n;
    const int tot = n1 * nc;
    for (int w = threadIdx.x; w < tot; w += blockDim.x) {
        const int k = w / nc, c = w - k * nc;
        cplx v = d[(int64_t)k * n2 + j0 + c];
        if (INV) v = cmul(v, unit_root((int64_t)k * (j0 + c), n, true));
        cur[k * TCs + c] = v;
    }
    __syncthreads();
    fft_lines_lds_ip<INV, A_EPT, const cplx*, LEAN>(cur, p1, p1.tw, nc, TCs, 1, nullptr, fastdiv_magic((unsigned)nc));
    for (int w = threadIdx.x; w < tot; w += blockDim.x) {
        const int k = w / nc, c = w - k * nc;
        cplx v = cur[k * TCs + c];
        if (!INV) v = cmul(v, unit_root((int64_t)k * (j0 + c), n, false));
        d[(int64_t)k * n2 + j0 + c] = v;
    }
}

// ---- Bluestein ------------------------------------------------------------------------------------------
__global__ void blue_chirp_kernel(cplx* __restrict__ chirp, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t e = ((int64_t)j * j) % (2 * (int64_t)n); // exact: the angle is pi e / n
    double s, c;
    sincospi((double)e / (double)n, &s, &c);
    chirp[j] = make_double2(c, -s);
}
// b[m] = conj(chirp[|m|]) for -(n-1) <= m <= n-1, wrapped into M >= 2n - 1 points
__global__ void blue_filter_kernel(cplx* __restrict__ b, const cplx* __restrict__ chirp, int n, int M) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    cplx v = make_double2(0.0, 0.0);
    if (m < n) v = cconj(chirp[m]);
    else if (M - m < n) v = cconj(chirp[M - m]);
    b[m] = v;
}
// the inverse (e^{+}) transform of a line is conj(forward(conj(line)))
template <bool INV>
__global__ void blue_pre_kernel(const cplx* __restrict__ lines, cplx* __restrict__ scr, const cplx* __restrict__ chirp,
                                int n, int M, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t l = i / M;
        const int m = (int)(i - l * M);
        cplx v = make_double2(0.0, 0.0);
        if (m < n) {
            const cplx x = lines[l * n + m];
            v = cmul(INV ? cconj(x) : x, chirp[m]);
        }
        scr[i] = v;
    }
}
__global__ void blue_mul_kernel(cplx* __restrict__ scr, const cplx* __restrict__ bhat, int M, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        scr[i] = cmul(scr[i], bhat[i % M]);
}
template <bool INV>
__global__ void blue_post_kernel(cplx* __restrict__ lines, const cplx* __restrict__ scr, const cplx* __restrict__ chirp,
                                 int n, int M, int64_t total) {
    const double inv_m = 1.0 / (double)M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t l = i / n;
        const int k = (int)(i - l * n);
        cplx v = cscale(cmul(scr[l * M + k], chirp[k]), inv_m);
        lines[i] = INV ? cconj(v) : v;
    }
}

// ---- the lattice around the transforms --------------------------------------------------------------------
template <int IN_MODE> // 0 = x, 1 = x - loc, 2 = aux
__global__ void anyn_load_kernel(LatArgs a, cplx* __restrict__ lines, int Hp, int64_t total) {
    const int H = a.H, W = a.W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % W);
        const int64_t t = i / W;
        const int r = (int)(t % Hp);
        const int64_t b = t / Hp;
        auto in = [&](int row) -> double {
            const int64_t off = (int64_t)row * W + j;
            if (IN_MODE == 2) return a.aux[b * a.aux_bs + off];
            const double v = a.x[b * a.x_bs + off];
            return (IN_MODE == 1 && a.loc) ? v - a.loc[b * a.loc_bs + off] : v;
        };
        lines[i] = make_double2(in(2 * r), (2 * r + 1 < H) ? in(2 * r + 1) : 0.0);
    }
}

// two-for-one split of the transformed lines into the packed spectrum [batch][H][NC] (gpt_common.h); n1, n2 = order of
// the transformed lines
__global__ void anyn_split_kernel(const cplx* __restrict__ lines, cplx* __restrict__ ws, int H, int W, int NC, int Hp,
                                  int n1, int n2, int64_t total) {
    const bool weven = (W % 2) == 0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int jp = (int)(idx % NC);
        const int64_t t = idx / NC;
        const int i = (int)(t % H);
        const int64_t b = t / H;
        const cplx* line = lines + (b * Hp + i / 2) * W;
        const bool odd = i & 1;
        auto rowspec = [&](int j) -> cplx {
            const cplx z = line[pos_of(j, n1, n2)];
            const cplx zc = line[pos_of(j == 0 ? 0 : W - j, n1, n2)];
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
        ws[idx] = v;
    }
}

// the way back: rows 2r, 2r + 1 of the packed spectrum -> one complex line, written in the order the inverse line
// transform takes
__global__ void anyn_merge_kernel(const cplx* __restrict__ ws, cplx* __restrict__ lines, int H, int W, int NC, int Hp,
                                  int n1, int n2, int64_t total) {
    const bool weven = (W % 2) == 0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % W);
        const int64_t t = idx / W;
        const int r = (int)(t % Hp);
        const int64_t b = t / Hp;
        const cplx* wsb = ws + b * (int64_t)H * NC;
        auto rowspec = [&](int i) -> cplx {
            const bool mir = j > W / 2;
            const int q = mir ? W - j : j;
            const cplx* row = wsb + (int64_t)i * NC;
            cplx v;
            if (q == 0) v = make_double2(row[0].x, 0.0);
            else if (weven && q == W / 2) v = make_double2(row[0].y, 0.0);
            else v = row[q];
            if (mir) v.y = -v.y;
            return v;
        };
        const cplx A = rowspec(2 * r);
        const cplx B = (2 * r + 1 < H) ? rowspec(2 * r + 1) : make_double2(0.0, 0.0);
        lines[(b * Hp + r) * W + pos_of(j, n1, n2)] = make_double2(A.x - B.y, A.y + B.x);
    }
}

__global__ void anyn_store_kernel(LatArgs a, const cplx* __restrict__ lines, int Hp, double alpha, int add_loc,
                                  int64_t total) {
    const int H = a.H, W = a.W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % W);
        const int64_t t = i / W;
        const int r = (int)(t % Hp);
        const int64_t b = t / Hp;
        const cplx v = lines[i];
        double* ob = a.out + b * (int64_t)H * W;
        for (int h = 0; h < 2; ++h) {
            const int row = 2 * r + h;
            if (row >= H) break;
            double o = alpha * (h ? v.y : v.x);
            if (add_loc && a.loc) o += a.loc[b * a.loc_bs + (int64_t)row * W + j];
            ob[(int64_t)row * W + j] = o;
        }
    }
}

// Elementwise stage on the transposed spectrum cols[b][jp][position of row k] (h1, h2 = order of the column
// transform). Tiles of TJ columns x TK positions are read along the columns, turned in LDS and processed along the
// rows of cov / d/dcov / z. Packed column 0 belongs to anyn_ew_col0_kernel.
template <int OP>
__global__ void __launch_bounds__(256) anyn_ew_kernel(LatArgs a, cplx* __restrict__ cols, int NC, int h1, int h2, int TK,
                                                       int TJ, int ntk, int64_t ntiles, double* __restrict__ partials,
                                                       int npart) {
    __shared__ cplx tile[2048];
    __shared__ double red[16];
    const int64_t b = blockIdx.y;
    const int H = a.H;
    const int TKs = TK | 1;
    const ZLayout L = make_layout(a);
    const EwPtrs P = make_ptrs(a, b);
    cplx* cb = cols + b * (int64_t)NC * H;
    EwAcc acc;
    acc_init(acc);
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tj = (int)(t / ntk), tk = (int)(t - (int64_t)tj * ntk);
        const int j0 = tj * TJ, p0 = tk * TK;
        const int nj = min(TJ, NC - j0), nk = min(TK, H - p0);
        const int tot = nj * nk;
        if (op_has_fwd(OP)) {
            for (int e = threadIdx.x; e < tot; e += blockDim.x) {
                const int jl = e / nk, pl = e - jl * nk;
                tile[jl * TKs + pl] = cb[(int64_t)(j0 + jl) * H + p0 + pl];
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < tot; e += blockDim.x) {
            const int pl = e / nj, jl = e - pl * nj;
            const int jp = j0 + jl;
            if (jp == 0) continue;
            const int p = p0 + pl;
            const int k = p / h2 + h1 * (p % h2); // the row whose coefficient sits at position p
            const cplx F = op_has_fwd(OP) ? tile[jl * TKs + pl] : make_double2(0.0, 0.0);
            const cplx G = ew_interior<OP>(a, L, P, k, jp, F, acc);
            if (op_has_inv(OP)) tile[jl * TKs + pl] = G;
            acc_renorm(acc);
        }
        __syncthreads();
        if (op_has_inv(OP)) {
            for (int e = threadIdx.x; e < tot; e += blockDim.x) {
                const int jl = e / nk, pl = e - jl * nk;
                if (j0 + jl == 0) continue;
                cb[(int64_t)(j0 + jl) * H + p0 + pl] = tile[jl * TKs + pl];
            }
            __syncthreads();
        }
    }
    acc_fold(acc);
    if (OP != OP_RFFT2 && OP != OP_INV_RFFT2) {
        double s = acc.sum, mn = acc.mn;
        block_sum_min(s, mn, red);
        if (threadIdx.x == 0) {
            double* p = partials + (b * npart + blockIdx.x) * GPT_PART;
            p[0] = s;
            p[1] = p[2] = p[3] = p[4] = 0.0;
            p[5] = mn;
        }
    }
}

template <int OP>
__global__ void __launch_bounds__(256) anyn_ew_col0_kernel(LatArgs a, cplx* __restrict__ cols, int NC, int h1, int h2,
                                                            double* __restrict__ partials, int npart, int slot0) {
    __shared__ double red[16];
    const int64_t b = blockIdx.y;
    const int H = a.H;
    const ZLayout L = make_layout(a);
    const EwPtrs P = make_ptrs(a, b);
    cplx* cb = cols + b * (int64_t)NC * H; // column 0: the first H elements
    EwAcc acc;
    acc_init(acc);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k <= H / 2; k += gridDim.x * blockDim.x) {
        const int km = (H - k) % H;
        const int pk = pos_of(k, h1, h2), pm = pos_of(km, h1, h2);
        cplx Zk = make_double2(0.0, 0.0), Zm = Zk;
        if (op_has_fwd(OP)) {
            Zk = cb[pk];
            Zm = cb[pm];
        }
        ew_packed0<OP>(a, L, P, k, Zk, Zm, acc);
        if (op_has_inv(OP)) {
            cb[pk] = Zk;
            if (km != k) cb[pm] = Zm;
        }
    }
    acc_fold(acc);
    if (OP != OP_RFFT2 && OP != OP_INV_RFFT2) {
        double s = acc.sum, mn = acc.mn;
        block_sum_min(s, mn, red);
        if (threadIdx.x == 0) {
            double* p = partials + (b * npart + slot0 + blockIdx.x) * GPT_PART;
            p[0] = s;
            p[1] = p[2] = p[3] = p[4] = 0.0;
            p[5] = mn;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------
int largest_prime_factor(int n) {
    int best = 1;
    for (int f = 2; (int64_t)f * f <= n; ++f)
        while (n % f == 0) {
            best = f;
            n /= f;
        }
    return n > 1 ? n : best;
}
// direct or four-step: n1 == 1 or n = n1 * n2 with both factors <= 4096
bool fft_shape(int n, int* n1, int* n2) {
    if (n < 1 || largest_prime_factor(n) > A_SMOOTH) return false;
    if (n <= A_BUF) {
        *n1 = 1;
        *n2 = n;
        return true;
    }
    for (int d = (n + A_BUF - 1) / A_BUF; d <= A_BUF; ++d)
        if (n % d == 0) { // the smallest first factor: the widest column tiles of the strided step
            *n1 = d;
            *n2 = n / d;
            return true;
        }
    return false;
}
int bluestein_length(int n) { // smallest power of two >= 2n - 1, 0 if beyond the longest four-step line
    int64_t m = 1;
    while (m < 2 * (int64_t)n - 1) m <<= 1;
    return m <= ((int64_t)1 << A_MAX_LOG2_M) ? (int)m : 0;
}
bool axis_ok(int n) {
    int n1, n2;
    return n >= 1 && (fft_shape(n, &n1, &n2) || bluestein_length(n) > 0);
}
unsigned blocks_for(int64_t total, int per = 256, int64_t cap = 65536) {
    int64_t nb = (total + per - 1) / per;
    if (nb < 1) nb = 1;
    return (unsigned)(nb < cap ? nb : cap);
}

} // namespace

struct gpt_anyn {
    std::map<int, cplx*> tw;
    std::map<int, Axis> axes;
    cplx *ws = nullptr, *tmp = nullptr, *bscr = nullptr;
    size_t ws_elems = 0, tmp_elems = 0, bscr_elems = 0;
    double* partials = nullptr;
    size_t partials_elems = 0;
    bool lds_raised = false;
};

namespace {

#define A_TRY(expr)                                                                                            \
    do {                                                                                                       \
        hipError_t e__ = (expr);                                                                               \
        if (e__ != hipSuccess) return e__;                                                                     \
    } while (0)

template <typename T> hipError_t need(T*& p, size_t& have, size_t want, hipStream_t st) {
    if (want <= have) return hipSuccess;
    if (p) {
        A_TRY(hipStreamSynchronize(st));
        A_TRY(hipFree(p));
        p = nullptr;
        have = 0;
    }
    void* q = nullptr;
    A_TRY(hipMalloc(&q, want * sizeof(T)));
    p = reinterpret_cast<T*>(q);
    have = want;
    return hipSuccess;
}

hipError_t line_plan(gpt_anyn* s, int n, LinePlan& p, hipStream_t st) {
    p.n = n;
    p.ns = 0;
    int m = n;
    auto push = [&](int r) {
        if (p.ns < GPT_MAX_STAGES) p.r[p.ns] = r;
        p.ns++;
    };
    int odd = n; // 5-smooth lengths keep to radices 8, 4, 2, 3, 5: the lean kernels (as plan_radices, gpt_api.hip)
    while (odd % 2 == 0) odd /= 2;
    while (odd % 3 == 0) odd /= 3;
    while (odd % 5 == 0) odd /= 5;
    if (odd == 1) {
        while (m % 8 == 0) { push(8); m /= 8; }
    } else {
        while (m % 16 == 0) { push(16); m /= 16; } // register butterflies: one pass through LDS for four bits
        if (m % 8 == 0) { push(8); m /= 8; }
    }
    while (m % 4 == 0) { push(4); m /= 4; }
    while (m % 2 == 0) { push(2); m /= 2; }
    for (int f = 3; (int64_t)f * f <= m; f += 2)
        while (m % f == 0) { push(f); m /= f; }
    if (m > 1) push(m);
    if (p.ns > GPT_MAX_STAGES) return hipErrorInvalidValue;
    auto it = s->tw.find(n);
    if (it == s->tw.end()) {
        void* d = nullptr;
        A_TRY(hipMalloc(&d, sizeof(cplx) * (size_t)n));
        hipLaunchKernelGGL(anyn_table_kernel, dim3((n + 255) / 256), dim3(256), 0, st, reinterpret_cast<cplx*>(d), n);
        A_TRY(hipGetLastError());
        A_TRY(hipStreamSynchronize(st)); // (tables outlive the stream the context runs on today)
        it = s->tw.emplace(n, reinterpret_cast<cplx*>(d)).first;
    }
    p.tw = it->second;
    return hipSuccess;
}

hipError_t make_fft(gpt_anyn* s, int n, Fft& f, hipStream_t st) {
    f.n = n;
    if (!fft_shape(n, &f.n1, &f.n2)) return hipErrorInvalidValue;
    if (f.n1 > 1) A_TRY(line_plan(s, f.n1, f.p1, st));
    else f.p1 = LinePlan{1, 0, {0}, nullptr};
    return line_plan(s, f.n2, f.p2, st);
}

hipError_t raise_lds(gpt_anyn* s) {
    if (s->lds_raised) return hipSuccess;
    const int a = A_BUF * (int)sizeof(cplx), b = A_BUF * (int)sizeof(cplx);
    const void* lines[4] = {reinterpret_cast<const void*>(&anyn_lines_fft<false, false>),
                            reinterpret_cast<const void*>(&anyn_lines_fft<true, false>),
                            reinterpret_cast<const void*>(&anyn_lines_fft<false, true>),
                            reinterpret_cast<const void*>(&anyn_lines_fft<true, true>)};
    const void* cols[4] = {reinterpret_cast<const void*>(&anyn_cols_fft<false, false>),
                           reinterpret_cast<const void*>(&anyn_cols_fft<true, false>),
                           reinterpret_cast<const void*>(&anyn_cols_fft<false, true>),
                           reinterpret_cast<const void*>(&anyn_cols_fft<true, true>)};
    for (int i = 0; i < 4; ++i) {
        A_TRY(hipFuncSetAttribute(lines[i], hipFuncAttributeMaxDynamicSharedMemorySize, a));
        A_TRY(hipFuncSetAttribute(cols[i], hipFuncAttributeMaxDynamicSharedMemorySize, b));
    }
    s->lds_raised = true;
    return hipSuccess;
}

hipError_t lines_fft(const LinePlan& p, cplx* data, int64_t nlines, bool inv, hipStream_t st) {
    if (p.n <= 1 || nlines <= 0) return hipSuccess;
    // power-of-two lines of 256 ... 4096 points: the register-tile kernel of the tuned path (bandwidth-bound, 3x the
    // LDS Stockham routine below)
    if (gpt_pow2_line_fft_supported(p.n)) return gpt_pow2_line_fft(p.n, inv, data, nlines, p.tw, st);
    const int LP = A_BUF / p.n > 1 ? A_BUF / p.n : 1;
    const dim3 grid((unsigned)((nlines + LP - 1) / LP));
    const size_t lds = (size_t)LP * p.n * sizeof(cplx); // (the tile itself: short lines leave room for more workgroups)
    const bool lean = lean_plan(p);
    if (inv && lean) hipLaunchKernelGGL((anyn_lines_fft<true, true>), grid, dim3(A_THREADS), lds, st, data, nlines, p, LP);
    else if (inv) hipLaunchKernelGGL((anyn_lines_fft<true, false>), grid, dim3(A_THREADS), lds, st, data, nlines, p, LP);
    else if (lean) hipLaunchKernelGGL((anyn_lines_fft<false, true>), grid, dim3(A_THREADS), lds, st, data, nlines, p, LP);
    else hipLaunchKernelGGL((anyn_lines_fft<false, false>), grid, dim3(A_THREADS), lds, st, data, nlines, p, LP);
    return hipGetLastError();
}

hipError_t cols_fft(const Fft& f, cplx* data, int64_t nlines, bool inv, hipStream_t st) {
    int TC = A_BUF / f.n1; // as many adjacent columns as the buffer holds: a short first factor gets a wide tile
    if (TC < 1) TC = 1;
    if (TC > f.n2) TC = f.n2;
    const int TCs = TC; // (no padding: the stages number their work items across the columns, see gpt_generic.hip)
    const int ntile = (f.n2 + TC - 1) / TC;
    const dim3 grid((unsigned)(nlines * ntile));
    const size_t lds = (size_t)f.n1 * TCs * sizeof(cplx);
    const bool lean = lean_plan(f.p1);
    if (inv && lean) hipLaunchKernelGGL((anyn_cols_fft<true, true>), grid, dim3(A_THREADS), lds, st, data, f.p1, f.n2, TC, TCs, ntile);
    else if (inv) hipLaunchKernelGGL((anyn_cols_fft<true, false>), grid, dim3(A_THREADS), lds, st, data, f.p1, f.n2, TC, TCs, ntile);
    else if (lean) hipLaunchKernelGGL((anyn_cols_fft<false, true>), grid, dim3(A_THREADS), lds, st, data, f.p1, f.n2, TC, TCs, ntile);
    else hipLaunchKernelGGL((anyn_cols_fft<false, false>), grid, dim3(A_THREADS), lds, st, data, f.p1, f.n2, TC, TCs, ntile);
    return hipGetLastError();
}

// in place on `nlines` contiguous lines of f.n points; forward leaves the four-step order, inverse takes it
hipError_t fft_run(const Fft& f, cplx* data, int64_t nlines, bool inv, hipStream_t st) {
    if (f.n <= 1 || nlines <= 0) return hipSuccess;
    if (f.n1 == 1) return lines_fft(f.p2, data, nlines, inv, st);
    if (!inv) {
        A_TRY(cols_fft(f, data, nlines, false, st));
        return lines_fft(f.p2, data, nlines * f.n1, false, st);
    }
    A_TRY(lines_fft(f.p2, data, nlines * f.n1, true, st));
    return cols_fft(f, data, nlines, true, st);
}

hipError_t get_axis(gpt_anyn* s, int n, const Axis** out, hipStream_t st) {
    auto it = s->axes.find(n);
    if (it != s->axes.end()) {
        *out = &it->second;
        return hipSuccess;
    }
    Axis ax;
    ax.n = n;
    ax.chirp = ax.bhat = nullptr;
    int n1, n2;
    ax.blue = !fft_shape(n, &n1, &n2);
    if (!ax.blue) {
        A_TRY(make_fft(s, n, ax.f, st));
        ax.pn1 = ax.f.n1;
        ax.pn2 = ax.f.n2;
    } else {
        const int M = bluestein_length(n);
        if (M == 0) return hipErrorInvalidValue;
        A_TRY(make_fft(s, M, ax.f, st));
        ax.pn1 = 1;
        ax.pn2 = n;
        void *c = nullptr, *b = nullptr;
        auto tables = [&]() -> hipError_t {
            A_TRY(hipMalloc(&c, sizeof(cplx) * (size_t)n));
            A_TRY(hipMalloc(&b, sizeof(cplx) * (size_t)M));
            hipLaunchKernelGGL(blue_chirp_kernel, dim3((n + 255) / 256), dim3(256), 0, st, reinterpret_cast<cplx*>(c), n);
            A_TRY(hipGetLastError());
            hipLaunchKernelGGL(blue_filter_kernel, dim3((M + 255) / 256), dim3(256), 0, st, reinterpret_cast<cplx*>(b),
                               reinterpret_cast<const cplx*>(c), n, M);
            A_TRY(hipGetLastError());
            A_TRY(fft_run(ax.f, reinterpret_cast<cplx*>(b), 1, false, st));
            return hipStreamSynchronize(st);
        };
        if (const hipError_t e = tables(); e != hipSuccess) {
            if (c) (void)hipFree(c);
            if (b) (void)hipFree(b);
            return e;
        }
        ax.chirp = reinterpret_cast<const cplx*>(c);
        ax.bhat = reinterpret_cast<const cplx*>(b);
    }
    *out = &s->axes.emplace(n, ax).first->second;
    return hipSuccess;
}

// transform of `nlines` contiguous lines of ax.n points, in place
hipError_t axis_run(gpt_anyn* s, const Axis& ax, cplx* data, int64_t nlines, bool inv, hipStream_t st) {
    if (ax.n <= 1 || nlines <= 0) return hipSuccess;
    if (!ax.blue) return fft_run(ax.f, data, nlines, inv, st);
    const int M = ax.f.n, n = ax.n;
    int64_t per = A_CHUNK_ELEMS / M;
    if (per < 1) per = 1;
    if (per > nlines) per = nlines;
    A_TRY(need(s->bscr, s->bscr_elems, (size_t)per * (size_t)M, st));
    for (int64_t l0 = 0; l0 < nlines; l0 += per) {
        const int64_t nl = nlines - l0 < per ? nlines - l0 : per;
        cplx* d = data + l0 * n;
        const dim3 gm(blocks_for(nl * M)), gn(blocks_for(nl * n));
        if (inv) hipLaunchKernelGGL(blue_pre_kernel<true>, gm, dim3(256), 0, st, d, s->bscr, ax.chirp, n, M, nl * M);
        else hipLaunchKernelGGL(blue_pre_kernel<false>, gm, dim3(256), 0, st, d, s->bscr, ax.chirp, n, M, nl * M);
        A_TRY(hipGetLastError());
        A_TRY(fft_run(ax.f, s->bscr, nl, false, st));
        hipLaunchKernelGGL(blue_mul_kernel, gm, dim3(256), 0, st, s->bscr, ax.bhat, M, nl * M);
        A_TRY(hipGetLastError());
        A_TRY(fft_run(ax.f, s->bscr, nl, true, st));
        if (inv) hipLaunchKernelGGL(blue_post_kernel<true>, gn, dim3(256), 0, st, d, s->bscr, ax.chirp, n, M, nl * n);
        else hipLaunchKernelGGL(blue_post_kernel<false>, gn, dim3(256), 0, st, d, s->bscr, ax.chirp, n, M, nl * n);
        A_TRY(hipGetLastError());
    }
    return hipSuccess;
}

struct EwGeo {
    int TK, TJ, ntk, nblk, nblk0;
    int64_t ntiles;
};
EwGeo ew_geo(int H, int NC) {
    EwGeo g;
    g.TK = H < 32 ? H : 32;
    g.TJ = 1024 / g.TK;
    g.ntk = (H + g.TK - 1) / g.TK;
    g.ntiles = (int64_t)g.ntk * ((NC + g.TJ - 1) / g.TJ);
    g.nblk = (int)(g.ntiles < 2048 ? g.ntiles : 2048);
    const int n0 = (H / 2 + 1 + 255) / 256;
    g.nblk0 = n0 < 64 ? n0 : 64;
    return g;
}

template <int OP>
hipError_t launch_ew(const LatArgs& a, cplx* cols, int NC, int h1, int h2, const EwGeo& g, double* partials,
                     hipStream_t st) {
    const int npart = g.nblk + g.nblk0;
    hipLaunchKernelGGL(anyn_ew_kernel<OP>, dim3(g.nblk, (unsigned)a.batch), dim3(256), 0, st, a, cols, NC, h1, h2, g.TK, g.TJ,
                       g.ntk, g.ntiles, partials, npart);
    A_TRY(hipGetLastError());
    hipLaunchKernelGGL(anyn_ew_col0_kernel<OP>, dim3(g.nblk0, (unsigned)a.batch), dim3(256), 0, st, a, cols, NC, h1, h2,
                       partials, npart, g.nblk);
    return hipGetLastError();
}

} // namespace

bool gpt_anyn_supported(int H, int W) {
    return H >= 1 && W >= 2 && (int64_t)H * W <= 2147483647LL && axis_ok(H) && axis_ok(W);
}

// An axis with a large prime factor costs the LDS Stockham routine of the generic path O(n p) per line; Bluestein here
// costs O(n log n): such shapes come to this path even when their lines would fit a workgroup.
bool gpt_anyn_preferred(int H, int W) {
    return gpt_anyn_supported(H, W) && (largest_prime_factor(H) > A_SMOOTH || largest_prime_factor(W) > A_SMOOTH);
}

int64_t gpt_anyn_chunk(int H, int W) { // lattices per call of gpt_launch_anyn
    const int64_t NC = 1 + (W - 1) / 2, Hp = (H + 1) / 2;
    const int64_t per = Hp * W > NC * H ? Hp * W : NC * H;
    int64_t chunk = A_CHUNK_ELEMS / per;
    if (chunk < 1) chunk = 1;
    return chunk < 32768 ? chunk : 32768;
}

gpt_anyn* gpt_anyn_create() { return new gpt_anyn(); }

void gpt_anyn_destroy(gpt_anyn* s) {
    if (!s) return;
    for (auto& kv : s->tw) (void)hipFree(kv.second);
    for (auto& kv : s->axes) {
        if (kv.second.chirp) (void)hipFree(const_cast<cplx*>(kv.second.chirp));
        if (kv.second.bhat) (void)hipFree(const_cast<cplx*>(kv.second.bhat));
    }
    if (s->ws) (void)hipFree(s->ws);
    if (s->tmp) (void)hipFree(s->tmp);
    if (s->bscr) (void)hipFree(s->bscr);
    if (s->partials) (void)hipFree(s->partials);
    delete s;
}

hipError_t gpt_launch_anyn(gpt_anyn* s, int op, const LatArgs& a, hipStream_t st) {
    if (!s || !gpt_anyn_supported(a.H, a.W) || a.batch > gpt_anyn_chunk(a.H, a.W)) return hipErrorInvalidValue;
    if (a.batch <= 0) return hipSuccess;
    const int H = a.H, W = a.W, NC = 1 + (W - 1) / 2, Hp = (H + 1) / 2;
    const int64_t nb = a.batch;
    const Axis *aw = nullptr, *ah = nullptr;
    A_TRY(raise_lds(s));
    A_TRY(get_axis(s, W, &aw, st));
    A_TRY(get_axis(s, H, &ah, st));
    const int64_t per = (int64_t)Hp * W > (int64_t)NC * H ? (int64_t)Hp * W : (int64_t)NC * H;
    const EwGeo g = ew_geo(H, NC);
    A_TRY(need(s->ws, s->ws_elems, (size_t)(nb * H * NC), st));
    A_TRY(need(s->tmp, s->tmp_elems, (size_t)(nb * per), st));
    A_TRY(need(s->partials, s->partials_elems, (size_t)nb * (size_t)(g.nblk + g.nblk0) * GPT_PART, st));
    cplx *ws = s->ws, *tmp = s->tmp;
    const int64_t nline_elems = nb * Hp * W, nspec = nb * (int64_t)H * NC;
    const double n = (double)H * (double)W;
    if (op_has_fwd(op)) {
        const dim3 gl(blocks_for(nline_elems));
        if (op == OP_RFFT2) hipLaunchKernelGGL(anyn_load_kernel<0>, gl, dim3(256), 0, st, a, tmp, Hp, nline_elems);
        else if (op == OP_INV_VJP) hipLaunchKernelGGL(anyn_load_kernel<2>, gl, dim3(256), 0, st, a, tmp, Hp, nline_elems);
        else hipLaunchKernelGGL(anyn_load_kernel<1>, gl, dim3(256), 0, st, a, tmp, Hp, nline_elems);
        A_TRY(hipGetLastError());
        A_TRY(axis_run(s, *aw, tmp, nb * Hp, false, st));
        hipLaunchKernelGGL(anyn_split_kernel, dim3(blocks_for(nspec)), dim3(256), 0, st, tmp, ws, H, W, NC, Hp, aw->pn1,
                           aw->pn2, nspec);
        A_TRY(hipGetLastError());
        if (H > 1) {
            A_TRY(gpt_launch_transpose(ws, tmp, H, NC, nb, (int)sizeof(cplx), st));
            A_TRY(axis_run(s, *ah, tmp, nb * NC, false, st));
        }
    }
    cplx* cols = H > 1 ? tmp : ws; // (H == 1: [NC][1] is [1][NC])
    const int h1 = H > 1 ? ah->pn1 : 1, h2 = H > 1 ? ah->pn2 : 1;
    hipError_t e;
    switch (op) {
    case OP_RFFT2: e = launch_ew<OP_RFFT2>(a, cols, NC, h1, h2, g, s->partials, st); break;
    case OP_INV_RFFT2: e = launch_ew<OP_INV_RFFT2>(a, cols, NC, h1, h2, g, s->partials, st); break;
    case OP_GP_RFFT2: e = launch_ew<OP_GP_RFFT2>(a, cols, NC, h1, h2, g, s->partials, st); break;
    case OP_GP_INV_RFFT2: e = launch_ew<OP_GP_INV_RFFT2>(a, cols, NC, h1, h2, g, s->partials, st); break;
    case OP_LPDF: e = launch_ew<OP_LPDF>(a, cols, NC, h1, h2, g, s->partials, st); break;
    case OP_LPDF_VJP: e = launch_ew<OP_LPDF_VJP>(a, cols, NC, h1, h2, g, s->partials, st); break;
    case OP_INV_VJP: e = launch_ew<OP_INV_VJP>(a, cols, NC, h1, h2, g, s->partials, st); break;
    case OP_FWD_VJP: e = launch_ew<OP_FWD_VJP>(a, cols, NC, h1, h2, g, s->partials, st); break;
    case OP_LPDF_VJP_NOCOV: e = launch_ew<OP_LPDF_VJP_NOCOV>(a, cols, NC, h1, h2, g, s->partials, st); break;
    default: return hipErrorInvalidValue;
    }
    A_TRY(e);
    if (op != OP_RFFT2 && op != OP_INV_RFFT2) {
        const bool lp = (op == OP_LPDF || op_is_lpdf_vjp(op) || op == OP_INV_VJP); // (OP_INV_VJP: the z prior)
        A_TRY(gpt_launch_finalize(s->partials, g.nblk + g.nblk0, nb, 1, -0.5 * n * 1.8378770664093454836,
                                  lp ? a.out_lpdf : nullptr, a.min_eig, st));
    }
    if (op_has_inv(op)) {
        if (H > 1) {
            A_TRY(axis_run(s, *ah, tmp, nb * NC, true, st));
            A_TRY(gpt_launch_transpose(tmp, ws, NC, H, nb, (int)sizeof(cplx), st));
        }
        hipLaunchKernelGGL(anyn_merge_kernel, dim3(blocks_for(nline_elems)), dim3(256), 0, st, ws, tmp, H, W, NC, Hp,
                           aw->pn1, aw->pn2, nline_elems);
        A_TRY(hipGetLastError());
        A_TRY(axis_run(s, *aw, tmp, nb * Hp, true, st));
        double alpha = 1.0;
        int add_loc = 0;
        if (op == OP_INV_RFFT2) alpha = 1.0 / n;
        else if (op == OP_GP_INV_RFFT2) { alpha = 1.0 / n; add_loc = 1; }
        else if (op_is_lpdf_vjp(op)) alpha = -1.0 / n;
        hipLaunchKernelGGL(anyn_store_kernel, dim3(blocks_for(nline_elems)), dim3(256), 0, st, a, tmp, Hp, alpha, add_loc,
                           nline_elems);
        A_TRY(hipGetLastError());
    }
    return hipSuccess;
}
