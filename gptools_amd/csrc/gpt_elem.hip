// Elementwise and reduction kernels of the Fourier-GP path: scales, packing, Jacobian log-determinant,
// spectral kernels and their reverse-mode reductions. All are HBM-streaming kernels: one pass over the
// H x Wf (or H x W) array, grid-stride, coalesced along the width, wave-shuffle + LDS reductions and an
// atomic-free second stage (finalize) so results are bitwise reproducible.
#include "gpt_common.h"
#include "gpt_host.h"

namespace {

constexpr double PI = 3.14159265358979323846;

__device__ __forceinline__ int entry_kind(int H, int W, int i, int j) {
    // 2 = complex pair, 1 = real point, 0 = never read (fft2.stan:55-74, 143-173)
    const bool edge = (j == 0) || ((W % 2 == 0) && (j == W / 2));
    if (!edge) return 2;
    if (i == 0 || 2 * i == H) return 1;
    return (i <= (H - 1) / 2) ? 2 : 0;
}

__global__ void __launch_bounds__(256) scale_kernel(ElemArgs e, double* partials) {
    const int64_t b = blockIdx.y;
    const int Wf = e.W / 2 + 1;
    const int64_t tot = (int64_t)e.H * Wf;
    const double n = (double)e.H * (double)e.W;
    const double* c = e.in + b * e.in_bs;
    double* o = e.out + b * tot;
    double mn = INFINITY;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < tot; w += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(w / Wf), j = (int)(w - (int64_t)i * Wf);
        double r = 0.5 * n * c[w];
        mn = fmin(mn, r);
        // the x2 adjustments of fft2.stan:26-38 / fft1.stan:25-28 sit at the (up to four) real points
        const bool edge = (j == 0) || ((e.W % 2 == 0) && (j == e.W / 2));
        if (edge && (i == 0 || 2 * i == e.H)) r *= 2.0;
        o[w] = sqrt(r);
    }
    double s[1] = {0.0};
    block_reduce_store<1>(s, mn, partials, b);
}

__global__ void __launch_bounds__(256) ladj_kernel(ElemArgs e, double* partials) {
    const int64_t b = blockIdx.y;
    const int Wf = e.W / 2 + 1;
    const int64_t tot = (int64_t)e.H * Wf;
    const double n = (double)e.H * (double)e.W;
    const double* c = e.in + b * e.in_bs;
    double mn = INFINITY;
    double s[1] = {0.0};
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < tot; w += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(w / Wf), j = (int)(w - (int64_t)i * Wf);
        const double r = 0.5 * n * c[w];
        mn = fmin(mn, r);
        const int kind = entry_kind(e.H, e.W, i, j);
        // -log(s) per whitened entry: pair -> 2 * (-1/2 log r); real point -> -1/2 log(2 r)
        if (kind == 2) s[0] -= log(r);
        else if (kind == 1) s[0] -= 0.5 * log(2.0 * r);
    }
    block_reduce_store<1>(s, mn, partials, b);
}

__global__ void __launch_bounds__(256) ladj_vjp_kernel(ElemArgs e) {
    const int64_t b = blockIdx.y;
    const int Wf = e.W / 2 + 1;
    const int64_t tot = (int64_t)e.H * Wf;
    const double* c = e.in + b * e.in_bs;
    double* o = e.out + b * tot;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < tot; w += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(w / Wf), j = (int)(w - (int64_t)i * Wf);
        o[w] = -0.5 * (double)entry_kind(e.H, e.W, i, j) / c[w];
    }
}

__global__ void __launch_bounds__(256) unpack_kernel(ElemArgs e) {
    const int64_t b = blockIdx.y;
    ZLayout L;
    L.H = e.H; L.W = e.W; L.Wf = e.W / 2 + 1; L.Hf = e.H / 2 + 1; L.wc = (e.W - 1) / 2; L.hc = (e.H - 1) / 2;
    L.dim1 = e.dim1;
    const int64_t tot = (int64_t)e.H * L.Wf;
    const cplx* z = e.cin + b * e.cin_bs;
    double* o = e.out + b * (int64_t)e.H * e.W;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < tot; w += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(w / L.Wf), j = (int)(w - (int64_t)i * L.Wf);
        const int kind = entry_kind(e.H, e.W, i, j);
        const cplx v = z[w];
        const bool edge = (j == 0) || ((e.W % 2 == 0) && (j == e.W / 2));
        if (!edge) {
            o[(int64_t)i * e.W + L.re_col(j)] = v.x;
            o[(int64_t)i * e.W + L.im_col(j)] = v.y;
        } else if (kind != 0) {
            const int jz = L.edge_col(j);
            o[(int64_t)i * e.W + jz] = v.x;
            if (kind == 2) o[(int64_t)(L.Hf + i - 1) * e.W + jz] = v.y;
        }
    }
}

__global__ void __launch_bounds__(256) pack_kernel(ElemArgs e) {
    const int64_t b = blockIdx.y;
    ZLayout L;
    L.H = e.H; L.W = e.W; L.Wf = e.W / 2 + 1; L.Hf = e.H / 2 + 1; L.wc = (e.W - 1) / 2; L.hc = (e.H - 1) / 2;
    L.dim1 = e.dim1;
    const int64_t tot = (int64_t)e.H * L.Wf;
    const double* z = e.in + b * e.in_bs;
    cplx* o = e.cout + b * tot;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < tot; w += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(w / L.Wf), j = (int)(w - (int64_t)i * L.Wf);
        const bool edge = (j == 0) || ((e.W % 2 == 0) && (j == e.W / 2));
        cplx v;
        if (!edge) {
            v = make_double2(z[(int64_t)i * e.W + L.re_col(j)], z[(int64_t)i * e.W + L.im_col(j)]);
        } else {
            // expand_rfft(gp_pack_rfft(column)) along the height (fft2.stan:91,97)
            const int jz = L.edge_col(j);
            const int k = (i <= e.H / 2) ? i : e.H - i;
            const bool real_pt = (k == 0) || (2 * k == e.H);
            v.x = z[(int64_t)k * e.W + jz];
            v.y = real_pt ? 0.0 : z[(int64_t)(L.Hf + k - 1) * e.W + jz];
            if (i != k) v.y = -v.y;
        }
        o[w] = v;
    }
}

__global__ void __launch_bounds__(256) expand_kernel(ElemArgs e) {
    // expand_rfft (util.stan:271-281): 1-D, n = e.W
    const int64_t b = blockIdx.y;
    const int n = e.W, nrfft = n / 2 + 1;
    const cplx* y = e.cin + b * e.cin_bs;
    cplx* o = e.cout + b * (int64_t)n;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        cplx v;
        if (k < nrfft) v = y[k];
        else v = cconj(y[n - k]);
        o[k] = v;
    }
}

// Spectral kernels. par[b] = (sigma, l1, l2, p1, p2) in 2-D, (sigma, l, p) in 1-D.
template <bool MATERN>
__device__ __forceinline__ double spec_value(const ElemArgs& e, const double* par, int i, int j) {
    if (e.dim1) {
        const double sigma = par[0], l = par[1], p = par[2];
        const double n = (double)e.W;
        if (!MATERN) { // fft1.stan:154-158
            const double t = PI * (double)j * l / p;
            return n * sigma * sigma * l / p * sqrt(2.0 * PI) * exp(-2.0 * t * t);
        }
        // fft1.stan:170-176
        const double t = PI * l / p * (double)j;
        return sigma * sigma * n * sqrt(2.0 * PI / e.nu) * e.gamma_ratio *
               pow(1.0 + 2.0 / e.nu * t * t, -(e.nu + 0.5)) * l / p;
    }
    const double sigma = par[0], l1 = par[1], l2 = par[2], p1 = par[3], p2 = par[4];
    const int ki = (i <= e.H / 2) ? i : e.H - i; // rows mirrored (fft2.stan:240-242, expand_rfft in :212)
    if (!MATERN) { // fft2.stan:206-213: outer product of two 1-D spectra
        const double t1 = PI * (double)ki * l1 / p1, t2 = PI * (double)j * l2 / p2;
        const double m = (double)e.H * sigma * sigma * l1 / p1 * sqrt(2.0 * PI) * exp(-2.0 * t1 * t1);
        const double nn = (double)e.W * l2 / p2 * sqrt(2.0 * PI) * exp(-2.0 * t2 * t2);
        return m * nn;
    }
    // fft2.stan:228-247, ndim = 2
    const double a1 = (double)ki * l1 / p1, a2 = (double)j * l2 / p2;
    const double arg = 1.0 + 2.0 / e.nu * PI * PI * (a1 * a1 + a2 * a2);
    return sigma * sigma * (double)e.H * (double)e.W * 4.0 * (PI / (2.0 * e.nu)) * e.gamma_ratio *
           pow(arg, -(e.nu + 1.0)) * (l1 / p1) * (l2 / p2);
}

template <bool MATERN>
__global__ void __launch_bounds__(256) spec_kernel(ElemArgs e) {
    const int64_t b = blockIdx.y;
    const int Wf = e.W / 2 + 1;
    const int64_t tot = (int64_t)e.H * Wf;
    const double* par = e.par + b * (e.dim1 ? 3 : 5);
    double* o = e.out + b * tot;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < tot; w += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(w / Wf), j = (int)(w - (int64_t)i * Wf);
        o[w] = spec_value<MATERN>(e, par, i, j) + e.nugget;
    }
}

// Reverse mode of the spectral kernels: covbar (e.in) -> partial sums of
// (sigma_bar, l1_bar, l2_bar, p1_bar, p2_bar) [2-D] or (sigma_bar, l_bar, p_bar) [1-D].
template <bool MATERN>
__global__ void __launch_bounds__(256) spec_vjp_kernel(ElemArgs e, double* partials) {
    const int64_t b = blockIdx.y;
    const int Wf = e.W / 2 + 1;
    const int64_t tot = (int64_t)e.H * Wf;
    const double* par = e.par + b * (e.dim1 ? 3 : 5);
    const double* cb = e.in + b * e.in_bs;
    double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < tot; w += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(w / Wf), j = (int)(w - (int64_t)i * Wf);
        const double wgt = cb[w] * spec_value<MATERN>(e, par, i, j);
        if (e.dim1) {
            const double sigma = par[0], l = par[1], p = par[2];
            const double a = (double)j * l / p;
            double f; // d log c / d log l = 1 - f
            if (!MATERN) f = 4.0 * PI * PI * a * a;
            else {
                const double arg = 1.0 + 2.0 / e.nu * PI * PI * a * a;
                f = (e.nu + 0.5) * (4.0 * PI * PI / e.nu) / arg * a * a;
            }
            s[0] += wgt * 2.0 / sigma;
            s[1] += wgt * (1.0 - f) / l;
            s[2] += wgt * (f - 1.0) / p;
        } else {
            const double sigma = par[0], l1 = par[1], l2 = par[2], p1 = par[3], p2 = par[4];
            const int ki = (i <= e.H / 2) ? i : e.H - i;
            const double a1 = (double)ki * l1 / p1, a2 = (double)j * l2 / p2;
            double f1, f2;
            if (!MATERN) {
                f1 = 4.0 * PI * PI * a1 * a1;
                f2 = 4.0 * PI * PI * a2 * a2;
            } else {
                const double arg = 1.0 + 2.0 / e.nu * PI * PI * (a1 * a1 + a2 * a2);
                const double g = (e.nu + 1.0) * (4.0 * PI * PI / e.nu) / arg;
                f1 = g * a1 * a1;
                f2 = g * a2 * a2;
            }
            s[0] += wgt * 2.0 / sigma;
            s[1] += wgt * (1.0 - f1) / l1;
            s[2] += wgt * (1.0 - f2) / l2;
            s[3] += wgt * (f1 - 1.0) / p1;
            s[4] += wgt * (f2 - 1.0) / p2;
        }
    }
    block_reduce_store<5>(s, INFINITY, partials, b);
}

__global__ void __launch_bounds__(256) finalize_kernel(const double* partials, int nblk, int K, double add,
                                                       double* out_sum, double* out_min) {
    const int64_t b = blockIdx.x;
    const double* p = partials + b * (int64_t)nblk * GPT_PART;
    double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double mn = INFINITY;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] += p[(int64_t)i * GPT_PART + k];
        mn = fmin(mn, p[(int64_t)i * GPT_PART + 5]);
    }
    __shared__ double red[6 * 4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] = wave_sum(s[k]);
    mn = wave_min(mn);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) red[k * 4 + w] = s[k];
        red[5 * 4 + w] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int k = 0; k < K && out_sum; ++k) {
            double t = 0.0;
            for (int i = 0; i < nw; ++i) t += red[k * 4 + i];
            out_sum[b * K + k] = t + (k == 0 ? add : 0.0);
        }
        if (out_min) {
            double m = INFINITY;
            for (int i = 0; i < nw; ++i) m = fmin(m, red[5 * 4 + i]);
            out_min[b] = m;
        }
    }
}

// Tiled transpose through LDS: in is [batch][R][C], out is [batch][C][R]; T = 8- or 16-byte elements.
template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(const T* in, T* out, int R, int C, int nbx) {
    __shared__ T tile[32][33];
    const int64_t b = blockIdx.y;
    const T* ib = in + b * (int64_t)R * C;
    T* ob = out + b * (int64_t)R * C;
    // (tiles numbered along grid.x alone: a matrix of millions of rows has more row tiles than grid.y allows)
    const int by = blockIdx.x / nbx, bx = blockIdx.x - by * nbx;
    const int c0 = bx * 32, r0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        if (r < R && c < C) tile[k][tx] = ib[(int64_t)r * C + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < R && c < C) ob[(int64_t)c * R + r] = tile[tx][k];
    }
}

// up to GPT_XPOSE_MAX transposes of different shapes in one launch (the small calls of the host-pointer entry points:
// a launch costs more than the transposes themselves)
template <typename T> __global__ void __launch_bounds__(256) transpose_multi_kernel(XposeBatch b) {
    __shared__ T tile[32][33];
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.first_tile[k + 1]) ++k;
    const int t = blockIdx.x - b.first_tile[k];
    const int R = b.R[k], C = b.C[k], nbx = (C + 31) / 32;
    const T* ib = static_cast<const T*>(b.in[k]);
    T* ob = static_cast<T*>(b.out[k]);
    const int by = t / nbx, bx = t - by * nbx;
    const int c0 = bx * 32, r0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int q = ty; q < 32; q += 8) {
        const int r = r0 + q, c = c0 + tx;
        if (r < R && c < C) tile[q][tx] = ib[(int64_t)r * C + c];
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int c = c0 + q, r = r0 + tx;
        if (r < R && c < C) ob[(int64_t)c * R + r] = tile[tx][q];
    }
}

} // namespace

hipError_t gpt_launch_transpose_multi(XposeBatch b, int esz, hipStream_t st) {
    if (b.n <= 0) return hipSuccess;
    if (b.n > GPT_XPOSE_MAX) return hipErrorInvalidValue;
    int tiles = 0;
    for (int k = 0; k < b.n; ++k) {
        b.first_tile[k] = tiles;
        tiles += ((b.C[k] + 31) / 32) * ((b.R[k] + 31) / 32);
    }
    if (esz == 8) hipLaunchKernelGGL(transpose_multi_kernel<double>, dim3((unsigned)tiles), dim3(256), 0, st, b);
    else if (esz == 16) hipLaunchKernelGGL(transpose_multi_kernel<cplx>, dim3((unsigned)tiles), dim3(256), 0, st, b);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

int gpt_elem_blocks(int H, int W) {
    const int64_t tot = (int64_t)H * (W / 2 + 1);
    int64_t nb = (tot + 2047) / 2048;
    if (nb < 1) nb = 1;
    if (nb > 512) nb = 512;
    return (int)nb;
}

hipError_t gpt_launch_elem(int op, const ElemArgs& e, double* partials, hipStream_t st) {
    if (e.batch <= 0) return hipSuccess;
    if (e.batch > 65535) return hipErrorInvalidValue; // callers chunk the batch
    dim3 grid((unsigned)gpt_elem_blocks(e.H, e.W), (unsigned)e.batch), block(256);
    switch (op) {
    case EL_SCALE: hipLaunchKernelGGL(scale_kernel, grid, block, 0, st, e, partials); break;
    case EL_LADJ: hipLaunchKernelGGL(ladj_kernel, grid, block, 0, st, e, partials); break;
    case EL_LADJ_VJP: hipLaunchKernelGGL(ladj_vjp_kernel, grid, block, 0, st, e); break;
    case EL_UNPACK: hipLaunchKernelGGL(unpack_kernel, grid, block, 0, st, e); break;
    case EL_PACK: hipLaunchKernelGGL(pack_kernel, grid, block, 0, st, e); break;
    case EL_EXPAND: hipLaunchKernelGGL(expand_kernel, grid, block, 0, st, e); break;
    case EL_EXPQUAD: hipLaunchKernelGGL(spec_kernel<false>, grid, block, 0, st, e); break;
    case EL_MATERN: hipLaunchKernelGGL(spec_kernel<true>, grid, block, 0, st, e); break;
    case EL_EXPQUAD_VJP: hipLaunchKernelGGL(spec_vjp_kernel<false>, grid, block, 0, st, e, partials); break;
    case EL_MATERN_VJP: hipLaunchKernelGGL(spec_vjp_kernel<true>, grid, block, 0, st, e, partials); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t gpt_launch_finalize(const double* partials, int nblk, int64_t batch, int K, double add,
                               double* out_sum, double* out_min, hipStream_t st) {
    if (batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)batch), dim3(256), 0, st, partials, nblk, K, add, out_sum,
                       out_min);
    return hipGetLastError();
}

hipError_t gpt_launch_transpose(const void* in, void* out, int R, int C, int64_t batch, int esz, hipStream_t st) {
    if (batch <= 0 || R <= 0 || C <= 0) return hipSuccess;
    if (batch > 65535) return hipErrorInvalidValue; // callers chunk the batch
    const int nbx = (C + 31) / 32;
    const int64_t tiles = (int64_t)nbx * ((R + 31) / 32);
    if (tiles > 2147483647LL) return hipErrorInvalidValue;
    dim3 grid((unsigned)tiles, (unsigned)batch), block(256);
    if (esz == 8)
        hipLaunchKernelGGL(transpose_kernel<double>, grid, block, 0, st, (const double*)in, (double*)out, R, C, nbx);
    else if (esz == 16)
        hipLaunchKernelGGL(transpose_kernel<cplx>, grid, block, 0, st, (const cplx*)in, (cplx*)out, R, C, nbx);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
