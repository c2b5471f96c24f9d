// Lines longer than one workgroup can hold (32768-point rows / columns of BASELINE config 5): the four-step
// pieces that surround the single-workgroup line transforms.
//
//   line of N = R * NS points, decimation in frequency:
//       u_q[n]     = w_N^{n q} * sum_p x[n + NS p] w_R^{p q}           (radix-R step over elements NS apart)
//       X[R k + q] = FFT_NS(u_q)[k]                                    (R independent lines of NS points)
//   and the mirror image (decimation in time) for the inverse.
//
//   columns (spectrum slab [H][ld], H = R * HS): rowstep_kernel does the radix-R step in place over rows HS apart
//       (forward: u_q -> row q HS + r; inverse: rows q HS + r -> rows r + HS p); the sub-line column kernel of
//       gpt_pow2_sub.h works on the rows in between.
//   rows (two real rows of W = 8 * WS points as one complex line): longrow_fwd1 (radix-8 step, writes the eight
//       sub-lines into a scratch line), line_fft_kernel (gpt_sub_n.hip), longrow_split (two-for-one split and
//       packing into the half spectrum); longrow_merge, line_fft_kernel, longrow_inv3 on the way back.
//
// Correctness first: every step is a separate streaming kernel (three sweeps per row pass instead of one).
#include "gpt_pow2_impl.h"

namespace {

using pow2::Bfly;

constexpr int RW = 8; // radix of the step along a long row

template <int R, bool INV>
__global__ void __launch_bounds__(256) rowstep_kernel(cplx* __restrict__ spec, int64_t ld, int HS, int ncols,
                                                      const cplx* __restrict__ twH) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= ncols || r >= HS) return;
    cplx x[R];
    if (!INV) {
#pragma unroll
        for (int p = 0; p < R; ++p) x[p] = spec[(int64_t)(r + HS * p) * ld + j];
        Bfly<R, false>::run(x);
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const cplx w = twH[(int64_t)r * q]; // r q < H
            spec[(int64_t)(q * HS + r) * ld + j] = q ? cmul(x[q], w) : x[q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const cplx v = spec[(int64_t)(q * HS + r) * ld + j];
            x[q] = q ? cmulc(v, twH[(int64_t)r * q]) : v;
        }
        Bfly<R, true>::run(x);
#pragma unroll
        for (int p = 0; p < R; ++p) spec[(int64_t)(r + HS * p) * ld + j] = x[p];
    }
}

// rows 2 pair, 2 pair + 1 of y (- loc) -> tmp[pair][q][n], n < WS
__global__ void __launch_bounds__(256) longrow_fwd1(const double* __restrict__ y, const double* __restrict__ loc, int W,
                                                    cplx* __restrict__ tmp, const cplx* __restrict__ twW) {
    const int WS = W / RW;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int64_t pair = blockIdx.y;
    if (n >= WS) return;
    const double* r0 = y + 2 * pair * W;
    const double* r1 = r0 + W;
    cplx x[RW];
#pragma unroll
    for (int p = 0; p < RW; ++p) x[p] = make_double2(r0[n + WS * p], r1[n + WS * p]);
    if (loc) {
        const double* l0 = loc + 2 * pair * W;
        const double* l1 = l0 + W;
#pragma unroll
        for (int p = 0; p < RW; ++p) {
            x[p].x -= l0[n + WS * p];
            x[p].y -= l1[n + WS * p];
        }
    }
    Bfly<RW, false>::run(x);
    cplx* o = tmp + pair * W;
#pragma unroll
    for (int q = 0; q < RW; ++q) o[(int64_t)q * WS + n] = q ? cmul(x[q], twW[(int64_t)n * q]) : x[q];
}

// tmp[pair][j % 8][j / 8] = Z[j] -> packed half spectrum rows 2 pair, 2 pair + 1 (column 0 = DC + i Nyquist)
__global__ void __launch_bounds__(256) longrow_split(const cplx* __restrict__ tmp, int W, cplx* __restrict__ spec,
                                                     int64_t ld, int cshift, int blk) {
    const int WS = W / RW;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int64_t pair = blockIdx.y;
    if (j >= W / 2) return;
    const cplx* z = tmp + pair * W;
    auto at = [&](int jj) { return z[(int64_t)(jj % RW) * WS + jj / RW]; };
    cplx* s0 = spec + 2 * pair * ld;
    cplx* s1 = s0 + ld;
    const cplx a = at(j);
    // (blocked layout of the exchange: column j lives at (j >> cshift) * blk + (j & mask), LatArgs::spec_cshift)
    const unsigned jo = cshift ? ((unsigned)j >> cshift) * (unsigned)blk + ((unsigned)j & ((1u << cshift) - 1u)) : (unsigned)j;
    if (j == 0) {
        const cplx zn = at(W / 2);
        s0[0] = make_double2(a.x, zn.x);
        s1[0] = make_double2(a.y, zn.y);
    } else {
        const cplx c = at(W - j);
        s0[jo] = make_double2(0.5 * (a.x + c.x), 0.5 * (a.y - c.y));
        s1[jo] = make_double2(0.5 * (a.y + c.y), -0.5 * (a.x - c.x));
    }
}

// packed half spectrum rows 2 pair, 2 pair + 1 -> tmp[pair][j % 8][j / 8] = Z[j] (the full Hermitian-completed line).
// One thread = one frequency j: a wave reads 64 consecutive frequencies (the mirrored half descending) and writes eight
// 128-byte pieces, one per sub-line -- the access pattern of longrow_split, mirrored.
__global__ void __launch_bounds__(256) longrow_merge(const cplx* __restrict__ spec, int64_t ld, int W,
                                                     cplx* __restrict__ tmp, int cshift, int blk) {
    const int WS = W / RW;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int64_t pair = blockIdx.y;
    if (j >= W) return;
    const cplx* s0 = spec + 2 * pair * ld;
    const cplx* s1 = s0 + ld;
    cplx v;
    auto so = [&](int c) -> unsigned {
        return cshift ? ((unsigned)c >> cshift) * (unsigned)blk + ((unsigned)c & ((1u << cshift) - 1u)) : (unsigned)c;
    };
    if (j < W / 2) {
        const cplx A = s0[so(j)], B = s1[so(j)];
        v = j == 0 ? make_double2(A.x, B.x) : make_double2(A.x - B.y, A.y + B.x);
    } else {
        const int m = W - j; // 1 .. W/2
        const unsigned mo = so(m == W / 2 ? 0 : m);
        const cplx A = s0[mo], B = s1[mo];
        v = m == W / 2 ? make_double2(A.y, B.y) : make_double2(A.x + B.y, -A.y + B.x);
    }
    tmp[pair * W + (int64_t)(j % RW) * WS + j / RW] = v;
}

// tmp[pair][q][n] (inverse sub-line transforms done) -> rows 2 pair, 2 pair + 1 of out = alpha * value
__global__ void __launch_bounds__(256) longrow_inv3(const cplx* __restrict__ tmp, int W, double alpha,
                                                    double* __restrict__ out, const cplx* __restrict__ twW) {
    const int WS = W / RW;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int64_t pair = blockIdx.y;
    if (n >= WS) return;
    const cplx* z = tmp + pair * W;
    cplx x[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        const cplx v = z[(int64_t)q * WS + n];
        x[q] = q ? cmulc(v, twW[(int64_t)n * q]) : v;
    }
    Bfly<RW, true>::run(x);
    double* o0 = out + 2 * pair * W;
    double* o1 = o0 + W;
#pragma unroll
    for (int p = 0; p < RW; ++p) {
        o0[n + WS * p] = alpha * x[p].x;
        o1[n + WS * p] = alpha * x[p].y;
    }
}

template <bool INV>
hipError_t rowstep(int R, cplx* spec, int64_t ld, int HS, int ncols, const cplx* twH, hipStream_t st) {
    dim3 grid((unsigned)((ncols + 63) / 64), (unsigned)((HS + 3) / 4));
    switch (R) {
    case 2: hipLaunchKernelGGL((rowstep_kernel<2, INV>), grid, dim3(256), 0, st, spec, ld, HS, ncols, twH); break;
    case 4: hipLaunchKernelGGL((rowstep_kernel<4, INV>), grid, dim3(256), 0, st, spec, ld, HS, ncols, twH); break;
    case 8: hipLaunchKernelGGL((rowstep_kernel<8, INV>), grid, dim3(256), 0, st, spec, ld, HS, ncols, twH); break;
    case 16: hipLaunchKernelGGL((rowstep_kernel<16, INV>), grid, dim3(256), 0, st, spec, ld, HS, ncols, twH); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace

hipError_t gpt_long_rowstep(bool inverse, int R, cplx* spec, int64_t ld, int HS, int ncols, const cplx* twH,
                            hipStream_t st) {
    return inverse ? rowstep<true>(R, spec, ld, HS, ncols, twH, st) : rowstep<false>(R, spec, ld, HS, ncols, twH, st);
}

int gpt_long_row_radix() { return RW; }

// step 1 / 3 of the long-row passes (the line transforms in between are gpt_line_fft)
hipError_t gpt_long_row_fwd1(const double* y, const double* loc, int64_t rows, int W, cplx* tmp, const cplx* twW,
                             hipStream_t st) {
    dim3 grid((unsigned)((W / RW + 255) / 256), (unsigned)(rows / 2));
    hipLaunchKernelGGL(longrow_fwd1, grid, dim3(256), 0, st, y, loc, W, tmp, twW);
    return hipGetLastError();
}
hipError_t gpt_long_row_split(const cplx* tmp, int64_t rows, int W, cplx* spec, int64_t ld, int cshift, int blk,
                              hipStream_t st) {
    dim3 grid((unsigned)((W / 2 + 255) / 256), (unsigned)(rows / 2));
    hipLaunchKernelGGL(longrow_split, grid, dim3(256), 0, st, tmp, W, spec, ld, cshift, blk);
    return hipGetLastError();
}
hipError_t gpt_long_row_merge(const cplx* spec, int64_t ld, int64_t rows, int W, cplx* tmp, int cshift, int blk,
                              hipStream_t st) {
    dim3 grid((unsigned)((W + 255) / 256), (unsigned)(rows / 2));
    hipLaunchKernelGGL(longrow_merge, grid, dim3(256), 0, st, spec, ld, W, tmp, cshift, blk);
    return hipGetLastError();
}
hipError_t gpt_long_row_inv3(const cplx* tmp, int64_t rows, int W, double alpha, double* out, const cplx* twW,
                             hipStream_t st) {
    dim3 grid((unsigned)((W / RW + 255) / 256), (unsigned)(rows / 2));
    hipLaunchKernelGGL(longrow_inv3, grid, dim3(256), 0, st, tmp, W, alpha, out, twW);
    return hipGetLastError();
}
