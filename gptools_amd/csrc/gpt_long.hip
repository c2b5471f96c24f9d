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
//   rows of 32768 points no longer come here: one real row is one complex line of 16384 points, a single kernel each
//       way (rowreal_fwd_kernel / rowreal_inv_kernel, gpt_pow2_impl.h; round 2 took three sweeps per row pass).
//
// The column side is still three sweeps (radix-R step, sub-line column kernel, inverse radix-R step).
#include "gpt_pow2_impl.h"

namespace {

using pow2::Bfly;

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
