// One translation unit per line length: compiled with -DGPT_N=<256|512|1024|2048|4096|8192>.
#include "gpt_pow2_impl.h"

#ifndef GPT_N
#error "compile with -DGPT_N=<power of two>"
#endif

namespace pow2 {

namespace {
// split exchange: TC regions of (H + skew) doubles; the packed column 0 needs H complex values once
template <int H> constexpr size_t col_lds_bytes() {
    const size_t a = (size_t)Geo<H>::TC * (H + Geo<H>::SKEW) * sizeof(double), b = (size_t)H * sizeof(cplx);
    return a > b ? a : b;
}
template <typename K> hipError_t raise_lds(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds);
}
} // namespace

namespace {
template <bool SUB2, bool NTS>
hipError_t row_fwd_g(const LatArgs& a, int in_mode, cplx* ws, int wstride, double* edge, hipStream_t st) {
    constexpr int W = GPT_N;
    const size_t lds = (size_t)Geo<W>::ROW_LINES * W * sizeof(double);
    hipError_t e = raise_lds(row_fwd_kernel<W, SUB2, NTS>, lds);
    if (e != hipSuccess) return e;
    const int npairs = a.H / 2;
    dim3 grid((npairs + Geo<W>::ROW_LINES - 1) / Geo<W>::ROW_LINES, (unsigned)a.batch);
    hipLaunchKernelGGL((row_fwd_kernel<W, SUB2, NTS>), grid, dim3(Geo<W>::ROW_THREADS), lds, st, a, in_mode, ws, wstride,
                       edge);
    return hipGetLastError();
}
} // namespace

template <>
hipError_t Launch<GPT_N>::row_fwd(const LatArgs& a, int in_mode, cplx* ws, int wstride, double* edge, bool sub2,
                                  hipStream_t st) {
    // a workspace beyond the Infinity Cache is streamed past it (see rows_st)
    const bool cold = (size_t)a.batch * (size_t)a.H * (size_t)wstride * sizeof(cplx) > ((size_t)256 << 20);
    if (cold) return sub2 ? row_fwd_g<true, true>(a, in_mode, ws, wstride, edge, st)
                          : row_fwd_g<false, true>(a, in_mode, ws, wstride, edge, st);
    return sub2 ? row_fwd_g<true, false>(a, in_mode, ws, wstride, edge, st)
                : row_fwd_g<false, false>(a, in_mode, ws, wstride, edge, st);
}

template <>
hipError_t Launch<GPT_N>::row_inv(const LatArgs& a, double alpha, int add_loc, const cplx* ws, int wstride,
                                  const FinArgs& fin, const double* edge, bool sub2, hipStream_t st) {
    constexpr int W = GPT_N;
    const size_t lds = (size_t)Geo<W>::ROW_LINES * W * sizeof(double);
    hipError_t e = sub2 ? raise_lds(row_inv_kernel<W, true>, lds) : raise_lds(row_inv_kernel<W, false>, lds);
    if (e != hipSuccess) return e;
    const int npairs = a.H / 2;
    dim3 grid((npairs + Geo<W>::ROW_LINES - 1) / Geo<W>::ROW_LINES, (unsigned)a.batch);
    if (sub2)
        hipLaunchKernelGGL((row_inv_kernel<W, true>), grid, dim3(Geo<W>::ROW_THREADS), lds, st, a, alpha, add_loc, ws,
                           wstride, fin, edge);
    else
        hipLaunchKernelGGL((row_inv_kernel<W, false>), grid, dim3(Geo<W>::ROW_THREADS), lds, st, a, alpha, add_loc, ws,
                           wstride, fin, edge);
    return hipGetLastError();
}

template <>
hipError_t Launch<GPT_N>::rowreal_fwd(const LatArgs& a, cplx* spec, int ld, const cplx* twN, hipStream_t st) {
    constexpr int N = GPT_N;
    if (a.W != 2 * N || a.batch != 1 || !twN) return hipErrorInvalidValue;
    const size_t lds = (size_t)GeoR<N>::LINES * N * sizeof(double);
    hipError_t e = raise_lds(rowreal_fwd_kernel<N>, lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)((a.H + GeoR<N>::LINES - 1) / GeoR<N>::LINES));
    hipLaunchKernelGGL((rowreal_fwd_kernel<N>), grid, dim3(GeoR<N>::THREADS), lds, st, a, spec, ld, twN);
    return hipGetLastError();
}
template <>
hipError_t Launch<GPT_N>::rowreal_inv(const LatArgs& a, double alpha, const cplx* spec, int ld, const cplx* twN,
                                      hipStream_t st) {
    constexpr int N = GPT_N;
    if (a.W != 2 * N || a.batch != 1 || !twN) return hipErrorInvalidValue;
    const size_t lds = (size_t)GeoR<N>::LINES * N * sizeof(double);
    hipError_t e = raise_lds(rowreal_inv_kernel<N>, lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)((a.H + GeoR<N>::LINES - 1) / GeoR<N>::LINES));
    hipLaunchKernelGGL((rowreal_inv_kernel<N>), grid, dim3(GeoR<N>::THREADS), lds, st, a, alpha, spec, ld, twN);
    return hipGetLastError();
}

#if GPT_N <= 8192
template <>
hipError_t Launch<GPT_N>::row_obs(const LatArgs& a, double alpha, cplx* ws, int wstride, const FinArgs& fin, double* edge,
                                  const ObsRowArgs& o, bool sub2, hipStream_t st) {
    constexpr int W = GPT_N;
    const size_t lds = (size_t)Geo<W>::ROW_LINES * W * sizeof(double);
    hipError_t e = sub2 ? raise_lds(row_obs_kernel<W, true>, lds) : raise_lds(row_obs_kernel<W, false>, lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)row_obs_blocks(a.H), (unsigned)a.batch);
    if (sub2)
        hipLaunchKernelGGL((row_obs_kernel<W, true>), grid, dim3(Geo<W>::ROW_THREADS), lds, st, a, alpha, ws, wstride, fin,
                           edge, o);
    else
        hipLaunchKernelGGL((row_obs_kernel<W, false>), grid, dim3(Geo<W>::ROW_THREADS), lds, st, a, alpha, ws, wstride, fin,
                           edge, o);
    return hipGetLastError();
}
#else
template <>
hipError_t Launch<GPT_N>::row_obs(const LatArgs&, double, cplx*, int, const FinArgs&, double*, const ObsRowArgs&, bool,
                                  hipStream_t) {
    return hipErrorInvalidValue;
}
#endif

#if GPT_N <= 8192
template <int OP>
static hipError_t col_t(const LatArgs& a, cplx* ws, int wstride, double* partials, int nwg, double* edge,
                        int edge_flags, hipStream_t st) {
    constexpr int H = GPT_N;
    const size_t lds = col_lds_bytes<H>();
    hipError_t e = raise_lds(col_kernel<H, OP>, lds);
    if (e != hipSuccess) return e;
    const int ntiles = (a.ncols > 0 ? a.ncols : a.W / 2) / Geo<H>::TC;
    dim3 grid((unsigned)nwg, (unsigned)a.batch);
    hipLaunchKernelGGL((col_kernel<H, OP>), grid, dim3(Geo<H>::COL_THREADS), lds, st, a, ws, wstride, partials,
                       ntiles, edge, edge_flags);
    return hipGetLastError();
}

template <>
hipError_t Launch<GPT_N>::col(int op, const LatArgs& a, cplx* ws, int wstride, double* partials, int nwg,
                              double* edge, int edge_flags, hipStream_t st) {
    switch (op) {
    case OP_RFFT2: return col_t<OP_RFFT2>(a, ws, wstride, partials, nwg, edge, edge_flags, st);
    case OP_INV_RFFT2: return col_t<OP_INV_RFFT2>(a, ws, wstride, partials, nwg, edge, edge_flags, st);
    case OP_GP_RFFT2: return col_t<OP_GP_RFFT2>(a, ws, wstride, partials, nwg, edge, edge_flags, st);
    case OP_GP_INV_RFFT2: return col_t<OP_GP_INV_RFFT2>(a, ws, wstride, partials, nwg, edge, edge_flags, st);
    case OP_LPDF: return col_t<OP_LPDF>(a, ws, wstride, partials, nwg, edge, edge_flags, st);
    case OP_LPDF_VJP: return col_t<OP_LPDF_VJP>(a, ws, wstride, partials, nwg, edge, edge_flags, st);
    case OP_INV_VJP: return col_t<OP_INV_VJP>(a, ws, wstride, partials, nwg, edge, edge_flags, st);
    case OP_FWD_VJP: return col_t<OP_FWD_VJP>(a, ws, wstride, partials, nwg, edge, edge_flags, st);
    case OP_LPDF_VJP_NOCOV: return col_t<OP_LPDF_VJP_NOCOV>(a, ws, wstride, partials, nwg, edge, edge_flags, st);
    default: return hipErrorInvalidValue;
    }
}
#else
// no column kernel for this line length (gpt_pow2_supported)
template <>
hipError_t Launch<GPT_N>::col(int, const LatArgs&, cplx*, int, double*, int, double*, int, hipStream_t) {
    return hipErrorInvalidValue;
}
#endif

} // namespace pow2

#if GPT_TRACE
// debug builds only: copy the column kernel's phase timestamps to the host (scripts/trace_col.py)
extern "C" int gpt_debug_col_trace(unsigned long long* dst, size_t n) {
    if (n > sizeof(g_col_trace) / sizeof(g_col_trace[0])) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_col_trace), n * sizeof(unsigned long long));
}
#endif
