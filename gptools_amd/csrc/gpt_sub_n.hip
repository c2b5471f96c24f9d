// Sub-line kernels (gpt_pow2_sub.h), one translation unit per line length: compiled with -DGPT_N=<power of two>.
// GPT_N is the SUB-LINE length of colsub (and the line length of line_fft).
#include "gpt_pow2_sub.h"
#include <cstdlib>

#ifndef GPT_N
#error "compile with -DGPT_N=<power of two>"
#endif

namespace pow2 {

namespace {
template <typename K> hipError_t raise_lds_sub(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds);
}
// two register tiles per thread at 256 VGPRs; 256-thread workgroups (two share a CU) while a sub-line fits them
#ifndef GPT_SUB_V
#define GPT_SUB_V 2
#endif
constexpr int SUB_V = GPT_SUB_V;
#ifndef GPT_SUB_T2048
#define GPT_SUB_T2048 256
#endif
constexpr int SUB_THREADS = GPT_N >= 4096 ? 512 : (GPT_N == 2048 ? GPT_SUB_T2048 : 256);
#ifndef GPT_SUB_WGS_PER_CU
#define GPT_SUB_WGS_PER_CU (SUB_THREADS == 256 ? 2 : 1)
#endif
constexpr int SUB_WGS_PER_CU = GPT_SUB_WGS_PER_CU; // workgroups the register / LDS budget lets share a CU
using GS = GeoS<GPT_N, SUB_V, SUB_THREADS>;
} // namespace

template <>
hipError_t LaunchSub<GPT_N>::line_fft(bool inverse, cplx* data, int64_t nlines, const cplx* tw, hipStream_t st) {
    constexpr int N = GPT_N;
    const size_t lds = (size_t)Geo<N>::ROW_LINES * N * sizeof(double);
    hipError_t e = inverse ? raise_lds_sub(line_fft_kernel<N, true>, lds) : raise_lds_sub(line_fft_kernel<N, false>, lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)((nlines + Geo<N>::ROW_LINES - 1) / Geo<N>::ROW_LINES));
    if (inverse) hipLaunchKernelGGL((line_fft_kernel<N, true>), grid, dim3(Geo<N>::ROW_THREADS), lds, st, data, nlines, tw);
    else hipLaunchKernelGGL((line_fft_kernel<N, false>), grid, dim3(Geo<N>::ROW_THREADS), lds, st, data, nlines, tw);
    return hipGetLastError();
}

#if GPT_N <= 4096
// Ops whose elementwise stage reads or writes the real H x W layout of the whitened coefficients z: a workgroup's
// pieces of those rows are TC doubles wide. Sub-lines of 2048 points give TC = 4 (32 bytes) with 256 threads; these
// ops run 512-thread workgroups there (TC = 8, one workgroup per CU): +4..8 % on the non-centred model density,
// while the ops that only touch the complex workspace and cov are faster with two small workgroups per CU.
constexpr bool op_z_layout(int op) {
    return op == OP_GP_RFFT2 || op == OP_GP_INV_RFFT2 || op == OP_INV_VJP || op == OP_FWD_VJP;
}
template <int OP, int THREADS, bool NT = false>
static hipError_t colsub_g(const LatArgs& a, int R, cplx* ws, int wstride, const cplx* tws, double* partials, int nwg,
                           double* edge, int edge_flags, hipStream_t st) {
    constexpr int HS = GPT_N;
    using G = GeoS<HS, SUB_V, THREADS>;
    const size_t lds = G::lds_bytes();
    hipError_t e = raise_lds_sub(colsub_kernel<HS, SUB_V, THREADS, OP, NT>, lds);
    if (e != hipSuccess) return e;
    const int ncols = a.ncols > 0 ? a.ncols : a.W / 2;
    if (ncols % G::TC != 0) return hipErrorInvalidValue;
    const int ntq = ncols / G::TC;
    // nwg is the number of partial slots per lattice (the same for every geometry); a geometry that fits one
    // workgroup per CU launches half as many
    int nact = nwg;
    if (THREADS != SUB_THREADS && SUB_WGS_PER_CU == 2 && nwg >= 16) nact = (nwg / 2) & ~7;
    if (nact < R / 2 + 1) nact = nwg;
    dim3 grid((unsigned)nact, (unsigned)a.batch);
    hipLaunchKernelGGL((colsub_kernel<HS, SUB_V, THREADS, OP, NT>), grid, dim3(THREADS), lds, st, a, R, ws, wstride, tws,
                       partials, ntq, edge, edge_flags, nwg);
    return hipGetLastError();
}

template <int OP>
static hipError_t colsub_t(const LatArgs& a, int R, cplx* ws, int wstride, const cplx* tws, double* partials, int nwg,
                           double* edge, int edge_flags, hipStream_t st) {
#if GPT_N == 2048
    // GPT_SUB_GEOM (tuning knob, same results): 1 = 256 threads (two workgroups per CU), 2 = the same with non-temporal
    // workspace accesses, 3 = 512 threads with ordinary accesses, 4 = 512 threads non-temporal; 0 / unset = the rule below
    static const int geom = [] {
        const char* e = getenv("GPT_SUB_GEOM");
        return e ? atoi(e) : 0;
    }();
    if (geom == 1) return colsub_g<OP, SUB_THREADS>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    if (geom == 2) return colsub_g<OP, SUB_THREADS, true>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    if (geom == 3) return colsub_g<OP, 512>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    if (geom == 4) return colsub_g<OP, 512, true>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    // 512-thread workgroups (8 columns: whole 128-byte lines of the workspace, 64-byte pieces of cov) for the ops on
    // the real z layout, and for every op when the launch's workspace cannot be cache-resident (units of many
    // lattices, gpt_api.hip): 4096^2 column pass 124.5 -> 118 us there, but 123 -> 130 us on a warm workspace.
    const bool cold = (size_t)a.batch * (size_t)a.H * (size_t)wstride * sizeof(cplx) > ((size_t)256 << 20);
    // (the kernel-fused ops load no cov and store no d/dcov: nothing of theirs comes in narrow pieces, and their
    // arithmetic is what bounds them -- two small workgroups per CU overlap it with the workspace traffic:
    // 4096^2 x 16 column pass 126 -> 105 us, profiles/r03_a_geom_ab.jsonl)
    if (cold && !op_is_kvjp(OP)) // ... and streams the workspace past the caches
        return colsub_g<OP, 512, true>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    if (op_z_layout(OP)) return colsub_g<OP, 512>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
#endif
    return colsub_g<OP, SUB_THREADS>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
}

template <>
hipError_t LaunchSub<GPT_N>::colsub(int op, const LatArgs& a, int R, cplx* ws, int wstride, const cplx* tws,
                                    double* partials, int nwg, double* edge, int edge_flags, hipStream_t st) {
    switch (op) {
    case OP_RFFT2: return colsub_t<OP_RFFT2>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    case OP_INV_RFFT2: return colsub_t<OP_INV_RFFT2>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    case OP_GP_RFFT2: return colsub_t<OP_GP_RFFT2>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    case OP_GP_INV_RFFT2: return colsub_t<OP_GP_INV_RFFT2>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    case OP_LPDF: return colsub_t<OP_LPDF>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    case OP_LPDF_VJP: return colsub_t<OP_LPDF_VJP>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    case OP_INV_VJP: return colsub_t<OP_INV_VJP>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    case OP_FWD_VJP: return colsub_t<OP_FWD_VJP>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    case OP_LPDF_VJP_NOCOV:
        return colsub_t<OP_LPDF_VJP_NOCOV>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    case OP_LPDF_KVJP: return colsub_t<OP_LPDF_KVJP>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    case OP_LPDF_KVJP_EQ:
        return colsub_t<OP_LPDF_KVJP_EQ>(a, R, ws, wstride, tws, partials, nwg, edge, edge_flags, st);
    default: return hipErrorInvalidValue;
    }
}
template <> int LaunchSub<GPT_N>::col_tile() { return GS::TC; }
template <> int LaunchSub<GPT_N>::col_blocks(int R, int ncols, int cus) {
    const int ntiles = R * (ncols / GS::TC);
    int cap = cus * SUB_WGS_PER_CU;
    if (cap >= 8) cap &= ~7;
    if (cap < R / 2 + 1) cap = R / 2 + 1; // the packed column 0 needs its R/2 + 1 workgroups
    return ntiles < cap ? (ntiles > R / 2 + 1 ? ntiles : R / 2 + 1) : cap;
}
#else
template <>
hipError_t LaunchSub<GPT_N>::colsub(int, const LatArgs&, int, cplx*, int, const cplx*, double*, int, double*, int,
                                    hipStream_t) {
    return hipErrorInvalidValue;
}
template <> int LaunchSub<GPT_N>::col_tile() { return 0; }
template <> int LaunchSub<GPT_N>::col_blocks(int, int, int) { return 0; }
#endif

} // namespace pow2

#if GPT_TRACE
// debug builds only: copy the column kernel's phase timestamps to the host (scripts/trace_sub.py)
extern "C" int gpt_debug_sub_trace(unsigned long long* dst, size_t n) {
    if (n > sizeof(g_col_trace) / sizeof(g_col_trace[0])) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_col_trace), n * sizeof(unsigned long long));
}
#endif
