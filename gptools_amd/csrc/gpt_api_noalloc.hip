// The allocation-free entry point gpt_gp_rfft2_lpdf_vjp_noalloc (include/gptools_hip.h): the built-in scratch array,
// its per-device address (resolved here, in the array's own source file, when a context is created) and the call.
#include "gpt_ctx.h"
using namespace gptd;

// Built-in scratch of the allocation-free entry point (gpt_gp_rfft2_lpdf_vjp_noalloc): twiddle tables, per-workgroup
// partials and the packed-spectrum workspace of the lattices of one unit. Every GPU the library runs on holds its own
// copy at its own address until the process ends. In host code the name below is only a handle for
// hipGetSymbolAddress (gpt_create); kernels and copies get the per-device address kept in gpt_ctx::scratch.
static __device__ __align__(256) unsigned char scratch[GPT_NOALLOC_SCRATCH_BYTES];

// w_n^m = exp(-2 pi i m / n), m = 0 .. n-1, generated on the device (sincospi is exact at the octants)
static __global__ void noalloc_twiddle_kernel(cplx* __restrict__ tw, int n) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    double sn, cs;
    sincospi(2.0 * (double)m / (double)n, &sn, &cs);
    tw[m] = make_double2(cs, -sn);
}

namespace gptd {
int noalloc_resolve(gpt_ctx* c) {
    void* sp = nullptr;
    if (hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)) != hipSuccess || !sp) return GPT_ERR_HIP;
    c->scratch = reinterpret_cast<unsigned char*>(sp);
    return GPT_OK;
}
} // namespace gptd

extern "C" {

// gpt_gp_rfft2_lpdf_vjp for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid it):
// same arguments, same results, but twiddles, partials and workspace live in the built-in scratch array, so the call
// makes no hipMalloc / hipFree and uses none of the context's cached buffers. Every call on a GPU shares that GPU's
// array: calls that overlap in time on one GPU race (serialise them on one stream). Nothing in the array has to start
// at zero: each call regenerates its tables and every kernel writes what it later reads.
int gpt_gp_rfft2_lpdf_vjp_noalloc(gpt_ctx* c, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs,
                                  const double* cov, int64_t cov_bs, int64_t H, int64_t W, int64_t batch, int dim1,
                                  double* lpdf, double* grad_y, double* grad_cov, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if (!y || !cov || !lpdf || !grad_y) return fail(c, GPT_ERR_SHAPE, "y, cov, lpdf and grad_y must not be null");
    if (!c->scratch) return fail(c, GPT_ERR_HIP, "the built-in scratch array was not resolved on this device");
    HIP_TRY(c, hipSetDevice(c->device));
    if (batch == 0) return GPT_OK;
    const int op = grad_cov ? OP_LPDF_VJP : OP_LPDF_VJP_NOCOV;
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = y; a.x_bs = y_bs;
    a.loc = loc; a.loc_bs = loc_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.out = grad_y;
    a.out_cov = grad_cov;
    a.out_lpdf = lpdf;
    a.min_eig = min_eig;
    const int path = path_for_shape(H, W);
    if (path < 0) {
        char buf[160];
        snprintf(buf, sizeof buf, "no kernel for a %d x %d lattice in this build", a.H, a.W);
        return fail(c, GPT_ERR_UNSUPPORTED, buf);
    }
    // carve the scratch array: tables, then partials, then the workspace
    size_t off = 0;
    bool fits = true;
    auto take = [&](size_t bytes) -> unsigned char* {
        unsigned char* p = c->scratch + off;
        off += (bytes + 255) & ~(size_t)255;
        if (off > GPT_NOALLOC_SCRATCH_BYTES) fits = false;
        return p;
    };
    auto table = [&](int n, const cplx** out) -> int {
        cplx* t = reinterpret_cast<cplx*>(take(sizeof(cplx) * (size_t)n));
        if (!fits) return GPT_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(noalloc_twiddle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, t, n);
        *out = t;
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? GPT_OK : hip_fail(c, e, "twiddle table");
    };
    const char* too_big = "this lattice needs more than the 40 MiB of built-in scratch of the allocation-free entry point "
                          "(three-pass lattices up to 2048 x 2048); use gpt_gp_rfft2_lpdf_vjp";
    if (path == 3) // (its plans, tables and Bluestein scratch are allocated per shape)
        return fail(c, GPT_ERR_UNSUPPORTED,
                    "the allocation-free entry point does not take any-length shapes (long, prime or non-power-of-two "
                    "axes beyond 4096 points: their line plans and scratch are allocated); use gpt_gp_rfft2_lpdf_vjp");
    if ((rc = plan_radices(c, a.W, a.pw)) != GPT_OK || (rc = plan_radices(c, a.H, a.ph)) != GPT_OK) return rc;
    if ((rc = table(a.W, &a.pw.tw)) != GPT_OK) return fits ? rc : fail(c, GPT_ERR_UNSUPPORTED, too_big);
    if (a.H == a.W) a.ph.tw = a.pw.tw;
    else if ((rc = table(a.H, &a.ph.tw)) != GPT_OK) return fits ? rc : fail(c, GPT_ERR_UNSUPPORTED, too_big);
    if (path == 0) {
        HIP_TRY(c, gpt_launch_small(op, a, c->stream));
        return GPT_OK;
    }
    if (path == 1) {
        const int R = gpt_pow2_sub_radix(op, a.H, a.W);
        if (R && (rc = table(a.H / R, &a.tw_sub)) != GPT_OK) return fits ? rc : fail(c, GPT_ERR_UNSUPPORTED, too_big);
    }
    const size_t per = path == 1 ? gpt_pow2_workspace_bytes(a.H, a.W) : gpt_generic_workspace_bytes(a.H, a.W);
    const int nblk = path == 1 ? gpt_pow2_col_blocks(a.H, a.W, c->total_cus > 0 ? c->total_cus : 256)
                               : gpt_generic_col_blocks(a.H, a.W);
    if (c->total_cus == 0) c->total_cus = 256;
    const size_t per_lattice = ((per + 255) & ~(size_t)255) + sizeof(double) * GPT_PART * (size_t)nblk;
    const size_t left = off + 512 <= GPT_NOALLOC_SCRATCH_BYTES ? GPT_NOALLOC_SCRATCH_BYTES - off - 512 : 0;
    int64_t chunk = (int64_t)(left / per_lattice);
    if (chunk < 1) return fail(c, GPT_ERR_UNSUPPORTED, too_big);
    if (chunk > batch) chunk = batch;
    double* partials = reinterpret_cast<double*>(take(sizeof(double) * GPT_PART * (size_t)nblk * (size_t)chunk));
    cplx* ws = reinterpret_cast<cplx*>(take(per * (size_t)chunk));
    if (!fits) return fail(c, GPT_ERR_UNSUPPORTED, too_big);
    return run_units(c, op, a, path, chunk, ws, partials, false);
}

} // extern "C"
