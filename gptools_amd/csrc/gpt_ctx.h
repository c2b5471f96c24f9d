// The context behind the C ABI (include/gptools_hip.h) and the host-side helpers its translation units share:
// gpt_api.hip (context, dispatch, device-pointer entry points), gpt_api_noalloc.hip, gpt_api_slab.hip, gpt_api_obs.hip,
// gpt_api_host.hip (host-pointer staging, what the Stan header binds).
#pragma once
#include "../../include/gptools_hip.h"
#include "gpt_common.h"
#include "gpt_host.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

struct HostPipe;

struct gpt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::map<int, cplx*> tw; // twiddle tables w_n^m, keyed by n
    cplx* ws = nullptr;      // packed-spectrum workspace of the three-pass paths
    size_t ws_bytes = 0;
    double* partials = nullptr;
    size_t partials_bytes = 0;
    double* sink = nullptr; // write-only dump for gradients the caller did not ask for
    size_t sink_bytes = 0;
    double* kscr = nullptr; // kernel-fused log density: per-lattice sums, or the spectrum + d/dcov of the fallback
    size_t kscr_bytes = 0;
    double* oscr = nullptr; // observation models: f, d/df and per-lattice sums when the caller does not keep them
    size_t oscr_bytes = 0;
    char* arena = nullptr; // device staging arena of the host-pointer entry points
    size_t arena_bytes = 0;
    double* pinned = nullptr; // small pinned scratch for scalar read-backs
    std::string err;
    double last_min_eig = NAN;
    bool profiling = false; // record events around the passes of the three-pass paths
    struct ProfRec {
        hipEvent_t t0, t1;
        int pass; // 0 = A, 1 = B, 3 = C (slot 2 is the separate finalize of older builds: always 0 now)
    };
    std::vector<ProfRec> prof;
    int total_cus = 0;
    HostPipe* pipe = nullptr; // pinned staging ring of the host-pointer entry points (created on first large transfer)
    unsigned char* scratch = nullptr; // this device's copy of the built-in scratch array (gpt_gp_rfft2_lpdf_vjp_noalloc)
    gpt_anyn* anyn = nullptr; // plans, tables and scratch of the any-length path (gpt_anyn.hip), created on first use
    // Pinned, device-mapped bounce area of the host-pointer entry points for transfers too small for the ring (Stager):
    // small calls let the kernels read and write it directly, mid-size ones stage their asynchronous copies in it.
    char* bounce = nullptr;     // host view
    char* bounce_dev = nullptr; // device view of the same memory
    size_t bounce_bytes = 0;
    // Completion of a small host call, published by its one kernel (gpt_small.hip) instead of waited for through the
    // runtime: a sequence number in the last 256 bytes of the bounce allocation, stored by the kernel with a system-scope
    // release after its last result, spun on by the host (launch + hipStreamSynchronize of an empty kernel: 11.1 us on
    // this box; kernel-stored flag + spin: 6.0 -- profiles/r03_m_sync_latency.txt). done_armed: the entry point asks for
    // it (only those whose one launch is the last thing on the stream); done_published: run_lattice() did hand the flag
    // to a one-lattice small launch.
    unsigned long long* done_host = nullptr;
    unsigned long long* done_dev = nullptr;
    unsigned long long done_seq = 0;
    bool done_armed = false, done_published = false;
};

// Built-in scratch of the allocation-free entry point (gpt_api_noalloc.hip): bytes per GPU
#define GPT_NOALLOC_SCRATCH_BYTES ((size_t)40 << 20)

#define HIP_TRY(c, expr)                                                                                       \
    do {                                                                                                       \
        hipError_t e__ = (expr);                                                                               \
        if (e__ != hipSuccess) return gptd::hip_fail((c), e__, #expr);                                         \
    } while (0)

// Host-side helpers shared by the translation units of the C ABI (gpt_api*.hip); defined in gpt_api.hip.
namespace gptd {
int fail(gpt_ctx* c, int code, const std::string& msg);
int hip_fail(gpt_ctx* c, hipError_t e, const char* what);
template <typename T> int ensure(gpt_ctx* c, T*& ptr, size_t& have, size_t need) {
    if (need <= have) return GPT_OK;
    if (ptr) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipFree(ptr));
        ptr = nullptr;
        have = 0;
    }
    void* p = nullptr;
    HIP_TRY(c, hipMalloc(&p, need));
    ptr = reinterpret_cast<T*>(p);
    have = need;
    return GPT_OK;
}
int get_twiddles(gpt_ctx* c, int n, const cplx** out);
int plan_radices(gpt_ctx* c, int n, LinePlan& p, bool wide = false);
int make_plan(gpt_ctx* c, int n, LinePlan& p, bool wide = false);
int path_for_shape(int64_t H, int64_t W);
int check_shape(gpt_ctx* c, int64_t H, int64_t W, int64_t batch, int dim1);
int run_units(gpt_ctx* c, int op, const LatArgs& a, int path, int64_t chunk, cplx* ws, double* partials, bool may_profile);
int run_lattice(gpt_ctx* c, int op, LatArgs a);
LatArgs base_args(int64_t H, int64_t W, int64_t batch, int dim1);
int run_elem(gpt_ctx* c, int op, ElemArgs e, bool reduces, int K, double add, double* out_sum, double* out_min);
ElemArgs elem_args(int64_t H, int64_t W, int64_t batch, int dim1);
int inv_rfft2_vjp(gpt_ctx* c, const double* z, int64_t z_bs, const double* cov, int64_t cov_bs, const double* fbar,
                  int64_t fbar_bs, int64_t H, int64_t W, int64_t batch, int dim1, double* zbar, double* covbar, double* prior_lp);
int device_cus(gpt_ctx* c); // CU count of the context's device (cached)
// this device's address of the built-in scratch array (gpt_api_noalloc.hip: the array and its hipGetSymbolAddress
// live in one source file); called by gpt_create with the device current
int noalloc_resolve(gpt_ctx* c);
} // namespace gptd
