// Cross-chain batching: the chains of one sampler process (cmdstan threads under
// STAN_THREADS, or any other threads) each evaluate gp_rfft(2)_lpdf once per gradient evaluation. Alone, a chain
// pays a launch, a synchronisation and three small PCIe copies per call; here every chain hands its lattice to
// its SLOT of a shared batcher, the call blocks until every active slot has arrived, and the last arrival runs
// ONE fused launch over all of them (the batched device entry points of gpt_api.hip) and wakes the others.
// No counterpart in the reference, where every chain evaluates on its own CPU core.
#include "../../include/gptools_hip.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

// batched row-major <-> column-major transposes (gpt_elem.hip)
hipError_t gpt_launch_transpose(const void* in, void* out, int R, int C, int64_t batch, int esz, hipStream_t st);

struct gpt_batcher {
    gpt_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    int device = 0, nslots = 0, layout = GPT_ROW_MAJOR, dim1 = 0;
    int64_t H = 0, W = 0, hw = 0, hwf = 0;
    // slot-major staging: pinned host and device copies of (y, loc, cov) and (lpdf, min_eig, grad_y, grad_cov)
    double *h_y = nullptr, *h_loc = nullptr, *h_cov = nullptr, *h_lp = nullptr, *h_mn = nullptr, *h_gy = nullptr,
           *h_gc = nullptr;
    double *d_y = nullptr, *d_loc = nullptr, *d_cov = nullptr, *d_lp = nullptr, *d_mn = nullptr, *d_gy = nullptr,
           *d_gc = nullptr, *d_t = nullptr; // d_t: transposed twin (column-major callers)
    std::mutex m;
    std::condition_variable cv;
    std::vector<char> active;
    int nactive = 0, arrived = 0;
    bool want_gy = false, want_gc = false;
    unsigned long long generation = 0, launches = 0;
    int batch_rc = GPT_OK;
    std::string err;
    double timeout_s = 0.0; // longest wait for the other chains (0: for ever), gpt_batcher_set_timeout
};

namespace {

bool ok(gpt_batcher* b, hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    b->err = std::string(what) + ": " + hipGetErrorString(e);
    b->batch_rc = e == hipErrorOutOfMemory ? GPT_ERR_ALLOC : GPT_ERR_HIP;
    return false;
}

// upload `count` slot-major matrices of R x C doubles; column-major callers get them transposed on the device
bool upload(gpt_batcher* b, double* dst, const double* src, int64_t R, int64_t C) {
    const size_t bytes = sizeof(double) * (size_t)b->nslots * R * C;
    const bool tr = b->layout == GPT_COL_MAJOR && R > 1 && C > 1;
    if (!ok(b, hipMemcpyAsync(tr ? b->d_t : dst, src, bytes, hipMemcpyHostToDevice, b->stream), "upload")) return false;
    if (tr && !ok(b, gpt_launch_transpose(b->d_t, dst, (int)C, (int)R, b->nslots, 8, b->stream), "transpose")) return false;
    return true;
}
bool download(gpt_batcher* b, double* dst, double* src, int64_t R, int64_t C) {
    const size_t bytes = sizeof(double) * (size_t)b->nslots * R * C;
    const bool tr = b->layout == GPT_COL_MAJOR && R > 1 && C > 1;
    if (tr && !ok(b, gpt_launch_transpose(src, b->d_t, (int)R, (int)C, b->nslots, 8, b->stream), "transpose")) return false;
    return ok(b, hipMemcpyAsync(dst, tr ? b->d_t : src, bytes, hipMemcpyDeviceToHost, b->stream), "download");
}

// one fused evaluation of every slot; called with the lock held by the last arrival
void run_batch(gpt_batcher* b) {
    b->batch_rc = GPT_OK;
    b->err.clear();
    const int64_t Wf = b->W / 2 + 1;
    do {
        if (!ok(b, hipSetDevice(b->device), "hipSetDevice")) break;
        if (!upload(b, b->d_y, b->h_y, b->H, b->W) || !upload(b, b->d_loc, b->h_loc, b->H, b->W) ||
            !upload(b, b->d_cov, b->h_cov, b->H, Wf))
            break;
        int rc;
        if (b->want_gy)
            rc = gpt_gp_rfft2_lpdf_vjp(b->ctx, b->d_y, b->hw, b->d_loc, b->hw, b->d_cov, b->hwf, b->H, b->W, b->nslots,
                                       b->dim1, b->d_lp, b->d_gy, b->want_gc ? b->d_gc : nullptr, b->d_mn);
        else
            rc = gpt_gp_rfft2_lpdf(b->ctx, b->d_y, b->hw, b->d_loc, b->hw, b->d_cov, b->hwf, b->H, b->W, b->nslots,
                                   b->dim1, b->d_lp, b->d_mn);
        if (rc != GPT_OK) {
            b->batch_rc = rc;
            b->err = gpt_last_error(b->ctx);
            break;
        }
        if (!download(b, b->h_lp, b->d_lp, 1, 1) || !download(b, b->h_mn, b->d_mn, 1, 1)) break;
        if (b->want_gy && !download(b, b->h_gy, b->d_gy, b->H, b->W)) break;
        if (b->want_gy && b->want_gc && !download(b, b->h_gc, b->d_gc, b->H, Wf)) break;
    } while (false);
    // on every exit, failed rounds included: the pinned staging areas are rewritten by the chains as soon as they
    // wake, so no copy of this round may still be in flight
    {
        const hipError_t e = hipStreamSynchronize(b->stream);
        if (b->batch_rc == GPT_OK) ok(b, e, "hipStreamSynchronize");
    }
    b->launches++;
    b->want_gy = b->want_gc = false;
    b->arrived = 0;
    b->generation++;
    b->cv.notify_all();
}

} // namespace

extern "C" {

int gpt_batcher_create(int device, int nslots, int layout, int64_t height, int64_t width, int dim1,
                       gpt_batcher** out) {
    if (!out) return GPT_ERR_SHAPE;
    *out = nullptr;
    if (nslots < 1 || nslots > 65536 || height < 1 || width < 1 || (dim1 && height != 1) ||
        height * width > 2147483647LL || gpt_path_for_shape(height, width) < 0)
        return GPT_ERR_SHAPE;
    if (hipSetDevice(device) != hipSuccess) return GPT_ERR_HIP;
    gpt_batcher* b = new gpt_batcher();
    b->device = device;
    b->nslots = nslots;
    b->layout = layout;
    b->dim1 = dim1 ? 1 : 0;
    b->H = height;
    b->W = width;
    b->hw = height * width;
    b->hwf = height * (width / 2 + 1);
    b->active.assign((size_t)nslots, 1);
    b->nactive = nslots;
    bool good = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) == hipSuccess &&
                gpt_create(device, b->stream, &b->ctx) == GPT_OK;
    const size_t nhw = sizeof(double) * (size_t)nslots * b->hw, nhwf = sizeof(double) * (size_t)nslots * b->hwf,
                 ns = sizeof(double) * (size_t)nslots;
    auto host = [&](double** p, size_t bytes) { good = good && hipHostMalloc((void**)p, bytes, hipHostMallocDefault) == hipSuccess; };
    auto dev = [&](double** p, size_t bytes) { good = good && hipMalloc((void**)p, bytes) == hipSuccess; };
    host(&b->h_y, nhw); host(&b->h_loc, nhw); host(&b->h_cov, nhwf); host(&b->h_lp, ns); host(&b->h_mn, ns);
    host(&b->h_gy, nhw); host(&b->h_gc, nhwf);
    dev(&b->d_y, nhw); dev(&b->d_loc, nhw); dev(&b->d_cov, nhwf); dev(&b->d_lp, ns); dev(&b->d_mn, ns);
    dev(&b->d_gy, nhw); dev(&b->d_gc, nhwf); dev(&b->d_t, nhw);
    if (!good) {
        gpt_batcher_destroy(b);
        return GPT_ERR_ALLOC;
    }
    // slots that never call (or have left) are evaluated on benign data
    memset(b->h_y, 0, nhw);
    memset(b->h_loc, 0, nhw);
    for (size_t i = 0; i < (size_t)nslots * b->hwf; ++i) b->h_cov[i] = 1.0;
    *out = b;
    return GPT_OK;
}

int gpt_batcher_destroy(gpt_batcher* b) {
    if (!b) return GPT_OK;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->ctx) gpt_destroy(b->ctx);
    for (double* p : {b->h_y, b->h_loc, b->h_cov, b->h_lp, b->h_mn, b->h_gy, b->h_gc})
        if (p) (void)hipHostFree(p);
    for (double* p : {b->d_y, b->d_loc, b->d_cov, b->d_lp, b->d_mn, b->d_gy, b->d_gc, b->d_t})
        if (p) (void)hipFree(p);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return GPT_OK;
}

int gpt_batcher_lpdf_host(gpt_batcher* b, int slot, const double* y, const double* loc, const double* cov,
                          double* lpdf, double* grad_y, double* grad_cov) {
    if (!b || slot < 0 || slot >= b->nslots || !y || !cov || !lpdf || (grad_cov && !grad_y)) return GPT_ERR_SHAPE;
    const size_t hw = (size_t)b->hw, hwf = (size_t)b->hwf;
    {
        // a slot that has left holds benign data that a running batch may be uploading: it must not be written
        std::unique_lock<std::mutex> lk(b->m);
        if (!b->active[(size_t)slot]) return GPT_ERR_SHAPE;
    }
    // The staging area of an ACTIVE slot is private to its chain until the chain arrives: a batch runs only once
    // every active slot has arrived, this one included.
    memcpy(b->h_y + slot * hw, y, sizeof(double) * hw);
    if (loc) memcpy(b->h_loc + slot * hw, loc, sizeof(double) * hw);
    else memset(b->h_loc + slot * hw, 0, sizeof(double) * hw);
    memcpy(b->h_cov + slot * hwf, cov, sizeof(double) * hwf);
    int rc;
    {
        std::unique_lock<std::mutex> lk(b->m);
        b->want_gy = b->want_gy || grad_y != nullptr;
        b->want_gc = b->want_gc || grad_cov != nullptr;
        const unsigned long long gen = b->generation;
        if (++b->arrived >= b->nactive) {
            run_batch(b);
        } else if (b->timeout_s > 0.0) {
            const auto limit = std::chrono::duration<double>(b->timeout_s);
            if (!b->cv.wait_for(lk, limit, [&] { return b->generation != gen; })) {
                // a chain ended without gpt_batcher_leave: this call gives up (its slot stays active and may call again)
                b->arrived--;
                b->err = "timed out waiting for the other chains: " + std::to_string(b->nactive - b->arrived - 1) +
                         " active slot(s) did not call (a chain that ends must gpt_batcher_leave its slot)";
                return GPT_ERR_TIMEOUT;
            }
        } else {
            b->cv.wait(lk, [&] { return b->generation != gen; });
        }
        rc = b->batch_rc;
    }
    if (rc != GPT_OK) return rc;
    *lpdf = b->h_lp[slot];
    if (grad_y) memcpy(grad_y, b->h_gy + slot * hw, sizeof(double) * hw);
    if (grad_cov) memcpy(grad_cov, b->h_gc + slot * hwf, sizeof(double) * hwf);
    // the reference's reject (fft1.stan:19-21, fft2.stan:21-23) concerns this chain alone
    return b->h_mn[slot] < 0 ? GPT_ERR_NOT_PD : GPT_OK;
}

int gpt_batcher_leave(gpt_batcher* b, int slot) {
    if (!b || slot < 0 || slot >= b->nslots) return GPT_ERR_SHAPE;
    std::unique_lock<std::mutex> lk(b->m);
    if (!b->active[(size_t)slot]) return GPT_OK;
    b->active[(size_t)slot] = 0;
    b->nactive--;
    // benign data from now on
    memset(b->h_y + (size_t)slot * b->hw, 0, sizeof(double) * (size_t)b->hw);
    memset(b->h_loc + (size_t)slot * b->hw, 0, sizeof(double) * (size_t)b->hw);
    for (int64_t i = 0; i < b->hwf; ++i) b->h_cov[(size_t)slot * b->hwf + i] = 1.0;
    if (b->nactive > 0 && b->arrived >= b->nactive) run_batch(b); // the others were waiting for this chain only
    return GPT_OK;
}

int gpt_batcher_set_timeout(gpt_batcher* b, double seconds) {
    if (!b || !(seconds >= 0.0)) return GPT_ERR_SHAPE;
    std::unique_lock<std::mutex> lk(b->m);
    b->timeout_s = seconds;
    return GPT_OK;
}

double gpt_batcher_min_eig(gpt_batcher* b, int slot) { return (b && slot >= 0 && slot < b->nslots) ? b->h_mn[slot] : 0.0; }
long long gpt_batcher_launches(gpt_batcher* b) { return b ? (long long)b->launches : 0; }
const char* gpt_batcher_last_error(gpt_batcher* b) { return b ? b->err.c_str() : "null batcher"; }

} // extern "C"
