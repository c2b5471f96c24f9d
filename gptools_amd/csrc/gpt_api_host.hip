// Host-pointer entry points of the C ABI (gpt_*_host: what the Stan user header binds): staging of the caller's
// column-major (Stan / Eigen) or row-major host arrays, the device path, the reference's `reject`.
#include <chrono>
#include "gpt_ctx.h"
#include "gpt_hostpipe.h"

#include <functional>
using namespace gptd;

// Slice `id` of `nt` of an n-byte copy: ceil(n / nt) rounded up to a page, the last slices shortened or empty, so that
// the slices tile [0, n) exactly for every n and nt (round 2 rounded floor(n / nt) up and lost the last n % nt bytes
// when that was already a multiple of 4096 and nt >= 9). Exported for its unit test (tests/test_host_copy_slices.py).
extern "C" void gpt_host_copy_slice(size_t n, int nt, int id, size_t* lo, size_t* len) {
    if (nt < 1) nt = 1;
    const size_t slice = (((n + (size_t)nt - 1) / (size_t)nt) + 4095) & ~(size_t)4095;
    const size_t a = (size_t)id * slice;
    *lo = a < n ? a : n;
    *len = a < n ? (a + slice <= n ? slice : n - a) : 0;
}

// ===================================================================================================

// ===================================================================================================
// Host-pointer entry points: stage through a device arena, transposing Stan's column-major matrices.
namespace {

struct Stager {
    gpt_ctx* c;
    size_t off = 0;
    int layout;
    int rc = GPT_OK;
    bool drained = false;
    // Every exit of a host entry point, early error returns included, leaves no copy from or to the caller's
    // (pageable) arrays in flight.
    ~Stager() {
        if (!drained) (void)hipStreamSynchronize(c->stream);
    }
    // Small calls (the docs models: 35 x 60, n = 100 ... 200; everything up to TINY_BYTES of arena): what costs there is
    // the number of operations on the stream and above all the copy <-> kernel hand-overs (~12 us each), not bytes. So
    // these calls make NO copies: the caller's arrays are memcpy'd into the context's pinned, device-mapped bounce area
    // and the kernels read them there (the transposes of column-major matrices, or the transform kernels themselves);
    // outputs and scalars are written by the kernels straight into the bounce area and memcpy'd out after the one
    // synchronisation of the call. Only kernels on the stream: 64 x 64 column-major 225 -> 94 (one copy each way) -> ~50 us.
    static constexpr size_t TINY_BYTES = (size_t)64 << 20; // = BOUNCE_BYTES: lattices up to 512 x 512
    bool tiny = false;
    bool gpu_xpose = false; // a matrix of this call changes layout in a kernel (too large for the CPU to do it in passing)
    Stager(gpt_ctx* ctx, int lay, size_t need) : c(ctx), layout(lay) {
        if (hipSetDevice(c->device) != hipSuccess) {
            rc = fail(c, GPT_ERR_HIP, "hipSetDevice failed");
            return;
        }
        rc = ensure(c, c->arena, c->arena_bytes, need);
        tiny = rc == GPT_OK && need <= TINY_BYTES && bounce_ready() && c->bounce_dev && need <= c->bounce_bytes;
        c->done_armed = c->done_published = false;
    }
    // For the entry points whose ONE lattice launch is the last operation of the call: ask that launch to publish its
    // completion (small calls only; run_lattice() takes it up when the one-workgroup kernel runs, finish() then spins on
    // the flag instead of going through hipStreamSynchronize). Call right before the library call.
    void arm() {
        static const bool off = [] { const char* e = getenv("GPT_HOST_FLAG_SYNC"); return e && e[0] == '0'; }();
        if (tiny && rc == GPT_OK && !off && !gpu_xpose) c->done_armed = true; // (transposes on the stream: not the last launch)
    }
    // true: the flag arrived; false: not within ~5 ms (a kernel that long is no small call: wait the ordinary way, which
    // also reports a fault)
    bool spin_done() {
        const unsigned long long want = c->done_seq;
        for (int round = 0; round < 5000; ++round) {
            const auto t0 = std::chrono::steady_clock::now();
            do {
                for (int i = 0; i < 64; ++i)
                    if (__atomic_load_n(c->done_host, __ATOMIC_ACQUIRE) == want) return true;
            } while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(1));
        }
        return false;
    }
    bool host_visible(const void* dev_ptr) const {
        const char* p = static_cast<const char*>(dev_ptr);
        return tiny && p >= c->bounce_dev && p < c->bounce_dev + c->bounce_bytes;
    }
    // the two views of the bounce area
    void* to_dev(char* host_ptr) { return c->bounce_dev + (host_ptr - c->bounce); }
    char* to_host(const void* dev_ptr) { return c->bounce + (static_cast<const char*>(dev_ptr) - c->bounce_dev); }
    bool is_matrix(int64_t R, int64_t C) const { return layout == GPT_COL_MAJOR && R > 1 && C > 1; }
    // transposes of a small call, gathered: inputs are launched together by go(), outputs by finish()
    XposeBatch xq{};
    int xq_esz = 0;
    void xflush() {
        if (xq.n == 0) return;
        const hipError_t e = gpt_launch_transpose_multi(xq, xq_esz, c->stream);
        if (e != hipSuccess && rc == GPT_OK) rc = hip_fail(c, e, "transpose");
        xq.n = 0;
    }
    void xpose(const void* in, void* out, int R, int C, int esz) {
        if (xq.n == GPT_XPOSE_MAX || (xq.n && xq_esz != esz)) xflush();
        xq_esz = esz;
        xq.in[xq.n] = in;
        xq.out[xq.n] = out;
        xq.R[xq.n] = R;
        xq.C[xq.n] = C;
        xq.n++;
    }
    // after the uploads and before the first kernel of the call: returns rc
    int go() {
        xflush();
        return rc;
    }
    // a scalar input next to device-side results (d from dev())
    void put(double* d, double v) {
        if (rc != GPT_OK) return;
        if (tiny) {
            *reinterpret_cast<double*>(to_host(d)) = v; // (the stream is idle between calls: nothing reads this yet)
            return;
        }
        c->pinned[1] = v; // (slot 0 is the min_eig read-back of finish())
        const hipError_t e = hipMemcpyAsync(d, c->pinned + 1, sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) rc = hip_fail(c, e, "scalar upload");
    }
    // large transfers go through the context's pinned ring (HostPipe), small ones through the runtime's pageable path
    bool piped(size_t bytes) {
        if (bytes < HostPipe::MIN_BYTES) return false;
        if (!c->pipe) c->pipe = new HostPipe();
        if (!c->pipe->ready() && c->pipe->init() != 0) {
            c->pipe->destroy();
            return false; // no pinned memory to be had: the plain path still works
        }
        return true;
    }
    // small transfers: a piece of the context's pinned bounce area (null: none to be had, the pageable path still works)
    static constexpr size_t BOUNCE_BYTES = 32 * HostPipe::MIN_BYTES; // 64 MiB
    size_t boff = 0;
    struct Pending {
        void* host;
        const char* src;
        size_t bytes;
        int R = 0, C = 0, esz = 0; // R > 0: src is an R x C row-major matrix, the caller wants it column-major
    };
    // Matrices of a small call up to this many elements change layout on the CPU, inside the copy into / out of the
    // bounce area that happens anyway (35 x 60 doubles: well under a microsecond each, from L1) -- a transpose kernel on
    // either side of the lattice kernel costs two more hand-overs on the stream, and keeps the lattice launch from
    // being the last one (arm()).
    static constexpr int64_t CPU_XPOSE_ELEMS = 4096;
    bool cpu_xpose(int64_t R, int64_t C) const { return tiny && is_matrix(R, C) && R * C <= CPU_XPOSE_ELEMS; }
    // in: R x C row-major -> out: C x R row-major, elements of esz = 8 or 16 bytes
    static void xpose_host(const void* in, void* out, int64_t R, int64_t C, int esz) {
        if (esz == 8) {
            const double* a = static_cast<const double*>(in);
            double* b = static_cast<double*>(out);
            for (int64_t c = 0; c < C; ++c)
                for (int64_t r = 0; r < R; ++r) b[c * R + r] = a[r * C + c];
        } else {
            const cplx* a = static_cast<const cplx*>(in);
            cplx* b = static_cast<cplx*>(out);
            for (int64_t c = 0; c < C; ++c)
                for (int64_t r = 0; r < R; ++r) b[c * R + r] = a[r * C + c];
        }
    }
    std::vector<Pending> pending; // downloads that sit in the bounce area until finish() has synchronised
    bool bounce_ready() {
        if (c->bounce) return true;
        void* p = nullptr;
        if (hipHostMalloc(&p, BOUNCE_BYTES, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return false;
        c->bounce = static_cast<char*>(p);
        c->bounce_bytes = BOUNCE_BYTES - 256; // (the tail holds the completion flag)
        void* d = nullptr;
        c->bounce_dev = hipHostGetDevicePointer(&d, p, 0) == hipSuccess ? static_cast<char*>(d) : nullptr;
        c->done_host = reinterpret_cast<unsigned long long*>(c->bounce + c->bounce_bytes);
        *c->done_host = 0;
        c->done_dev = c->bounce_dev ? reinterpret_cast<unsigned long long*>(c->bounce_dev + c->bounce_bytes) : nullptr;
        return true;
    }
    char* bounce(size_t bytes) {
        if (!bounce_ready()) return nullptr;
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (boff + need > c->bounce_bytes) {
            // (cannot happen for a small call: its arrays fit arena_need() <= the bounce area; the caller falls back or fails)
            if (tiny && rc == GPT_OK) rc = fail(c, GPT_ERR_ALLOC, "pinned bounce area exhausted");
            return tiny ? c->bounce : nullptr; // a valid address, never used once rc is set
        }
        char* p = c->bounce + boff;
        boff += need;
        return p;
    }
    hipError_t h2d(void* dev, const void* host, size_t bytes) {
        if (piped(bytes)) return c->pipe->up(dev, host, bytes, c->stream);
        if (char* b = bounce(bytes)) {
            memcpy(b, host, bytes);
            return hipMemcpyAsync(dev, b, bytes, hipMemcpyHostToDevice, c->stream);
        }
        return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream);
    }
    hipError_t d2h(void* host, const void* dev, size_t bytes) {
        if (piped(bytes)) return c->pipe->down(host, dev, bytes, c->stream);
        if (char* b = bounce(bytes)) {
            pending.push_back({host, b, bytes});
            return hipMemcpyAsync(b, dev, bytes, hipMemcpyDeviceToHost, c->stream);
        }
        return hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream);
    }
    void* bump(size_t bytes) {
        void* p = c->arena + off;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    }
    // upload an R x C matrix of esz-byte elements; returns a row-major device copy
    const void* up(const void* host, int64_t R, int64_t C, int esz) {
        if (rc != GPT_OK || !host) return nullptr;
        const size_t bytes = (size_t)R * C * esz;
        if (tiny) {
            char* b = bounce(bytes);
            if (cpu_xpose(R, C)) { // the caller's column-major R x C = a row-major C x R
                xpose_host(host, b, C, R, esz);
                return to_dev(b);
            }
            memcpy(b, host, bytes);
            if (!is_matrix(R, C)) return to_dev(b); // the kernels read it where it is
            gpu_xpose = true;
            void* d = bump(bytes);
            xpose(to_dev(b), d, (int)C, (int)R, esz);
            return d;
        }
        void* d = bump(bytes);
        hipError_t e;
        if (layout == GPT_COL_MAJOR && R > 1 && C > 1) {
            void* t = bump(bytes);
            e = h2d(t, host, bytes);
            if (e == hipSuccess) e = gpt_launch_transpose(t, d, (int)C, (int)R, 1, esz, c->stream);
        } else {
            e = h2d(d, host, bytes);
        }
        if (e != hipSuccess) rc = hip_fail(c, e, "upload");
        return d;
    }
    // room for an R x C result of the kernels (small calls: what needs no transpose lives in the bounce area already)
    void* dev(int64_t R, int64_t C, int esz) {
        const size_t bytes = (size_t)R * C * esz;
        if (tiny && (!is_matrix(R, C) || cpu_xpose(R, C))) return to_dev(bounce(bytes));
        return bump(bytes);
    }
    // download a row-major device R x C matrix into the caller's layout
    void down(void* host, void* d, int64_t R, int64_t C, int esz) {
        if (rc != GPT_OK || !host) return;
        const size_t bytes = (size_t)R * C * esz;
        hipError_t e;
        if (tiny) {
            if (cpu_xpose(R, C)) {
                Pending p{host, to_host(d), bytes};
                p.R = (int)R, p.C = (int)C, p.esz = esz;
                pending.push_back(p);
            } else if (is_matrix(R, C)) {
                gpu_xpose = true;
                char* b = bounce(bytes);
                xpose(d, to_dev(b), (int)R, (int)C, esz);
                pending.push_back({host, b, bytes});
            } else {
                pending.push_back({host, to_host(d), bytes}); // (d came from dev(): a piece of the bounce area)
            }
            return;
        }
        if (layout == GPT_COL_MAJOR && R > 1 && C > 1) {
            void* t = bump(bytes);
            e = gpt_launch_transpose(d, t, (int)R, (int)C, 1, esz, c->stream);
            if (e == hipSuccess) e = d2h(host, t, bytes);
        } else {
            e = d2h(host, d, bytes);
        }
        if (e != hipSuccess) rc = hip_fail(c, e, "download");
    }
    int finish(double* d_min_eig) {
        const bool kernels_after = xq.n != 0; // output transposes: the lattice launch was not the last one
        xflush();
        c->done_armed = false;
        if (rc != GPT_OK) return rc;
        if (tiny && c->done_published && !kernels_after && spin_done()) drained = true;
        c->done_published = false;
        double mn = INFINITY;
        if (!tiny && d_min_eig) {
            hipError_t e = hipMemcpyAsync(c->pinned, d_min_eig, sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) return hip_fail(c, e, "min_eig read-back");
        }
        if (!drained) {
            hipError_t e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) return hip_fail(c, e, "hipStreamSynchronize");
            drained = true;
        }
        for (const Pending& p : pending) {
            if (p.R > 0) xpose_host(p.src, p.host, p.R, p.C, p.esz);
            else memcpy(p.host, p.src, p.bytes);
        }
        pending.clear();
        if (d_min_eig) {
            mn = tiny ? *reinterpret_cast<const double*>(to_host(d_min_eig)) : c->pinned[0];
            c->last_min_eig = mn;
            if (mn < 0) {
                char buf[128];
                snprintf(buf, sizeof buf, "covariance matrix is not positive-definite (minimum eigenvalue is %g)", mn);
                return fail(c, GPT_ERR_NOT_PD, buf);
            }
        }
        return GPT_OK;
    }
};

size_t arena_need(int64_t H, int64_t W, int narrays) {
    // every staged array may need a transposed twin; complex arrays count double
    return (size_t)narrays * 2 * ((size_t)H * W * sizeof(double) + 256) + 4096;
}

} // namespace

extern "C" {

int gpt_gp_rfft2_host(gpt_ctx* c, int layout, const double* y, const double* loc, const double* cov, int64_t H,
                      int64_t W, int dim1, double* z_out) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 5));
    const double* dy = (const double*)s.up(y, H, W, 8);
    const double* dl = (const double*)s.up(loc, H, W, 8);
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    double* dz = (double*)s.dev(H, W, 8);
    double* dm = (double*)s.dev(1, 1, 8);
    if (s.go()) return s.rc;
    s.arm();
    if ((rc = gpt_gp_rfft2(c, dy, 0, dl, 0, dc, 0, H, W, 1, dim1, dz, dm))) return rc;
    s.down(z_out, dz, H, W, 8);
    return s.finish(dm);
}

int gpt_gp_inv_rfft2_host(gpt_ctx* c, int layout, const double* z, const double* loc, const double* cov,
                          int64_t H, int64_t W, int dim1, double* f_out) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 5));
    const double* dz = (const double*)s.up(z, H, W, 8);
    const double* dl = (const double*)s.up(loc, H, W, 8);
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    double* df = (double*)s.dev(H, W, 8);
    double* dm = (double*)s.dev(1, 1, 8);
    if (s.go()) return s.rc;
    s.arm();
    if ((rc = gpt_gp_inv_rfft2(c, dz, 0, dl, 0, dc, 0, H, W, 1, dim1, df, dm))) return rc;
    s.down(f_out, df, H, W, 8);
    return s.finish(dm);
}

int gpt_gp_inv_rfft2_obs_lpdf_host(gpt_ctx* c, int layout, int family, const double* z, const double* loc,
                                   double loc_scalar, const double* cov, const double* obs, int64_t oh, int64_t ow,
                                   double theta, int prior, int64_t H, int64_t W, int dim1, double* lp, double* grad_z,
                                   double* grad_loc_sum, double* grad_cov, double* grad_theta) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    if (!z || !cov || !obs || !lp || !grad_z) return fail(c, GPT_ERR_SHAPE, "z, cov, obs, lp and grad_z must not be null");
    if (oh < 1 || ow < 1 || oh > H || ow > W)
        return fail(c, GPT_ERR_SHAPE, "the observed window must lie inside the lattice");
    // Stan Math rejects these (normal_lpdf: "Scale parameter", neg_binomial_2_lpmf: "Precision parameter")
    if (family != GPT_OBS_POISSON_LOG && !(theta > 0 && std::isfinite(theta)))
        return fail(c, GPT_ERR_SHAPE, family == GPT_OBS_NORMAL ? "Scale parameter must be positive finite"
                                                               : "Precision parameter must be positive finite");
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 8));
    const double* dz = (const double*)s.up(z, H, W, 8);
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    const double* dob = (const double*)s.up(obs, oh, ow, 8);
    const double* dl = (const double*)s.up(loc, H, W, 8);
    double* dfill = loc ? nullptr : (double*)s.dev(H, W, 8);
    double* dgz = (double*)s.dev(H, W, 8);
    double* dgc = grad_cov ? (double*)s.dev(H, Wf, 8) : nullptr;
    double* dsc = (double*)s.dev(1, 8, 8); // theta, lp, d/dloc sum, d/dtheta, min_eig
    if (s.go()) return s.rc;
    if (dfill) {
        if (s.host_visible(dfill)) { // (a small call: its arrays live in the pinned bounce area -- fill it from here)
            double* hf = reinterpret_cast<double*>(s.to_host(dfill));
            for (int64_t i = 0; i < H * W; ++i) hf[i] = loc_scalar;
        } else {
            HIP_TRY(c, gpt_launch_fill(dfill, H * W, loc_scalar, c->stream));
        }
        dl = dfill;
    }
    s.put(dsc, theta);
    if (s.rc) return s.rc;
    s.arm(); // (taken up only by the one-launch route for lattices that fit a workgroup, gpt_api_obs.hip)
    if ((rc = gpt_gp_inv_rfft2_obs_lpdf_vjp(c, family, dz, 0, dl, 0, dc, 0, dob, 0, oh, ow,
                                            family == GPT_OBS_POISSON_LOG ? nullptr : dsc, prior, H, W, 1, dim1, dsc + 1,
                                            nullptr, dgz, nullptr, dsc + 2, dgc, dsc + 3, dsc + 4)))
        return rc;
    s.down(grad_z, dgz, H, W, 8);
    s.down(grad_cov, dgc, H, Wf, 8);
    s.down(lp, dsc + 1, 1, 1, 8);
    s.down(grad_loc_sum, dsc + 2, 1, 1, 8);
    if (family != GPT_OBS_POISSON_LOG) s.down(grad_theta, dsc + 3, 1, 1, 8);
    else if (grad_theta) *grad_theta = 0.0;
    return s.finish(dsc + 4);
}

// The same with the spectrum given by its kernel parameters (gpt_gp_inv_rfft2_obs_lpdf_kernel_vjp): five numbers go up
// and five come back where the array form moves two H x (W/2+1) arrays each way.
int gpt_gp_inv_rfft2_obs_lpdf_kernel_host(gpt_ctx* c, int layout, int family, int matern, double nu, const double* z,
                                          const double* loc, double loc_scalar, const double* params, double nugget,
                                          const double* obs, int64_t oh, int64_t ow, double theta, int prior, int64_t H,
                                          int64_t W, int dim1, double* lp, double* grad_z, double* grad_loc_sum,
                                          double* params_bar, double* grad_theta) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    if (!z || !params || !obs || !lp || !grad_z)
        return fail(c, GPT_ERR_SHAPE, "z, params, obs, lp and grad_z must not be null");
    if (oh < 1 || ow < 1 || oh > H || ow > W)
        return fail(c, GPT_ERR_SHAPE, "the observed window must lie inside the lattice");
    if (family != GPT_OBS_POISSON_LOG && !(theta > 0 && std::isfinite(theta)))
        return fail(c, GPT_ERR_SHAPE, family == GPT_OBS_NORMAL ? "Scale parameter must be positive finite"
                                                               : "Precision parameter must be positive finite");
    const int npar = dim1 ? 3 : 5;
    Stager s(c, layout, arena_need(H, W, 6));
    const double* dz = (const double*)s.up(z, H, W, 8);
    const double* dob = (const double*)s.up(obs, oh, ow, 8);
    const double* dl = (const double*)s.up(loc, H, W, 8);
    const double* dpar = (const double*)s.up(params, 1, npar, 8);
    double* dfill = loc ? nullptr : (double*)s.dev(H, W, 8);
    double* dgz = (double*)s.dev(H, W, 8);
    double* dsc = (double*)s.dev(1, 16, 8); // theta, lp, d/dloc sum, d/dtheta, min_eig, params_bar[5]
    if (s.go()) return s.rc;
    if (dfill) {
        if (s.host_visible(dfill)) {
            double* hf = reinterpret_cast<double*>(s.to_host(dfill));
            for (int64_t i = 0; i < H * W; ++i) hf[i] = loc_scalar;
        } else {
            HIP_TRY(c, gpt_launch_fill(dfill, H * W, loc_scalar, c->stream));
        }
        dl = dfill;
    }
    s.put(dsc, theta);
    if (s.rc) return s.rc;
    s.arm(); // (taken up by the one-launch route for lattices that fit a workgroup, gpt_api_obs.hip)
    if ((rc = gpt_gp_inv_rfft2_obs_lpdf_kernel_vjp(c, family, matern, nu, dz, 0, dl, 0, dpar, nugget, dob, 0, oh, ow,
                                                   family == GPT_OBS_POISSON_LOG ? nullptr : dsc, prior, H, W, 1, dim1,
                                                   dsc + 1, dgz, dsc + 2, params_bar ? dsc + 5 : nullptr, dsc + 3, dsc + 4)))
        return rc;
    s.down(grad_z, dgz, H, W, 8);
    s.down(lp, dsc + 1, 1, 1, 8);
    s.down(grad_loc_sum, dsc + 2, 1, 1, 8);
    if (params_bar) s.down(params_bar, dsc + 5, 1, npar, 8);
    if (family != GPT_OBS_POISSON_LOG) s.down(grad_theta, dsc + 3, 1, 1, 8);
    else if (grad_theta) *grad_theta = 0.0;
    return s.finish(dsc + 4);
}

int gpt_gp_rfft2_log_abs_det_jac_host(gpt_ctx* c, int layout, const double* cov, int64_t H, int64_t W, int dim1,
                                      double* out, double* grad_cov) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 4));
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    double* dout = (double*)s.dev(1, 1, 8);
    double* dm = (double*)s.dev(1, 1, 8);
    double* dg = grad_cov ? (double*)s.dev(H, Wf, 8) : nullptr;
    if (s.go()) return s.rc;
    if ((rc = gpt_gp_rfft2_log_abs_det_jac(c, dc, 0, H, W, 1, dim1, dout, dm))) return rc;
    if (grad_cov && (rc = gpt_gp_rfft2_log_abs_det_jac_vjp(c, dc, 0, H, W, 1, dim1, dg))) return rc;
    s.down(out, dout, 1, 1, 8);
    if (grad_cov) s.down(grad_cov, dg, H, Wf, 8);
    return s.finish(dm);
}

int gpt_gp_rfft2_lpdf_host(gpt_ctx* c, int layout, const double* y, const double* loc, const double* cov,
                           int64_t H, int64_t W, int dim1, double* lpdf, double* grad_y, double* grad_cov) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 7));
    const double* dy = (const double*)s.up(y, H, W, 8);
    const double* dl = (const double*)s.up(loc, H, W, 8);
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    double* dlp = (double*)s.dev(1, 1, 8);
    double* dm = (double*)s.dev(1, 1, 8);
    double* dgy = grad_y ? (double*)s.dev(H, W, 8) : nullptr;
    double* dgc = (grad_y && grad_cov) ? (double*)s.dev(H, Wf, 8) : nullptr;
    if (s.go()) return s.rc;
    s.arm();
    if (grad_y) rc = gpt_gp_rfft2_lpdf_vjp(c, dy, 0, dl, 0, dc, 0, H, W, 1, dim1, dlp, dgy, dgc, dm);
    else rc = gpt_gp_rfft2_lpdf(c, dy, 0, dl, 0, dc, 0, H, W, 1, dim1, dlp, dm);
    if (rc) return rc;
    s.down(lpdf, dlp, 1, 1, 8);
    if (grad_y) s.down(grad_y, dgy, H, W, 8);
    if (dgc) s.down(grad_cov, dgc, H, Wf, 8);
    return s.finish(dm);
}

// gp_rfft2_lpdf over a spectrum given by its kernel parameters (gpt_gp_rfft2_lpdf_kernel_vjp): y, loc and five numbers
// go up; the log density, d/dy and the five adjoints come back. loc may be NULL (0). grad_y NULL: the value only.
int gpt_gp_rfft2_lpdf_kernel_host(gpt_ctx* c, int layout, int matern, double nu, const double* y, const double* loc,
                                  const double* params, double nugget, int64_t H, int64_t W, int dim1, double* lpdf,
                                  double* grad_y, double* params_bar) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    if (!y || !params || !lpdf) return fail(c, GPT_ERR_SHAPE, "y, params and lpdf must not be null");
    const int npar = dim1 ? 3 : 5;
    Stager s(c, layout, arena_need(H, W, 5));
    const double* dy = (const double*)s.up(y, H, W, 8);
    const double* dl = (const double*)s.up(loc, H, W, 8);
    const double* dpar = (const double*)s.up(params, 1, npar, 8);
    double* dgy = (double*)s.dev(H, W, 8); // (the fused evaluation always produces d/dy)
    double* dsc = (double*)s.dev(1, 8, 8); // lpdf, min_eig, params_bar[5]
    if (s.go()) return s.rc;
    s.arm(); // (taken up when the call is one launch on the one-workgroup path)
    if ((rc = gpt_gp_rfft2_lpdf_kernel_vjp(c, matern, nu, dy, 0, dl, 0, dpar, nugget, H, W, 1, dim1, dsc, dgy,
                                           params_bar ? dsc + 2 : nullptr, dsc + 1)))
        return rc;
    s.down(lpdf, dsc, 1, 1, 8);
    if (grad_y) s.down(grad_y, dgy, H, W, 8);
    if (params_bar) s.down(params_bar, dsc + 2, 1, npar, 8);
    return s.finish(dsc + 1);
}

int gpt_gp_inv_rfft2_vjp_host(gpt_ctx* c, int layout, const double* z, const double* cov, const double* fbar,
                              int64_t H, int64_t W, int dim1, double* zbar, double* covbar) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 7));
    const double* dz = (const double*)s.up(z, H, W, 8);
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    const double* dfb = (const double*)s.up(fbar, H, W, 8);
    double* dzb = (double*)s.dev(H, W, 8);
    double* dcb = covbar ? (double*)s.dev(H, Wf, 8) : nullptr;
    if (s.go()) return s.rc;
    s.arm();
    if ((rc = gpt_gp_inv_rfft2_vjp(c, dz, 0, dc, 0, dfb, 0, H, W, 1, dim1, dzb, dcb))) return rc;
    s.down(zbar, dzb, H, W, 8);
    if (covbar) s.down(covbar, dcb, H, Wf, 8);
    return s.finish(nullptr);
}

int gpt_gp_rfft2_vjp_host(gpt_ctx* c, int layout, const double* y, const double* loc, const double* cov,
                          const double* zbar, int64_t H, int64_t W, int dim1, double* ybar, double* covbar) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 8));
    const double* dy = (const double*)s.up(y, H, W, 8);
    const double* dl = (const double*)s.up(loc, H, W, 8);
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    const double* dzb = (const double*)s.up(zbar, H, W, 8);
    double* dyb = (double*)s.dev(H, W, 8);
    double* dcb = covbar ? (double*)s.dev(H, Wf, 8) : nullptr;
    if (s.go()) return s.rc;
    s.arm();
    if ((rc = gpt_gp_rfft2_vjp(c, dy, 0, dl, 0, dc, 0, dzb, 0, H, W, 1, dim1, dyb, dcb))) return rc;
    s.down(ybar, dyb, H, W, 8);
    if (covbar) s.down(covbar, dcb, H, Wf, 8);
    return s.finish(nullptr);
}

// ---- host-pointer versions of the real-FFT layer and of the layout helpers (double only) ---------------
int gpt_rfft2_host(gpt_ctx* c, int layout, const double* y, int64_t H, int64_t W, double* out_complex) {
    int rc = check_shape(c, H, W, 1, 0);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 6));
    const double* dy = (const double*)s.up(y, H, W, 8);
    double* dout = (double*)s.dev(H, Wf, 16);
    if (s.go()) return s.rc;
    s.arm();
    if ((rc = gpt_rfft2(c, dy, H, W, 1, dout))) return rc;
    s.down(out_complex, dout, H, Wf, 16);
    return s.finish(nullptr);
}

int gpt_inv_rfft2_host(gpt_ctx* c, int layout, const double* z_complex, int64_t H, int64_t W, double* out) {
    int rc = check_shape(c, H, W, 1, 0);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 8));
    const double* dz = (const double*)s.up(z_complex, H, Wf, 16);
    double* dout = (double*)s.dev(H, W, 8);
    if (s.go()) return s.rc;
    s.arm();
    if ((rc = gpt_inv_rfft2(c, dz, H, W, 1, dout))) return rc;
    s.down(out, dout, H, W, 8);
    return s.finish(nullptr);
}

int gpt_expand_rfft_host(gpt_ctx* c, const double* y_complex, int64_t n, double* out_complex) {
    int rc = check_shape(c, 1, n, 1, 1);
    if (rc) return rc;
    Stager s(c, GPT_ROW_MAJOR, arena_need(1, n, 8));
    const double* dy = (const double*)s.up(y_complex, 1, n / 2 + 1, 16);
    double* dout = (double*)s.dev(1, n, 16);
    if (s.go()) return s.rc;
    if ((rc = gpt_expand_rfft(c, dy, n, 1, dout))) return rc;
    s.down(out_complex, dout, 1, n, 16);
    return s.finish(nullptr);
}

int gpt_gp_evaluate_rfft2_scale_host(gpt_ctx* c, int layout, const double* cov, int64_t H, int64_t W, int dim1,
                                     double* out) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 4));
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    double* dout = (double*)s.dev(H, Wf, 8);
    double* dm = (double*)s.dev(1, 1, 8);
    if (s.go()) return s.rc;
    if ((rc = gpt_gp_evaluate_rfft2_scale(c, dc, 0, H, W, 1, dim1, dout, dm))) return rc;
    s.down(out, dout, H, Wf, 8);
    return s.finish(dm);
}

int gpt_gp_unpack_rfft2_host(gpt_ctx* c, int layout, const double* z_complex, int64_t H, int64_t W, int dim1,
                             double* out) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 8));
    const double* dz = (const double*)s.up(z_complex, H, Wf, 16);
    double* dout = (double*)s.dev(H, W, 8);
    if (s.go()) return s.rc;
    if ((rc = gpt_gp_unpack_rfft2(c, dz, H, W, 1, dim1, dout))) return rc;
    s.down(out, dout, H, W, 8);
    return s.finish(nullptr);
}

int gpt_gp_pack_rfft2_host(gpt_ctx* c, int layout, const double* z, int64_t H, int64_t W, int dim1,
                           double* out_complex) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 8));
    const double* dz = (const double*)s.up(z, H, W, 8);
    double* dout = (double*)s.dev(H, Wf, 16);
    if (s.go()) return s.rc;
    if ((rc = gpt_gp_pack_rfft2(c, dz, H, W, 1, dim1, dout))) return rc;
    s.down(out_complex, dout, H, Wf, 16);
    return s.finish(nullptr);
}

static int spec_host(gpt_ctx* c, int layout, bool matern, double nu, int64_t H, int64_t W, const double* params,
                     int dim1, double* out) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 3));
    const double* dp = (const double*)s.up(params, 1, dim1 ? 3 : 5, 8);
    double* dout = (double*)s.dev(H, Wf, 8);
    if (s.go()) return s.rc;
    rc = matern ? gpt_gp_periodic_matern_cov_rfft2(c, nu, H, W, dp, 1, dim1, dout)
                : gpt_gp_periodic_exp_quad_cov_rfft2(c, H, W, dp, 1, dim1, dout);
    if (rc) return rc;
    s.down(out, dout, H, Wf, 8);
    return s.finish(nullptr);
}

static int spec_vjp_host(gpt_ctx* c, int layout, bool matern, double nu, int64_t H, int64_t W,
                         const double* params, const double* covbar, int dim1, double* params_bar) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 4));
    const double* dp = (const double*)s.up(params, 1, dim1 ? 3 : 5, 8);
    const double* dcb = (const double*)s.up(covbar, H, Wf, 8);
    double* dpb = (double*)s.dev(1, 8, 8);
    if (s.go()) return s.rc;
    rc = matern ? gpt_gp_periodic_matern_cov_rfft2_vjp(c, nu, H, W, dp, dcb, 0, 1, dim1, dpb)
                : gpt_gp_periodic_exp_quad_cov_rfft2_vjp(c, H, W, dp, dcb, 0, 1, dim1, dpb);
    if (rc) return rc;
    s.down(params_bar, dpb, 1, dim1 ? 3 : 5, 8);
    return s.finish(nullptr);
}

int gpt_gp_periodic_exp_quad_cov_rfft2_host(gpt_ctx* c, int layout, int64_t H, int64_t W, const double* params,
                                            int dim1, double* out) {
    return spec_host(c, layout, false, 0.0, H, W, params, dim1, out);
}
int gpt_gp_periodic_matern_cov_rfft2_host(gpt_ctx* c, int layout, double nu, int64_t H, int64_t W,
                                          const double* params, int dim1, double* out) {
    return spec_host(c, layout, true, nu, H, W, params, dim1, out);
}
int gpt_gp_periodic_exp_quad_cov_rfft2_vjp_host(gpt_ctx* c, int layout, int64_t H, int64_t W,
                                                const double* params, const double* covbar, int dim1,
                                                double* params_bar) {
    return spec_vjp_host(c, layout, false, 0.0, H, W, params, covbar, dim1, params_bar);
}
int gpt_gp_periodic_matern_cov_rfft2_vjp_host(gpt_ctx* c, int layout, double nu, int64_t H, int64_t W,
                                              const double* params, const double* covbar, int dim1,
                                              double* params_bar) {
    return spec_vjp_host(c, layout, true, nu, H, W, params, covbar, dim1, params_bar);
}

} // extern "C"
