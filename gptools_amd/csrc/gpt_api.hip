// C ABI of libgptools_hip.so (see include/gptools_hip.h): context, plan cache, path dispatch and the device-pointer
// entry points. The allocation-free entry point, the slab passes, the observation models and the host-pointer staging
// live in gpt_api_noalloc.hip, gpt_api_slab.hip, gpt_api_obs.hip and gpt_api_host.hip.
#include "gpt_ctx.h"
#include "gpt_hostpipe.h"

namespace gptd {

int fail(gpt_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
int hip_fail(gpt_ctx* c, hipError_t e, const char* what) {
    return fail(c, e == hipErrorOutOfMemory ? GPT_ERR_ALLOC : GPT_ERR_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}
int get_twiddles(gpt_ctx* c, int n, const cplx** out) {
    auto it = c->tw.find(n);
    if (it != c->tw.end()) {
        *out = it->second;
        return GPT_OK;
    }
    std::vector<cplx> h((size_t)n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int m = 0; m < n; ++m) {
        // exact octant values where they exist, long-double elsewhere
        const long double ang = two_pi * (long double)m / (long double)n;
        long double cr = cosl(ang), si = -sinl(ang);
        if ((4LL * m) % n == 0) {
            const int q = (int)((4LL * m) / n) & 3;
            cr = (q == 0) ? 1.0L : (q == 2) ? -1.0L : 0.0L;
            si = (q == 1) ? -1.0L : (q == 3) ? 1.0L : 0.0L;
        }
        h[(size_t)m] = make_double2((double)cr, (double)si);
    }
    void* d = nullptr;
    HIP_TRY(c, hipMalloc(&d, sizeof(cplx) * (size_t)n));
    HIP_TRY(c, hipMemcpy(d, h.data(), sizeof(cplx) * (size_t)n, hipMemcpyHostToDevice));
    c->tw[n] = reinterpret_cast<cplx*>(d);
    *out = reinterpret_cast<cplx*>(d);
    return GPT_OK;
}

// Stockham radices of a line of n points (host arithmetic only)
// wide: radix-16 / radix-8 stages first (the generic three-pass path: fewer passes through LDS); the one-workgroup
// kernels keep radix 4 and 2 (their register budget)
int plan_radices(gpt_ctx* c, int n, LinePlan& p, bool wide) {
    p.n = n;
    p.ns = 0;
    p.tw = nullptr;
    int m = n;
    auto push = [&](int r) {
        if (p.ns < GPT_MAX_STAGES) p.r[p.ns] = r;
        p.ns++;
    };
    if (wide) {
        // 5-smooth lengths (what a padded lattice usually is) keep to radices 8, 4, 2, 3, 5: their kernels are the lean
        // instantiations of gpt_generic.hip (half the registers, twice the waves per SIMD); anything else takes
        // radix-16 steps first
        int odd = n;
        while (odd % 2 == 0) odd /= 2;
        while (odd % 3 == 0) odd /= 3;
        while (odd % 5 == 0) odd /= 5;
        if (odd == 1) {
            while (m % 8 == 0) { push(8); m /= 8; }
        } else {
            while (m % 16 == 0) { push(16); m /= 16; }
            if (m % 8 == 0) { push(8); m /= 8; }
        }
    }
    while (m % 4 == 0) { push(4); m /= 4; }
    while (m % 2 == 0) { push(2); m /= 2; }
    for (int f = 3; (int64_t)f * f <= m; f += 2)
        while (m % f == 0) { push(f); m /= f; }
    if (m > 1) push(m);
    if (p.ns > GPT_MAX_STAGES) return fail(c, GPT_ERR_UNSUPPORTED, "too many FFT stages");
    return GPT_OK;
}

int make_plan(gpt_ctx* c, int n, LinePlan& p, bool wide) {
    const int rc = plan_radices(c, n, p, wide);
    return rc != GPT_OK ? rc : get_twiddles(c, n, &p.tw);
}

int path_for_shape(int64_t H, int64_t W) {
    if (H < 1 || W < 1 || H * W > 2147483647LL) return -1;
    if (gpt_small_fits((int)H, (int)W, nullptr, nullptr)) return 0;
    if (gpt_pow2_supported((int)H, (int)W)) return 1;
    if (gpt_generic_supported((int)H, (int)W) && !gpt_anyn_preferred((int)H, (int)W)) return 2;
    if (gpt_anyn_supported((int)H, (int)W)) return 3;
    return -1;
}

int check_shape(gpt_ctx* c, int64_t H, int64_t W, int64_t batch, int dim1) {
    if (!c) return GPT_ERR_SHAPE;
    if (H < 1 || W < 1 || batch < 0) return fail(c, GPT_ERR_SHAPE, "height, width >= 1 and batch >= 0 required");
    if (dim1 && H != 1) return fail(c, GPT_ERR_SHAPE, "1-D functions take height == 1");
    // n = height * width is a 32-bit Stan int in the reference (fft2.stan:15)
    if (H * W > 2147483647LL) return fail(c, GPT_ERR_SHAPE, "height * width exceeds the reference's 32-bit int");
    return GPT_OK;
}

// The three passes over `a.batch` lattices in units of `chunk` lattices that share the workspace `ws` (chunk times
// the path's per-lattice workspace) and `partials`, one unit after the other on the context's stream.
int run_units(gpt_ctx* c, int op, const LatArgs& a, int path, int64_t chunk, cplx* ws, double* partials, bool may_profile) {
    const int64_t hw = (int64_t)a.H * a.W, hwf = (int64_t)a.H * (a.W / 2 + 1);
    for (int64_t b0 = 0; b0 < a.batch; b0 += chunk) {
        LatArgs s = a;
        s.batch = (a.batch - b0 < chunk) ? a.batch - b0 : chunk;
        if (s.x) s.x += b0 * a.x_bs;
        if (s.loc) s.loc += b0 * a.loc_bs;
        if (s.cov) s.cov += b0 * a.cov_bs;
        if (s.aux) s.aux += b0 * a.aux_bs;
        if (s.cin) s.cin += b0 * a.cin_bs;
        if (s.out) s.out += b0 * hw;
        if (s.out_cov) s.out_cov += b0 * hwf;
        if (s.out_c) s.out_c += b0 * hwf;
        if (s.out_lpdf) s.out_lpdf += b0;
        if (s.min_eig) s.min_eig += b0;
        if (s.kpar) s.kpar += b0 * 5;
        if (s.kbar) s.kbar += b0 * 5;
        if (path == 3) {
            HIP_TRY(c, gpt_launch_anyn(c->anyn, op, s, c->stream));
            continue;
        }
        if (path != 1) {
            HIP_TRY(c, gpt_launch_generic(op, s, ws, partials, c->stream));
            continue;
        }
        for (int pass = 0; pass < 3; ++pass) {
            // each pass bracketed by timing events in profiling mode
            const bool present = pass == 1 || (pass == 0 ? op_has_fwd(op) : op_has_inv(op));
            gpt_ctx::ProfRec pr = {nullptr, nullptr, pass == 2 ? 3 : pass};
            if (may_profile && c->profiling && present) {
                HIP_TRY(c, hipEventCreate(&pr.t0));
                HIP_TRY(c, hipEventCreate(&pr.t1));
                HIP_TRY(c, hipEventRecord(pr.t0, c->stream));
            }
            HIP_TRY(c, gpt_pow2_pass(pass, op, s, ws, partials, c->total_cus, c->stream));
            if (pr.t0) {
                HIP_TRY(c, hipEventRecord(pr.t1, c->stream));
                c->prof.push_back(pr);
            }
        }
    }
    return GPT_OK;
}

// A lattice that fills more than half of a CU's LDS runs one workgroup per CU on the one-workgroup path: fine for a few
// chains (one launch, ~30 us), but a launch of many such lattices is then one latency-bound workgroup after the other on
// every CU, and the generic three-pass path (several small workgroups per CU, lean kernels) has the higher throughput.
// Measured (profiles/r03_s_small_vs_generic_many.jsonl, value + gradients, us per launch, one-workgroup / generic):
//   72 x 72 (83 KB):  256 chains 29.5 / 43.1, 384: 54.4 / 49.7, 512: 89.8 / 56.1, 1024: 177 / 94.6, 4096: 801 / 359
//   96 x 100 (150 KB): 256: 40.4 / 51.6, 384: 74.9 / 68.5, 512: 137 / 83.9, 4096: 1266 / 648
//   60 x 60 (58 KB, two workgroups per CU): 512: 37.7 / 46.7, 1024: 72.9 / 67.7, 4096: 333 / 261
//   50 x 50 (40 KB) and smaller, 35 x 60: the one-workgroup path at every count (4096: 154 / 219, 121 / 200)
// -- up to a lattice per CU the single launch wins; from 1.5 per CU (one workgroup per CU) or 8 per CU (two) the three
// passes do. GPT_SMALL_TO_GENERIC_BATCH=<n>: that first count (default 384; 0 = never re-route).
static bool many_large_small_lattices(int op, const LatArgs& a) {
    static const int64_t from = [] {
        const char* e = getenv("GPT_SMALL_TO_GENERIC_BATCH");
        return (int64_t)(e ? atol(e) : 384);
    }();
    if (from <= 0 || a.batch < from || a.dim1 || op_is_kvjp(op)) return false;
    if (a.H < 32 || a.W < 32 || !gpt_generic_supported(a.H, a.W)) return false;
    int cap = 0;
    if (!gpt_small_fits(a.H, a.W, &cap, nullptr)) return false;
    const size_t lds = 2 * (size_t)cap * sizeof(cplx);
    if (lds > 80 * 1024) return true;                      // one workgroup per CU
    return lds > 160 * 1024 / 3 && a.batch >= 2048;         // two
}

// The kernel-fused log density on power-of-two lattices WITHOUT a sub-line column kernel (256 rows): their whole-column
// kernel has no OP_LPDF_KVJP, the generic path's has (256 x 256 x 64 lattices: 88 us there against 170 us for the
// composition of spectrum, log density and spectrum reverse mode on the power-of-two path).
static bool kvjp_on_generic(int op, int H, int W) {
    return op_is_kvjp(op) && gpt_pow2_supported(H, W) && gpt_pow2_sub_radix(op, H, W) == 0 && gpt_generic_supported(H, W);
}

// The power-of-two kernels are built for throughput: few, fat workgroups (a 256 x 256 lattice is 8 row and a handful of
// column workgroups), so a launch of one or a few SMALL power-of-two lattices -- what a single Stan chain sends -- is a
// long chain of phases on a handful of CUs. The generic path's kernels (many small workgroups, lean instantiations: every
// power of two is 5-smooth) have the lower latency there. Measured (profiles/r03_x_pow2_vs_generic_few.jsonl, value +
// gradients, us per launch, power-of-two / generic kernels): 256^2: 1 lattice 60.7 / 31.8, 16: 74.9 / 40.8, 64: 98.1 /
// 74.3, 1024: 1039 / 1073; 512^2: 1: 40.0 / 35.0, 4: 48.8 / 43.5, 16: 64.4 / 75.4; 1024^2: 1: 66.0 / 47.8, 4: 83.1 / 84.0,
// 16: 297 / 322; 1024 x 256: 1: 52.1 / 37.9, 4: 62.3 / 47.0; 2048^2 and larger: the power-of-two kernels at every count.
// Rule: launches of up to 2^20 lattice points in all (2^23 at 256 rows, whose power-of-two column kernel is the
// whole-column one) take the generic kernels. GPT_POW2_SMALL_TO_GENERIC=0 (read per call) keeps them on the tuned path.
static bool few_pow2_lattices(const LatArgs& a) {
    const char* e = getenv("GPT_POW2_SMALL_TO_GENERIC");
    if ((e && e[0] == '0') || a.H > 1024 || !gpt_generic_supported(a.H, a.W)) return false;
    const int64_t points = a.batch * (int64_t)a.H * (int64_t)a.W;
    return points <= (a.H <= 256 ? (int64_t)1 << 23 : (int64_t)1 << 20);
}

int run_lattice(gpt_ctx* c, int op, LatArgs a) {
    const bool armed = c->done_armed; // (one request, one launch: whatever happens below, it does not outlive this call)
    c->done_armed = false;
    HIP_TRY(c, hipSetDevice(c->device));
    if (a.batch == 0) return GPT_OK;
    int rc;
    int path = path_for_shape(a.H, a.W);
    if (path == 0 && many_large_small_lattices(op, a)) path = 2;
    if (path == 1 && kvjp_on_generic(op, a.H, a.W)) path = 2;
    if (path == 1 && few_pow2_lattices(a)) path = 2;
    if (path == 3) { // any-length path: its own plans (four-step / Bluestein), tables and scratch
        if (!c->anyn) c->anyn = gpt_anyn_create();
        return run_units(c, op, a, path, gpt_anyn_chunk(a.H, a.W), nullptr, nullptr, false);
    }
    if ((rc = make_plan(c, a.W, a.pw, path == 2)) != GPT_OK) return rc;
    if ((rc = make_plan(c, a.H, a.ph, path == 2)) != GPT_OK) return rc;
    if (path == 0) {
        unsigned long long* flag = nullptr;
        if (armed && a.batch == 1 && c->done_dev) { // (a host entry point whose only launch this is: see gpt_ctx::done_*)
            flag = c->done_dev;
            c->done_published = true;
            ++c->done_seq;
        }
        HIP_TRY(c, gpt_launch_small(op, a, c->stream, flag, c->done_seq));
        return GPT_OK;
    }
    if (path < 0) {
        char buf[160];
        snprintf(buf, sizeof buf, "no kernel for a %d x %d lattice in this build", a.H, a.W);
        return fail(c, GPT_ERR_UNSUPPORTED, buf);
    }
    // Three-pass paths: lattices are processed in units (chunks of lattices that share one packed-spectrum
    // workspace of <= ~128 MiB, so that it can live in the 256 MiB Infinity Cache between the passes), one
    // unit after the other on the context's stream. (Overlapping pass B of one lattice with the row passes of
    // its neighbours on a second stream, with and without CU masks, was measured and is slower: pass B has no
    // idle CU resources to share and slows down in proportion to the CUs it gives up -- DESIGN.md section 6.)
    if (path == 1) {
        const int R = gpt_pow2_sub_radix(op, a.H, a.W);
        if (R && (rc = get_twiddles(c, a.H / R, &a.tw_sub)) != GPT_OK) return rc;
    }
    const size_t per = path == 1 ? gpt_pow2_workspace_bytes(a.H, a.W) : gpt_generic_workspace_bytes(a.H, a.W);
    // Ops with a forward row pass (power-of-two path) take units of up to GPT_UNIT_MIB (default 2560) of workspace
    // instead: that pass numbers its workgroups lattice-fastest, so the rows of a loc shared by the batch are read
    // from HBM once per unit, and launches of many lattices have no tails. At 4096^2 with 16 lattices per unit the
    // row pass takes 52 instead of 76 us per lattice, the column pass (its workspace now comes from HBM) 125 instead
    // of 117, the inverse row pass 54 instead of 50: +3 % overall; units of 2..9 lattices lose.
    static const size_t unit_budget = [] {
        const char* e = getenv("GPT_UNIT_MIB");
        const long v = e ? atol(e) : 0;
        return (size_t)(v > 0 ? v : 2560) << 20;
    }();
    const size_t small_budget = (size_t)128 << 20;
    int64_t chunk = (int64_t)(small_budget / per);
    if (chunk < 1) chunk = 1;
    static const int small_unit_max = [] { // GPT_UNIT_SMALL_MAX: cache-sized units of up to this many lattices give way to large ones
        const char* e = getenv("GPT_UNIT_SMALL_MAX");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 3;
    }();
    if (path == 1 && op_has_fwd(op) && chunk <= small_unit_max) {
        // (lattices so large that a cache-sized unit holds only a few: launch tails matter.) A unit that is neither
        // cache-sized nor many lattices is the worst of both: below 1.5 GiB of workspace the small units stay.
        int64_t big = (int64_t)(unit_budget / per);
        if (big < 1) big = 1;
        const int64_t nunits = (a.batch + big - 1) / big;
        big = (a.batch + nunits - 1) / nunits; // units of equal size
        static const size_t unit_min = [] { // GPT_UNIT_MIN_MIB: smallest unit worth giving up cache residency for
            const char* e = getenv("GPT_UNIT_MIN_MIB");
            const long v = e ? atol(e) : 0;
            return (size_t)(v > 0 ? v : 1536) << 20;
        }();
        if (per * (size_t)big >= unit_min) chunk = big;
    }
    if (chunk > a.batch) chunk = a.batch;
    if (chunk > 32768) chunk = 32768;
    if (c->total_cus == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
        c->total_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int nblk = path == 1 ? gpt_pow2_col_blocks(a.H, a.W, c->total_cus) : gpt_generic_col_blocks(a.H, a.W);
    if ((rc = ensure(c, c->ws, c->ws_bytes, per * (size_t)chunk)) != GPT_OK) return rc;
    if ((rc = ensure(c, c->partials, c->partials_bytes, sizeof(double) * GPT_PART * (size_t)nblk * (size_t)chunk)) !=
        GPT_OK)
        return rc;
    return run_units(c, op, a, path, chunk, c->ws, c->partials, true);
}

LatArgs base_args(int64_t H, int64_t W, int64_t batch, int dim1) {
    LatArgs a;
    memset(&a, 0, sizeof a);
    a.H = (int)H;
    a.W = (int)W;
    a.batch = batch;
    a.dim1 = dim1 ? 1 : 0;
    return a;
}

int run_elem(gpt_ctx* c, int op, ElemArgs e, bool reduces, int K, double add, double* out_sum, double* out_min) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (e.batch == 0) return GPT_OK;
    const int nblk = gpt_elem_blocks(e.H, e.W);
    const int64_t chunk_max = 32768;
    int rc;
    if (reduces) {
        const int64_t chunk = e.batch < chunk_max ? e.batch : chunk_max;
        if ((rc = ensure(c, c->partials, c->partials_bytes, sizeof(double) * GPT_PART * (size_t)nblk * (size_t)chunk)) !=
            GPT_OK)
            return rc;
    }
    const int64_t hw = (int64_t)e.H * e.W, hwf = (int64_t)e.H * (e.W / 2 + 1);
    const int npar = e.dim1 ? 3 : 5;
    for (int64_t b0 = 0; b0 < e.batch; b0 += chunk_max) {
        ElemArgs s = e;
        s.batch = (e.batch - b0 < chunk_max) ? e.batch - b0 : chunk_max;
        if (s.in) s.in += b0 * e.in_bs;
        if (s.cin) s.cin += b0 * e.cin_bs;
        if (s.par) s.par += b0 * npar;
        if (s.out) {
            if (op == EL_UNPACK) s.out += b0 * hw;
            else s.out += b0 * hwf;
        }
        if (s.cout) s.cout += b0 * (op == EL_EXPAND ? (int64_t)e.W : hwf);
        HIP_TRY(c, gpt_launch_elem(op, s, c->partials, c->stream));
        if (reduces)
            HIP_TRY(c, gpt_launch_finalize(c->partials, nblk, s.batch, K, add, out_sum ? out_sum + b0 * K : nullptr,
                                           out_min ? out_min + b0 : nullptr, c->stream));
    }
    return GPT_OK;
}

ElemArgs elem_args(int64_t H, int64_t W, int64_t batch, int dim1) {
    ElemArgs e;
    memset(&e, 0, sizeof e);
    e.H = (int)H;
    e.W = (int)W;
    e.batch = batch;
    e.dim1 = dim1 ? 1 : 0;
    return e;
}

int device_cus(gpt_ctx* c) {
    if (c->total_cus == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return 256;
        c->total_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return c->total_cus;
}

} // namespace gptd
using namespace gptd;

// ===================================================================================================
extern "C" {

const char* gpt_version(void) { return "gptools_hip 0.1 (gfx950)"; }

int gpt_create(int device, void* stream, gpt_ctx** out) {
    if (!out) return GPT_ERR_SHAPE;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return GPT_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return GPT_ERR_HIP;
    gpt_ctx* c = new gpt_ctx();
    c->device = device;
    c->stream = reinterpret_cast<hipStream_t>(stream); // NULL = HIP's default stream
    void* p = nullptr;
    if (hipHostMalloc(&p, 4096, hipHostMallocDefault) != hipSuccess) {
        delete c;
        return GPT_ERR_ALLOC;
    }
    c->pinned = reinterpret_cast<double*>(p);
    // this device's copy of the built-in scratch array (`device` is current) and its CU count: nothing after this
    // point of an allocation-free call has to ask the runtime for either
    if (noalloc_resolve(c) != GPT_OK) {
        (void)hipHostFree(p);
        delete c;
        return GPT_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        c->total_cus = prop.multiProcessorCount;
    *out = c;
    return GPT_OK;
}

int gpt_destroy(gpt_ctx* c) {
    if (!c) return GPT_OK;
    // (may run on whichever thread collects the context, e.g. a finalizer: leave that thread's current device alone)
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& r : c->prof) {
        (void)hipEventDestroy(r.t0);
        (void)hipEventDestroy(r.t1);
    }
    for (auto& kv : c->tw) (void)hipFree(kv.second);
    if (c->ws) (void)hipFree(c->ws);
    if (c->partials) (void)hipFree(c->partials);
    if (c->arena) (void)hipFree(c->arena);
    if (c->kscr) (void)hipFree(c->kscr);
    if (c->oscr) (void)hipFree(c->oscr);
    if (c->sink) (void)hipFree(c->sink);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pipe) {
        c->pipe->destroy();
        delete c->pipe;
    }
    if (c->bounce) (void)hipHostFree(c->bounce);
    gpt_anyn_destroy(c->anyn);
    delete c;
    if (prev >= 0) (void)hipSetDevice(prev);
    return GPT_OK;
}

int gpt_set_stream(gpt_ctx* c, void* stream) {
    if (!c) return GPT_ERR_SHAPE;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = reinterpret_cast<hipStream_t>(stream);
    return GPT_OK;
}

int gpt_synchronize(gpt_ctx* c) {
    if (!c) return GPT_ERR_SHAPE;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GPT_OK;
}

int gpt_profile_begin(gpt_ctx* c) {
    if (!c) return GPT_ERR_SHAPE;
    for (auto& r : c->prof) {
        (void)hipEventDestroy(r.t0);
        (void)hipEventDestroy(r.t1);
    }
    c->prof.clear();
    c->profiling = true;
    return GPT_OK;
}

int gpt_profile_end(gpt_ctx* c, double* pass_ms, int64_t* launches) {
    if (!c) return GPT_ERR_SHAPE;
    c->profiling = false;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double acc[4] = {0, 0, 0, 0};
    int64_t nb = 0;
    for (auto& r : c->prof) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, r.t0, r.t1));
        acc[r.pass] += ms;
        if (r.pass == 1) ++nb;
    }
    for (auto& r : c->prof) {
        (void)hipEventDestroy(r.t0);
        (void)hipEventDestroy(r.t1);
    }
    c->prof.clear();
    if (pass_ms)
        for (int k = 0; k < 4; ++k) pass_ms[k] = acc[k];
    if (launches) *launches = nb;
    return GPT_OK;
}

const char* gpt_last_error(const gpt_ctx* c) { return c ? c->err.c_str() : "null context"; }
double gpt_last_min_eig(const gpt_ctx* c) { return c ? c->last_min_eig : NAN; }
int gpt_path_for_shape(int64_t H, int64_t W) { return path_for_shape(H, W); }


// ---- real-FFT layer --------------------------------------------------------------------------------
int gpt_rfft2(gpt_ctx* c, const double* y, int64_t H, int64_t W, int64_t batch, double* out_complex) {
    int rc = check_shape(c, H, W, batch, 0);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, 0);
    a.x = y;
    a.x_bs = H * W;
    a.out_c = reinterpret_cast<cplx*>(out_complex);
    return run_lattice(c, OP_RFFT2, a);
}

int gpt_inv_rfft2(gpt_ctx* c, const double* z_complex, int64_t H, int64_t W, int64_t batch, double* out) {
    int rc = check_shape(c, H, W, batch, 0);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, 0);
    a.cin = reinterpret_cast<const cplx*>(z_complex);
    a.cin_bs = H * (W / 2 + 1);
    a.out = out;
    return run_lattice(c, OP_INV_RFFT2, a);
}

int gpt_expand_rfft(gpt_ctx* c, const double* y_complex, int64_t n, int64_t batch, double* out_complex) {
    int rc = check_shape(c, 1, n, batch, 1);
    if (rc) return rc;
    ElemArgs e = elem_args(1, n, batch, 1);
    e.cin = reinterpret_cast<const cplx*>(y_complex);
    e.cin_bs = n / 2 + 1;
    e.cout = reinterpret_cast<cplx*>(out_complex);
    return run_elem(c, EL_EXPAND, e, false, 0, 0.0, nullptr, nullptr);
}

// ---- GP layer --------------------------------------------------------------------------------------
int gpt_gp_evaluate_rfft2_scale(gpt_ctx* c, const double* cov, int64_t cov_bs, int64_t H, int64_t W,
                                int64_t batch, int dim1, double* out, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.in = cov;
    e.in_bs = cov_bs;
    e.out = out;
    return run_elem(c, EL_SCALE, e, true, 0, 0.0, nullptr, min_eig);
}

int gpt_gp_unpack_rfft2(gpt_ctx* c, const double* z_complex, int64_t H, int64_t W, int64_t batch, int dim1,
                        double* out) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.cin = reinterpret_cast<const cplx*>(z_complex);
    e.cin_bs = H * (W / 2 + 1);
    e.out = out;
    return run_elem(c, EL_UNPACK, e, false, 0, 0.0, nullptr, nullptr);
}

int gpt_gp_pack_rfft2(gpt_ctx* c, const double* z, int64_t H, int64_t W, int64_t batch, int dim1,
                      double* out_complex) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.in = z;
    e.in_bs = H * W;
    e.cout = reinterpret_cast<cplx*>(out_complex);
    return run_elem(c, EL_PACK, e, false, 0, 0.0, nullptr, nullptr);
}

int gpt_gp_rfft2(gpt_ctx* c, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs, const double* cov,
                 int64_t cov_bs, int64_t H, int64_t W, int64_t batch, int dim1, double* z_out, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = y; a.x_bs = y_bs;
    a.loc = loc; a.loc_bs = loc_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.out = z_out;
    a.min_eig = min_eig;
    return run_lattice(c, OP_GP_RFFT2, a);
}

int gpt_gp_inv_rfft2(gpt_ctx* c, const double* z, int64_t z_bs, const double* loc, int64_t loc_bs,
                     const double* cov, int64_t cov_bs, int64_t H, int64_t W, int64_t batch, int dim1,
                     double* f_out, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = z; a.x_bs = z_bs;
    a.loc = loc; a.loc_bs = loc_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.out = f_out;
    a.min_eig = min_eig;
    return run_lattice(c, OP_GP_INV_RFFT2, a);
}

int gpt_gp_rfft2_log_abs_det_jac(gpt_ctx* c, const double* cov, int64_t cov_bs, int64_t H, int64_t W,
                                 int64_t batch, int dim1, double* out, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.in = cov;
    e.in_bs = cov_bs;
    const int64_t n = H * W;
    int64_t nterms = (n - 1) / 2;
    if (H % 2 == 0 && W % 2 == 0) nterms -= 1;
    // fft2.stan:167-171 / fft1.stan:81
    const double add = -std::log(2.0) * (double)nterms + (double)n * std::log((double)n) / 2.0;
    return run_elem(c, EL_LADJ, e, true, 1, add, out, min_eig);
}

int gpt_gp_rfft2_log_abs_det_jac_vjp(gpt_ctx* c, const double* cov, int64_t cov_bs, int64_t H, int64_t W,
                                     int64_t batch, int dim1, double* grad_cov) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.in = cov;
    e.in_bs = cov_bs;
    e.out = grad_cov;
    return run_elem(c, EL_LADJ_VJP, e, false, 0, 0.0, nullptr, nullptr);
}

int gpt_gp_rfft2_lpdf(gpt_ctx* c, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs,
                      const double* cov, int64_t cov_bs, int64_t H, int64_t W, int64_t batch, int dim1,
                      double* lpdf, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = y; a.x_bs = y_bs;
    a.loc = loc; a.loc_bs = loc_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.out_lpdf = lpdf;
    a.min_eig = min_eig;
    return run_lattice(c, OP_LPDF, a);
}

int gpt_gp_rfft2_lpdf_vjp(gpt_ctx* c, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs,
                          const double* cov, int64_t cov_bs, int64_t H, int64_t W, int64_t batch, int dim1,
                          double* lpdf, double* grad_y, double* grad_cov, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if (!grad_y) return fail(c, GPT_ERR_SHAPE, "grad_y must not be null (use gpt_gp_rfft2_lpdf for the value)");
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = y; a.x_bs = y_bs;
    a.loc = loc; a.loc_bs = loc_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.out = grad_y;
    a.out_cov = grad_cov;
    a.out_lpdf = lpdf;
    a.min_eig = min_eig;
    return run_lattice(c, grad_cov ? OP_LPDF_VJP : OP_LPDF_VJP_NOCOV, a);
}


} // extern "C"
namespace gptd {
// prior_lp != null: the white-noise prior of the non-centred models rides along (zbar -= z, prior_lp[b] =
// std_normal_lpdf(z[b])): the column pass reads z and writes zbar anyway
int inv_rfft2_vjp(gpt_ctx* c, const double* z, int64_t z_bs, const double* cov, int64_t cov_bs,
                         const double* fbar, int64_t fbar_bs, int64_t H, int64_t W, int64_t batch, int dim1,
                         double* zbar, double* covbar, double* prior_lp) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = z; a.x_bs = z_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.aux = fbar; a.aux_bs = fbar_bs;
    a.out = zbar;
    a.out_cov = covbar;
    a.zprior = prior_lp ? 1 : 0;
    a.out_lpdf = prior_lp;
    if (!covbar) {
        if ((rc = ensure(c, c->sink, c->sink_bytes, sizeof(double) * (size_t)batch * H * (W / 2 + 1)))) return rc;
        a.out_cov = c->sink;
    }
    return run_lattice(c, OP_INV_VJP, a);
}
} // namespace gptd
extern "C" {

int gpt_gp_inv_rfft2_vjp(gpt_ctx* c, const double* z, int64_t z_bs, const double* cov, int64_t cov_bs,
                         const double* fbar, int64_t fbar_bs, int64_t H, int64_t W, int64_t batch, int dim1,
                         double* zbar, double* covbar) {
    return inv_rfft2_vjp(c, z, z_bs, cov, cov_bs, fbar, fbar_bs, H, W, batch, dim1, zbar, covbar, nullptr);
}

int gpt_gp_rfft2_vjp(gpt_ctx* c, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs,
                     const double* cov, int64_t cov_bs, const double* zbar, int64_t zbar_bs, int64_t H, int64_t W,
                     int64_t batch, int dim1, double* ybar, double* covbar) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = y; a.x_bs = y_bs;
    a.loc = loc; a.loc_bs = loc_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.aux = zbar; a.aux_bs = zbar_bs;
    a.out = ybar;
    a.out_cov = covbar;
    if (!covbar) {
        if ((rc = ensure(c, c->sink, c->sink_bytes, sizeof(double) * (size_t)batch * H * (W / 2 + 1)))) return rc;
        a.out_cov = c->sink;
    }
    return run_lattice(c, OP_FWD_VJP, a);
}

// ---- spectral kernels ------------------------------------------------------------------------------
static double gamma_ratio(double nu, int dim1) {
    const double half_ndim = dim1 ? 0.5 : 1.0;
    return std::tgamma(nu + half_ndim) / std::tgamma(nu); // fft1.stan:173, fft2.stan:245
}

// Kernel-fused log density: lpdf(y | loc, kernel(params) + nugget) with the reverse mode down
// to the kernel parameters. One-workgroup lattices, power-of-two lattices of 512..8192 rows and the generic three-pass
// path evaluate the spectrum inside their elementwise stage (OP_LPDF_KVJP: no cov read, no d/dcov write, the
// parameter gradient finished by the last kernel); every other shape (and smoothness values other than 0.5 ... 5.5)
// composes the existing kernels on library-owned scratch (spectrum -> fused value + gradient -> kernel reverse mode),
// same results.
int gpt_gp_rfft2_lpdf_kernel_vjp(gpt_ctx* c, int matern, double nu, const double* y, int64_t y_bs, const double* loc,
                                 int64_t loc_bs, const double* params, double nugget, int64_t H, int64_t W,
                                 int64_t batch, int dim1, double* lpdf, double* grad_y, double* params_bar,
                                 double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if (!y || !params || !lpdf || !grad_y) return fail(c, GPT_ERR_SHAPE, "y, params, lpdf and grad_y must not be null");
    if (matern && !(nu > 0)) return fail(c, GPT_ERR_SHAPE, "nu must be positive");
    // (a host entry point's request to let the last launch publish its completion, gpt_ctx::done_*: the composed route
    // below is several launches and must not hand it to its first one)
    const bool armed = c->done_armed;
    c->done_armed = false;
    if (batch == 0) return GPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    // the fused evaluation knows exp-quad and half-integer Matern smoothness (nu = 0.5, 1.5, ... 5.5)
    const bool nu_ok = !matern || (nu <= 5.5 && nu - std::floor(nu) == 0.5 /* kern_eval */);
    const int path = path_for_shape(H, W);
    // (the 1-D functions: the one-workgroup kernel only -- its kern_prepare knows fft1.stan:154-176)
    const bool fused = nu_ok && (path == 0 || (!dim1 && path == 1 && gpt_pow2_sub_radix(OP_LPDF_KVJP, (int)H, (int)W) != 0) ||
                                 (!dim1 && path == 2) || (!dim1 && path == 1 && kvjp_on_generic(OP_LPDF_KVJP, (int)H, (int)W)));
    if (fused) {
        // both paths finish inside their last kernel (the one-workgroup kernel's last thread, finalize_block of pass C):
        // no combine launch, no scratch
        LatArgs a = base_args(H, W, batch, dim1);
        a.x = y; a.x_bs = y_bs;
        a.loc = loc; a.loc_bs = loc_bs;
        a.out = grad_y;
        a.min_eig = min_eig;
        a.kpar = params;
        a.kmatern = matern ? 1 : 0;
        a.knu = nu;
        a.kgamma = matern ? gamma_ratio(nu, dim1) : 0.0;
        a.knugget = nugget;
        a.out_lpdf = lpdf;
        a.kbar = params_bar;
        c->done_armed = armed && path == 0; // (one launch on the one-workgroup path: it may publish its completion)
        return run_lattice(c, matern ? OP_LPDF_KVJP : OP_LPDF_KVJP_EQ, a);
    }
    // composed: cov and d/dcov live in scratch, a chunk of lattices at a time (<= ~256 MiB)
    const size_t hwf = (size_t)H * (size_t)(W / 2 + 1);
    int64_t chunk = (int64_t)((size_t)(128u << 20) / (hwf * sizeof(double)));
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if ((rc = ensure(c, c->kscr, c->kscr_bytes, 2 * sizeof(double) * hwf * (size_t)chunk)) != GPT_OK) return rc;
    double* cov = c->kscr;
    double* dcov = c->kscr + hwf * (size_t)chunk;
    const int npar = dim1 ? 3 : 5;
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = batch - b0 < chunk ? batch - b0 : chunk;
        ElemArgs e = elem_args(H, W, nb, dim1);
        e.par = params + b0 * npar;
        e.out = cov;
        e.nu = nu;
        e.gamma_ratio = matern ? gamma_ratio(nu, dim1) : 0.0;
        e.nugget = nugget;
        if ((rc = run_elem(c, matern ? EL_MATERN : EL_EXPQUAD, e, false, 0, 0.0, nullptr, nullptr)) != GPT_OK) return rc;
        if ((rc = gpt_gp_rfft2_lpdf_vjp(c, y + b0 * y_bs, y_bs, loc ? loc + b0 * loc_bs : nullptr, loc_bs, cov, (int64_t)hwf,
                                        H, W, nb, dim1, lpdf + b0, grad_y + b0 * H * W, params_bar ? dcov : nullptr,
                                        min_eig ? min_eig + b0 : nullptr)) != GPT_OK)
            return rc;
        if (params_bar) {
            ElemArgs v = elem_args(H, W, nb, dim1);
            v.par = params + b0 * npar;
            v.in = dcov;
            v.in_bs = (int64_t)hwf;
            v.nu = nu;
            v.gamma_ratio = e.gamma_ratio;
            if ((rc = run_elem(c, matern ? EL_MATERN_VJP : EL_EXPQUAD_VJP, v, true, npar, 0.0, params_bar + b0 * npar,
                               nullptr)) != GPT_OK)
                return rc;
        }
    }
    return GPT_OK;
}


int gpt_gp_periodic_exp_quad_cov_rfft2(gpt_ctx* c, int64_t H, int64_t W, const double* params, int64_t batch,
                                       int dim1, double* out) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.par = params;
    e.out = out;
    return run_elem(c, EL_EXPQUAD, e, false, 0, 0.0, nullptr, nullptr);
}

int gpt_gp_periodic_matern_cov_rfft2(gpt_ctx* c, double nu, int64_t H, int64_t W, const double* params,
                                     int64_t batch, int dim1, double* out) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if (!(nu > 0)) return fail(c, GPT_ERR_SHAPE, "nu must be positive");
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.par = params;
    e.out = out;
    e.nu = nu;
    e.gamma_ratio = gamma_ratio(nu, dim1);
    return run_elem(c, EL_MATERN, e, false, 0, 0.0, nullptr, nullptr);
}

int gpt_gp_periodic_exp_quad_cov_rfft2_vjp(gpt_ctx* c, int64_t H, int64_t W, const double* params,
                                           const double* covbar, int64_t covbar_bs, int64_t batch, int dim1,
                                           double* params_bar) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.par = params;
    e.in = covbar;
    e.in_bs = covbar_bs;
    return run_elem(c, EL_EXPQUAD_VJP, e, true, dim1 ? 3 : 5, 0.0, params_bar, nullptr);
}

int gpt_gp_periodic_matern_cov_rfft2_vjp(gpt_ctx* c, double nu, int64_t H, int64_t W, const double* params,
                                         const double* covbar, int64_t covbar_bs, int64_t batch, int dim1,
                                         double* params_bar) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if (!(nu > 0)) return fail(c, GPT_ERR_SHAPE, "nu must be positive");
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.par = params;
    e.in = covbar;
    e.in_bs = covbar_bs;
    e.nu = nu;
    e.gamma_ratio = gamma_ratio(nu, dim1);
    return run_elem(c, EL_MATERN_VJP, e, true, dim1 ? 3 : 5, 0.0, params_bar, nullptr);
}

} // extern "C"
