// BASELINE configs 1-3 at the C ABI, where a sampler calls (no Python in the loop): microseconds per call of the fused
// value + gradient entry point
//   dev     gpt_gp_rfft2_lpdf_vjp, device pointers, `iters` back-to-back calls, one synchronisation at the end
//   graph   gpt_gp_rfft2_lpdf_vjp_noalloc captured into a HIP graph once, `iters` replays (the entry point that makes
//           no allocation is the one a capture can hold)
//   host    gpt_gp_rfft2_lpdf_host, host pointers, column-major (what the Stan header calls; synchronous)
// One JSON object per line.
//   build: hipcc -O2 -std=c++17 -I include scripts/micro/bench_small_abi.cpp -o scripts/micro/bench_small_abi \
//          -L gptools_amd -lgptools_hip -Wl,-rpath,'$ORIGIN/../../gptools_amd'
#include "gptools_hip.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define GK(x) do { int r_ = (x); if (r_ != GPT_OK) { printf("%s -> %d: %s\n", #x, r_, gpt_last_error(ctx)); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    struct Case { const char* label; int64_t H, W, batch; };
    const Case cases[] = {{"config 1: 1-D n=128, one chain", 1, 128, 1},        {"config 1 (docs): 1-D n=100, one chain", 1, 100, 1},
                          {"config 2: 35x60, 1 chain per launch", 35, 60, 1},   {"config 2: 35x60, 64 chains per launch", 35, 60, 64},
                          {"config 2: 35x60, 4096 chains per launch", 35, 60, 4096},
                          {"config 3: 1-D n=128, 64 chains per launch", 1, 128, 64}, {"config 3: 1-D n=144, 64 chains per launch", 1, 144, 64},
                          {"config 3: 1-D n=160, 64 chains per launch", 1, 160, 64}, {"config 3: 1-D n=176, 64 chains per launch", 1, 176, 64},
                          {"config 3: 1-D n=192, 64 chains per launch", 1, 192, 64}};
    hipStream_t st;
    CK(hipStreamCreate(&st));
    gpt_ctx* ctx = nullptr;
    if (gpt_create(0, st, &ctx) != GPT_OK) return 1;
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd;
    const int iters = 2000;
    for (const Case& c : cases) {
        const int dim1 = c.H == 1;
        const size_t hw = (size_t)(c.H * c.W), hwf = (size_t)(c.H * (c.W / 2 + 1)), B = (size_t)c.batch;
        std::vector<double> y(B * hw), loc(hw), cov(B * hwf), lp(B), gy(B * hw), gc(B * hwf);
        for (auto& v : y) v = nd(g);
        for (auto& v : loc) v = nd(g);
        for (auto& v : cov) v = 1.0 + 0.5 * std::fabs(nd(g)); // every chain its own spectrum
        double *dy, *dl, *dc, *dlp, *dgy, *dgc, *dmn;
        CK(hipMalloc(&dy, B * hw * 8)); CK(hipMalloc(&dl, hw * 8)); CK(hipMalloc(&dc, B * hwf * 8));
        CK(hipMalloc(&dlp, B * 8)); CK(hipMalloc(&dgy, B * hw * 8)); CK(hipMalloc(&dgc, B * hwf * 8)); CK(hipMalloc(&dmn, B * 8));
        CK(hipMemcpy(dy, y.data(), B * hw * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dl, loc.data(), hw * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dc, cov.data(), B * hwf * 8, hipMemcpyHostToDevice));
        auto dev_call = [&] {
            GK(gpt_gp_rfft2_lpdf_vjp(ctx, dy, (int64_t)hw, dl, 0, dc, (int64_t)hwf, c.H, c.W, c.batch, dim1, dlp, dgy, dgc, dmn));
        };
        for (int i = 0; i < 20; ++i) dev_call();
        CK(hipStreamSynchronize(st));
        double t0 = now();
        for (int i = 0; i < iters; ++i) dev_call();
        CK(hipStreamSynchronize(st));
        const double us_dev = 1e6 * (now() - t0) / iters;
        // the allocation-free entry point under stream capture
        double us_graph = -1.0;
        {
            hipGraph_t graph;
            hipGraphExec_t exec;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            const int rc = gpt_gp_rfft2_lpdf_vjp_noalloc(ctx, dy, (int64_t)hw, dl, 0, dc, (int64_t)hwf, c.H, c.W, c.batch, dim1, dlp, dgy, dgc, dmn);
            CK(hipStreamEndCapture(st, &graph));
            if (rc == GPT_OK) {
                CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(exec, st));
                CK(hipStreamSynchronize(st));
                t0 = now();
                for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(exec, st));
                CK(hipStreamSynchronize(st));
                us_graph = 1e6 * (now() - t0) / iters;
                CK(hipGraphExecDestroy(exec));
            }
            CK(hipGraphDestroy(graph));
        }
        // the Stan-side call: one chain, host pointers, column-major, synchronous
        // (and row-major: what a C or NumPy caller passes; a 1-D call has no layout)
        double us_host = -1.0, us_host_rm = -1.0;
        if (c.batch == 1) {
            for (int layout : {GPT_COL_MAJOR, GPT_ROW_MAJOR}) {
                auto host_call = [&] {
                    const int rc = gpt_gp_rfft2_lpdf_host(ctx, layout, y.data(), loc.data(), cov.data(), c.H, c.W, dim1, lp.data(), gy.data(), gc.data());
                    if (rc != GPT_OK) { printf("host call -> %d: %s\n", rc, gpt_last_error(ctx)); exit(1); }
                };
                for (int i = 0; i < 20; ++i) host_call();
                t0 = now();
                for (int i = 0; i < 500; ++i) host_call();
                (layout == GPT_COL_MAJOR ? us_host : us_host_rm) = 1e6 * (now() - t0) / 500;
            }
        }
        // the same density from kernel parameters (Matern 3/2 + nugget): one host call (gpt_gp_rfft2_lpdf_kernel_host)
        // against the three the array interface needs (spectrum, density, the spectrum's reverse mode)
        double us_khost = -1.0, us_3host = -1.0;
        if (c.batch == 1) {
            const double par1[3] = {1.1, 0.1 * c.W, (double)c.W}, par2[5] = {1.1, 2.5, 3.0, (double)c.H, (double)c.W};
            const double* par = dim1 ? par1 : par2;
            double kb[5];
            std::vector<double> cv(cov.size());
            auto khost = [&] {
                const int rc = gpt_gp_rfft2_lpdf_kernel_host(ctx, GPT_COL_MAJOR, 1, 1.5, y.data(), loc.data(), par, 1e-9, c.H, c.W, dim1, lp.data(),
                                                             gy.data(), kb);
                if (rc != GPT_OK) { printf("host call -> %d: %s\n", rc, gpt_last_error(ctx)); exit(1); }
            };
            auto three = [&] {
                GK(gpt_gp_periodic_matern_cov_rfft2_host(ctx, GPT_COL_MAJOR, 1.5, c.H, c.W, par, dim1, cv.data()));
                for (auto& v : cv) v += 1e-9;
                GK(gpt_gp_rfft2_lpdf_host(ctx, GPT_COL_MAJOR, y.data(), loc.data(), cv.data(), c.H, c.W, dim1, lp.data(), gy.data(), gc.data()));
                GK(gpt_gp_periodic_matern_cov_rfft2_vjp_host(ctx, GPT_COL_MAJOR, 1.5, c.H, c.W, par, gc.data(), dim1, kb));
            };
            for (int i = 0; i < 20; ++i) khost();
            t0 = now();
            for (int i = 0; i < 500; ++i) khost();
            us_khost = 1e6 * (now() - t0) / 500;
            for (int i = 0; i < 20; ++i) three();
            t0 = now();
            for (int i = 0; i < 500; ++i) three();
            us_3host = 1e6 * (now() - t0) / 500;
        }
        printf("{\"case\": \"%s\", \"us_per_call_device_ptr\": %.2f, \"evals_per_s_device_ptr\": %.0f, \"us_per_graph_replay_noalloc\": %.2f, "
               "\"us_per_call_host_ptr_col_major\": %.2f, \"us_per_call_host_ptr_row_major\": %.2f, "
               "\"from_kernel_parameters\": {\"us_per_call_host_ptr_col_major\": %.2f, \"us_three_host_calls_array_interface\": %.2f}}\n",
               c.label, us_dev, c.batch / (1e-6 * us_dev), us_graph, us_host, us_host_rm, us_khost, us_3host);
        fflush(stdout);
        CK(hipFree(dy)); CK(hipFree(dl)); CK(hipFree(dc)); CK(hipFree(dlp)); CK(hipFree(dgy)); CK(hipFree(dgc)); CK(hipFree(dmn));
    }
    // The non-centred model density of docs/trees (trees.stan:31-49): z -> f = gp_inv_rfft2(z, loc, cov) -> negative-binomial
    // counts on the 25 x 50 corner of the 35 x 60 padded lattice (10 % masked) -> reverse mode, white-noise prior included.
    // Device pointers (1 / 64 / 4096 chains per launch) and the host-pointer call Stan makes (column-major, one chain).
    for (int64_t batch : {1, 64, 4096}) {
        const int64_t H = 35, W = 60, oh = 25, ow = 50, Wf = W / 2 + 1, n = H * W;
        std::vector<double> z(batch * n), cov(H * Wf), obs(oh * ow), th(batch, 1.7);
        for (auto& v : z) v = nd(g);
        for (auto& v : cov) v = std::exp(0.3 * nd(g));
        for (auto& v : obs) { v = std::floor(std::fabs(2.0 * nd(g))); if (std::fabs(nd(g)) > 1.64) v = -1.0; }
        double *dz, *dc, *dob, *dth, *dlp, *dgz, *dgc, *dgt, *dgl, *dmn, *dloc;
        CK(hipMalloc(&dz, z.size() * 8)); CK(hipMalloc(&dc, cov.size() * 8)); CK(hipMalloc(&dob, obs.size() * 8));
        CK(hipMalloc(&dth, batch * 8)); CK(hipMalloc(&dlp, batch * 8)); CK(hipMalloc(&dgz, z.size() * 8));
        CK(hipMalloc(&dgc, batch * H * Wf * 8)); CK(hipMalloc(&dgt, batch * 8)); CK(hipMalloc(&dgl, batch * 8));
        CK(hipMalloc(&dmn, batch * 8)); CK(hipMalloc(&dloc, n * 8));
        std::vector<double> loc(n, 0.5);
        CK(hipMemcpy(dz, z.data(), z.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, cov.data(), cov.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dob, obs.data(), obs.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dth, th.data(), batch * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dloc, loc.data(), n * 8, hipMemcpyHostToDevice));
        auto call = [&] {
            GK(gpt_gp_inv_rfft2_obs_lpdf_vjp(ctx, GPT_OBS_NEG_BINOMIAL_2_LOG, dz, n, dloc, 0, dc, 0, dob, 0, oh, ow, dth, 1, H, W, batch, 0,
                                             dlp, nullptr, dgz, nullptr, dgl, dgc, dgt, dmn));
        };
        for (int i = 0; i < 20; ++i) call();
        CK(hipStreamSynchronize(st));
        const int it = batch > 1000 ? 300 : 2000;
        double t0 = now();
        for (int i = 0; i < it; ++i) call();
        CK(hipStreamSynchronize(st));
        const double us_dev = 1e6 * (now() - t0) / it;
        double us_host = -1.0;
        if (batch == 1) {
            double lp, gt, gl;
            std::vector<double> gz(n), gc(H * Wf);
            auto host_call = [&] {
                const int rc = gpt_gp_inv_rfft2_obs_lpdf_host(ctx, GPT_COL_MAJOR, GPT_OBS_NEG_BINOMIAL_2_LOG, z.data(), nullptr, 0.5, cov.data(),
                                                              obs.data(), oh, ow, 1.7, 1, H, W, 0, &lp, gz.data(), &gl, gc.data(), &gt);
                if (rc != GPT_OK) { printf("host call -> %d: %s\n", rc, gpt_last_error(ctx)); exit(1); }
            };
            for (int i = 0; i < 20; ++i) host_call();
            t0 = now();
            for (int i = 0; i < 500; ++i) host_call();
            us_host = 1e6 * (now() - t0) / 500;
        }
        // ... and from the kernel parameters (Matern 3/2 + nugget, trees.stan:26-29): one call instead of spectrum + density +
        // the spectrum's reverse mode
        std::vector<double> par(batch * 5);
        for (int64_t q = 0; q < batch; ++q) { par[5 * q] = 1.1; par[5 * q + 1] = 2.5; par[5 * q + 2] = 3.0; par[5 * q + 3] = (double)H; par[5 * q + 4] = (double)W; }
        double *dpar, *dkb;
        CK(hipMalloc(&dpar, par.size() * 8)); CK(hipMalloc(&dkb, par.size() * 8));
        CK(hipMemcpy(dpar, par.data(), par.size() * 8, hipMemcpyHostToDevice));
        auto kcall = [&] {
            GK(gpt_gp_inv_rfft2_obs_lpdf_kernel_vjp(ctx, GPT_OBS_NEG_BINOMIAL_2_LOG, 1, 1.5, dz, n, dloc, 0, dpar, 1e-9, dob, 0, oh, ow, dth, 1,
                                                    H, W, batch, 0, dlp, dgz, dgl, dkb, dgt, dmn));
        };
        for (int i = 0; i < 20; ++i) kcall();
        CK(hipStreamSynchronize(st));
        t0 = now();
        for (int i = 0; i < it; ++i) kcall();
        CK(hipStreamSynchronize(st));
        const double us_kdev = 1e6 * (now() - t0) / it;
        double us_khost = -1.0, us_3host = -1.0;
        if (batch == 1) {
            double lp, gt, gl, kb[5];
            std::vector<double> gz(n), gc(H * Wf), cv(H * Wf);
            auto khost = [&] {
                const int rc = gpt_gp_inv_rfft2_obs_lpdf_kernel_host(ctx, GPT_COL_MAJOR, GPT_OBS_NEG_BINOMIAL_2_LOG, 1, 1.5, z.data(), nullptr, 0.5,
                                                                     par.data(), 1e-9, obs.data(), oh, ow, 1.7, 1, H, W, 0, &lp, gz.data(), &gl, kb, &gt);
                if (rc != GPT_OK) { printf("host call -> %d: %s\n", rc, gpt_last_error(ctx)); exit(1); }
            };
            for (int i = 0; i < 20; ++i) khost();
            t0 = now();
            for (int i = 0; i < 500; ++i) khost();
            us_khost = 1e6 * (now() - t0) / 500;
            // what the array interface costs a Stan model per gradient evaluation: three host calls
            auto three = [&] {
                GK(gpt_gp_periodic_matern_cov_rfft2_host(ctx, GPT_COL_MAJOR, 1.5, H, W, par.data(), 0, cv.data()));
                for (auto& v : cv) v += 1e-9;
                GK(gpt_gp_inv_rfft2_obs_lpdf_host(ctx, GPT_COL_MAJOR, GPT_OBS_NEG_BINOMIAL_2_LOG, z.data(), nullptr, 0.5, cv.data(), obs.data(), oh, ow,
                                                  1.7, 1, H, W, 0, &lp, gz.data(), &gl, gc.data(), &gt));
                GK(gpt_gp_periodic_matern_cov_rfft2_vjp_host(ctx, GPT_COL_MAJOR, 1.5, H, W, par.data(), gc.data(), 0, kb));
            };
            for (int i = 0; i < 20; ++i) three();
            t0 = now();
            for (int i = 0; i < 500; ++i) three();
            us_3host = 1e6 * (now() - t0) / 500;
        }
        CK(hipFree(dpar)); CK(hipFree(dkb));
        printf("{\"case\": \"trees model density: 35x60, 25x50 negative-binomial counts, %lld chain(s) per launch\", "
               "\"us_per_call_device_ptr\": %.2f, \"evals_per_s_device_ptr\": %.0f, \"us_per_call_host_ptr_col_major\": %.2f, "
               "\"from_kernel_parameters\": {\"us_per_call_device_ptr\": %.2f, \"evals_per_s_device_ptr\": %.0f, "
               "\"us_per_call_host_ptr_col_major\": %.2f, \"us_three_host_calls_array_interface\": %.2f}}\n",
               (long long)batch, us_dev, batch / (1e-6 * us_dev), us_host, us_kdev, batch / (1e-6 * us_kdev), us_khost, us_3host);
        fflush(stdout);
        CK(hipFree(dz)); CK(hipFree(dc)); CK(hipFree(dob)); CK(hipFree(dth)); CK(hipFree(dlp)); CK(hipFree(dgz)); CK(hipFree(dgc));
        CK(hipFree(dgt)); CK(hipFree(dgl)); CK(hipFree(dmn)); CK(hipFree(dloc));
    }
    gpt_destroy(ctx);
    return 0;
}
