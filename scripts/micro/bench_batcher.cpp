// Cross-chain batching without Python in the loop: NCHAINS std::threads, each a chain calling
// gpt_batcher_lpdf_host (batched) or gpt_gp_rfft2_lpdf_host on a context of its own (unbatched), ROUNDS times.
//   build: g++ -O2 -std=c++17 -I include scripts/micro/bench_batcher.cpp -o scripts/micro/bench_batcher \
//          -L gptools_amd -lgptools_hip -Wl,-rpath,'$ORIGIN/../../gptools_amd' -lpthread
//   run:   scripts/micro/bench_batcher <height> <width> <nchains> <rounds>     (height 1: the 1-D functions)
#include "gptools_hip.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

int main(int argc, char** argv) {
    const int64_t H = argc > 1 ? atoll(argv[1]) : 35, W = argc > 2 ? atoll(argv[2]) : 60;
    const int nch = argc > 3 ? atoi(argv[3]) : 64, rounds = argc > 4 ? atoi(argv[4]) : 500;
    const int dim1 = H == 1;
    const size_t hw = (size_t)(H * W), hwf = (size_t)(H * (W / 2 + 1));
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd;
    std::vector<std::vector<double>> y(nch, std::vector<double>(hw)), cov(nch, std::vector<double>(hwf));
    std::vector<double> loc(hw);
    for (auto& v : loc) v = nd(g);
    for (int c = 0; c < nch; ++c) {
        for (auto& v : y[c]) v = nd(g);
        for (auto& v : cov[c]) v = 1.0 + 0.5 * std::fabs(nd(g));
    }
    auto timed = [&](auto fn) {
        std::vector<std::thread> ts;
        const auto t0 = std::chrono::steady_clock::now();
        for (int c = 0; c < nch; ++c) ts.emplace_back(fn, c);
        for (auto& t : ts) t.join();
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    gpt_batcher* b = nullptr;
    if (gpt_batcher_create(0, nch, GPT_ROW_MAJOR, H, W, dim1, &b) != GPT_OK) return 1;
    std::vector<double> check(nch);
    auto batched = [&](int c) {
        std::vector<double> gy(hw), gc(hwf);
        double lp = 0;
        for (int r = 0; r < rounds; ++r)
            if (gpt_batcher_lpdf_host(b, c, y[c].data(), loc.data(), cov[c].data(), &lp, gy.data(), gc.data()) != GPT_OK) exit(2);
        check[c] = lp;
    };
    timed([&](int c) { double lp; std::vector<double> gy(hw), gc(hwf); gpt_batcher_lpdf_host(b, c, y[c].data(), loc.data(), cov[c].data(), &lp, gy.data(), gc.data()); });
    const double tb = timed(batched);
    const long long launches = gpt_batcher_launches(b);
    gpt_batcher_destroy(b);
    std::vector<double> check2(nch);
    auto unbatched = [&](int c) {
        gpt_ctx* ctx = nullptr;
        if (gpt_create(0, nullptr, &ctx) != GPT_OK) exit(3);
        std::vector<double> gy(hw), gc(hwf);
        double lp = 0;
        for (int r = 0; r < rounds + 1; ++r)
            if (gpt_gp_rfft2_lpdf_host(ctx, GPT_ROW_MAJOR, y[c].data(), loc.data(), cov[c].data(), H, W, dim1, &lp, gy.data(), gc.data()) != GPT_OK) exit(4);
        check2[c] = lp;
        gpt_destroy(ctx);
    };
    const double tu = timed(unbatched);
    double dmax = 0;
    for (int c = 0; c < nch; ++c) dmax = std::fmax(dmax, std::fabs(check[c] - check2[c]) / std::fabs(check2[c]));
    printf("{\"case\": \"%lldx%lld, %d chains x %d rounds, C++ threads, host pointers\", \"batched_evals_per_s\": %.0f, "
           "\"us_per_round_batched\": %.1f, \"fused_launches\": %lld, \"unbatched_evals_per_s\": %.0f, "
           "\"max_rel_diff_batched_vs_unbatched\": %.2e}\n",
           (long long)H, (long long)W, nch, rounds, nch * (double)rounds / tb, 1e6 * tb / rounds, launches,
           nch * (double)(rounds + 1) / tu, dmax);
    return 0;
}
