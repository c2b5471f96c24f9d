// Drives every function of include/gptools_hip_bind.hpp (the Stan-free part of the cmdstan binding) on the GPU and
// writes the results to a file that tests/test_gpu_bind_cpp.py compares with the CPU oracle. Inputs come from a
// 64-bit LCG that the Python side replays, arrays are COLUMN-major (Stan / Eigen order).
//   usage: bind_driver <output file>
// Record format: int64 name length, name bytes, int64 count, count doubles.
#include <gptools_hip_bind.hpp>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace g = gptools_hip;
using g::cplx;

static uint64_t lcg_state = 0;
static double lcg() { // uniform in (-1, 1)
    lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(lcg_state >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}
static std::vector<double> rnd(size_t n) {
    std::vector<double> v(n);
    for (auto& x : v) x = lcg();
    return v;
}
static std::vector<double> rnd_pos(size_t n) { // spectra: 0.5 .. 2.5
    std::vector<double> v(n);
    for (auto& x : v) x = 1.5 + lcg();
    return v;
}
static FILE* out_file = nullptr;
static void put(const std::string& name, const double* p, size_t n) {
    const int64_t ln = (int64_t)name.size(), cnt = (int64_t)n;
    fwrite(&ln, 8, 1, out_file);
    fwrite(name.data(), 1, name.size(), out_file);
    fwrite(&cnt, 8, 1, out_file);
    fwrite(p, 8, n, out_file);
}
static void put(const std::string& name, const std::vector<double>& v) { put(name, v.data(), v.size()); }
static void put(const std::string& name, const std::vector<cplx>& v) { put(name, g::re(v.data()), 2 * v.size()); }
static std::vector<cplx> as_cplx(const std::vector<double>& v) {
    std::vector<cplx> c(v.size() / 2);
    memcpy(static_cast<void*>(c.data()), v.data(), v.size() * 8);
    return c;
}

static void run_case(int64_t H, int64_t W, int dim1, uint64_t seed) {
    const g::Shape s{H, W, dim1};
    const std::string tag = (dim1 ? "n" + std::to_string(W) : std::to_string(H) + "x" + std::to_string(W)) + "/";
    lcg_state = seed;
    const size_t n = (size_t)s.n(), nf = (size_t)s.nf();
    // inputs, in this order (the Python side draws the same stream)
    const auto y = rnd(n), loc = rnd(n), z = rnd(n), bar = rnd(n);
    const auto cov = rnd_pos(nf), barf = rnd(nf);
    const auto zc = as_cplx(rnd(2 * nf)), barc = as_cplx(rnd(2 * nf));
    // real-FFT layer
    std::vector<cplx> F(nf);
    g::rfft2(y.data(), s, F.data());
    put(tag + "rfft2", F);
    std::vector<double> x(n);
    g::inv_rfft2(zc.data(), s, x.data());
    put(tag + "inv_rfft2", x);
    g::rfft2_vjp(barc.data(), s, x.data());
    put(tag + "rfft2_vjp", x);
    g::inv_rfft2_vjp(bar.data(), s, F.data());
    put(tag + "inv_rfft2_vjp", F);
    if (dim1) {
        std::vector<cplx> full((size_t)W), half((size_t)s.wf());
        g::expand_rfft(zc.data(), s.wf(), W, full.data());
        put(tag + "expand_rfft", full);
        const auto fullbar = as_cplx(rnd(2 * (size_t)W));
        g::expand_rfft_vjp(fullbar.data(), W, half.data());
        put(tag + "expand_rfft_vjp", half);
    }
    // scales and layout helpers
    std::vector<double> sc(nf), cb(nf);
    g::gp_evaluate_scale(cov.data(), s, sc.data());
    put(tag + "scale", sc);
    g::gp_evaluate_scale_vjp(cov.data(), sc.data(), barf.data(), s, cb.data());
    put(tag + "scale_vjp", cb);
    g::gp_unpack(zc.data(), s, x.data());
    put(tag + "unpack", x);
    g::gp_pack(z.data(), s, F.data());
    put(tag + "pack", F);
    g::gp_unpack_vjp(bar.data(), s, F.data());
    put(tag + "unpack_vjp", F);
    g::gp_pack_vjp(barc.data(), s, x.data());
    put(tag + "pack_vjp", x);
    // GP functions and their reverse modes
    std::vector<double> abar(n);
    g::gp_transform(true, y.data(), loc.data(), cov.data(), s, x.data());
    put(tag + "gp_rfft", x);
    g::gp_transform_vjp(true, y.data(), loc.data(), cov.data(), bar.data(), s, abar.data(), cb.data());
    put(tag + "gp_rfft_vjp_y", abar);
    put(tag + "gp_rfft_vjp_cov", cb);
    g::gp_transform(false, z.data(), loc.data(), cov.data(), s, x.data());
    put(tag + "gp_inv_rfft", x);
    g::gp_transform_vjp(false, z.data(), loc.data(), cov.data(), bar.data(), s, abar.data(), cb.data());
    put(tag + "gp_inv_rfft_vjp_z", abar);
    put(tag + "gp_inv_rfft_vjp_cov", cb);
    double v = g::gp_lpdf(y.data(), loc.data(), cov.data(), s, nullptr, nullptr);
    put(tag + "lpdf_value_only", &v, 1);
    v = g::gp_lpdf(y.data(), loc.data(), cov.data(), s, abar.data(), cb.data());
    put(tag + "lpdf", &v, 1);
    put(tag + "lpdf_grad_y", abar);
    put(tag + "lpdf_grad_cov", cb);
    v = g::gp_ladj(cov.data(), s, cb.data());
    put(tag + "ladj", &v, 1);
    put(tag + "ladj_grad_cov", cb);
    // spectral kernels
    const double par1[3] = {1.3, 0.11 * W, 1.0 * W}, par2[5] = {1.3, 0.09 * H, 0.12 * W, 1.0 * H, 1.1 * W};
    const double* par = dim1 ? par1 : par2;
    const size_t np = dim1 ? 3 : 5;
    double pbar[5];
    for (int matern = 0; matern < 2; ++matern) {
        const std::string k = matern ? "matern32" : "expquad";
        g::spectrum(matern, 1.5, s, par, sc.data());
        put(tag + k, sc);
        g::spectrum_vjp(matern, 1.5, s, par, barf.data(), pbar);
        put(tag + k + "_vjp", pbar, np);
    }
    // gp_rfft(2)_lpdf over the Matern-3/2 spectrum + a nugget, from the kernel parameters
    v = g::gp_kernel_lpdf(true, 1.5, y.data(), loc.data(), par, 1e-4, s, abar.data(), pbar);
    put(tag + "klpdf", &v, 1);
    put(tag + "klpdf_grad_y", abar);
    put(tag + "klpdf_grad_par", pbar, np);
    // the non-centred model density of docs/trees/trees.stan:31-49 (negative binomial, white-noise prior)
    if (!dim1) {
        const int64_t oh = H - H / 4, ow = W - W / 4;
        std::vector<double> obs((size_t)(oh * ow));
        for (auto& o : obs) o = (double)(int)(3.0 + 3.0 * lcg()); // counts 0..5
        obs[1] = -1.0;                                           // a held-out quadrat (trees.stan:45)
        g::ObsGrad og;
        v = g::obs_lpdf(GPT_OBS_NEG_BINOMIAL_2_LOG, z.data(), 0.35, cov.data(), obs.data(), oh, ow, 2.5, 1, s, abar.data(),
                        cb.data(), &og);
        put(tag + "obs", obs);
        put(tag + "obs_lpdf", &v, 1);
        put(tag + "obs_grad_z", abar);
        put(tag + "obs_grad_cov", cb);
        const double gg[2] = {og.loc_sum, og.theta};
        put(tag + "obs_grad_loc_theta", gg, 2);
        // ... the same with the spectrum given by its Matern-3/2 parameters + a nugget (trees.stan:26-29)
        double kb[5];
        v = g::obs_kernel_lpdf(GPT_OBS_NEG_BINOMIAL_2_LOG, true, 1.5, z.data(), 0.35, par2, 1e-4, obs.data(), oh, ow, 2.5, 1, s,
                               abar.data(), kb, &og);
        put(tag + "obsk_lpdf", &v, 1);
        put(tag + "obsk_grad_z", abar);
        put(tag + "obsk_grad_par", kb, 5);
        const double gk[2] = {og.loc_sum, og.theta};
        put(tag + "obsk_grad_loc_theta", gk, 2);
    }
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    out_file = fopen(argv[1], "wb");
    if (!out_file) return 3;
    try {
        run_case(1, 8, 1, 11);
        run_case(1, 7, 1, 12);
        run_case(1, 128, 1, 13);
        run_case(5, 8, 0, 21);
        run_case(6, 7, 0, 22);
        run_case(35, 60, 0, 23);
        run_case(256, 512, 0, 24); // three-pass path
        // error mapping: the reference's rejects
        double status[3] = {0, 0, 0};
        {
            const g::Shape s = g::shape2(4, 6);
            std::vector<double> cov(16, 1.0), y(24, 0.0), sc(16);
            cov[5] = -0.25;
            try {
                g::gp_evaluate_scale(cov.data(), s, sc.data());
            } catch (const std::domain_error& e) {
                status[0] = std::string(e.what()).find("covariance matrix is not positive-definite (minimum eigenvalue is -3)") !=
                            std::string::npos;
            }
            try {
                g::gp_lpdf(y.data(), y.data(), cov.data(), s, nullptr, nullptr);
            } catch (const std::domain_error& e) {
                status[1] = std::string(e.what()).find("not positive-definite") != std::string::npos;
            }
            std::vector<cplx> half(4), full(4);
            try {
                g::expand_rfft(half.data(), 4, 4, full.data());
            } catch (const std::domain_error& e) {
                status[2] = std::string(e.what()) == "expected complex vector with 3 elements but got 4";
            }
        }
        put("rejects", status, 3);
    } catch (const std::exception& e) {
        fprintf(stderr, "bind_driver: %s\n", e.what());
        fclose(out_file);
        return 1;
    }
    fclose(out_file);
    return 0;
}
