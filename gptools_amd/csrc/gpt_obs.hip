// Observation models on the device: the docs models of the reference draw the latent field
// with the non-centred parameterisation, f = gp_inv_rfft2(z, loc, cov), and then score data against f inside Stan:
//     docs/trees/trees.stan:43-49          frequency[i, j] ~ neg_binomial_2(exp(f[i, j]), 1 / kappa), masked where < 0,
//                                          on the num_rows x num_cols corner of the padded lattice
//     docs/padding/padded.stan:34          y[:observe_first] ~ normal(f[:observe_first], kappa)
//     docs/poisson_regression/poisson_regression_fourier_non_centered.stan:26     y ~ poisson_log(eta)
// With these kernels f and df never cross PCIe: one streaming pass reads f and the data, writes d lp / d f for the
// whole (padded) lattice and reduces the log density and the derivative with respect to the family's parameter.
// HBM-bound: 8 bytes read + 8 written per lattice point, + 8 read per observed point.
#include "../../include/gptools_hip.h"
#include "gpt_common.h"
#include "gpt_host.h"
#include "gpt_obs_dev.h"

namespace {

template <int FAM> __global__ void __launch_bounds__(256) obs_kernel(ObsArgs a, double* partials) {
    const int64_t b = blockIdx.y;
    const int64_t tot = (int64_t)a.H * a.W;
    const double* f = a.f + b * tot;
    const double* y = a.obs + b * a.obs_bs;
    double* fb = a.fbar + b * tot;
    const double th = a.theta ? a.theta[b] : 0.0;
    double k[3];
    obs_consts(FAM, th, k);
    double s[4] = {0.0, 0.0, 0.0, 0.0}; // log density, d/dtheta, sum of d/df, observed points
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < tot; w += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(w / a.W), j = (int)(w - (int64_t)i * a.W);
        double g = 0.0;
        if (i < a.oh && j < a.ow) {
            const double yv = y[(int64_t)i * a.ow + j];
            // masked: NaN for every family; negative counts (trees.stan:45)
            const bool seen = yv == yv && (FAM == GPT_OBS_NORMAL || yv >= 0.0);
            if (seen) {
                const ObsTerm t = obs_term<FAM>(yv, f[w], th, k);
                g = t.df;
                s[0] += t.lp;
                s[1] += t.dth;
                s[2] += t.df;
                s[3] += 1.0;
            }
        }
        fb[w] = g;
    }
    block_reduce_store<4>(s, INFINITY, partials, b);
}

__global__ void obs_combine_kernel(const double* sums, const double* psum, int64_t batch, double* lp, double* grad_theta,
                                   double* grad_loc_sum, double* nobs) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    if (lp) lp[b] = sums[4 * b] + (psum ? psum[b] : 0.0);
    if (grad_theta) grad_theta[b] = sums[4 * b + 1];
    if (grad_loc_sum) grad_loc_sum[b] = sums[4 * b + 2];
    if (nobs) nobs[b] = sums[4 * b + 3];
}

__global__ void fill_kernel(double* p, int64_t n, double v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

} // namespace

int gpt_obs_blocks(int H, int W) {
    int64_t nb = ((int64_t)H * W + 1023) / 1024;
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;
    return (int)nb;
}

hipError_t gpt_launch_obs(const ObsArgs& a, double* partials, hipStream_t st) {
    if (a.batch <= 0) return hipSuccess;
    if (a.batch > 65535) return hipErrorInvalidValue; // callers chunk the batch
    dim3 grid((unsigned)gpt_obs_blocks(a.H, a.W), (unsigned)a.batch), block(256);
    switch (a.family) {
    case GPT_OBS_NORMAL: hipLaunchKernelGGL(obs_kernel<GPT_OBS_NORMAL>, grid, block, 0, st, a, partials); break;
    case GPT_OBS_POISSON_LOG: hipLaunchKernelGGL(obs_kernel<GPT_OBS_POISSON_LOG>, grid, block, 0, st, a, partials); break;
    case GPT_OBS_NEG_BINOMIAL_2_LOG:
        hipLaunchKernelGGL(obs_kernel<GPT_OBS_NEG_BINOMIAL_2_LOG>, grid, block, 0, st, a, partials);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t gpt_launch_obs_combine(const double* sums, const double* psum, int64_t batch, double* lp, double* grad_theta,
                                  double* grad_loc_sum, double* nobs, hipStream_t st) {
    if (batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(obs_combine_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st, sums, psum, batch, lp,
                       grad_theta, grad_loc_sum, nobs);
    return hipGetLastError();
}

// rep_matrix(loc, H, W) of trees.stan:32 for callers that hold a scalar
hipError_t gpt_launch_fill(double* p, int64_t n, double v, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    int64_t nb = (n + 1023) / 1024;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)nb), dim3(256), 0, st, p, n, v);
    return hipGetLastError();
}
