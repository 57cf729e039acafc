// update(alpha, beta, lA, mu, sigma, x) for an ARBITRARY transition list, on device, from
// materialised alpha/beta (reference src/baumwelch.jl:205-309).  Used for overlap models and for
// API parity of `update`; the ring engine never materialises alpha/beta.
//
// The reference's sequential log-sum-exp folds over states/time are replaced by max+sum-exp
// reductions (same quantity; differences at the 1e-15 relative level, tolerance 1e-6).
#include <cmath>

#include "generic_dev.h"
#include "hmmsort_internal.h"

namespace hmmsort {

__device__ __forceinline__ double wave_max(double v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// block-wide reductions; result valid in every thread.  red: >= 16 doubles of LDS.
__device__ double block_max(double v, double *red)
{
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double r = red[0];
    for (int i = 1; i < nw; i++) r = fmax(r, red[i]);
    return r;
}
__device__ double block_sum(double v, double *red)
{
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double r = red[0];
    for (int i = 1; i < nw; i++) r += red[i];
    return r;
}

// (1) gamma  baumwelch.jl:216-224: one block iteration per t
__global__ void upd_gamma(const double *__restrict__ alpha, const double *__restrict__ beta,
                          int64_t T, int S, double *__restrict__ gam)
{
    __shared__ double red[16];
    for (int64_t t = blockIdx.x; t < T; t += gridDim.x) {
        const double *a = alpha + (int64_t)S * t, *b = beta + (int64_t)S * t;
        double m = -INFINITY;
        for (int j = threadIdx.x; j < S; j += blockDim.x) m = fmax(m, a[j] + b[j]);
        m = block_max(m, red);
        double s = 0.0;
        for (int j = threadIdx.x; j < S; j += blockDim.x) s += exp((a[j] + b[j]) - m);
        s = block_sum(s, red);
        const double g = m + log(s);
        for (int j = threadIdx.x; j < S; j += blockDim.x)
            gam[j + (int64_t)S * t] = (a[j] + b[j]) - g;
    }
}

// (2) xi rows for the transitions leaving state 1, normalised by the all-transition total
//     baumwelch.jl:229-253.  tr_* is the source-major list itself (out_* CSR arrays).
__global__ void upd_xi(const double *__restrict__ alpha, const double *__restrict__ beta,
                       const double *__restrict__ x, int64_t T, int S, int R,
                       const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ out_dst,
                       const double *__restrict__ out_lp, const double *__restrict__ mean,
                       double c0, double den, int nsrc1, double *__restrict__ xi)
{
    __shared__ double red[16];
    extern __shared__ int32_t srcof[];  // R entries: source state of list entry e
    for (int j = threadIdx.x; j < S; j += blockDim.x)
        for (int e = out_ptr[j]; e < out_ptr[j + 1]; e++) srcof[e] = j;
    __syncthreads();
    for (int64_t t = blockIdx.x; t < T - 1; t += gridDim.x) {
        const double *a = alpha + (int64_t)S * t, *b = beta + (int64_t)S * (t + 1);
        const double xv = x[t + 1];
        double m = -INFINITY;
        for (int e = threadIdx.x; e < R; e += blockDim.x) {
            const int j = out_dst[e];
            const double dd = xv - mean[j];
            const double v = ((a[srcof[e]] + out_lp[e]) + b[j]) + (c0 - (dd * dd) / den);
            m = fmax(m, v);
        }
        m = block_max(m, red);
        double s = 0.0;
        for (int e = threadIdx.x; e < R; e += blockDim.x) {
            const int j = out_dst[e];
            const double dd = xv - mean[j];
            const double v = ((a[srcof[e]] + out_lp[e]) + b[j]) + (c0 - (dd * dd) / den);
            s += exp(v - m);
        }
        s = block_sum(s, red);
        const double q = m + log(s);
        for (int i = threadIdx.x; i < nsrc1; i += blockDim.x) {  // entries 0..nsrc1-1 leave state 1
            const int j = out_dst[i];
            const double dd = xv - mean[j];
            xi[i + (int64_t)nsrc1 * t] = (((a[0] + out_lp[i]) + b[j]) + (c0 - (dd * dd) / den)) - q;
        }
    }
}

// (3) log-sum-exp over time of one row: row r < nsrc1 -> xi[r,:], row nsrc1 -> gamma[1,:]
//     baumwelch.jl:254-261; both over t = 1..T-1
__global__ void upd_lse_time(const double *__restrict__ xi, const double *__restrict__ gam,
                             int64_t T, int S, int nsrc1, double *__restrict__ out)
{
    __shared__ double red[16];
    const int r = blockIdx.x;
    const double *base = (r < nsrc1) ? xi + r : gam;
    const int64_t stride = (r < nsrc1) ? nsrc1 : S;
    double m = -INFINITY;
    for (int64_t t = threadIdx.x; t < T - 1; t += blockDim.x) m = fmax(m, base[stride * t]);
    m = block_max(m, red);
    double s = 0.0;
    for (int64_t t = threadIdx.x; t < T - 1; t += blockDim.x) s += exp(base[stride * t] - m);
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[r] = m + log(s);
}

// (4) per-state sums over time: num[j] = sum_t x_t e^gamma, den[j] = sum_t e^gamma
//     baumwelch.jl:270-282
__global__ void upd_state_sums(const double *__restrict__ gam, const double *__restrict__ x,
                               int64_t T, int S, double *__restrict__ num,
                               double *__restrict__ den)
{
    __shared__ double red[16];
    const int j = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int64_t t = threadIdx.x; t < T; t += blockDim.x) {
        const double eg = exp(gam[j + (int64_t)S * t]);
        a += x[t] * eg;
        b += eg;
    }
    a = block_sum(a, red);
    b = block_sum(b, red);
    if (threadIdx.x == 0) { num[j] = a; den[j] = b; }
}

// (5) new mu (in place semantics: zero, accumulate, divide rows 2..K) and new per-state means
//     baumwelch.jl:266-293.  One thread; O(S*N).
__global__ void upd_finish_mu(const int16_t *__restrict__ states, int N, int K, int S,
                              const double *__restrict__ num, const double *__restrict__ den,
                              double *__restrict__ gg, double *__restrict__ mu_new,
                              double *__restrict__ mean_new)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < K * N; i++) { mu_new[i] = 0.0; gg[i] = 0.0; }
    for (int j = 0; j < S; j++) {
        int nact = 0;
        for (int l = 0; l < N; l++) nact += (states[l + N * j] >= 2);
        if (nact != 1) continue;  // :269 sidx
        for (int l = 0; l < N; l++) {
            const int ss = states[l + N * j];
            if (ss > 1) {
                mu_new[(ss - 1) + K * l] += num[j];
                gg[(ss - 1) + K * l] += den[j];
            }
        }
    }
    for (int l = 0; l < N; l++)
        for (int j = 1; j < K; j++) mu_new[j + K * l] /= gg[j + K * l];
    for (int j = 0; j < S; j++) {
        double a = 0.0;
        for (int l = 0; l < N; l++) a += mu_new[(states[l + N * j] - 1) + K * l];
        mean_new[j] = a;
    }
}

// (6) variance sums  baumwelch.jl:295-305: per-block partials
__global__ void upd_sigma_partials(const double *__restrict__ gam, const double *__restrict__ x,
                                   int64_t T, int S, const double *__restrict__ mean_new,
                                   double *__restrict__ part)
{
    __shared__ double red[16];
    double x2 = 0.0, qq = 0.0;
    for (int64_t t = blockIdx.x; t < T; t += gridDim.x) {
        const double xv = x[t];
        for (int j = threadIdx.x; j < S; j += blockDim.x) {
            const double eg = exp(gam[j + (int64_t)S * t]);
            const double d = xv - mean_new[j];
            x2 += (d * d) * eg;
            qq += eg;
        }
    }
    x2 = block_sum(x2, red);
    qq = block_sum(qq, red);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = x2; part[2 * blockIdx.x + 1] = qq; }
}

// (7) pack [mu | sigma | xb[2:end] | pp]
__global__ void upd_pack(const double *__restrict__ mu_new, int KN, const double *__restrict__ part,
                         int nparts, const double *__restrict__ lse, int nsrc1,
                         const double *__restrict__ gam, int S, double *__restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < KN; i++) out[i] = mu_new[i];
    double x2 = 0.0, qq = 0.0;
    for (int i = 0; i < nparts; i++) { x2 += part[2 * i]; qq += part[2 * i + 1]; }
    out[KN] = sqrt(x2 / qq);                               // :306-307
    const double bb = lse[nsrc1];
    for (int i = 1; i < nsrc1; i++) out[KN + i] = lse[i] - bb;  // :264 xb[2:end]
    for (int j = 0; j < S; j++) out[KN + nsrc1 + j] = gam[j];   // :263 pp = gammaf[:,1]
}

int generic_update(GenericDev *g, const double *d_alpha, const double *d_beta, const double *d_y,
                   double *d_out, hipStream_t st)
{
    const int64_t T = g->T, S = g->S, R = g->R, K = g->K, N = g->N;
    const int nsrc1 = g->nsrc1;
    HS_CHECK(T >= 2, HMMSORT_EINVAL, "update: need T >= 2");
    HS_CHECK(nsrc1 >= 1, HMMSORT_EINVAL, "update: no transition leaves state 1");
    const int nparts = 512;
    // scratch: gamma S*T | xi nsrc1*(T-1) | lse nsrc1+1 | num S | den S | gg KN | mu KN | mean S | part
    const int64_t n_gam = S * T, n_xi = (int64_t)nsrc1 * (T - 1);
    const int64_t total = n_gam + n_xi + (nsrc1 + 1) + 3 * S + 2 * K * N + 2 * nparts;
    if (g->upd_bytes < total * 8) {
        if (g->d_upd) (void)hipFree(g->d_upd);
        g->d_upd = nullptr;
        if (hipMalloc((void **)&g->d_upd, total * 8) != hipSuccess) {
            (void)hipGetLastError();
            set_error("update: hipMalloc of %.2f GB scratch failed", total * 8 / 1e9);
            return HMMSORT_ENOMEM;
        }
        g->upd_bytes = total * 8;
        g->bytes += total * 8;
    }
    double *gam = g->d_upd, *xi = gam + n_gam, *lse = xi + n_xi, *num = lse + nsrc1 + 1,
           *den = num + S, *mean_new = den + S, *gg = mean_new + S, *mu_new = gg + K * N,
           *part = mu_new + K * N;
    const double c0 = -kLog2Pi - g->lsig;
    const double dn = 2.0 * (g->sigma * g->sigma);
    const int gb = (int)std::min<int64_t>(T, 2048);
    hipLaunchKernelGGL(upd_gamma, dim3(gb), dim3(256), 0, st, d_alpha, d_beta, T, (int)S, gam);
    hipLaunchKernelGGL(upd_xi, dim3(gb), dim3(256), R * sizeof(int32_t), st, d_alpha, d_beta, d_y,
                       T, (int)S, (int)R, g->d_out_ptr, g->d_out_dst, g->d_out_lp, g->d_mean, c0,
                       dn, nsrc1, xi);
    hipLaunchKernelGGL(upd_lse_time, dim3(nsrc1 + 1), dim3(256), 0, st, xi, gam, T, (int)S, nsrc1,
                       lse);
    hipLaunchKernelGGL(upd_state_sums, dim3((int)S), dim3(256), 0, st, gam, d_y, T, (int)S, num,
                       den);
    hipLaunchKernelGGL(upd_finish_mu, dim3(1), dim3(64), 0, st, g->d_states, (int)N, (int)K,
                       (int)S, num, den, gg, mu_new, mean_new);
    hipLaunchKernelGGL(upd_sigma_partials, dim3(nparts), dim3(256), 0, st, gam, d_y, T, (int)S,
                       mean_new, part);
    hipLaunchKernelGGL(upd_pack, dim3(1), dim3(64), 0, st, mu_new, (int)(K * N), part, nparts, lse,
                       nsrc1, gam, (int)S, d_out);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

}  // namespace hmmsort
