// Generic ("strict") engine: one sequential sweep per signal over an ARBITRARY transition list
// (overlap models included), in the reference's operation order.
//
//   viterbi  : reference src/viterbi.jl:44-98   -- bit-exact path and ll by construction
//              (same fp64 operations in the same order; only + - * / > are used in the loop)
//   forward  : reference src/baumwelch.jl:25-51 -- same fold order; exp/log1p are ROCm's
//   backward : reference src/baumwelch.jl:73-98
//
// One workgroup owns the whole state vector: delta/alpha live in LDS (two S-vectors, ping-pong),
// one barrier per sample.  The kernels are latency-bound by design; they exist for API parity on
// short signals, for overlap models, and as the on-GPU bit-exact check of the ring engine.
// Compiled with -ffp-contract=off (Julia does not contract).
#include <cmath>
#include <cstdio>

#include "generic_dev.h"
#include "hmmsort_internal.h"

namespace hmmsort {

// ------------------------------------------------------------------------------------------
// device math restating reference utils.jl
// ------------------------------------------------------------------------------------------
// funcl(x, mu, sigma, lsigma) utils.jl:4 with the loop-invariant parts hoisted (same values):
//   c0 = -log2pi - lsigma,  den = 2*sigma^2,  result = c0 - (dd*dd)/den
__device__ __forceinline__ double funcl_dev(double x, double mu, double c0, double den)
{
    double dd = x - mu;
    return c0 - (dd * dd) / den;
}

// logsumexpl(xp, yp) utils.jl:24-32
__device__ __forceinline__ double logsumexpl_dev(double xp, double yp)
{
    if (xp > yp) return xp + log1p(exp(yp - xp));
    return yp + log1p(exp(xp - yp));
}

// ------------------------------------------------------------------------------------------
// Viterbi forward sweep.  grid = 1 block.  LDS: 2*S doubles.
// ------------------------------------------------------------------------------------------
__global__ void gen_viterbi_sweep(const double *__restrict__ y, int64_t T, int S,
                                  const double *__restrict__ mean,
                                  const int32_t *__restrict__ in_ptr,
                                  const int32_t *__restrict__ in_src,
                                  const double *__restrict__ in_lp, double c0, double den,
                                  const int32_t *__restrict__ psidx, int npsi,
                                  int16_t *__restrict__ T2, double *__restrict__ last,
                                  double *gbuf)
{
    // state vectors in LDS, or -- when 2*S doubles exceed it (large overlap models) -- in a global
    // scratch that stays in this CU's L1/L2; one workgroup, so __syncthreads() orders the accesses
    extern __shared__ double sh[];
    double *prev = gbuf ? gbuf : sh, *cur = prev + S;
    const int tid = threadIdx.x, nt = blockDim.x;
    // viterbi.jl:55-63  first column: emission only, then T1[1,1] = 0
    {
        const double y0 = y[0];
        for (int j = tid; j < S; j += nt) cur[j] = (j == 0) ? 0.0 : funcl_dev(y0, mean[j], c0, den);
        // (the first column of T2 is never read: viterbi.jl:93-94 walks i = nobs .. 2)
    }
    for (int64_t t = 1; t < T; t++) {
        __syncthreads();
        double *tmp = prev; prev = cur; cur = tmp;
        const double yt = y[t];
        int16_t *psi = T2 + (int64_t)npsi * t;
        for (int j = tid; j < S; j += nt) {
            double best = -INFINITY;  // viterbi.jl:52 fill(-Inf)
            int arg = 1;              // viterbi.jl:53 ones(Int16)
            const int e1 = in_ptr[j + 1];
            for (int e = in_ptr[j]; e < e1; e++) {  // list order == source ascending
                double tt = prev[in_src[e]] + in_lp[e];  // :79
                if (tt > best) {                           // :80 strict
                    best = tt;
                    arg = in_src[e] + 1;
                }
            }
            cur[j] = best + funcl_dev(yt, mean[j], c0, den);  // :85-87
            const int pi = psidx[j];
            if (pi >= 0) psi[pi] = (int16_t)arg;   // single-source states: the pointer is implied
        }
    }
    __syncthreads();
    for (int j = tid; j < S; j += nt) last[j] = cur[j];
}

// argmax (first maximum, viterbi.jl:90) + backtrace (:93-94).  One block; T2 columns are staged
// through LDS in coalesced bulk so the serial walk never waits on HBM.
__global__ void gen_viterbi_backtrace(const int16_t *__restrict__ T2, const double *__restrict__ last,
                                      const int32_t *__restrict__ psidx, int npsi, int idx_in_lds,
                                      int64_t T, int S, int W, int16_t *__restrict__ x)
{
    extern __shared__ int16_t shp[];  // W columns of npsi entries | (idx_in_lds) psidx[S]
    __shared__ int cur_state;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int32_t *pidx = psidx;
    if (idx_in_lds) {
        int32_t *li = reinterpret_cast<int32_t *>(shp + (((size_t)W * npsi + 1) & ~(size_t)1));
        for (int j = tid; j < S; j += nt) li[j] = psidx[j];
        pidx = li;
    }
    if (tid == 0) {
        int best = 0;
        for (int j = 1; j < S; j++)
            if (last[j] > last[best]) best = j;
        cur_state = best + 1;
        x[T - 1] = (int16_t)(best + 1);
    }
    // walk i = T-1 .. 1 in chunks [lo, hi]
    for (int64_t hi = T - 1; hi >= 1; hi -= W) {
        int64_t lo = hi - W + 1;
        if (lo < 1) lo = 1;
        const int64_t ncol = hi - lo + 1;
        __syncthreads();
        const int64_t nel = ncol * npsi;
        const int16_t *srcp = T2 + (int64_t)npsi * lo;
        for (int64_t e = tid; e < nel; e += nt) shp[e] = srcp[e];
        __syncthreads();
        if (tid == 0) {
            int xs = cur_state;
            for (int64_t i = hi; i >= lo; i--) {
                const int pi = pidx[xs - 1];
                xs = pi >= 0 ? (int)shp[(i - lo) * npsi + pi] : -pi;
                x[i - 1] = (int16_t)xs;
            }
            cur_state = xs;
        }
    }
}

// Path values T1[x[t],t] re-accumulated along the decoded path with the reference's op order
// ((T1[k]+lp)+q), then ll summed from i = nobs down to 2 (viterbi.jl:92-96).  One thread.
__global__ void gen_viterbi_ll(const double *__restrict__ y, const int16_t *__restrict__ x,
                               int64_t T, const double *__restrict__ mean,
                               const int32_t *__restrict__ in_ptr,
                               const int32_t *__restrict__ in_src,
                               const double *__restrict__ in_lp, double c0, double den,
                               double *__restrict__ pv, double *__restrict__ ll_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int xp = x[0] - 1;
    double p = (xp == 0) ? 0.0 : funcl_dev(y[0], mean[xp], c0, den);
    pv[0] = p;
    for (int64_t t = 1; t < T; t++) {
        const int xc = x[t] - 1;
        double lp = -INFINITY;
        const int e1 = in_ptr[xc + 1];
        for (int e = in_ptr[xc]; e < e1; e++)
            if (in_src[e] == xp) { lp = in_lp[e]; break; }
        p = (p + lp) + funcl_dev(y[t], mean[xc], c0, den);
        pv[t] = p;
        xp = xc;
    }
    double ll = 0.0;
    for (int64_t i = T - 1; i >= 1; i--) ll += pv[i];
    *ll_out = ll;
}

// ------------------------------------------------------------------------------------------
// forward / backward (materialising, S x T column-major like the reference)
// ------------------------------------------------------------------------------------------
__global__ void gen_forward_sweep(const double *__restrict__ y, int64_t T, int S,
                                  const double *__restrict__ mean,
                                  const int32_t *__restrict__ in_ptr,
                                  const int32_t *__restrict__ in_src,
                                  const double *__restrict__ in_lp, double c0, double den,
                                  double *__restrict__ alpha, double *gbuf)
{
    extern __shared__ double sh[];
    double *prev = gbuf ? gbuf : sh, *cur = prev + S;
    const int tid = threadIdx.x, nt = blockDim.x;
    {
        const double y0 = y[0];
        for (int j = tid; j < S; j += nt) {  // baumwelch.jl:36
            double v = funcl_dev(y0, mean[j], c0, den);
            cur[j] = v;
            alpha[j] = v;
        }
    }
    for (int64_t t = 1; t < T; t++) {
        __syncthreads();
        double *tmp = prev; prev = cur; cur = tmp;
        const double yt = y[t];
        double *at = alpha + (int64_t)S * t;
        for (int j = tid; j < S; j += nt) {
            const double b = funcl_dev(yt, mean[j], c0, den);
            double acc = -INFINITY;  // :28 fill(-Inf)
            const int e1 = in_ptr[j + 1];
            for (int e = in_ptr[j]; e < e1; e++)
                acc = logsumexpl_dev(acc, (prev[in_src[e]] + in_lp[e]) + b);  // :47
            cur[j] = acc;
            at[j] = acc;
        }
    }
}

__global__ void gen_backward_sweep(const double *__restrict__ y, int64_t T, int S,
                                   const double *__restrict__ mean,
                                   const int32_t *__restrict__ out_ptr,
                                   const int32_t *__restrict__ out_dst,
                                   const double *__restrict__ out_lp, double c0, double den,
                                   double *__restrict__ beta, double *gbuf)
{
    extern __shared__ double sh[];
    double *nxt = gbuf ? gbuf : sh, *cur = nxt + S, *bq = nxt + 2 * S;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int j = tid; j < S; j += nt) {  // baumwelch.jl:80
        cur[j] = 0.0;
        beta[j + (int64_t)S * (T - 1)] = 0.0;
    }
    for (int64_t t = T - 2; t >= 0; t--) {
        __syncthreads();
        double *tmp = nxt; nxt = cur; cur = tmp;
        const double v = y[t + 1];
        for (int j = tid; j < S; j += nt) bq[j] = funcl_dev(v, mean[j], c0, den);
        __syncthreads();
        double *bt = beta + (int64_t)S * t;
        for (int j = tid; j < S; j += nt) {
            double acc = -INFINITY;  // :79
            const int e1 = out_ptr[j + 1];
            for (int e = out_ptr[j]; e < e1; e++) {
                const int k = out_dst[e];
                acc = logsumexpl_dev(acc, (nxt[k] + out_lp[e]) + bq[k]);  // :94
            }
            cur[j] = acc;
            bt[j] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------
// reconstruct_signal (reconstruction.jl:1-10) and unroll_mlseq (extraction.jl:4-13)
// ------------------------------------------------------------------------------------------
__global__ void k_reconstruct(const int16_t *__restrict__ x, int64_t T,
                              const int16_t *__restrict__ states, int N, int S,
                              const double *__restrict__ mu, int K, double *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < T; i += stride) {
        int xs = x[i];
        double a = 0.0;
        if (xs >= 1 && xs <= S)
            for (int j = 0; j < N; j++) a += mu[(states[j + N * (xs - 1)] - 1) + K * j];
        else
            a = NAN;
        out[i] = a;
    }
}

__global__ void k_unroll(const int16_t *__restrict__ x, int64_t T,
                         const int16_t *__restrict__ states, int N, int S,
                         int16_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < T; i += stride) {
        int xs = x[i];
        for (int j = 0; j < N; j++)
            out[j + (int64_t)N * i] = (xs >= 1 && xs <= S) ? states[j + N * (xs - 1)] : (int16_t)0;
    }
}

// ------------------------------------------------------------------------------------------
// extract_spiketimes (extraction.jl:15-24): ordered stream compaction.  match[j] has bit i set
// when state j puts neuron i at the row of its template minimum.  Block b owns samples
// [b*CH, (b+1)*CH); pass 0 counts per (block, neuron), the host turns the counts into offsets,
// pass 1 writes the 1-based sample indices in ascending order.
// ------------------------------------------------------------------------------------------
constexpr int kSpikeChunk = 4096;

__global__ __launch_bounds__(256) void k_spike_compact(const int16_t *__restrict__ x, int64_t T,
                                                       const uint32_t *__restrict__ match, int N,
                                                       int S, int pass, int64_t *__restrict__ cnt,
                                                       const int64_t *__restrict__ offs,
                                                       int64_t *__restrict__ times, int64_t cap)
{
    __shared__ int wsum[4];
    __shared__ long long run;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * kSpikeChunk;
    for (int i = 0; i < N; i++) {
        if (threadIdx.x == 0) run = 0;
        __syncthreads();
        for (int sub = 0; sub < kSpikeChunk; sub += 256) {
            const int64_t t = base + sub + threadIdx.x;
            int xs = (t < T) ? x[t] : 0;
            const bool hit = xs >= 1 && xs <= S && ((match[xs - 1] >> i) & 1u);
            const unsigned long long bm = __ballot(hit);
            const int before = __popcll(bm & ((1ull << lane) - 1ull));
            if (lane == 0) wsum[wv] = __popcll(bm);
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wv; w++) woff += wsum[w];
            const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (pass == 1 && hit) {
                const int64_t pos = offs[(int64_t)blockIdx.x * N + i] + run + woff + before;
                if (pos < cap) times[(int64_t)i * cap + pos] = t + 1;
            }
            __syncthreads();
            if (threadIdx.x == 0) run += tot;
            __syncthreads();
        }
        if (pass == 0 && threadIdx.x == 0) cnt[(int64_t)blockIdx.x * N + i] = run;
        __syncthreads();
    }
}

int dev_spike_compact(const int16_t *d_x, int64_t T, const uint32_t *d_match, int N, int S, int pass,
                      int64_t *d_cnt, const int64_t *d_offs, int64_t *d_times, int64_t cap,
                      hipStream_t st)
{
    const int nb = (int)((T + kSpikeChunk - 1) / kSpikeChunk);
    hipLaunchKernelGGL(k_spike_compact, dim3(nb), dim3(256), 0, st, d_x, T, d_match, N, S, pass, d_cnt,
                       d_offs, d_times, cap);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int dev_reconstruct(const int16_t *d_x, int64_t T, const int16_t *d_states, int64_t N, int64_t S,
                    const double *d_mu, int64_t K, double *d_out, hipStream_t st)
{
    if (T <= 0) return HMMSORT_OK;
    int blocks = (int)std::min<int64_t>((T + 255) / 256, 4096);
    hipLaunchKernelGGL(k_reconstruct, dim3(blocks), dim3(256), 0, st, d_x, T, d_states, (int)N,
                       (int)S, d_mu, (int)K, d_out);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int dev_unroll(const int16_t *d_x, int64_t T, const int16_t *d_states, int64_t N, int64_t S,
               int16_t *d_out, hipStream_t st)
{
    if (T <= 0) return HMMSORT_OK;
    int blocks = (int)std::min<int64_t>((T + 255) / 256, 4096);
    hipLaunchKernelGGL(k_unroll, dim3(blocks), dim3(256), 0, st, d_x, T, d_states, (int)N, (int)S,
                       d_out);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

// raw acquisition samples -> fp64 (the reference widens on the host, src/hmmsort.jl:84-88, and hands
// 8 bytes per sample to fit; here 2-byte samples cross PCIe and are widened in HBM).  Every source type
// converts exactly, so the result is the host conversion's bit for bit.
template <typename Tin>
__global__ void k_widen(const Tin *__restrict__ in, int64_t T, int64_t stride, double *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (double)in[i * stride];
}

int dev_widen(const void *d_in, int dtype, int64_t T, int64_t stride, double *d_out, hipStream_t st)
{
    if (T <= 0) return HMMSORT_OK;
    const int blocks = (int)std::min<int64_t>((T + 255) / 256, 8192);
    switch (dtype) {
    case HMMSORT_SAMPLES_I16:
        hipLaunchKernelGGL(k_widen<int16_t>, dim3(blocks), dim3(256), 0, st, (const int16_t *)d_in, T, stride, d_out);
        break;
    case HMMSORT_SAMPLES_I32:
        hipLaunchKernelGGL(k_widen<int32_t>, dim3(blocks), dim3(256), 0, st, (const int32_t *)d_in, T, stride, d_out);
        break;
    case HMMSORT_SAMPLES_F32:
        hipLaunchKernelGGL(k_widen<float>, dim3(blocks), dim3(256), 0, st, (const float *)d_in, T, stride, d_out);
        break;
    case HMMSORT_SAMPLES_F64:
        hipLaunchKernelGGL(k_widen<double>, dim3(blocks), dim3(256), 0, st, (const double *)d_in, T, stride, d_out);
        break;
    default:
        set_error("samples_to_f64: unknown sample type %d", dtype);
        return HMMSORT_EINVAL;
    }
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <typename Tv>
static int upload(Tv **dst, const std::vector<Tv> &v, int64_t *bytes)
{
    if (!*dst) {
        HS_HIP(hipMalloc((void **)dst, std::max<size_t>(v.size(), 1) * sizeof(Tv)));
        *bytes += v.size() * sizeof(Tv);
    }
    if (!v.empty()) HS_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(Tv), hipMemcpyHostToDevice));
    return HMMSORT_OK;
}

int generic_set_model(GenericDev *g, const HostModel &m)
{
    HS_CHECK(m.N == g->N && m.K == g->K && m.S == g->S && m.R == g->R, HMMSORT_EINVAL,
             "set_model: model shape changed");
    g->sigma = m.sigma;
    g->lsig = std::log(m.sigma);  // viterbi.jl:47, on the host so it equals the CPU value
    int rc;
    if ((rc = upload(&g->d_mean, m.mean, &g->bytes))) return rc;
    if ((rc = upload(&g->d_in_ptr, m.in_ptr, &g->bytes))) return rc;
    if ((rc = upload(&g->d_in_src, m.in_src, &g->bytes))) return rc;
    if ((rc = upload(&g->d_in_lp, m.in_lp, &g->bytes))) return rc;
    if ((rc = upload(&g->d_out_ptr, m.out_ptr, &g->bytes))) return rc;
    if ((rc = upload(&g->d_out_dst, m.out_dst, &g->bytes))) return rc;
    if ((rc = upload(&g->d_out_lp, m.out_lp, &g->bytes))) return rc;
    if ((rc = upload(&g->d_mu, m.mu, &g->bytes))) return rc;
    if ((rc = upload(&g->d_states, m.states, &g->bytes))) return rc;
    g->nsrc1 = m.out_ptr[1] - m.out_ptr[0];
    {
        std::vector<int32_t> idx(m.S);
        int np = 0;
        for (int64_t j = 0; j < m.S; j++) {
            const int deg = m.in_ptr[j + 1] - m.in_ptr[j];
            idx[j] = deg > 1 ? np++ : -(deg == 1 ? m.in_src[m.in_ptr[j]] + 1 : 1);
        }
        g->npsi = np > 0 ? np : 1;
        if ((rc = upload(&g->d_psidx, idx, &g->bytes))) return rc;
    }
    if (g->blocked) return blocked_set_model(g, m);
    return HMMSORT_OK;
}

int generic_create(GenericDev **out, const HostModel &m, int64_t T, bool blocked,
                   int64_t block_req, int64_t halo_req)
{
    HS_CHECK(T >= 1, HMMSORT_EINVAL, "generic engine: T must be >= 1");
    GenericDev *g = new GenericDev();
    g->N = m.N; g->K = m.K; g->S = m.S; g->R = m.R; g->T = T;
    int th = (int)((m.S + 63) / 64 * 64);
    g->threads = th < 64 ? 64 : (th > 1024 ? 1024 : th);
    int rc = generic_set_model(g, m);
    if (rc) { generic_destroy(g); return rc; }
    // LDS budget: 3*S doubles (backward sweep) within 150 KiB, else a global scratch
    g->use_global = 3 * m.S * 8 > 150 * 1024;
    if (g->use_global && hipMalloc((void **)&g->d_gbuf, 3 * m.S * sizeof(double)) != hipSuccess) {
        set_error("generic engine: hipMalloc failed");
        generic_destroy(g);
        return HMMSORT_ENOMEM;
    }
    if (hipMalloc((void **)&g->d_last, m.S * sizeof(double)) != hipSuccess) {
        set_error("generic engine: hipMalloc failed");
        generic_destroy(g);
        return HMMSORT_ENOMEM;
    }
    g->bytes += m.S * sizeof(double);
    if (blocked && (rc = blocked_create(g, m, block_req, halo_req))) {
        generic_destroy(g);
        return rc;
    }
    *out = g;
    return HMMSORT_OK;
}

void generic_destroy(GenericDev *g)
{
    if (!g) return;
    void *ptrs[] = {g->d_mean, g->d_in_lp, g->d_out_lp, g->d_mu, g->d_in_ptr, g->d_in_src,
                    g->d_out_ptr, g->d_out_dst, g->d_states, g->d_T2, g->d_pv, g->d_last,
                    g->d_upd, g->d_gbuf, g->d_psidx};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    blocked_destroy(g);
    blocked_estep_destroy(g);
    delete g;
}

int64_t generic_workspace_bytes(const GenericDev *g) { return g->bytes; }
int64_t generic_n_lp(const GenericDev *g) { return g->nsrc1 - 1; }
bool generic_is_blocked(const GenericDev *g) { return g->blocked; }
int64_t blocked_min_samples() { return 4096; }

void generic_geometry(const GenericDev *g, int64_t *block, int64_t *halo, int64_t *nblocks)
{
    *block = g->blocked ? g->B : g->T;
    *halo = g->blocked ? g->H : 0;
    *nblocks = g->blocked ? g->nblk : 1;
}

int generic_diagnostics(GenericDev *g, hipStream_t st, int64_t diag[8])
{
    if (g->blocked) {
        int rc = blocked_diagnostics(g, st, diag);
        if (rc) return rc;
        return blocked_estep_diagnostics(g, st, diag);
    }
    return HMMSORT_OK;
}

static int set_lds_limit(const void *fn, size_t bytes)
{
    if (bytes > 64 * 1024)
        HS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return HMMSORT_OK;
}

int generic_viterbi(GenericDev *g, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st)
{
    const int64_t T = g->T, S = g->S;
    if (g->blocked) return blocked_viterbi(g, d_y, d_x, d_ll, st);
    if (!g->d_T2 || g->t2_rows < g->npsi) {
        if (g->d_T2) { (void)hipFree(g->d_T2); g->d_T2 = nullptr; g->bytes -= g->t2_rows * T * 2; }
        const double need = (double)g->npsi * (double)T * 2.0 + (double)T * 8.0;
        HS_CHECK(need < 200e9, HMMSORT_ENOMEM,
                 "strict Viterbi needs %.1f GB of back-pointers; decode in chunks", need / 1e9);
        if (hipMalloc((void **)&g->d_T2, (size_t)g->npsi * T * sizeof(int16_t)) != hipSuccess ||
            (!g->d_pv && hipMalloc((void **)&g->d_pv, (size_t)T * sizeof(double)) != hipSuccess)) {
            (void)hipGetLastError();
            set_error("strict Viterbi: hipMalloc of %.1f GB failed", need / 1e9);
            return HMMSORT_ENOMEM;
        }
        g->t2_rows = g->npsi;
        g->bytes += (int64_t)need;
    }
    const double c0 = -kLog2Pi - g->lsig;
    const double den = 2.0 * (g->sigma * g->sigma);
    size_t lds = g->use_global ? 0 : 2 * S * sizeof(double);
    int rc = set_lds_limit((const void *)gen_viterbi_sweep, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(gen_viterbi_sweep, dim3(1), dim3(g->threads), lds, st, d_y, T, (int)S,
                       g->d_mean, g->d_in_ptr, g->d_in_src, g->d_in_lp, c0, den, g->d_psidx, g->npsi,
                       g->d_T2, g->d_last, g->use_global ? g->d_gbuf : nullptr);
    HS_HIP(hipGetLastError());
    int W = (int)std::max<int64_t>(1, std::min<int64_t>(4096, (48 * 1024) / ((int64_t)g->npsi * 2)));
    const int idx_in_lds = S * 4 <= 96 * 1024 ? 1 : 0;   // the per-state row index beside the staged columns
    size_t lds2 = (((size_t)W * g->npsi + 1) & ~(size_t)1) * sizeof(int16_t) + (idx_in_lds ? (size_t)S * 4 : 0);
    rc = set_lds_limit((const void *)gen_viterbi_backtrace, lds2);
    if (rc) return rc;
    hipLaunchKernelGGL(gen_viterbi_backtrace, dim3(1), dim3(256), lds2, st, g->d_T2, g->d_last, g->d_psidx,
                       g->npsi, idx_in_lds, T, (int)S, W, d_x);
    HS_HIP(hipGetLastError());
    hipLaunchKernelGGL(gen_viterbi_ll, dim3(1), dim3(64), 0, st, d_y, d_x, T, g->d_mean,
                       g->d_in_ptr, g->d_in_src, g->d_in_lp, c0, den, g->d_pv, d_ll);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int generic_forward(GenericDev *g, const double *d_y, double *d_alpha, hipStream_t st)
{
    // funcl 3-arg form (utils.jl:3): -log2pi - log(sigma) - ...; same constant as the 4-arg form
    const double c0 = -kLog2Pi - g->lsig;
    const double den = 2.0 * (g->sigma * g->sigma);
    size_t lds = g->use_global ? 0 : 2 * g->S * sizeof(double);
    int rc = set_lds_limit((const void *)gen_forward_sweep, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(gen_forward_sweep, dim3(1), dim3(g->threads), lds, st, d_y, g->T, (int)g->S,
                       g->d_mean, g->d_in_ptr, g->d_in_src, g->d_in_lp, c0, den, d_alpha,
                       g->use_global ? g->d_gbuf : nullptr);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int generic_backward(GenericDev *g, const double *d_y, double *d_beta, hipStream_t st)
{
    const double c0 = -kLog2Pi - g->lsig;
    const double den = 2.0 * (g->sigma * g->sigma);
    size_t lds = g->use_global ? 0 : 3 * g->S * sizeof(double);
    int rc = set_lds_limit((const void *)gen_backward_sweep, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(gen_backward_sweep, dim3(1), dim3(g->threads), lds, st, d_y, g->T,
                       (int)g->S, g->d_mean, g->d_out_ptr, g->d_out_dst, g->d_out_lp, c0, den,
                       d_beta, g->use_global ? g->d_gbuf : nullptr);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

}  // namespace hmmsort
