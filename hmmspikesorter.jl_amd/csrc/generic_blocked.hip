// Blocked generic engine: the Viterbi recursion of reference src/viterbi.jl:44-98 over an ARBITRARY
// transition list (overlap models, reference types.jl:78-90), run time-parallel.
//
// The signal is cut into blocks of B samples; one workgroup sweeps one block with the reference's
// own fp64 operations in the reference's order ((T1[src]+lp) > best strict, then + funcl), after a
// warm-up of H samples that starts from a flat column (emissions only).  A block therefore works in
// its own additive frame: its trellis column differs from the sequential one by a constant once the
// warm-up has forgotten its start, and back-pointers (which only see differences) are the same.
// That claim is CHECKED, not assumed: k_block_check compares the warm column of block c with the
// column block c-1 ended on -- the spread of their difference over all states must stay below
// kSpreadTol -- and the count of failing boundaries is returned through plan_diagnostics (the
// host-buffer entry points then retry with a longer warm-up and finally with the strict engine).
//
// Backtrace (viterbi.jl:90-94) is exact whatever the block length: k_block_map walks ALL S end
// states of a block back to its first sample (they merge after a few hundred samples; the merged
// tail is written straight to x), k_block_compose chains the per-block maps from the global argmax
// and k_block_finish fills the unmerged heads.  ll (viterbi.jl:92-96) is re-accumulated along the
// decoded path per block with the reference's op order and combined across blocks.
//
// HBM layout: T2 S x T int16 column-major as in the strict engine; per-block columns endv/warmv
// nblk x S doubles; maps nblk x S int16.
#include <cmath>

#include "generic_dev.h"
#include "hmmsort_internal.h"

namespace hmmsort {

constexpr double kSpreadTol = 1e-6;

__device__ __forceinline__ double funcl_b(double x, double mu, double c0, double den)
{
    double dd = x - mu;
    return c0 - (dd * dd) / den;  // utils.jl:4 with the invariants hoisted (see generic_engine.hip)
}

struct BlockArgs {
    const double *y;
    int64_t T;
    int S, B, H;
    const double *mean;   // [S]
    const double *lp0;    // [S] first incoming transition (list order), -inf when none
    const int32_t *src0;  // [S]
    const int32_t *tinfo; // [S] tail offset << 8 | tail count (incoming transitions after the first)
    const int32_t *tsrc;  // [ntail]
    const double *tlp;    // [ntail]
    int ntail, tail_lds;
    double c0, den;
    int16_t *T2;
    double *endv, *warmv; // [nblk][S]
    double *gbuf;         // [nblk][2S] when the columns do not fit LDS, else null
};

// One workgroup = one block.  SPT = states per thread; CACHE keeps the per-state constants in
// registers, otherwise they are re-read (coalesced, L2-resident) every sample.
template <int SPT, bool CACHE>
__global__ __launch_bounds__(1024) void gen_vit_block(BlockArgs a)
{
    extern __shared__ double sh[];
    const int c = blockIdx.x, S = a.S, tid = threadIdx.x, nt = blockDim.x;
    double *prev = a.gbuf ? a.gbuf + (int64_t)c * 2 * S : sh;
    double *cur = prev + S;
    const int32_t *tsrc = a.tsrc;
    const double *tlp = a.tlp;
    if (a.tail_lds) {
        double *l_tlp = sh + (a.gbuf ? 0 : 2 * S);
        int32_t *l_tsrc = (int32_t *)(l_tlp + a.ntail);
        for (int i = tid; i < a.ntail; i += nt) { l_tlp[i] = a.tlp[i]; l_tsrc[i] = a.tsrc[i]; }
        tsrc = l_tsrc; tlp = l_tlp;
    }
    const int64_t s = (int64_t)c * a.B;
    const int64_t e = (s + a.B < a.T) ? s + a.B : a.T;
    const int64_t w = (s - a.H > 0) ? s - a.H : 0;
    const double c0 = a.c0, den = a.den;

    double mean_r[SPT], lp0_r[SPT];
    int src0_r[SPT], ti_r[SPT];
    if (CACHE) {
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            const int j = tid + i * nt;
            const bool ok = j < S;
            mean_r[i] = ok ? a.mean[j] : 0.0;
            lp0_r[i] = ok ? a.lp0[j] : 0.0;
            src0_r[i] = ok ? a.src0[j] : 0;
            ti_r[i] = ok ? a.tinfo[j] : 0;
        }
    }
    {   // first column: viterbi.jl:55-63 at the start of the signal, flat (emissions) elsewhere
        const double y0 = a.y[w];
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            const int j = tid + i * nt;
            if (j < S) {
                const double m = CACHE ? mean_r[i] : a.mean[j];
                cur[j] = (w == 0 && j == 0) ? 0.0 : funcl_b(y0, m, c0, den);
                if (s == 0) a.T2[j] = 1;
            }
        }
        if (s > 0 && w == s - 1) {
#pragma unroll
            for (int i = 0; i < SPT; i++) {
                const int j = tid + i * nt;
                if (j < S) a.warmv[(int64_t)c * S + j] = cur[j];
            }
        }
    }
    for (int64_t t = w + 1; t < e; t++) {
        __threadfence_block();
        __syncthreads();
        double *tmp = prev; prev = cur; cur = tmp;
        const double yt = a.y[t];
        const bool own = t >= s;
        int16_t *psi = a.T2 + (int64_t)S * t;
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            const int j = tid + i * nt;
            if (j < S) {
                const double m = CACHE ? mean_r[i] : a.mean[j];
                const double l0 = CACHE ? lp0_r[i] : a.lp0[j];
                const int s0 = CACHE ? src0_r[i] : a.src0[j];
                const int ti = CACHE ? ti_r[i] : a.tinfo[j];
                double best = -INFINITY;  // viterbi.jl:52
                int arg = 1;              // viterbi.jl:53
                double tt = prev[s0] + l0;  // :79
                if (tt > best) { best = tt; arg = s0 + 1; }  // :80 strict, list order
                const int tn = ti & 255;
                int tp = ti >> 8;
                for (int q = 0; q < tn; q++, tp++) {
                    const int sq = tsrc[tp];
                    tt = prev[sq] + tlp[tp];
                    if (tt > best) { best = tt; arg = sq + 1; }
                }
                const double v = best + funcl_b(yt, m, c0, den);  // :85-87
                cur[j] = v;
                if (own) psi[j] = (int16_t)arg;
                if (t == s - 1) a.warmv[(int64_t)c * S + j] = v;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SPT; i++) {
        const int j = tid + i * nt;
        if (j < S) a.endv[(int64_t)c * S + j] = cur[j];
    }
}

// Boundary certificate: spread over the states of (warm column of block c) - (end column of block
// c-1), both at sample c*B-1.  diag[0] counts failing boundaries, diag[2] holds the largest spread.
__global__ __launch_bounds__(256) void k_block_check(const double *__restrict__ endv,
                                                     const double *__restrict__ warmv, int S,
                                                     unsigned long long *diag)
{
    __shared__ double smin[256], smax[256];
    __shared__ int sbad[256];
    const int c = blockIdx.x + 1, tid = threadIdx.x;
    const double *wv = warmv + (int64_t)c * S, *ev = endv + (int64_t)(c - 1) * S;
    double lo = INFINITY, hi = -INFINITY;
    int bad = 0;
    for (int j = tid; j < S; j += 256) {
        const double a = wv[j], b = ev[j];
        const bool fa = a > -INFINITY && a < INFINITY, fb = b > -INFINITY && b < INFINITY;
        if (fa && fb) {
            const double d = a - b;
            lo = d < lo ? d : lo;
            hi = d > hi ? d : hi;
        } else if (fa != fb || a != a || b != b) {
            bad = 1;  // reachable in one frame only, or NaN
        }
    }
    smin[tid] = lo; smax[tid] = hi; sbad[tid] = bad;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            smin[tid] = smin[tid + o] < smin[tid] ? smin[tid + o] : smin[tid];
            smax[tid] = smax[tid + o] > smax[tid] ? smax[tid + o] : smax[tid];
            sbad[tid] |= sbad[tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        double spread = (smax[0] >= smin[0]) ? smax[0] - smin[0] : 0.0;
        if (sbad[0]) spread = INFINITY;
        if (!(spread <= kSpreadTol)) atomicAdd(&diag[0], 1ull);
        atomicMax(&diag[2], (unsigned long long)__double_as_longlong(spread));
    }
}

// All S end states of block c walked back to its first sample.  fmap[c][j] = state at sample
// c*B-1 when sample e-1 is in state j+1 (block 0: unused).  Once every walker sits in the same
// state the rest of the block's path is known: it is written to x and merged[c] records the last
// sample written (c*B-1 when nothing was).
__global__ __launch_bounds__(256) void k_block_map(const int16_t *__restrict__ T2, int64_t T, int S,
                                                   int B, int16_t *__restrict__ fmap,
                                                   int64_t *__restrict__ merged,
                                                   int16_t *__restrict__ x)
{
    extern __shared__ int16_t wk[];  // S walkers
    __shared__ int mn, mx;
    const int c = blockIdx.x, tid = threadIdx.x;
    const int64_t s = (int64_t)c * B;
    const int64_t e = (s + B < T) ? s + B : T;
    for (int j = tid; j < S; j += 256) wk[j] = (int16_t)(j + 1);
    int64_t t = e - 1;  // walkers hold the state at sample t
    bool one = (S == 1);
    int step = 0;
    while (t > s && !one) {
        const int16_t *psi = T2 + (int64_t)S * t;
        for (int j = tid; j < S; j += 256) wk[j] = psi[wk[j] - 1];
        t--;
        if ((++step & 15) == 0) {
            if (tid == 0) { mn = 32767; mx = 0; }
            __syncthreads();
            int lo = 32767, hi = 0;
            for (int j = tid; j < S; j += 256) {
                const int v = wk[j];
                lo = v < lo ? v : lo;
                hi = v > hi ? v : hi;
            }
            atomicMin(&mn, lo);
            atomicMax(&mx, hi);
            __syncthreads();
            one = (mn == mx);
            __syncthreads();
        }
    }
    __syncthreads();
    if (one) {
        if (tid == 0) {
            int v = wk[0];
            merged[c] = t;
            x[t] = (int16_t)v;
            while (t > s) {
                v = T2[(int64_t)S * t + (v - 1)];
                t--;
                x[t] = (int16_t)v;
            }
            wk[0] = (int16_t)((c > 0) ? T2[(int64_t)S * s + (v - 1)] : 1);
        }
        __syncthreads();
        const int16_t f = wk[0];
        __syncthreads();
        for (int j = tid; j < S; j += 256) fmap[(int64_t)c * S + j] = f;
    } else {
        if (tid == 0) merged[c] = s - 1;
        const int16_t *psi = T2 + (int64_t)S * s;
        for (int j = tid; j < S; j += 256)
            fmap[(int64_t)c * S + j] = (c > 0) ? psi[wk[j] - 1] : (int16_t)1;
    }
}

// argmax of the last column (first maximum, viterbi.jl:90), then the end state of every block
__global__ void k_block_compose(const double *__restrict__ endv, const int16_t *__restrict__ fmap,
                                int S, int nblk, int16_t *__restrict__ endstate)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double *last = endv + (int64_t)(nblk - 1) * S;
    int best = 0;
    for (int j = 1; j < S; j++)
        if (last[j] > last[best]) best = j;
    int v = best + 1;
    endstate[nblk - 1] = (int16_t)v;
    for (int c = nblk - 1; c >= 1; c--) {
        v = fmap[(int64_t)c * S + (v - 1)];
        endstate[c - 1] = (int16_t)v;
    }
}

// the part of each block's path above the merge point
__global__ void k_block_finish(const int16_t *__restrict__ T2, int64_t T, int S, int B, int nblk,
                               const int16_t *__restrict__ endstate,
                               const int64_t *__restrict__ merged, int16_t *__restrict__ x)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nblk) return;
    const int64_t s = (int64_t)c * B;
    const int64_t e = (s + B < T) ? s + B : T;
    const int64_t stop = merged[c] + 1 > s ? merged[c] + 1 : s;  // first sample still unknown
    int v = endstate[c];
    int64_t t = e - 1;
    if (t >= stop) x[t] = (int16_t)v;
    while (t > stop) {
        v = T2[(int64_t)S * t + (v - 1)];
        t--;
        x[t] = (int16_t)v;
    }
}

// Path values of one block in the block's own frame (p = 0 before its first sample), accumulated
// as ((p + lp) + q) like viterbi.jl:85-87.  part[c] = {last p, sum of p over samples >= 1, count}.
constexpr int kLLTile = 1024;
__global__ __launch_bounds__(256) void k_block_ll(const double *__restrict__ y,
                                                  const int16_t *__restrict__ x, int64_t T, int S,
                                                  int B, const double *__restrict__ mean,
                                                  const int32_t *__restrict__ in_ptr,
                                                  const int32_t *__restrict__ in_src,
                                                  const double *__restrict__ in_lp, double c0,
                                                  double den, double *__restrict__ part)
{
    __shared__ double slp[kLLTile], sq[kLLTile];
    __shared__ double carry[2];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int64_t s = (int64_t)c * B;
    const int64_t e = (s + B < T) ? s + B : T;
    if (tid == 0) { carry[0] = 0.0; carry[1] = 0.0; }
    for (int64_t base = s; base < e; base += kLLTile) {
        const int n = (int)((e - base < kLLTile) ? e - base : kLLTile);
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            const int64_t t = base + i;
            const int xc = x[t] - 1;
            double lp = 0.0, q = funcl_b(y[t], mean[xc], c0, den);
            if (t == 0) {
                if (xc == 0) q = 0.0;  // T1[1,1] = 0, viterbi.jl:63
            } else {
                const int xp = x[t - 1] - 1;
                lp = -INFINITY;
                const int e1 = in_ptr[xc + 1];
                for (int ed = in_ptr[xc]; ed < e1; ed++)
                    if (in_src[ed] == xp) { lp = in_lp[ed]; break; }
            }
            slp[i] = lp; sq[i] = q;
        }
        __syncthreads();
        if (tid == 0) {
            double p = carry[0], sum = carry[1];
            for (int i = 0; i < n; i++) {
                p = (base + i == 0) ? sq[i] : (p + slp[i]) + sq[i];
                if (base + i >= 1) sum += p;
            }
            carry[0] = p; carry[1] = sum;
        }
    }
    __syncthreads();
    if (tid == 0) {
        part[3 * c] = carry[0];
        part[3 * c + 1] = carry[1];
        part[3 * c + 2] = (double)((e - s) - (s == 0 ? 1 : 0));
    }
}

__global__ void k_block_ll_sum(const double *__restrict__ part, int nblk, double *__restrict__ ll)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double off = 0.0, acc = 0.0;
    for (int c = 0; c < nblk; c++) {
        acc += part[3 * c + 2] * off + part[3 * c + 1];
        off += part[3 * c];
    }
    *ll = acc;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
void blocked_geometry(int64_t T, int64_t L, int64_t block_req, int64_t halo_req, int64_t *B,
                      int64_t *H, int64_t *nblk)
{
    int64_t h = halo_req > 0 ? halo_req : std::max<int64_t>(256, 4 * L);
    h = (h + 63) / 64 * 64;
    int64_t b = block_req > 0 ? block_req : std::max<int64_t>(2 * h, (T + 2047) / 2048);
    b = (b + 63) / 64 * 64;
    if (b < 64) b = 64;
    *B = b; *H = h; *nblk = (T + b - 1) / b;
}

template <typename Tv>
static int dalloc(Tv **p, size_t n, int64_t *bytes)
{
    if (hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(Tv)) != hipSuccess) {
        (void)hipGetLastError();
        set_error("blocked engine: hipMalloc of %.2f GB failed", (double)n * sizeof(Tv) / 1e9);
        return HMMSORT_ENOMEM;
    }
    *bytes += (int64_t)(n * sizeof(Tv));
    return HMMSORT_OK;
}

int blocked_set_model(GenericDev *g, const HostModel &m)
{
    const int64_t S = m.S;
    std::vector<double> lp0(S, -INFINITY), tlp;
    std::vector<int32_t> src0(S, 0), tinfo(S, 0), tsrc;
    for (int64_t j = 0; j < S; j++) {
        const int b = m.in_ptr[j], e = m.in_ptr[j + 1];
        if (e > b) { lp0[j] = m.in_lp[b]; src0[j] = m.in_src[b]; }
        const int nt = e > b ? e - b - 1 : 0;
        HS_CHECK(nt <= 255 && tsrc.size() < (1u << 22), HMMSORT_EUNSUP,
                 "blocked engine: in-degree %d of state %lld too large", nt + 1, (long long)j + 1);
        tinfo[j] = (int32_t)((tsrc.size() << 8) | (unsigned)nt);
        for (int q = b + 1; q < e; q++) { tsrc.push_back(m.in_src[q]); tlp.push_back(m.in_lp[q]); }
    }
    if (g->ntail < 0) {  // first call: allocate
        g->ntail = (int)tsrc.size();
        int rc;
        if ((rc = dalloc(&g->d_lp0, S, &g->bytes)) || (rc = dalloc(&g->d_src0, S, &g->bytes)) ||
            (rc = dalloc(&g->d_tinfo, S, &g->bytes)) ||
            (rc = dalloc(&g->d_tsrc, tsrc.size(), &g->bytes)) ||
            (rc = dalloc(&g->d_tlp, tlp.size(), &g->bytes)))
            return rc;
    }
    HS_CHECK((int)tsrc.size() == g->ntail, HMMSORT_EINVAL, "set_model: transition structure changed");
    HS_HIP(hipMemcpy(g->d_lp0, lp0.data(), S * sizeof(double), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(g->d_src0, src0.data(), S * sizeof(int32_t), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(g->d_tinfo, tinfo.data(), S * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!tsrc.empty()) {
        HS_HIP(hipMemcpy(g->d_tsrc, tsrc.data(), tsrc.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HS_HIP(hipMemcpy(g->d_tlp, tlp.data(), tlp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return HMMSORT_OK;
}

int blocked_create(GenericDev *g, const HostModel &m, int64_t block_req, int64_t halo_req)
{
    blocked_geometry(g->T, m.K - 1, block_req, halo_req, &g->B, &g->H, &g->nblk);
    g->blocked = true;
    g->ntail = -1;
    int rc = blocked_set_model(g, m);
    if (rc) return rc;
    const size_t S = (size_t)g->S, nb = (size_t)g->nblk;
    // LDS: two columns + the multi-source tails, else columns in a per-block global scratch
    const size_t tail_b = (size_t)g->ntail * 12 + 8;
    g->blk_cols_lds = 2 * S * 8 <= 150 * 1024;
    g->blk_tail_lds = (g->blk_cols_lds ? 2 * S * 8 : 0) + tail_b <= 150 * 1024;
    if ((rc = dalloc(&g->d_endv, nb * S, &g->bytes)) || (rc = dalloc(&g->d_warmv, nb * S, &g->bytes)) ||
        (rc = dalloc(&g->d_fmap, nb * S, &g->bytes)) || (rc = dalloc(&g->d_merged, nb, &g->bytes)) ||
        (rc = dalloc(&g->d_endstate, nb, &g->bytes)) || (rc = dalloc(&g->d_llpart, 3 * nb, &g->bytes)) ||
        (rc = dalloc(&g->d_bdiag, 8, &g->bytes)))
        return rc;
    if (!g->blk_cols_lds && (rc = dalloc(&g->d_blkbuf, nb * 2 * S, &g->bytes))) return rc;
    HS_HIP(hipMemset(g->d_bdiag, 0, 8 * sizeof(unsigned long long)));
    return HMMSORT_OK;
}

void blocked_destroy(GenericDev *g)
{
    void *ptrs[] = {g->d_lp0, g->d_src0, g->d_tinfo, g->d_tsrc, g->d_tlp, g->d_endv, g->d_warmv,
                    g->d_fmap, g->d_merged, g->d_endstate, g->d_llpart, g->d_bdiag, g->d_blkbuf};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
}

template <int SPT, bool CACHE>
static int launch_block_sweep(GenericDev *g, const BlockArgs &a, int threads, size_t lds,
                              hipStream_t st)
{
    if (lds > 64 * 1024)
        HS_HIP(hipFuncSetAttribute((const void *)gen_vit_block<SPT, CACHE>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((gen_vit_block<SPT, CACHE>), dim3((unsigned)g->nblk), dim3(threads), lds, st, a);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int blocked_viterbi(GenericDev *g, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st)
{
    const int64_t T = g->T, S = g->S;
    if (!g->d_T2) {
        const double need = (double)S * (double)T * 2.0;
        HS_CHECK(need < 220e9, HMMSORT_ENOMEM,
                 "Viterbi needs %.1f GB of back-pointers; decode in chunks", need / 1e9);
        int rc = dalloc(&g->d_T2, (size_t)S * T, &g->bytes);
        if (rc) return rc;
    }
    BlockArgs a;
    a.y = d_y; a.T = T; a.S = (int)S; a.B = (int)g->B; a.H = (int)g->H;
    a.mean = g->d_mean; a.lp0 = g->d_lp0; a.src0 = g->d_src0; a.tinfo = g->d_tinfo;
    a.tsrc = g->d_tsrc; a.tlp = g->d_tlp; a.ntail = g->ntail; a.tail_lds = g->blk_tail_lds;
    a.c0 = -kLog2Pi - g->lsig;
    a.den = 2.0 * (g->sigma * g->sigma);
    a.T2 = g->d_T2; a.endv = g->d_endv; a.warmv = g->d_warmv;
    a.gbuf = g->blk_cols_lds ? nullptr : g->d_blkbuf;
    size_t lds = (g->blk_cols_lds ? 2 * S * 8 : 0) + (g->blk_tail_lds ? (size_t)g->ntail * 12 + 8 : 0);
    int threads = (int)std::min<int64_t>(1024, (S + 63) / 64 * 64);
    const int spt = (int)((S + threads - 1) / threads);
    HS_HIP(hipMemsetAsync(g->d_bdiag, 0, 8 * sizeof(unsigned long long), st));
    int rc;
    if (spt <= 1) rc = launch_block_sweep<1, true>(g, a, threads, lds, st);
    else if (spt <= 2) rc = launch_block_sweep<2, true>(g, a, threads, lds, st);
    else if (spt <= 4) rc = launch_block_sweep<4, true>(g, a, threads, lds, st);
    else if (spt <= 8) rc = launch_block_sweep<8, true>(g, a, threads, lds, st);
    else if (spt <= 12) rc = launch_block_sweep<12, true>(g, a, threads, lds, st);
    else if (spt <= 16) rc = launch_block_sweep<16, false>(g, a, threads, lds, st);
    else if (spt <= 24) rc = launch_block_sweep<24, false>(g, a, threads, lds, st);
    else rc = launch_block_sweep<32, false>(g, a, threads, lds, st);
    if (rc) return rc;
    const int nb = (int)g->nblk;
    if (nb > 1) {
        hipLaunchKernelGGL(k_block_check, dim3(nb - 1), dim3(256), 0, st, g->d_endv, g->d_warmv, (int)S,
                           g->d_bdiag);
        HS_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_block_map, dim3(nb), dim3(256), (size_t)S * sizeof(int16_t), st, g->d_T2, T,
                       (int)S, (int)g->B, g->d_fmap, g->d_merged, d_x);
    HS_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_block_compose, dim3(1), dim3(64), 0, st, g->d_endv, g->d_fmap, (int)S, nb,
                       g->d_endstate);
    HS_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_block_finish, dim3((nb + 63) / 64), dim3(64), 0, st, g->d_T2, T, (int)S,
                       (int)g->B, nb, g->d_endstate, g->d_merged, d_x);
    HS_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_block_ll, dim3(nb), dim3(256), 0, st, d_y, d_x, T, (int)S, (int)g->B,
                       g->d_mean, g->d_in_ptr, g->d_in_src, g->d_in_lp, a.c0, a.den, g->d_llpart);
    HS_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_block_ll_sum, dim3(1), dim3(64), 0, st, g->d_llpart, nb, d_ll);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int blocked_diagnostics(GenericDev *g, hipStream_t st, int64_t diag[8])
{
    unsigned long long h[8];
    HS_HIP(hipMemcpyAsync(h, g->d_bdiag, sizeof(h), hipMemcpyDeviceToHost, st));
    HS_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < 8; i++) diag[i] = (int64_t)h[i];
    return HMMSORT_OK;
}

}  // namespace hmmsort
