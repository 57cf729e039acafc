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
    int ntail;
    double c0, den, rden;  // rden = RN(1/den)
    int16_t *T2;
    double *endv, *warmv; // [nblk][S]
    double *gbuf;         // [nblk][2S] when the columns do not fit LDS, else null
};

// (dd*dd)/den correctly rounded without the division sequence: with r = RN(1/den) from the host,
// q = RN(x*r), then q + RN(x - q*den)*r in one fma is the correctly rounded quotient (Markstein's
// theorem; x and den are normal and far from the exponent limits here).  Same double as utils.jl:4.
__device__ __forceinline__ double funcl_m(double x, double mu, double c0, double den, double rden)
{
    const double dd = x - mu;
    const double sq = dd * dd;
    const double q = sq * rden;
    const double rem = __builtin_fma(-q, den, sq);
    return c0 - __builtin_fma(rem, rden, q);
}

// Barrier that orders LDS traffic only: the back-pointer stores of the previous sample stay in
// flight (a __syncthreads() would drain them to L2 every sample, which is what bounds the sweep).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// the incoming transitions after the first one (multi-source states only): viterbi.jl:76-84
template <typename PrevPtr, typename SrcPtr, typename LpPtr>
__device__ __forceinline__ void block_tail(PrevPtr prev, int ti, SrcPtr tsrc, LpPtr tlp,
                                           double &best, int &arg)
{
    const int tn = ti & 255;
    int tp = ti >> 8;
    for (int k = 0; k < tn; k++, tp++) {
        const int sq = tsrc[tp];
        const double tt = prev[sq] + tlp[tp];
        if (tt > best) { best = tt; arg = sq + 1; }  // :80 strict, list order
    }
}

// One workgroup = one block.  With SPT > 0 every thread keeps the constants of its SPT states in
// registers (S <= 4096); SPT = 0 re-reads them each sample (coalesced, L2-resident).  GCOL: the
// two trellis columns live in global memory (S too large for LDS); TLDS: tails staged in LDS.
// Address spaces are fixed at compile time so that column accesses are ds_* (not flat) operations.
template <int SPT, bool GCOL, bool TLDS>
__global__ __launch_bounds__(1024) void gen_vit_block(BlockArgs a)
{
    extern __shared__ double sh[];
    constexpr bool CACHE = SPT > 0;
    constexpr int NS = CACHE ? SPT : 1;
    const int c = blockIdx.x, S = a.S, tid = threadIdx.x, nt = blockDim.x;
    double *l_tlp = sh + (GCOL ? 0 : 2 * S);
    int32_t *l_tsrc = (int32_t *)(l_tlp + a.ntail);
    if (TLDS)
        for (int i = tid; i < a.ntail; i += nt) { l_tlp[i] = a.tlp[i]; l_tsrc[i] = a.tsrc[i]; }
    const int64_t s = (int64_t)c * a.B;
    const int64_t e = (s + a.B < a.T) ? s + a.B : a.T;
    const int64_t w = (s - a.H > 0) ? s - a.H : 0;
    const double c0 = a.c0, den = a.den, rden = a.rden;
    double *warm = a.warmv + (int64_t)c * S;
    double *gcol = GCOL ? a.gbuf + (int64_t)c * 2 * S : nullptr;

    double mean_r[NS], lp0_r[NS];
    int src0_r[NS], ti_r[NS];
    if (CACHE) {
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const int j = tid + i * nt < S ? tid + i * nt : S - 1;
            mean_r[i] = a.mean[j];
            lp0_r[i] = a.lp0[j];
            src0_r[i] = a.src0[j];
            ti_r[i] = a.tinfo[j];
        }
    }
    int par = 0;  // cur = column par, prev = column par ^ 1
    {   // first column: viterbi.jl:55-63 at the start of the signal, flat (emissions) elsewhere
        const double y0 = a.y[w];
        for (int j = tid; j < S; j += nt) {
            const double v = (w == 0 && j == 0) ? 0.0 : funcl_m(y0, a.mean[j], c0, den, rden);
            if (GCOL) gcol[j] = v; else sh[j] = v;
            if (s == 0) a.T2[j] = 1;
            if (s > 0 && w == s - 1) warm[j] = v;
        }
    }
    for (int64_t t = w + 1; t < e; t++) {
        if (GCOL) { __threadfence_block(); __syncthreads(); }
        else lds_barrier();
        par ^= 1;
        const double yt = a.y[t];
        const bool own = t >= s, at_warm = t == s - 1;
        int16_t *psi = a.T2 + (int64_t)S * t;
        if (CACHE) {  // columns in LDS
            const double *prev = sh + (par ^ 1) * S;
            double *cur = sh + par * S;
            double best[NS];
            int arg[NS];
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const double tt = prev[src0_r[i]] + lp0_r[i];  // :79
                const bool up = tt > -INFINITY;                // :80 against fill(-Inf)
                best[i] = up ? tt : -INFINITY;
                arg[i] = up ? src0_r[i] + 1 : 1;
            }
#pragma unroll
            for (int i = 0; i < NS; i++) {
                if (ti_r[i] & 255) {
                    if (TLDS) block_tail(prev, ti_r[i], l_tsrc, l_tlp, best[i], arg[i]);
                    else block_tail(prev, ti_r[i], a.tsrc, a.tlp, best[i], arg[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const int j = tid + i * nt;
                const double v = best[i] + funcl_m(yt, mean_r[i], c0, den, rden);  // :85-87
                if (j < S) {
                    cur[j] = v;
                    if (own) psi[j] = (int16_t)arg[i];
                    if (at_warm) warm[j] = v;
                }
            }
        } else {
#pragma unroll 2
            for (int j = tid; j < S; j += nt) {
                const int s0 = a.src0[j], ti = a.tinfo[j];
                const double l0 = a.lp0[j], q = funcl_m(yt, a.mean[j], c0, den, rden);
                double v;
                int arg;
                if (GCOL) {
                    const double *prev = gcol + (par ^ 1) * S;
                    const double tt = prev[s0] + l0;
                    const bool up = tt > -INFINITY;
                    double best = up ? tt : -INFINITY;
                    arg = up ? s0 + 1 : 1;
                    if (ti & 255) {
                        if (TLDS) block_tail(prev, ti, l_tsrc, l_tlp, best, arg);
                        else block_tail(prev, ti, a.tsrc, a.tlp, best, arg);
                    }
                    v = best + q;
                    gcol[par * S + j] = v;
                } else {
                    const double *prev = sh + (par ^ 1) * S;
                    const double tt = prev[s0] + l0;
                    const bool up = tt > -INFINITY;
                    double best = up ? tt : -INFINITY;
                    arg = up ? s0 + 1 : 1;
                    if (ti & 255) {
                        if (TLDS) block_tail(prev, ti, l_tsrc, l_tlp, best, arg);
                        else block_tail(prev, ti, a.tsrc, a.tlp, best, arg);
                    }
                    v = best + q;
                    sh[par * S + j] = v;
                }
                if (own) psi[j] = (int16_t)arg;
                if (at_warm) warm[j] = v;
            }
        }
    }
    __syncthreads();
    for (int j = tid; j < S; j += nt)
        a.endv[(int64_t)c * S + j] = GCOL ? gcol[par * S + j] : sh[par * S + j];
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

template <int SPT, bool GCOL, bool TLDS>
static int launch_block_sweep(GenericDev *g, const BlockArgs &a, int threads, size_t lds,
                              hipStream_t st)
{
    if (lds > 64 * 1024)
        HS_HIP(hipFuncSetAttribute((const void *)gen_vit_block<SPT, GCOL, TLDS>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((gen_vit_block<SPT, GCOL, TLDS>), dim3((unsigned)g->nblk), dim3(threads), lds, st, a);
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
    a.tsrc = g->d_tsrc; a.tlp = g->d_tlp; a.ntail = g->ntail;
    a.c0 = -kLog2Pi - g->lsig;
    a.den = 2.0 * (g->sigma * g->sigma);
    a.rden = 1.0 / a.den;
    a.T2 = g->d_T2; a.endv = g->d_endv; a.warmv = g->d_warmv;
    a.gbuf = g->blk_cols_lds ? nullptr : g->d_blkbuf;
    size_t lds = (g->blk_cols_lds ? 2 * S * 8 : 0) + (g->blk_tail_lds ? (size_t)g->ntail * 12 + 8 : 0);
    int threads = (int)std::min<int64_t>(1024, (S + 63) / 64 * 64);
    const int spt = (int)((S + threads - 1) / threads);
    HS_HIP(hipMemsetAsync(g->d_bdiag, 0, 8 * sizeof(unsigned long long), st));
    int rc;
    const bool gcol = !g->blk_cols_lds, tl = g->blk_tail_lds;
    if (!gcol && tl && spt <= 1) rc = launch_block_sweep<1, false, true>(g, a, threads, lds, st);
    else if (!gcol && tl && spt <= 2) rc = launch_block_sweep<2, false, true>(g, a, threads, lds, st);
    else if (!gcol && tl && spt <= 4) rc = launch_block_sweep<4, false, true>(g, a, threads, lds, st);
    else if (!gcol && tl) rc = launch_block_sweep<0, false, true>(g, a, threads, lds, st);
    else if (!gcol) rc = launch_block_sweep<0, false, false>(g, a, threads, lds, st);
    else if (tl) rc = launch_block_sweep<0, true, true>(g, a, threads, lds, st);
    else rc = launch_block_sweep<0, true, false>(g, a, threads, lds, st);
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
