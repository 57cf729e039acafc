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
// HBM layout: back-pointers are kept only for the nms states with more than one incoming transition
// (T2c, nms x T int16; 236 of 3600 states at N=2, K=60): a single-source state's pointer is its
// source whenever its value is finite, and the backtrace only visits finite values (columns of all
// -Inf/NaN, i.e. non-finite data, are the one case where this differs from viterbi.jl:53's
// ones()).  Per-block columns endv/warmv nblk x S doubles; maps nblk x S int16.
#include <cmath>
#include <cstring>

#include "generic_dev.h"
#include "hmmsort_internal.h"

namespace hmmsort {

constexpr double kSpreadTol = 1e-6;

__device__ __forceinline__ double funcl_b(double x, double mu, double c0, double den)
{
    double dd = x - mu;
    return c0 - (dd * dd) / den;  // utils.jl:4 with the invariants hoisted (see generic_engine.hip)
}

struct MsRec {  // constants of one multi-source state, phase B of the sweep
    double lp0, mean;
    int32_t j, s0, ti, pad;
};

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
    const MsRec *ms;      // [nms] states with more than one incoming transition
    int nms;
    int nbthr;            // register-cached sweep: leading threads that only do phase B
    double c0, den, rden;  // rden = RN(1/den)
    int16_t *T2c;         // [T][nms]
    double *endv, *warmv; // [nblk][S]
    unsigned long long *gapmin;  // [nblk] bit pattern of the smallest non-zero candidate gap
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
// (a 4-wide batched variant of this loop was measured slower: it costs the registers that keep the
// sweep at two workgroups per CU)
constexpr double kGapCoarse = 1e-4;  // gaps above this are never near-ties (|trellis| < 1e10)
// gmin collects the smallest NON-ZERO distance between a candidate and the running maximum: a block
// works in its own additive frame, where two candidates a few ulps apart are told apart although
// the reference, at the magnitude its trellis has by then, rounds them to a tie (and then takes the
// first in list order).  Exactly equal candidates behave the same in every frame.
template <typename PrevPtr, typename SrcPtr, typename LpPtr>
__device__ __forceinline__ void block_tail(PrevPtr prev, int ti, SrcPtr tsrc, LpPtr tlp,
                                           double &best, int &arg, double &gmin)
{
    const int tn = ti & 255;
    int tp = ti >> 8;
    for (int k = 0; k < tn; k++, tp++) {
        const int sq = tsrc[tp];
        const double tt = prev[sq] + tlp[tp];
        const double gd = fabs(tt - best);
        if (gd < kGapCoarse && gd > 0.0) gmin = gd < gmin ? gd : gmin;  // rare
        if (tt > best) { best = tt; arg = sq + 1; }  // :80 strict, list order
    }
}

// gaps are published as they occur, and only when small (rare): no loop-carried register
__device__ __forceinline__ void publish_gap(unsigned long long *gapmin, int c, double g)
{
    if (g < kGapCoarse) atomicMin(&gapmin[c], (unsigned long long)__double_as_longlong(g));
}

// One workgroup = one block.  With SPT > 0 every thread keeps the constants of its SPT states in
// registers (S <= 4096); SPT = 0 re-reads them each sample (coalesced, L2-resident).  GCOL: the
// two trellis columns live in global memory (S too large for LDS); TLDS: tails staged in LDS.
// Address spaces are fixed at compile time so that column accesses are ds_* (not flat) operations.
// Two workgroups of 16 waves per CU need 8 waves per SIMD, i.e. at most 64 VGPRs: pinned, because a
// 65th register silently halves the occupancy (measured: 11.8 -> 16.4 ms at 4 M samples).
template <int SPT, bool SPEC, bool GCOL, bool TLDS>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
void gen_vit_block(BlockArgs a)
{
    extern __shared__ double sh[];
    constexpr bool CACHE = SPT > 0;
    constexpr int NS = CACHE ? SPT : 1;
    const int c = blockIdx.x, S = a.S, tid = threadIdx.x, nt = blockDim.x;
    const int CS = S + 1;  // LDS column stride: entry S of a column is a write-only dummy slot
    double *l_tlp = sh + (GCOL ? 0 : 2 * CS);
    int32_t *l_tsrc = (int32_t *)(l_tlp + a.ntail);
    if (TLDS)
        for (int i = tid; i < a.ntail; i += nt) { l_tlp[i] = a.tlp[i]; l_tsrc[i] = a.tsrc[i]; }
    const int64_t s = (int64_t)c * a.B;
    const int64_t e = (s + a.B < a.T) ? s + a.B : a.T;
    const int64_t w = (s - a.H > 0) ? s - a.H : 0;
    const double c0 = a.c0, den = a.den, rden = a.rden;
    double *warm = a.warmv + (int64_t)c * S;
    double *gcol = GCOL ? a.gbuf + (int64_t)c * 2 * S : nullptr;

    // Register-cached sweep.  SPEC (small models, measured faster up to ~2 states per lane): wave-
    // specialised, the first nbthr threads (whole waves) own the states with several incoming
    // transitions (phase B, a chain of dependent LDS reads), the others the first transition of
    // every state (phase A); both run concurrently between two barriers.  Otherwise every thread
    // does phase A and the leading threads then add phase B.
    const int nbt = (CACHE && SPEC) ? a.nbthr : 0;
    const bool is_b = tid < nbt;
    const int ta = tid - nbt, na = nt - nbt;
    // The two roles never coexist in a wave, so they share one set of registers:
    //   phase A: cd[i] = mean, cd[ND+i] = lp of the first transition, ci[i] = source | store << 16
    //            (store index S = the dummy slot; S < 2^15)
    //   phase B: cd[0] = mean, cd[1..3] = lp of the first three transitions (-Inf when absent),
    //            ci[0] = source0 | state << 16, ci[1] = source1 | source2 << 16, ci[2] = the rest
    constexpr int ND = NS > 2 ? NS : 2, NI = NS > 3 ? NS : 3;
    double cd[2 * ND];
    unsigned ci[NI];
    if (CACHE && !is_b) {
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const int jj = ta + i * na;
            const int j = jj < S ? jj : S - 1;
            cd[i] = a.mean[j];
            cd[ND + i] = a.lp0[j];
            const int wj = (jj < S && (a.tinfo[j] & 255) == 0) ? j : S;
            ci[i] = (unsigned)a.src0[j] | ((unsigned)wj << 16);
        }
    }
    int bj = 0, bs0 = 0, bti = 0;  // !SPEC: this thread's multi-source state
    double bl0 = 0.0, bmean = 0.0;
    if (CACHE && !SPEC && tid < a.nms) {
        bj = a.ms[tid].j;
        bs0 = a.src0[bj]; bti = a.tinfo[bj]; bl0 = a.lp0[bj]; bmean = a.mean[bj];
    }
    if (CACHE && is_b) {
        cd[0] = 0.0; cd[1] = cd[2] = cd[3] = -INFINITY;
        ci[0] = (unsigned)S << 16; ci[1] = 0; ci[2] = 0;
        if (tid < a.nms) {
            const int j = a.ms[tid].j, ti = a.tinfo[j];
            const int tn = ti & 255, tp = ti >> 8;
            cd[0] = a.mean[j];
            cd[1] = a.lp0[j];
            ci[0] = (unsigned)a.src0[j] | ((unsigned)j << 16);
            if (tn >= 1) { ci[1] = (unsigned)a.tsrc[tp]; cd[2] = a.tlp[tp]; }
            if (tn >= 2) { ci[1] |= (unsigned)a.tsrc[tp + 1] << 16; cd[3] = a.tlp[tp + 1]; }
            ci[2] = (unsigned)(((tp + 2) << 8) | (tn > 2 ? tn - 2 : 0));  // left for the loop
        }
    }
    int par = 0;  // cur = column par, prev = column par ^ 1
    {   // first column: viterbi.jl:55-63 at the start of the signal, flat (emissions) elsewhere
        const double y0 = a.y[w];
        for (int j = tid; j < S; j += nt) {
            const double v = (w == 0 && j == 0) ? 0.0 : funcl_m(y0, a.mean[j], c0, den, rden);
            if (GCOL) gcol[j] = v; else sh[j] = v;  // column 0
        }
    }
    double ynext = a.y[w + 1 < e ? w + 1 : w];
    for (int64_t t = w + 1; t < e; t++) {
        double gl = INFINITY;
        const double yt = ynext;
        ynext = a.y[t + 1 < e ? t + 1 : t];  // one sample ahead: its latency hides behind this column
        if (GCOL) { __threadfence_block(); __syncthreads(); }
        else lds_barrier();
        par ^= 1;
        if (t == s && s > 0)  // the column of sample s-1 is what the boundary certificate compares
            for (int j = tid; j < S; j += nt)
                warm[j] = GCOL ? gcol[(par ^ 1) * S + j] : sh[(par ^ 1) * CS + j];
        const bool own = t >= s;
        int16_t *psi = a.T2c + (int64_t)a.nms * t;
        if (CACHE) {  // columns in LDS
            const double *prev = sh + (par ^ 1) * CS;
            double *cur = sh + par * CS;
            if (!is_b) {  // always true without SPEC
                // phase A: the first transition of every state, no branches.  A finite value always
                // beats fill(-Inf) (:80) and a -Inf stays -Inf, so the compare is dropped; states with
                // more transitions go to the dummy slot here and are computed by the phase-B waves.
                double pv[NS];  // all reads of the previous column before any write of this one
#pragma unroll
                for (int i = 0; i < NS; i++) pv[i] = prev[ci[i] & 0xffffu];
#pragma unroll
                for (int i = 0; i < NS; i++) {
                    const int wj = (int)(ci[i] >> 16);
                    const double v = (pv[i] + cd[ND + i]) + funcl_m(yt, cd[i], c0, den, rden);  // :79, :85-87
                    cur[wj] = v;
                }
            }
            if (!SPEC) {
                for (int m = tid; m < a.nms; m += nt) {
                    const int j = (m == tid) ? bj : a.ms[m].j;
                    const int s0 = (m == tid) ? bs0 : a.src0[j];
                    const double tt = prev[s0] + ((m == tid) ? bl0 : a.lp0[j]);
                    const bool up = tt > -INFINITY;
                    double best = up ? tt : -INFINITY;
                    int arg = up ? s0 + 1 : 1;
                    const int ti = (m == tid) ? bti : a.tinfo[j];
                    if (TLDS) block_tail(prev, ti, l_tsrc, l_tlp, best, arg, gl);
                    else block_tail(prev, ti, a.tsrc, a.tlp, best, arg, gl);
                    const double v = best + funcl_m(yt, (m == tid) ? bmean : a.mean[j], c0, den, rden);
                    cur[j] = v;
                    if (own) psi[m] = (int16_t)arg;
                }
            } else if (is_b) {
                // phase B: list order, strict '>' (:76-84); absent transitions carry lp = -Inf
                if (tid < a.nms) {
                    const int s0 = (int)(ci[0] & 0xffffu), j = (int)(ci[0] >> 16);
                    const int s1 = (int)(ci[1] & 0xffffu), s2 = (int)(ci[1] >> 16);
                    const double t0 = prev[s0] + cd[1], t1 = prev[s1] + cd[2], t2 = prev[s2] + cd[3];
                    double best = -INFINITY;
                    int arg = 1;
                    if (t0 > best) { best = t0; arg = s0 + 1; }
                    { const double gd = fabs(t1 - best); if (gd < kGapCoarse && gd > 0.0) gl = gd < gl ? gd : gl; }
                    if (t1 > best) { best = t1; arg = s1 + 1; }
                    { const double gd = fabs(t2 - best); if (gd < kGapCoarse && gd > 0.0) gl = gd < gl ? gd : gl; }
                    if (t2 > best) { best = t2; arg = s2 + 1; }
                    if (ci[2] & 255u) {
                        if (TLDS) block_tail(prev, (int)ci[2], l_tsrc, l_tlp, best, arg, gl);
                        else block_tail(prev, (int)ci[2], a.tsrc, a.tlp, best, arg, gl);
                    }
                    const double v = best + funcl_m(yt, cd[0], c0, den, rden);
                    cur[j] = v;
                    if (own) psi[tid] = (int16_t)arg;
                }
                for (int m = tid + nbt; m < a.nms; m += nbt) {  // more such states than threads
                    const int j = a.ms[m].j, s0 = a.src0[j], ti = a.tinfo[j];
                    const double tt = prev[s0] + a.lp0[j];
                    const bool up = tt > -INFINITY;
                    double best = up ? tt : -INFINITY;
                    int arg = up ? s0 + 1 : 1;
                    if (TLDS) block_tail(prev, ti, l_tsrc, l_tlp, best, arg, gl);
                    else block_tail(prev, ti, a.tsrc, a.tlp, best, arg, gl);
                    const double v = best + funcl_m(yt, a.mean[j], c0, den, rden);
                    cur[j] = v;
                    if (own) psi[m] = (int16_t)arg;
                }
            }
        } else {
            // same two phases with the constants re-read every sample
            const double *prevl = sh + (par ^ 1) * CS;
            const double *prevg = GCOL ? gcol + (par ^ 1) * S : nullptr;
#pragma unroll 2
            for (int j = tid; j < S; j += nt) {
                const int s0 = a.src0[j], ti = a.tinfo[j];
                const double l0 = a.lp0[j], q = funcl_m(yt, a.mean[j], c0, den, rden);
                const double tt = (GCOL ? prevg[s0] : prevl[s0]) + l0;
                const double v = (tt > -INFINITY ? tt : -INFINITY) + q;
                if ((ti & 255) == 0) {
                    if (GCOL) gcol[par * S + j] = v; else sh[par * CS + j] = v;
                }
            }
            for (int m = tid; m < a.nms; m += nt) {
                const MsRec r = a.ms[m];
                const int j = r.j, s0 = r.s0, ti = r.ti;
                const double l0 = r.lp0, q = funcl_m(yt, r.mean, c0, den, rden);
                const double tt = (GCOL ? prevg[s0] : prevl[s0]) + l0;
                const bool up = tt > -INFINITY;
                double best = up ? tt : -INFINITY;
                int arg = up ? s0 + 1 : 1;
                if (GCOL) {
                    if (TLDS) block_tail(prevg, ti, l_tsrc, l_tlp, best, arg, gl);
                    else block_tail(prevg, ti, a.tsrc, a.tlp, best, arg, gl);
                } else {
                    if (TLDS) block_tail(prevl, ti, l_tsrc, l_tlp, best, arg, gl);
                    else block_tail(prevl, ti, a.tsrc, a.tlp, best, arg, gl);
                }
                const double v = best + q;
                if (GCOL) gcol[par * S + j] = v; else sh[par * CS + j] = v;
                if (own) psi[m] = (int16_t)arg;
            }
        }
        if (own) publish_gap(a.gapmin, c, gl);
    }
    __syncthreads();
    for (int j = tid; j < S; j += nt)
        a.endv[(int64_t)c * S + j] = GCOL ? gcol[par * S + j] : sh[par * CS + j];
}

// Models of 4 097 .. 12 288 states (N=3, K=60 has 10 621): ONE column in LDS, updated in place.  Every thread first reads all the sources it needs into
// registers, a barrier, then writes its states.  One workgroup per CU, 128 VGPRs.
template <int SPT>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gen_vit_block1(BlockArgs a)
{
    extern __shared__ double sh[];  // column [S] + dummy slot, then the tails
    const int c = blockIdx.x, S = a.S, tid = threadIdx.x, nt = blockDim.x;
    double *col = sh;
    double *l_tlp = sh + (S + 1);
    int32_t *l_tsrc = (int32_t *)(l_tlp + a.ntail);
    for (int i = tid; i < a.ntail; i += nt) { l_tlp[i] = a.tlp[i]; l_tsrc[i] = a.tsrc[i]; }
    const int64_t s = (int64_t)c * a.B;
    const int64_t e = (s + a.B < a.T) ? s + a.B : a.T;
    const int64_t w = (s - a.H > 0) ? s - a.H : 0;
    const double c0 = a.c0, den = a.den, rden = a.rden;
    double *warm = a.warmv + (int64_t)c * S;
    double cd[2 * SPT];   // mean, lp of the first transition
    unsigned ci[SPT];     // source | store index << 16 (S = dummy: multi-source or past the end)
#pragma unroll
    for (int i = 0; i < SPT; i++) {
        const int jj = tid + i * nt;
        const int j = jj < S ? jj : S - 1;
        cd[i] = a.mean[j];
        cd[SPT + i] = a.lp0[j];
        const int wj = (jj < S && (a.tinfo[j] & 255) == 0) ? j : S;
        ci[i] = (unsigned)a.src0[j] | ((unsigned)wj << 16);
    }
    const bool has_b = tid < a.nms;  // host guarantees nms <= blockDim.x
    int bj = 0, bs0 = 0, bti = 0;
    double bl0 = 0.0, bmean = 0.0;
    if (has_b) {
        bj = a.ms[tid].j;
        bs0 = a.src0[bj]; bti = a.tinfo[bj]; bl0 = a.lp0[bj]; bmean = a.mean[bj];
    }
    {
        const double y0 = a.y[w];
        for (int j = tid; j < S; j += nt) {
            const double v = (w == 0 && j == 0) ? 0.0 : funcl_m(y0, a.mean[j], c0, den, rden);
            col[j] = v;
        }
    }
    double ynext = a.y[w + 1 < e ? w + 1 : w];
    for (int64_t t = w + 1; t < e; t++) {
        double gl = INFINITY;
        const double yt = ynext;
        ynext = a.y[t + 1 < e ? t + 1 : t];
        const bool own = t >= s;
        lds_barrier();  // the column of sample t-1 is complete
        if (t == s)     // ... and for t = s it is the warm column the certificate looks at
            for (int j = tid; j < S; j += nt) warm[j] = col[j];
        double pv[SPT];
#pragma unroll
        for (int i = 0; i < SPT; i++) pv[i] = col[ci[i] & 0xffffu];
        double best = -INFINITY;
        int arg = 1;
        if (has_b) {
            const double tt = col[bs0] + bl0;
            if (tt > best) { best = tt; arg = bs0 + 1; }
            block_tail(col, bti, l_tsrc, l_tlp, best, arg, gl);
        }
        lds_barrier();  // every read of sample t-1 has returned: overwrite in place
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            const int wj = (int)(ci[i] >> 16);
            const double v = (pv[i] + cd[SPT + i]) + funcl_m(yt, cd[i], c0, den, rden);
            col[wj] = v;
        }
        if (has_b) {
            const double v = best + funcl_m(yt, bmean, c0, den, rden);
            col[bj] = v;
            if (own) a.T2c[(int64_t)a.nms * t + tid] = (int16_t)arg;
        }
        if (own) publish_gap(a.gapmin, c, gl);
    }
    __syncthreads();
    for (int j = tid; j < S; j += nt) a.endv[(int64_t)c * S + j] = col[j];
}

// Models whose column does not fit LDS at all (N=4, K=60 with overlaps: 21 123 states, the largest
// the reference's Int16 ids and its CLI default of 4 templates allow): the two columns stay in a
// per-block global scratch (L2), but the per-state constants are held in registers instead of being
// re-read every sample -- mean (2 VGPRs) and one packed word: source (15 bits) | index into an LDS
// dictionary of the distinct first-transition log-probabilities (8 bits; overlap models have a
// handful) | "has more transitions" flag.  Per sample and workgroup the L2 traffic drops from
// 40 to 16 bytes per state.  One workgroup per CU, 128 VGPRs.
template <int SPT>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gen_vit_blockG(BlockArgs a, const double *__restrict__ lpdict_g, int ndict,
                    const uint8_t *__restrict__ lpidx)
{
    extern __shared__ double sh[];  // dictionary, then the tails
    constexpr int nt = 1024;  // launch contract; compile-time so that per-state addresses are not hoisted
    const int c = blockIdx.x, S = a.S, tid = threadIdx.x;
    double *lpd = sh;
    double *l_tlp = sh + 256;
    int32_t *l_tsrc = (int32_t *)(l_tlp + a.ntail);
    for (int i = tid; i < ndict; i += nt) lpd[i] = lpdict_g[i];
    for (int i = tid; i < a.ntail; i += nt) { l_tlp[i] = a.tlp[i]; l_tsrc[i] = a.tsrc[i]; }
    const int64_t s = (int64_t)c * a.B;
    const int64_t e = (s + a.B < a.T) ? s + a.B : a.T;
    const int64_t w = (s - a.H > 0) ? s - a.H : 0;
    const double c0 = a.c0, den = a.den, rden = a.rden;
    double *warm = a.warmv + (int64_t)c * S;
    double *gcol = a.gbuf + (int64_t)c * 2 * S;
    double mean_r[SPT];
    unsigned ci[SPT];  // source | dictionary index << 15 | single-source & in-range flag << 23
#pragma unroll
    for (int i = 0; i < SPT; i++) {
        const int jj = tid + i * nt;
        const int j = jj < S ? jj : S - 1;
        mean_r[i] = a.mean[j];
        const unsigned wr = (jj < S && (a.tinfo[j] & 255) == 0) ? 1u : 0u;
        ci[i] = (unsigned)a.src0[j] | ((unsigned)lpidx[j] << 15) | (wr << 23);
    }
    {
        const double y0 = a.y[w];
        for (int j = tid; j < S; j += nt)
            gcol[j] = (w == 0 && j == 0) ? 0.0 : funcl_m(y0, a.mean[j], c0, den, rden);
    }
    int par = 0;
    double ynext = a.y[w + 1 < e ? w + 1 : w];
    __syncthreads();
    for (int64_t t = w + 1; t < e; t++) {
        double gl = INFINITY;
        const double yt = ynext;
        ynext = a.y[t + 1 < e ? t + 1 : t];
        __threadfence_block();
        __syncthreads();
        par ^= 1;
        const double *prev = gcol + (par ^ 1) * S;
        double *cur = gcol + par * S;
        if (t == s && s > 0)
            for (int j = tid; j < S; j += nt) warm[j] = prev[j];
        const bool own = t >= s;
        // prev and cur are different arrays: batches of 7 loads in flight are enough
#pragma unroll
        for (int i0 = 0; i0 < SPT; i0 += 7) {
            double pv[7];
#pragma unroll
            for (int i = 0; i < 7; i++) pv[i] = prev[ci[i0 + i] & 0x7fffu];
#pragma unroll
            for (int i = 0; i < 7; i++) {
                const unsigned w_ = ci[i0 + i];
                const double v = (pv[i] + lpd[(w_ >> 15) & 255u]) + funcl_m(yt, mean_r[i0 + i], c0, den, rden);
                if (w_ >> 23) cur[tid + (i0 + i) * nt] = v;
            }
        }
        for (int m = tid; m < a.nms; m += nt) {
            const MsRec r = a.ms[m];
            const double tt = prev[r.s0] + r.lp0;
            const bool up = tt > -INFINITY;
            double best = up ? tt : -INFINITY;
            int arg = up ? r.s0 + 1 : 1;
            block_tail(prev, r.ti, l_tsrc, l_tlp, best, arg, gl);
            cur[r.j] = best + funcl_m(yt, r.mean, c0, den, rden);
            if (own) a.T2c[(int64_t)a.nms * t + m] = (int16_t)arg;
        }
        if (own) publish_gap(a.gapmin, c, gl);
    }
    __threadfence_block();
    __syncthreads();
    for (int j = tid; j < S; j += nt) a.endv[(int64_t)c * S + j] = gcol[par * S + j];
}

// Boundary certificate: spread over the states of (warm column of block c) - (end column of block
// c-1), both at sample c*B-1.  diag[0] counts failing boundaries, diag[2] holds the largest spread.
__global__ __launch_bounds__(256) void k_block_check(const double *__restrict__ endv,
                                                     const double *__restrict__ warmv, int S,
                                                     unsigned long long *diag,
                                                     double *__restrict__ frame)  // [nblk][2]
{
    __shared__ double smin[256], smax[256], sabs[256];
    __shared__ int sbad[256];
    const int c = blockIdx.x + 1, tid = threadIdx.x;
    const double *wv = warmv + (int64_t)c * S, *ev = endv + (int64_t)(c - 1) * S;
    double lo = INFINITY, hi = -INFINITY, mabs = 0.0;
    int bad = 0;
    for (int j = tid; j < S; j += 256) {
        const double a = wv[j], b = ev[j];
        const bool fa = a > -INFINITY && a < INFINITY, fb = b > -INFINITY && b < INFINITY;
        if (fa && fb) {
            const double d = a - b;
            lo = d < lo ? d : lo;
            hi = d > hi ? d : hi;
            mabs = fabs(a) > mabs ? fabs(a) : mabs;
        } else if (fa != fb || a != a || b != b) {
            bad = 1;  // reachable in one frame only, or NaN
        }
    }
    smin[tid] = lo; smax[tid] = hi; sbad[tid] = bad; sabs[tid] = mabs;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            smin[tid] = smin[tid + o] < smin[tid] ? smin[tid + o] : smin[tid];
            smax[tid] = smax[tid + o] > smax[tid] ? smax[tid + o] : smax[tid];
            sbad[tid] |= sbad[tid + o];
            sabs[tid] = sabs[tid + o] > sabs[tid] ? sabs[tid + o] : sabs[tid];
        }
        __syncthreads();
    }
    if (tid == 0) {
        double spread = (smax[0] >= smin[0]) ? smax[0] - smin[0] : 0.0;
        if (sbad[0]) spread = INFINITY;
        if (!(spread <= kSpreadTol)) atomicAdd(&diag[0], 1ull);
        atomicMax(&diag[2], (unsigned long long)__double_as_longlong(spread));
        // block c's frame = block c-1's frame + (end column of c-1) - (warm column of c)
        frame[2 * c] = (smax[0] >= smin[0]) ? -0.5 * (smax[0] + smin[0]) : 0.0;
        frame[2 * c + 1] = sabs[0];
    }
}

// Near-ties: a block whose smallest non-zero candidate gap is below the rounding granularity the
// REFERENCE's trellis has there (|its values| ~ |frame offset| + |block values|) may have told two
// candidates apart that the reference rounds to a tie.  diag[7] counts such blocks; the host-buffer
// entry point then decodes with the strict engine (duplicate templates are the systematic case).
__global__ void k_block_ties(const double *__restrict__ frame, const unsigned long long *__restrict__ gapmin,
                             int nblk, unsigned long long *diag)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double off = 0.0;
    unsigned long long n = 0;
    for (int c = 1; c < nblk; c++) {  // block 0 works in the reference's own frame
        off += frame[2 * c];
        const double mag = fabs(off) + frame[2 * c + 1] + 1.0;
        const double gap = __longlong_as_double((long long)gapmin[c]);
        if (gap <= 8.0 * 2.220446049250313e-16 * mag) n++;  // a few ulps of the reference's values
    }
    diag[7] = n;
}

// Back-pointer lookup.  bt[j] > 0: the only source of state j+1; bt[j] <= 0: minus the index of
// state j+1 among the stored pointers of a sample (a row of T2c).
__device__ __forceinline__ int back_step(const int32_t *bt, const int16_t *row, int v)
{
    const int code = bt[v - 1];
    return code > 0 ? code : ((int)row[-code] & 0x7fff);   // bit 15: near-tie flag of the pair sweep (pair_sweep.hip)
}

// All S end states of block c walked back to its first sample, rows of T2c staged through LDS W at
// a time.  Once every walker sits in the same state the rest of the block's path is known: thread 0
// writes it to x and merged[c] records the last sample written (c*B-1 when nothing was).
// fconst[c] = state at sample c*B-1 when it no longer depends on the end state (-1 otherwise, then
// fmap[c][j] holds it for end state j+1).  Block 0 has no predecessor: fconst[0] = 1, unused.
// BTM: where the per-state codes live -- 0 global (int32), 1 LDS (int32), 2 LDS as int16 (every code fits: sources are
// state ids <= 32767, rows are minus an index below nms); with 2 a model of 21 123 states keeps its codes beside the
// walkers and the staged rows (116 KB) instead of paying an L2 round trip per walker and sample.
template <int BTM>
__global__ __launch_bounds__(1024) void k_block_map(const int16_t *__restrict__ T2c,
                                                   const int32_t *__restrict__ btg, int nms, int W,
                                                   int bt_lds, int64_t T, int S, int B,
                                                   int16_t *__restrict__ fmap,
                                                   int16_t *__restrict__ fconst,
                                                   int64_t *__restrict__ merged,
                                                   int16_t *__restrict__ x)
{
    extern __shared__ int16_t lds16[];
    __shared__ int mn, mx;
    int16_t *wk = lds16;                                  // S walkers
    int16_t *rows = wk + ((S + 3) & ~3);                  // W staged rows
    int32_t *btl = (int32_t *)(rows + ((W * nms + 3) & ~3));
    int16_t *btl16 = (int16_t *)btl;
    const int nth = blockDim.x;
    const int c = blockIdx.x, tid = threadIdx.x;
    const int64_t s = (int64_t)c * B;
    const int64_t e = (s + B < T) ? s + B : T;
    if (BTM == 1)
        for (int j = tid; j < S; j += nth) btl[j] = btg[j];
    if (BTM == 2)
        for (int j = tid; j < S; j += nth) btl16[j] = (int16_t)btg[j];
    const int32_t *bt = BTM == 1 ? btl : btg;
    auto bstep = [&](const int16_t *row, int v) {
        const int code = BTM == 2 ? (int)btl16[v - 1] : bt[v - 1];
        return code > 0 ? code : ((int)row[-code] & 0x7fff);
    };
    auto bcode0 = [&]() { return BTM == 2 ? (int)btl16[0] : bt[0]; };
    for (int j = tid; j < S; j += nth) wk[j] = (int16_t)(j + 1);
    bool one = (S == 1);
    int cur = 1;              // thread 0: the merged walker
    int64_t mrg = s - 1;      // last sample written to x
    if (one && tid == 0 && e - 1 >= s) { mrg = e - 1; x[e - 1] = 1; }
    // row r maps the state at sample r to the state at sample r-1
    for (int64_t hi = e - 1; hi >= s + 1; hi -= W) {
        const int64_t lo = (hi - W + 1 > s + 1) ? hi - W + 1 : s + 1;
        const int n = (int)(hi - lo + 1);
        __syncthreads();
        const int16_t *src = T2c + (int64_t)nms * lo;
        for (int i = tid; i < n * nms; i += nth) rows[i] = src[i];
        __syncthreads();
        if (!one) {
            for (int j = tid; j < S; j += nth) {
                int v = wk[j];
                for (int r = n - 1; r >= 0; r--) v = bstep(rows + r * nms, v);
                wk[j] = (int16_t)v;
            }
            if (tid == 0) { mn = 32767; mx = 0; }
            __syncthreads();
            int l = 32767, h = 0;
            for (int j = tid; j < S; j += nth) {
                const int v = wk[j];
                l = v < l ? v : l;
                h = v > h ? v : h;
            }
            atomicMin(&mn, l);
            atomicMax(&mx, h);
            __syncthreads();
            one = (mn == mx);
            if (one && tid == 0) {  // all walkers agree on the state at sample lo-1
                cur = wk[0];
                mrg = lo - 1;
                x[lo - 1] = (int16_t)cur;
            }
        } else if (tid < 64) {
            // The single walker: two dependent LDS reads per sample.  Most samples sit in the silent state and stay
            // there: bit r of `stay` says that at row r the silent state's pointer is the silent state, and the walker
            // crosses such a stretch without touching the rows again (n <= 64 rows are staged at a time).
            const int m0 = bcode0() <= 0 ? -bcode0() : -1;
            const bool st = m0 >= 0 && tid < n && ((int)rows[tid * nms + m0] & 0x7fff) == 1;
            const unsigned long long stay = __ballot(st);
            if (tid == 0) {
                int r = n - 1;
                while (r >= 0) {
                    if (cur == 1) {
                        int run = __clzll((long long)~(stay << (63 - r)));   // rows r, r-1, ... that stay silent
                        run = run > r + 1 ? r + 1 : run;
                        for (int q = 0; q < run; q++) x[lo + r - q - 1] = 1;
                        r -= run;
                        if (r < 0) break;
                    }
                    cur = bstep(rows + r * nms, cur);
                    x[lo + r - 1] = (int16_t)cur;
                    r--;
                }
            }
        }
    }
    __syncthreads();
    // walkers (or cur) now hold the state at sample s; one more row reaches the previous block
    const int16_t *row_s = T2c + (int64_t)nms * s;
    if (one) {
        if (tid == 0) {
            merged[c] = mrg;
            fconst[c] = (int16_t)((c > 0) ? bstep(row_s, cur) : 1);
        }
    } else {
        if (tid == 0) { merged[c] = s - 1; fconst[c] = (int16_t)((c > 0) ? -1 : 1); }
        if (c > 0)
            for (int j = tid; j < S; j += nth)
                fmap[(int64_t)c * S + j] = (int16_t)bstep(row_s, wk[j]);
    }
}

// argmax of the last column (first maximum, viterbi.jl:90), then the end state of every block
__global__ __launch_bounds__(256) void k_block_compose(const double *__restrict__ endv,
                                                       const int16_t *__restrict__ fmap,
                                                       const int16_t *__restrict__ fconst, int S,
                                                       int nblk, int16_t *__restrict__ endstate, int chain)
{
    __shared__ double bv[256];
    __shared__ int bi[256];
    const int tid = threadIdx.x;
    const double *last = endv + (int64_t)(nblk - 1) * S;
    double v = -INFINITY;
    int idx = S;  // first maximum = smallest index among the largest values; NaN never wins (:90)
    for (int j = tid; j < S; j += 256) {
        const double a = last[j];
        if (idx == S ? a == a : a > v) { v = a; idx = j; }
    }
    bv[tid] = v; bi[tid] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const double a = bv[tid + o];
            const int ia = bi[tid + o];
            if (ia < S && (bi[tid] == S || a > bv[tid] || (a == bv[tid] && ia < bi[tid]))) {
                bv[tid] = a; bi[tid] = ia;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        // the reference's scan starts at state 1 and only moves on a strict '>': a NaN there stays
        int st = (bi[0] < S && last[0] == last[0] ? bi[0] : 0) + 1;
        endstate[nblk - 1] = (int16_t)st;
        for (int c = nblk - 1; c >= 1 && chain; c--) {
            const int f = fconst[c];
            st = f > 0 ? f : fmap[(int64_t)c * S + (st - 1)];
            endstate[c - 1] = (int16_t)st;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Backtrace by segments (models with wide back-pointer rows: the overlap models of three and more templates keep
// ~1 000 pointers per sample, and staging every row through LDS for ONE walker reads all of T2c).  The time axis is cut
// into segments of SEG samples, one THREAD per segment: it starts WI samples above its segment in the silent state
// (a guess), walks down -- one dependent 2-byte load per sample, thousands of walkers in flight hide the latency --
// and writes its segment from the state it arrived with (guess[k]); below[k] is the state it implies for the last
// sample of segment k-1.  The last segment starts from the arg-max (viterbi.jl:90).  Back-pointers are deterministic, so
// the path is the reference's as soon as every boundary is consistent, guess[k] == below[k+1] for all k (induction from
// the last segment); k_seg_fix re-walks a segment whose guess was wrong from the true state and stops where it meets
// the path it wrote before; boundaries still open after the fix passes are counted in diag[0] (the host-buffer entry
// points then fall back like for a failed warm-up certificate).  No assumption is made -- the check is exact.
template <bool BTL>
__device__ __forceinline__ int seg_back(const int16_t *__restrict__ T2c, const int32_t *__restrict__ btg,
                                        const int16_t *btl, int nms, int64_t t, int v)
{
    const int code = BTL ? (int)btl[v - 1] : btg[v - 1];
    return code > 0 ? code : ((int)T2c[t * nms - code] & 0x7fff);
}

template <bool BTL>
__global__ __launch_bounds__(256) void k_seg_walk(const int16_t *__restrict__ T2c, const int32_t *__restrict__ btg,
                                                  int nms, int64_t T, int S, int SEG, int WI, int nseg,
                                                  const int16_t *__restrict__ laststate, int16_t *__restrict__ x,
                                                  int16_t *__restrict__ guess, int16_t *__restrict__ below)
{
    extern __shared__ int16_t seg_bt[];
    if (BTL) {
        for (int j = threadIdx.x; j < S; j += 256) seg_bt[j] = (int16_t)btg[j];
        __syncthreads();
    }
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nseg) return;
    const int64_t lo = (int64_t)k * SEG;
    const int64_t hi = (lo + SEG < T ? lo + SEG : T) - 1;
    int v;
    if (k == nseg - 1) v = laststate[0];
    else {
        const int64_t t0 = hi + WI < T - 1 ? hi + WI : T - 1;
        v = 1;
        for (int64_t t = t0; t > hi; t--) v = seg_back<BTL>(T2c, btg, seg_bt, nms, t, v);
    }
    guess[k] = (int16_t)v;
    for (int64_t t = hi; t >= lo; t--) {
        x[t] = (int16_t)v;
        if (t > 0) v = seg_back<BTL>(T2c, btg, seg_bt, nms, t, v);
    }
    below[k] = (int16_t)v;
}

template <bool BTL>
__global__ __launch_bounds__(256) void k_seg_fix(const int16_t *__restrict__ T2c, const int32_t *__restrict__ btg,
                                                 int nms, int64_t T, int S, int SEG, int nseg,
                                                 int16_t *__restrict__ x, int16_t *__restrict__ guess,
                                                 int16_t *__restrict__ below, int final_pass, unsigned long long *diag)
{
    extern __shared__ int16_t seg_bt[];
    const int k = blockIdx.x * 256 + threadIdx.x;
    const bool todo = k < nseg - 1 && guess[k] != below[k + 1];
    if (final_pass) {
        if (todo) atomicAdd(&diag[0], 1ull);
        return;
    }
    if (!__syncthreads_or(todo)) return;
    if (BTL) {
        for (int j = threadIdx.x; j < S; j += 256) seg_bt[j] = (int16_t)btg[j];
        __syncthreads();
    }
    if (!todo) return;
    int v = below[k + 1];
    guess[k] = (int16_t)v;
    const int64_t lo = (int64_t)k * SEG, hi = lo + SEG - 1;
    for (int64_t t = hi; t >= lo; t--) {
        if (x[t] == (int16_t)v) return;        // met the path written before: everything below is unchanged
        x[t] = (int16_t)v;
        if (t > 0) v = seg_back<BTL>(T2c, btg, seg_bt, nms, t, v);
    }
    below[k] = (int16_t)v;
}

// the part of each block's path above the merge point
__global__ void k_block_finish(const int16_t *__restrict__ T2c, const int32_t *__restrict__ bt,
                               int nms, int64_t T, int S, int B, int nblk,
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
        v = back_step(bt, T2c + (int64_t)nms * t, v);
        t--;
        x[t] = (int16_t)v;
    }
}

// Path values of one block in the block's own frame (p = 0 before its first sample): the reference accumulates
// p_t = (p_(t-1) + lp) + q (viterbi.jl:85-87) and returns the sum of p_t over t >= 2 (:92-96).  part[c] = {last p, sum of
// p over the block's samples >= 1, count}.  With inc_t = lp + q:  sum_t p_t = n p_0 + sum_j inc_j (n - j): two plain
// reductions instead of one thread walking the block (0.9 -> 0.2 ms at 10 M samples); the roundings differ from the
// serial order at the 1e-16 level per term, ll is compared at 1e-9.
__global__ __launch_bounds__(256) void k_block_ll(const double *__restrict__ y,
                                                  const int16_t *__restrict__ x, int64_t T, int S,
                                                  int B, const double *__restrict__ mean,
                                                  const int32_t *__restrict__ in_ptr,
                                                  const int32_t *__restrict__ in_src,
                                                  const double *__restrict__ in_lp, double c0,
                                                  double den, double *__restrict__ part)
{
    __shared__ double red[8];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int64_t s = (int64_t)c * B;
    const int64_t e = (s + B < T) ? s + B : T;
    // samples that count in the sum: t >= 1; every p_t with t in [max(s,1), e) contains the increments of s..t
    double a_inc = 0.0, a_w = 0.0;
    for (int64_t t = s + tid; t < e; t += 256) {
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
        const double inc = lp + q;
        a_inc += inc;
        // p_u contains inc_t for every u >= t; the sum runs over u in [max(s, 1), e): u >= max(t, 1)
        const int64_t first = t > 1 ? t : 1;
        a_w += inc * (double)(e - (first > s ? first : s));
    }
    for (int o = 32; o > 0; o >>= 1) { a_inc += __shfl_xor(a_inc, o); a_w += __shfl_xor(a_w, o); }
    if ((tid & 63) == 0) { red[tid >> 6] = a_inc; red[4 + (tid >> 6)] = a_w; }
    __syncthreads();
    if (tid == 0) {
        part[3 * c] = (red[0] + red[1]) + (red[2] + red[3]);
        part[3 * c + 1] = (red[4] + red[5]) + (red[6] + red[7]);
        part[3 * c + 2] = (double)((e - s) - (s == 0 ? 1 : 0));
    }
}

__global__ __launch_bounds__(256) void k_block_ll_sum(const double *__restrict__ part, int nblk,
                                                      double *__restrict__ ll)
{
    extern __shared__ double sp[];  // 3*nblk when it fits (lds != 0)
    const bool staged = nblk * 3 * sizeof(double) <= 60 * 1024;
    if (staged) {
        for (int i = threadIdx.x; i < 3 * nblk; i += 256) sp[i] = part[i];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const double *p = staged ? sp : part;
    double off = 0.0, acc = 0.0;
    for (int c = 0; c < nblk; c++) {
        acc += p[3 * c + 2] * off + p[3 * c + 1];
        off += p[3 * c];
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
    int64_t b = block_req > 0 ? block_req : std::max<int64_t>(2 * h, (T + 1279) / 1280);   // ~1 280 blocks: five rounds of one workgroup per CU, or all the pair sweep's wavefronts (five per CU) at once
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
    std::vector<int32_t> src0(S, 0), tinfo(S, 0), tsrc, bt(S, 1);
    std::vector<MsRec> ms;
    std::vector<double> dict;
    std::vector<uint8_t> didx(S, 0);
    bool dict_ok = true;
    for (int64_t j = 0; j < S; j++) {
        const int b = m.in_ptr[j], e = m.in_ptr[j + 1];
        if (e > b) { lp0[j] = m.in_lp[b]; src0[j] = m.in_src[b]; }
        const int nt = e > b ? e - b - 1 : 0;
        HS_CHECK(nt <= 255 && tsrc.size() < (1u << 22), HMMSORT_EUNSUP,
                 "blocked engine: in-degree %d of state %lld too large", nt + 1, (long long)j + 1);
        tinfo[j] = (int32_t)((tsrc.size() << 8) | (unsigned)nt);
        for (int q = b + 1; q < e; q++) { tsrc.push_back(m.in_src[q]); tlp.push_back(m.in_lp[q]); }
        bt[j] = src0[j] + 1;
        if (dict_ok) {  // dictionary of first-transition log-probabilities (bit patterns)
            size_t q = 0;
            while (q < dict.size() && memcmp(&dict[q], &lp0[j], 8) != 0) q++;
            if (q == dict.size()) { if (dict.size() < 256) dict.push_back(lp0[j]); else dict_ok = false; }
            if (dict_ok) didx[j] = (uint8_t)q;
        }
        if (nt > 0) {
            bt[j] = -(int32_t)ms.size();
            ms.push_back(MsRec{lp0[j], m.mean[j], (int32_t)j, src0[j], tinfo[j], 0});
        }
    }
    g->ndict = dict_ok ? (int)dict.size() : 0;
    dict.resize(256, 0.0);
    if (g->ntail < 0) {  // first call: allocate
        g->ntail = (int)tsrc.size();
        int rc;
        if ((rc = dalloc(&g->d_lp0, S, &g->bytes)) || (rc = dalloc(&g->d_src0, S, &g->bytes)) ||
            (rc = dalloc(&g->d_tinfo, S, &g->bytes)) ||
            (rc = dalloc(&g->d_tsrc, tsrc.size(), &g->bytes)) ||
            (rc = dalloc(&g->d_tlp, tlp.size(), &g->bytes)) ||
            (rc = dalloc((MsRec **)&g->d_ms, ms.size(), &g->bytes)) ||
            (rc = dalloc(&g->d_bt, S, &g->bytes)) || (rc = dalloc(&g->d_lpdict, 256, &g->bytes)) ||
            (rc = dalloc(&g->d_lpidx, S, &g->bytes)))
            return rc;
        g->nms = (int)ms.size();
    }
    HS_CHECK((int)tsrc.size() == g->ntail && (int)ms.size() == g->nms, HMMSORT_EINVAL,
             "set_model: transition structure changed");
    HS_HIP(hipMemcpy(g->d_bt, bt.data(), S * sizeof(int32_t), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(g->d_lpdict, dict.data(), 256 * sizeof(double), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(g->d_lpidx, didx.data(), S, hipMemcpyHostToDevice));
    if (!ms.empty())
        HS_HIP(hipMemcpy(g->d_ms, ms.data(), ms.size() * sizeof(MsRec), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(g->d_lp0, lp0.data(), S * sizeof(double), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(g->d_src0, src0.data(), S * sizeof(int32_t), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(g->d_tinfo, tinfo.data(), S * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!tsrc.empty()) {
        HS_HIP(hipMemcpy(g->d_tsrc, tsrc.data(), tsrc.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HS_HIP(hipMemcpy(g->d_tlp, tlp.data(), tlp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // two-template overlap models: the sweep that treats pair runs as delays (pair_sweep.hip); its back-pointer
    // rows are the multi-source states silent, A_k, B_k in state order
    g->pair_ok = false;
    g->multi_ok = false;
    if (!g->pair_off && !(getenv("HMMSORT_PAIR") && atoi(getenv("HMMSORT_PAIR")) == 0)) {
        std::vector<double> tab;
        const int64_t L = m.K - 1;
        bool ok = false, multi = false;
        if (m.N == 2) {
            ok = pair_analyze(m, tab) && (int64_t)ms.size() == 2 * L + 1;
            for (size_t q = 0; ok && q < ms.size(); q++) ok = ms[q].j == (int32_t)q;
        } else if (m.N >= 3 && m.N <= 5 && multi_lds_bytes((int)m.N, (int)L) <= 160 * 1024) {
            // three to five templates (multi_sweep.hip): rows = Z, the singles, the pair entries, in state order
            std::vector<int32_t> msj(ms.size());
            for (size_t q = 0; q < ms.size(); q++) msj[q] = ms[q].j;
            ok = multi = multi_rows_ok(m, msj) && multi_analyze(m, tab);
        }
        if (ok) {
            int rc;
            if (g->d_pairtab && g->pairtab_len < tab.size()) { (void)hipFree(g->d_pairtab); g->d_pairtab = nullptr; }
            if (!g->d_pairtab) {
                if ((rc = dalloc(&g->d_pairtab, tab.size(), &g->bytes))) return rc;
                g->pairtab_len = tab.size();
            }
            HS_HIP(hipMemcpy(g->d_pairtab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
            g->pair_ok = true;
            g->multi_ok = multi;
        }
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
    g->blk_cols_lds = 2 * (S + 1) * 8 <= 150 * 1024;
    // above 4096 states the register-cached two-column kernel no longer fits 64 VGPRs; one column
    // updated in place with cached constants (1 workgroup per CU) beats re-reading them (measured
    // 90 vs 36 Msamples/s around 9 500 states)
    g->blk_onecol = S > 4096 && (S + 1) * 8 + tail_b <= 158 * 1024 && S <= 12 * 1024 &&
                    g->nms <= 1024;
    g->blk_tail_lds = (g->blk_cols_lds ? 2 * (S + 1) * 8 : 0) + tail_b <= 150 * 1024;
    if ((rc = dalloc(&g->d_endv, nb * S, &g->bytes)) || (rc = dalloc(&g->d_warmv, nb * S, &g->bytes)) ||
        (rc = dalloc(&g->d_fmap, nb * S, &g->bytes)) || (rc = dalloc(&g->d_merged, nb, &g->bytes)) ||
        (rc = dalloc(&g->d_endstate, nb, &g->bytes)) || (rc = dalloc(&g->d_fconst, nb, &g->bytes)) || (rc = dalloc(&g->d_llpart, 3 * nb, &g->bytes)) ||
        (rc = dalloc(&g->d_bdiag, 8, &g->bytes)) || (rc = dalloc(&g->d_gapmin, nb, &g->bytes)) ||
        (rc = dalloc(&g->d_frame, 2 * nb, &g->bytes)) || (rc = dalloc(&g->d_qsum, 8, &g->bytes)))   // [0] sum y^2 of the pair sweep
        return rc;
    HS_HIP(hipMemset(g->d_frame, 0, 2 * nb * sizeof(double)));
    if (!g->blk_cols_lds && !g->blk_onecol && (rc = dalloc(&g->d_blkbuf, nb * 2 * S, &g->bytes)))
        return rc;
    HS_HIP(hipMemset(g->d_bdiag, 0, 8 * sizeof(unsigned long long)));
    return HMMSORT_OK;
}

void blocked_destroy(GenericDev *g)
{
    void *ptrs[] = {g->d_lp0, g->d_src0, g->d_tinfo, g->d_tsrc, g->d_tlp, g->d_endv, g->d_warmv,
                    g->d_fmap, g->d_merged, g->d_endstate, g->d_llpart, g->d_bdiag, g->d_blkbuf, g->d_ms, g->d_bt, g->d_fconst, g->d_lpdict, g->d_lpidx, g->d_gapmin, g->d_frame, g->d_pairtab, g->d_qsum, g->d_segbuf};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
}

template <int SPT, bool SPEC, bool GCOL, bool TLDS>
static int launch_block_sweep(GenericDev *g, const BlockArgs &a, int threads, size_t lds,
                              hipStream_t st)
{
    if (lds > 64 * 1024)
        HS_HIP(hipFuncSetAttribute((const void *)gen_vit_block<SPT, SPEC, GCOL, TLDS>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((gen_vit_block<SPT, SPEC, GCOL, TLDS>), dim3((unsigned)g->nblk), dim3(threads), lds, st, a);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int blocked_viterbi(GenericDev *g, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st)
{
    const int64_t T = g->T, S = g->S;
    if (!g->d_T2) {
        const double need = (double)std::max(g->nms, 1) * (double)T * 2.0;
        HS_CHECK(need < 220e9, HMMSORT_ENOMEM,
                 "Viterbi needs %.1f GB of back-pointers; decode in chunks", need / 1e9);
        int rc = dalloc(&g->d_T2, (size_t)std::max(g->nms, 1) * T, &g->bytes);
        if (rc) return rc;
    }
    BlockArgs a;
    a.y = d_y; a.T = T; a.S = (int)S; a.B = (int)g->B; a.H = (int)g->H;
    a.mean = g->d_mean; a.lp0 = g->d_lp0; a.src0 = g->d_src0; a.tinfo = g->d_tinfo;
    a.tsrc = g->d_tsrc; a.tlp = g->d_tlp; a.ntail = g->ntail;
    a.ms = (const MsRec *)g->d_ms; a.nms = g->nms;
    a.c0 = -kLog2Pi - g->lsig;
    a.den = 2.0 * (g->sigma * g->sigma);
    a.rden = 1.0 / a.den;
    a.T2c = g->d_T2; a.endv = g->d_endv; a.warmv = g->d_warmv; a.gapmin = g->d_gapmin;
    a.gbuf = g->blk_cols_lds ? nullptr : g->d_blkbuf;
    size_t lds = (g->blk_cols_lds ? 2 * (S + 1) * 8 : 0) + (g->blk_tail_lds ? (size_t)g->ntail * 12 + 8 : 0);
    // register-cached sweep: wave-specialised (nbthr phase-B threads + phase-A threads with <= 2
    // states each) when that fits 1024 threads, else every thread takes spt <= 4 states
    int threads = (int)std::min<int64_t>(1024, (S + 63) / 64 * 64), spt = 0, nbthr = 0;
    if (g->blk_cols_lds && g->blk_tail_lds) {
        const int nbw = std::min((g->nms + 63) / 64 * 64, 512);
        for (int k = 1; k <= 2 && !spt && g->nms > 0; k++) {
            const int na = (int)(((S + k - 1) / k + 63) / 64 * 64);
            if (nbw + na <= 1024) { spt = k; threads = nbw + na; nbthr = nbw; }
        }
        if (!spt && S <= 4096) spt = (int)((S + threads - 1) / threads) <= 2 ? (int)((S + threads - 1) / threads) : 4;
    }
    a.nbthr = nbthr;
    HS_HIP(hipMemsetAsync(g->d_bdiag, 0, 8 * sizeof(unsigned long long), st));
    HS_HIP(hipMemsetAsync(g->d_gapmin, 0x7f, (size_t)g->nblk * sizeof(unsigned long long), st));  // huge
    int rc;
    const bool gcol = !g->blk_cols_lds, tl = g->blk_tail_lds;
    if (g->pair_ok) {
        rc = g->multi_ok ? multi_sweep_launch(g, d_y, st) : pair_sweep_launch(g, d_y, st);
    } else if (g->blk_onecol) {
        const size_t lds1 = (size_t)(S + 1) * 8 + (size_t)g->ntail * 12 + 8;
        const int spt1 = (int)((S + 1023) / 1024);
        auto go = [&](auto kern) -> int {
            HS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds1));
            hipLaunchKernelGGL(kern, dim3((unsigned)g->nblk), dim3(1024), lds1, st, a);
            HS_HIP(hipGetLastError());
            return HMMSORT_OK;
        };
        if (spt1 <= 6) rc = go(gen_vit_block1<6>);
        else if (spt1 <= 8) rc = go(gen_vit_block1<8>);
        else if (spt1 <= 10) rc = go(gen_vit_block1<10>);
        else if (spt1 <= 11) rc = go(gen_vit_block1<11>);
        else rc = go(gen_vit_block1<12>);
    } else if (gcol && g->ndict > 0 && S <= 21 * 1024 && (size_t)256 * 8 + (size_t)g->ntail * 12 + 8 <= 150 * 1024) {
        const size_t ldsg = (size_t)256 * 8 + (size_t)g->ntail * 12 + 8;
        const int sptg = (int)((S + 1023) / 1024);
        auto go = [&](auto kern) -> int {
            if (ldsg > 64 * 1024)
                HS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)ldsg));
            hipLaunchKernelGGL(kern, dim3((unsigned)g->nblk), dim3(1024), ldsg, st, a, g->d_lpdict, g->ndict,
                               g->d_lpidx);
            HS_HIP(hipGetLastError());
            return HMMSORT_OK;
        };
        if (sptg <= 14) rc = go(gen_vit_blockG<14>);
        else rc = go(gen_vit_blockG<21>);
    } else
    if (spt == 1 && nbthr) rc = launch_block_sweep<1, true, false, true>(g, a, threads, lds, st);
    else if (spt == 2 && nbthr) rc = launch_block_sweep<2, true, false, true>(g, a, threads, lds, st);
    else if (spt == 1) rc = launch_block_sweep<1, false, false, true>(g, a, threads, lds, st);
    else if (spt == 2) rc = launch_block_sweep<2, false, false, true>(g, a, threads, lds, st);
    else if (spt == 4) rc = launch_block_sweep<4, false, false, true>(g, a, threads, lds, st);
    else if (!gcol && tl) rc = launch_block_sweep<0, false, false, true>(g, a, threads, lds, st);
    else if (!gcol) rc = launch_block_sweep<0, false, false, false>(g, a, threads, lds, st);
    else if (tl) rc = launch_block_sweep<0, false, true, true>(g, a, threads, lds, st);
    else rc = launch_block_sweep<0, false, true, false>(g, a, threads, lds, st);
    if (rc) return rc;
    const int nb = (int)g->nblk;
    if (nb > 1) {
        hipLaunchKernelGGL(k_block_check, dim3(nb - 1), dim3(256), 0, st, g->d_endv, g->d_warmv, (int)S,
                           g->d_bdiag, g->d_frame);
        HS_HIP(hipGetLastError());
    }
    if (!g->pair_ok) {
        hipLaunchKernelGGL(k_block_ties, dim3(1), dim3(64), 0, st, g->d_frame, g->d_gapmin, nb, g->d_bdiag);
        HS_HIP(hipGetLastError());
    }
    if (g->nms >= 64 && S > 1) {
        // wide rows: backtrace by segments (k_seg_walk)
        // (test aids: HMMSORT_SEG_WI shortens the walk-in so that guesses fail and k_seg_fix has work,
        // HMMSORT_SEG_PASSES limits the fix passes so that open boundaries reach diag[0], HMMSORT_SEG_LEN sets the segment length)
        const char *ewi = getenv("HMMSORT_SEG_WI"), *eps = getenv("HMMSORT_SEG_PASSES");
        const char *esg = getenv("HMMSORT_SEG_LEN");
        // long recordings are bound by the walkers' total steps, T (1 + WI / SEG): long segments; a chunk of the CLI's
        // decode (100 000 samples, fit.jl:11-42) by the steps of ONE walker, SEG + WI: short segments (57 -> 65 Msamples/s)
        const bool short_sig = T < 2000000;
        const int SEG = esg ? std::max(16, atoi(esg)) : (short_sig ? 64 : 256);
        const int WI = ewi ? std::max(1, atoi(ewi)) : (int)std::max<int64_t>(short_sig ? 192 : 256, (short_sig ? 3 : 4) * (g->K - 1) + 16);
        const int npass = eps ? std::max(0, std::min(64, atoi(eps))) : 8;
        const int nseg = (int)((T + SEG - 1) / SEG);
        if (!g->d_segbuf) {
            if ((rc = dalloc(&g->d_segbuf, (size_t)2 * nseg + 2, &g->bytes))) return rc;
        }
        int16_t *guess = g->d_segbuf, *below = g->d_segbuf + nseg + 1;
        hipLaunchKernelGGL(k_block_compose, dim3(1), dim3(256), 0, st, g->d_endv, g->d_fmap, g->d_fconst,
                           (int)S, nb, g->d_endstate, 0);
        HS_HIP(hipGetLastError());
        const bool btl = (size_t)S * 2 <= 64 * 1024;
        const size_t lds_seg = btl ? (size_t)S * 2 : 0;
        const dim3 gs((unsigned)((nseg + 255) / 256));
        if (btl) hipLaunchKernelGGL(k_seg_walk<true>, gs, dim3(256), lds_seg, st, g->d_T2, g->d_bt, g->nms, T, (int)S, SEG, WI,
                                    nseg, g->d_endstate + (nb - 1), d_x, guess, below);
        else hipLaunchKernelGGL(k_seg_walk<false>, gs, dim3(256), 0, st, g->d_T2, g->d_bt, g->nms, T, (int)S, SEG, WI,
                                nseg, g->d_endstate + (nb - 1), d_x, guess, below);
        HS_HIP(hipGetLastError());
        for (int pass = 0; pass <= npass; pass++) {
            if (btl) hipLaunchKernelGGL(k_seg_fix<true>, gs, dim3(256), lds_seg, st, g->d_T2, g->d_bt, g->nms, T, (int)S, SEG,
                                        nseg, d_x, guess, below, pass == npass, g->d_bdiag);
            else hipLaunchKernelGGL(k_seg_fix<false>, gs, dim3(256), 0, st, g->d_T2, g->d_bt, g->nms, T, (int)S, SEG, nseg,
                                    d_x, guess, below, pass == npass, g->d_bdiag);
            HS_HIP(hipGetLastError());
        }
    } else {
    // backtrace: rows of T2c staged W at a time; bt in LDS when it fits
        const int nms1 = std::max(g->nms, 1);
        int W = (int)std::max<int64_t>(1, std::min<int64_t>(64, (32 * 1024) / (nms1 * 2)));
        size_t lds_map = (size_t)((S + 3) & ~3) * 2 + (size_t)((W * g->nms + 3) & ~3) * 2;
        const int btm = lds_map + (size_t)S * 4 <= 150 * 1024 ? 1 : (lds_map + (size_t)S * 2 + 8 <= 150 * 1024 ? 2 : 0);
        if (btm == 1) lds_map += (size_t)S * 4;
        if (btm == 2) lds_map += (size_t)S * 2 + 8;
        const int map_threads = S > 8192 ? 1024 : 256;
        auto go_map = [&](auto kern) -> int {
            if (lds_map > 64 * 1024)
                HS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_map));
            hipLaunchKernelGGL(kern, dim3(nb), dim3(map_threads), lds_map, st, g->d_T2, g->d_bt, g->nms, W,
                               btm, T, (int)S, (int)g->B, g->d_fmap, g->d_fconst, g->d_merged, d_x);
            HS_HIP(hipGetLastError());
            return HMMSORT_OK;
        };
        if ((rc = btm == 1 ? go_map(k_block_map<1>) : (btm == 2 ? go_map(k_block_map<2>) : go_map(k_block_map<0>)))) return rc;
        hipLaunchKernelGGL(k_block_compose, dim3(1), dim3(256), 0, st, g->d_endv, g->d_fmap, g->d_fconst,
                           (int)S, nb, g->d_endstate, 1);
        HS_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_block_finish, dim3((nb + 63) / 64), dim3(64), 0, st, g->d_T2, g->d_bt, g->nms,
                           T, (int)S, (int)g->B, nb, g->d_endstate, g->d_merged, d_x);
        HS_HIP(hipGetLastError());
    }
    if (g->pair_ok && (rc = pair_ties_launch(g, d_x, st))) return rc;   // flagged decisions ON the decoded path -> diag[7]
    hipLaunchKernelGGL(k_block_ll, dim3(nb), dim3(256), 0, st, d_y, d_x, T, (int)S, (int)g->B,
                       g->d_mean, g->d_in_ptr, g->d_in_src, g->d_in_lp, a.c0, a.den, g->d_llpart);
    HS_HIP(hipGetLastError());
    const size_t lds_ll = (size_t)nb * 24 <= 60 * 1024 ? (size_t)nb * 24 : 0;
    hipLaunchKernelGGL(k_block_ll_sum, dim3(1), dim3(256), lds_ll, st, g->d_llpart, nb, d_ll);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

bool generic_pair_active(const GenericDev *g) { return g && g->blocked && g->pair_ok; }
int64_t generic_overlap_sweep(const GenericDev *g) { return (g && g->blocked && g->pair_ok) ? (g->multi_ok ? g->N : 2) : 0; }
void generic_pair_disable(GenericDev *g) { g->pair_off = true; g->pair_ok = false; g->multi_ok = false; }

int blocked_diagnostics(GenericDev *g, hipStream_t st, int64_t diag[8])
{
    unsigned long long h[8];
    HS_HIP(hipMemcpyAsync(h, g->d_bdiag, sizeof(h), hipMemcpyDeviceToHost, st));
    HS_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < 8; i++) diag[i] = (int64_t)h[i];
    return HMMSORT_OK;
}

}  // namespace hmmsort
