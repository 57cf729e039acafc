// Wave engine, part 4: exact resolution of near-tie decisions (reference src/viterbi.jl:74-84).
//
// The reference decides  t = T1[k,i-1] + lp;  if t > T1[j,i]  in list order on an UNNORMALISED trellis
// whose values reach |T1| ~ 2e6 after 1e7 samples (ulp 5e-10) and 2e7 after 1e8 (ulp 4e-9): two candidates
// closer than that are told apart (or rounded into a tie that list order breaks) by the rounding of an O(t)
// serial sum that no other summation order reproduces.  The time-parallel sweep (wave_viterbi.hip) therefore
// FLAGS every junction decision whose margin in its own frame is below the worst-case bound wave_thr, and
// this file settles the flagged decisions on the decoded path with the reference's own arithmetic:
//
//   * T1 along the decoded path P is the serial fold  v_0 = T1[x_0,0] (emission only, 0 for the silent
//     state: viterbi.jl:55-63),  v_u = (v_{u-1} + lp(x_{u-1},x_u)) + q_{x_u}(y_u)  -- exactly the operations of
//     viterbi.jl:79,86 in their order.  It is computed EXACTLY but not serially: while v stays inside one
//     binade [2^e, 2^(e+1)) its grid is U = 2^(e-52) and fl(v + a) = v + RN_U(a) unless a is exactly half-way
//     between two grid points, in which case round-to-even looks at the parity of v/U.  A block of samples
//     is therefore a map  v -> v + c[parity(v/U)]  (kw_tie_btransfer runs every block of 512 samples from an
//     even and an odd grid point next to an approximate prefix value, one lane per block, and records
//     whether all intermediate values stayed well inside the binade); the resolver chains the blocks from
//     the exact start value and runs serially only the few blocks that cross a power of two (or whose path
//     it has changed).
//   * A flagged decision for junction state j at time t (kw_tie_resolve): every candidate k (silent, ring
//     exits, in list order) is walked back through the stored back-pointers until it meets P, T1[k,t-1] is
//     replayed op for op from the exact value at the earliest meeting point, `T1[k,t-1] + lp` is compared
//     with strict '>' in list order -- the reference's decision.  Flagged junctions met on a candidate's own
//     way back are settled first (explicit stack, times strictly decrease).  Decisions are processed in time
//     order, so the prefix of P a decision relies on is already the reference's.  When the winner changes,
//     the segment of P behind the decision is replaced by the winner's path.
//   * The final arg-max (viterbi.jl:90) is settled the same way among the end states within wave_thr of the
//     best.
//
// diag[7] = decisions this procedure could not settle (candidate walk longer than kTieWalk, more than
// kTieCap flagged decisions on one channel's path, recursion deeper than the stack): 0 on every signal
// seen; hmmsort_viterbi then falls back to the op-for-op sweep.
#include <algorithm>
#include <cmath>

#include "wave_common.h"

namespace hmmsort {

constexpr int kTieStack = 48;
constexpr double kTieMargin = 0.5;   // distance every intermediate value keeps from the binade's ends

struct TieChan {
    int N, L, EB, epw;
    int64_t T, planePsi;
    const double *y, *mean, *ctab;
    double A, den;
    int16_t *x;
    uint32_t *psi;   // word 0 of this channel
};

__device__ __forceinline__ TieChan tie_chan(const WaveGeom &g, const WaveConst *cst, const double *y,
                                            const double *mean, const double *ctab, int16_t *x, uint32_t *psi,
                                            int ch)
{
    TieChan c;
    c.N = g.N; c.L = g.L; c.EB = g.EB; c.epw = g.epw;
    c.T = g.T; c.planePsi = (int64_t)g.C * g.T;
    c.y = y + (int64_t)ch * g.T;
    c.mean = mean + (int64_t)ch * (1 + g.N * g.L);
    c.ctab = ctab + (int64_t)ch * (1 + 2 * g.N + g.N * g.N + g.N * g.L);
    c.A = cst[ch].A; c.den = cst[ch].den;
    c.x = x + (int64_t)ch * g.T;
    c.psi = psi + (int64_t)ch * g.T;
    return c;
}

// junction entry of a state: 0 silent, a + 1 for (a,1), -1 for a state with a single predecessor
__device__ __forceinline__ int tie_entry_of(const TieChan &c, int s)
{
    if (s == 1) return 0;
    return ((s - 2) % c.L == 0) ? (s - 2) / c.L + 1 : -1;
}
__device__ __forceinline__ int tie_entry_state(const TieChan &c, int e) { return e == 0 ? 1 : 2 + (e - 1) * c.L; }
// state at t-1 of back-pointer code p: 0 silent, b = last state of ring b-1
__device__ __forceinline__ int tie_pred_state(const TieChan &c, int p) { return p == 0 ? 1 : 1 + p * c.L; }

__device__ __forceinline__ uint32_t tie_psi_get(const TieChan &c, int e, int64_t t)
{
    const uint32_t w = c.psi[(int64_t)(e / c.epw) * c.planePsi + t];
    return (w >> ((e % c.epw) * c.EB)) & ((1u << c.EB) - 1u);
}
__device__ __forceinline__ void tie_psi_set(const TieChan &c, int e, int64_t t, uint32_t ent)
{
    uint32_t *p = c.psi + (int64_t)(e / c.epw) * c.planePsi + t;
    const int sh = (e % c.epw) * c.EB;
    *p = (*p & ~(((1u << c.EB) - 1u) << sh)) | (ent << sh);
}
// funcl(y_t, mean_s, sigma, log sigma) (utils.jl:4), the reference's operations
__device__ __forceinline__ double tie_q(const TieChan &c, int s, int64_t t)
{
    const double dd = c.y[t] - c.mean[s - 1];
    return c.A - (dd * dd) / c.den;
}
// first trellis column (viterbi.jl:55-63)
__device__ __forceinline__ double tie_base(const TieChan &c, int s) { return s == 1 ? 0.0 : tie_q(c, s, 0); }

// ---- flagged decisions on the decoded path, in time order ---------------------------------------------
// Ordered compaction in three small kernels: flagged decisions per tile of kTieTile samples (one wavefront
// per tile), exclusive scan over the tiles, then every tile writes its entries t * 32 + e at its offset.
constexpr int kTieTile = 4096;

__device__ __forceinline__ bool tie_flagged_at(const WaveGeom &g, const int16_t *__restrict__ xc,
                                               const uint32_t *__restrict__ pc, int64_t planePsi, int64_t t, int *e_out)
{
    if (t < 2 || t >= g.T) return false;   // psi(1) only decides x[0], which kw_first_state re-decides exactly
    const int s = xc[t];
    int e = -1;
    if (s == 1) e = 0;
    else if ((s - 2) % g.L == 0) e = (s - 2) / g.L + 1;
    if (e < 0) return false;
    const uint32_t w = pc[(int64_t)(e / g.epw) * planePsi + t];
    *e_out = e;
    return ((w >> ((e % g.epw) * g.EB + g.EB - 1)) & 1u) != 0;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void kw_tie_collect(WaveGeom g, const int16_t *__restrict__ x,
                                                      const uint32_t *__restrict__ psi, int64_t *__restrict__ tie_cnt,
                                                      int64_t *__restrict__ tie_off, int64_t *__restrict__ tie_list,
                                                      int64_t ntile)
{
    const int ch = blockIdx.y;
    if (tie_cnt[ch * 8 + kTieTrig] == 0) return;
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntile) return;
    const int64_t T = g.T, planePsi = (int64_t)g.C * T;
    const int16_t *xc = x + (int64_t)ch * T;
    const uint32_t *pc = psi + (int64_t)ch * T;
    int64_t *off = tie_off + (int64_t)ch * (ntile + 1);
    int64_t run = WRITE ? off[tile] : 0;
    for (int i = 0; i < kTieTile / 64; i++) {
        const int64_t t = tile * kTieTile + 64 * i + lane;
        int e = 0;
        const bool fl = tie_flagged_at(g, xc, pc, planePsi, t, &e);
        const unsigned long long m = __ballot(fl);
        if (WRITE && fl) {
            const int64_t slot = run + __popcll(m & ((1ull << lane) - 1ull));
            if (slot < kTieCap) tie_list[(int64_t)ch * kTieCap + slot] = t * 32 + e;
        }
        run += __popcll(m);
    }
    if (!WRITE && lane == 0) off[tile] = run;
}

__global__ __launch_bounds__(1024) void kw_tie_offsets(const int64_t *__restrict__ tie_cnt_in, int64_t *__restrict__ tie_cnt,
                                                       int64_t *__restrict__ tie_off, int64_t ntile)
{
    __shared__ int64_t part[1024];
    const int ch = blockIdx.x, tid = threadIdx.x;
    if (tie_cnt_in[ch * 8 + kTieTrig] == 0) return;
    int64_t *off = tie_off + (int64_t)ch * (ntile + 1);
    const int64_t per = (ntile + 1023) / 1024, lo = (int64_t)tid * per, hi = lo + per < ntile ? lo + per : ntile;
    int64_t acc = 0;
    for (int64_t b = lo; b < hi; b++) acc += off[b];
    part[tid] = acc;
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0;
        for (int i = 0; i < 1024; i++) { const int64_t v = part[i]; part[i] = run; run += v; }
        tie_cnt[ch * 8 + kTieListed] = run;
    }
    __syncthreads();
    int64_t run = part[tid];
    for (int64_t b = lo; b < hi; b++) { const int64_t v = off[b]; off[b] = run; run += v; }
}

// ---- approximate prefix: block sums, then their exclusive scan ---------------------------------------
__global__ __launch_bounds__(256) void kw_tie_bsum(WaveGeom g, const WaveConst *__restrict__ cst,
                                                   const double *__restrict__ y, const int16_t *__restrict__ x,
                                                   const double *__restrict__ mean, const double *__restrict__ ctab_all,
                                                   const int64_t *__restrict__ tie_cnt, double *__restrict__ guess,
                                                   int64_t nblk)
{
    const int ch = blockIdx.y;
    if (tie_cnt[ch * 8 + kTieTrig] == 0) return;
    const int N = g.N, L = g.L, S = 1 + N * L;
    const int64_t T = g.T;
    const double *yc = y + (int64_t)ch * T, *mc = mean + (int64_t)ch * S;
    const int16_t *xc = x + (int64_t)ch * T;
    const double *ctab = ctab_all + (int64_t)ch * (1 + 2 * N + N * N + N * L);
    const double A = cst[ch].A, den = cst[ch].den;
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nblk) return;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < kTieBlk / 64; i++) {
        const int64_t u = b * kTieBlk + 1 + lane + 64 * i;
        if (u < T) {
            const int xp = xc[u - 1], xn = xc[u];
            const double d = yc[u] - mc[xn - 1];
            acc += wpath_lp(N, L, ctab, xp, xn) + (A - (d * d) / den);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) guess[(int64_t)ch * nblk + b] = acc;
}

__global__ __launch_bounds__(1024) void kw_tie_bscan(WaveGeom g, const WaveConst *__restrict__ cst,
                                                     const double *__restrict__ y, const int16_t *__restrict__ x,
                                                     const double *__restrict__ mean,
                                                     const int64_t *__restrict__ tie_cnt, double *__restrict__ guess,
                                                     int64_t nblk)
{
    __shared__ double part[1024];
    const int ch = blockIdx.x, tid = threadIdx.x;
    if (tie_cnt[ch * 8 + kTieTrig] == 0) return;
    double *gc = guess + (int64_t)ch * nblk;
    const int64_t per = (nblk + 1023) / 1024, lo = (int64_t)tid * per, hi = lo + per < nblk ? lo + per : nblk;
    double acc = 0.0;
    for (int64_t b = lo; b < hi; b++) acc += gc[b];
    part[tid] = acc;
    __syncthreads();
    if (tid == 0) {
        const int x0 = x[(int64_t)ch * g.T];
        double run = 0.0;
        if (x0 != 1) {
            const double d = y[(int64_t)ch * g.T] - mean[(int64_t)ch * (1 + g.N * g.L) + x0 - 1];
            run = cst[ch].A - (d * d) / cst[ch].den;
        }
        for (int i = 0; i < 1024; i++) { const double v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    double run = part[tid];
    for (int64_t b = lo; b < hi; b++) { const double v = gc[b]; gc[b] = run; run += v; }
}

// ---- exact block increments ------------------------------------------------------------------------
// One lane per block: the serial fold of the block's samples from the grid points next to the approximate
// start value with an even and an odd last mantissa bit.
__global__ __launch_bounds__(64) void kw_tie_btransfer(WaveGeom g, const WaveConst *__restrict__ cst,
                                                       const double *__restrict__ y, const int16_t *__restrict__ x,
                                                       const double *__restrict__ mean,
                                                       const double *__restrict__ ctab_all,
                                                       const int64_t *__restrict__ tie_cnt,
                                                       const double *__restrict__ guess, double *__restrict__ tc,
                                                       int32_t *__restrict__ tok, int64_t nblk)
{
    const int ch = blockIdx.y;
    if (tie_cnt[ch * 8 + kTieTrig] == 0) return;
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= nblk) return;
    const int N = g.N, L = g.L, S = 1 + N * L;
    const int64_t T = g.T;
    const double *yc = y + (int64_t)ch * T, *mc = mean + (int64_t)ch * S;
    const int16_t *xc = x + (int64_t)ch * T;
    const double *ctab = ctab_all + (int64_t)ch * (1 + 2 * N + N * N + N * L);
    const double A = cst[ch].A, den = cst[ch].den;
    const double gv = guess[(int64_t)ch * nblk + b];
    const long long gb = __double_as_longlong(gv);
    const double s0 = __longlong_as_double(gb & ~1ll), s1 = __longlong_as_double(gb | 1ll);
    double v0 = s0, v1 = s1, lo = fabs(s0), hi = fabs(s1);
    const int64_t u0 = b * kTieBlk + 1, u1 = (u0 + kTieBlk) < T ? (u0 + kTieBlk) : T;
    int xp = xc[u0 - 1];
#pragma unroll 4
    for (int64_t u = u0; u < u1; u++) {
        const int xn = xc[u];
        const double d = yc[u] - mc[xn - 1];
        const double a = wpath_lp(N, L, ctab, xp, xn), q = A - (d * d) / den;
        v0 = v0 + a; v1 = v1 + a;
        lo = fmin(lo, fabs(v0)); hi = fmax(hi, fabs(v0));
        v0 = v0 + q; v1 = v1 + q;
        lo = fmin(lo, fabs(v0)); hi = fmax(hi, fabs(v0));
        xp = xn;
    }
    bool ok = gv == gv && fabs(gv) < INFINITY && gv != 0.0;
    if (ok) {
        const int e = ilogb(fabs(gv));
        ok = (lo - kTieMargin >= ldexp(1.0, e)) && (hi + kTieMargin < ldexp(1.0, e + 1));
    }
    tc[((int64_t)ch * nblk + b) * 2] = v0 - s0;
    tc[((int64_t)ch * nblk + b) * 2 + 1] = v1 - s1;
    tok[(int64_t)ch * nblk + b] = ok ? 1 : 0;
}

// ---- the resolver: one wavefront per channel ---------------------------------------------------------
struct TieRun {
    const double *guess;   // nblk
    const double *tc;      // nblk x 2
    int32_t *tok;          // nblk
    double *tv;            // nblk + 1
    int64_t nblk, valid;   // tv[0..valid] are exact
    double vcur;           // == tv[valid], kept in a register (same value in every lane)
    int16_t *walk;         // kTieLanes x kTieWalk
    int64_t *cnt;          // this channel's 8 counters
    double *inc;           // LDS: 128 doubles (a | q of 64 samples)
};

// index into ctab of the transition xp -> xc (wpath_lp without the branches around the load)
__device__ __forceinline__ int tie_lp_index(int N, int L, int xp, int xc)
{
    const int a = (xp - 2) / L, k = (xp - 2) - a * L + 1, b = (xc - 2) / L;
    const int from_silent = xc == 1 ? 0 : 1 + b;
    const int from_ring = k < L ? 1 + 2 * N + N * N + a * L + k : (xc == 1 ? 1 + N + a : 1 + 2 * N + a * N + b);
    return xp == 1 ? from_silent : from_ring;
}

// serial fold over samples (from, to] from v (the value at `from`), all lanes in step: the lanes fetch the
// increments of 64 samples in parallel (the next 64 while the current ones are added), then every lane adds
// them in order (same value in every lane)
__device__ double tie_fold(const TieChan &c, const TieRun &R, double v, int64_t from, int64_t to, int lane)
{
    if (to <= from) return v;
    auto fetch = [&](int64_t u0, double &a, double &q) {
        const int64_t u = u0 + lane;
        const int64_t uc = u <= to ? u : to;
        const int xp = c.x[uc - 1], xn = c.x[uc];
        const double av = c.ctab[tie_lp_index(c.N, c.L, xp, xn)];
        const double dd = c.y[uc] - c.mean[xn - 1];
        const double qv = c.A - (dd * dd) / c.den;
        a = u <= to ? av : 0.0;
        q = u <= to ? qv : 0.0;
    };
    double a, q;
    fetch(from + 1, a, q);
    for (int64_t u0 = from + 1; u0 <= to; u0 += 64) {
        R.inc[lane] = a; R.inc[64 + lane] = q;
        __syncthreads();
        if (u0 + 64 <= to) fetch(u0 + 64, a, q);
        // samples past `to` were fetched as zeros: v + 0.0 + 0.0 is v
#pragma unroll 16
        for (int i = 0; i < 64; i++) v = (v + R.inc[i]) + R.inc[64 + i];
        __syncthreads();
    }
    return v;
}

// Advance the exact prefix to block start bt.  64 blocks at a time: a block is the map
// v -> v + c[parity(v/U)] with parity(v'/U) = parity(v/U) xor parity(c/U), and such maps compose, so the start
// values of 64 consecutive blocks come from one lane scan over (increment, parity) pairs for both start
// parities.  Every lane then checks that ITS start value lies in the binade its increments were computed in
// and next to the approximate value they were computed from; the blocks before the first lane that fails are
// accepted, that block is folded serially.  Increments are multiples of U below 2^53 U (|v| >= 2^16 here), so
// every sum is exact.
__device__ void tie_chain(const TieChan &c, TieRun &R, int64_t bt, int lane)
{
    double v = R.vcur;
    int64_t b = R.valid;
    // the records of the NEXT 64 blocks are fetched while the current 64 are scanned
    double ngv = 0.0, nc0 = 0.0, nc1 = 0.0;
    int nok = 0;
    int64_t nbase = -1;
    auto fetch = [&](int64_t base) {
        const int64_t bi = (base + lane) < R.nblk ? (base + lane) : (R.nblk - 1);
        ngv = R.guess[bi]; nc0 = R.tc[2 * bi]; nc1 = R.tc[2 * bi + 1]; nok = R.tok[bi];
        nbase = base;
    };
    while (b < bt) {
        const int nb = (bt - b) < 64 ? (int)(bt - b) : 64;
        const bool in = lane < nb;
        if (nbase != b) fetch(b);
        const double gv = ngv, c0 = nc0, c1 = nc1;
        const int okb = nok;
        if (b + 64 < bt) fetch(b + 64);
        const bool vfin = v == v && fabs(v) < INFINITY && v != 0.0;
        const int ev = vfin ? ilogb(fabs(v)) : 0;
        const bool lane_ok = !in || (okb != 0 && gv == gv && gv != 0.0 && fabs(gv) < INFINITY &&
                                     ilogb(fabs(gv)) == ev && ((gv < 0) == (v < 0)));
        const unsigned long long bad = __ballot(!lane_ok);
        const int nfast = bad ? __ffsll((long long)bad) - 1 : nb;
        int ngood = 0;
        double inc_incl = 0.0;
        if (vfin && ev >= 16 && nfast > 0) {
            const double sU = ldexp(1.0, 52 - ev);   // 1 / U
            double C0 = lane < nfast ? c0 : 0.0, C1 = lane < nfast ? c1 : 0.0;
            int P0 = (int)((long long)(C0 * sU) & 1ll), P1 = 1 ^ (int)((long long)(C1 * sU) & 1ll);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const double aC0 = __shfl_up(C0, d), aC1 = __shfl_up(C1, d);
                const int aP0 = __shfl_up(P0, d), aP1 = __shfl_up(P1, d);
                if (lane >= d) {
                    const double n0 = aC0 + (aP0 ? C1 : C0), n1 = aC1 + (aP1 ? C1 : C0);
                    const int q0 = aP0 ? P1 : P0, q1 = aP1 ? P1 : P0;
                    C0 = n0; C1 = n1; P0 = q0; P1 = q1;
                }
            }
            inc_incl = (__double_as_longlong(v) & 1ll) ? C1 : C0;
            double inc_excl = __shfl_up(inc_incl, 1);
            if (lane == 0) inc_excl = 0.0;
            const double vi = v + inc_excl;          // exact value at the start of block b + lane
            const bool good = lane >= nfast || (ilogb(fabs(vi)) == ev && ((vi < 0) == (v < 0)) &&
                                                fabs(vi - gv) < 0.5 * kTieMargin);
            const unsigned long long nb2 = __ballot(!good);
            ngood = nb2 ? __ffsll((long long)nb2) - 1 : nfast;
        }
        if (ngood > 0) {
            const double ve = v + inc_incl;          // value at the END of block b + lane
            if (lane < ngood) R.tv[b + lane + 1] = ve;
            v = wave_bcast(ve, ngood - 1);
            b += ngood;
            continue;
        }
        {   // one block the slow way: per-block increment when its start value fits, else the serial fold
            const double gv0 = wave_bcast(gv, 0), t0 = wave_bcast(c0, 0), t1 = wave_bcast(c1, 0);
            const int ok0 = __shfl(okb, 0);
            bool fast = ok0 != 0 && vfin && gv0 != 0.0 && gv0 == gv0 && fabs(gv0) < INFINITY;
            if (fast) fast = ilogb(fabs(gv0)) == ev && ((v < 0) == (gv0 < 0)) && fabs(v - gv0) < 0.5 * kTieMargin;
            if (fast) {
                v = v + ((__double_as_longlong(v) & 1ll) ? t1 : t0);
            } else {
                const int64_t to = ((b + 1) * kTieBlk) < (c.T - 1) ? ((b + 1) * kTieBlk) : (c.T - 1);
                v = tie_fold(c, R, v, b * kTieBlk, to, lane);
                if (lane == 0) R.cnt[kTieSerial] += 1;
            }
            if (lane == 0) R.tv[b + 1] = v;
            b += 1;
        }
    }
    R.valid = bt;
    R.vcur = v;
}

// exact T1[x_u, u] of the decoded path
__device__ double tie_exact(const TieChan &c, TieRun &R, int64_t u, int lane)
{
    const int64_t bt = u / kTieBlk;
    if (bt < R.valid) {   // an earlier block start: its value was stored when the chain passed it
        __threadfence();
        return tie_fold(c, R, R.tv[bt], bt * kTieBlk, u, lane);
    }
    if (bt > R.valid) tie_chain(c, R, bt, lane);
    return tie_fold(c, R, R.vcur, bt * kTieBlk, u, lane);
}

struct TieWalk {
    int64_t u;     // time at which the walk met the decoded path (-1: it reached sample 0 off the path)
    int n;         // states stored, newest first: state at t-1-i in slot i
    int status;    // 0 met the path, 1 stopped at an unsettled flagged junction (fe, ft), 2 too long
    int fe;
    int64_t ft;
};

// eight samples of the decoded path per round trip (the loads do not depend on the walk's state)
__device__ TieWalk tie_walk(const TieChan &c, int16_t *scr, int s, int64_t tau)
{
    TieWalk w;
    w.n = 0; w.status = 0; w.fe = 0; w.ft = 0; w.u = 0;
    for (;;) {
        int xb[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { const int64_t tj = tau - j; xb[j] = c.x[tj > 0 ? tj : 0]; }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (xb[j] == s) { w.u = tau; return w; }
            if (w.n >= kTieWalk) { w.status = 2; return w; }
            scr[w.n++] = (int16_t)s;
            if (tau == 0) { w.u = -1; return w; }
            const int e = tie_entry_of(c, s);
            if (e < 0) {
                s = s - 1;
            } else {
                const uint32_t ent = tie_psi_get(c, e, tau);
                if (ent >> (c.EB - 1)) { w.status = 1; w.fe = e; w.ft = tau; return w; }
                s = tie_pred_state(c, (int)ent);
            }
            tau--;
        }
    }
}

// Candidates (one per lane; `valid` lanes hold state sC at time t-1 and the transition's log-probability
// lpC, or no transition when !addlp) -> the reference's winner: exact T1 of every candidate, `+ lp`,
// strict '>' in lane order.  Returns 0 and the winning lane (-1: no candidate is finite), 1 when a flagged
// junction on a candidate's way back has to be settled first (pe, pt), 2 when a walk is too long.
__device__ int tie_eval(const TieChan &c, TieRun &R, int lane, bool valid, int sC, double lpC, bool addlp,
                        int64_t t, int nl, int *winner, TieWalk *mine, int64_t *umin_out, double *vmin_out,
                        int *pe, int64_t *pt)
{
    int16_t *scr = R.walk + (int64_t)(lane < kTieLanes ? lane : 0) * kTieWalk;
    TieWalk w;
    w.u = t - 1; w.n = 0; w.status = 0; w.fe = 0; w.ft = 0;
    if (valid) w = tie_walk(c, scr, sC, t - 1);
    *mine = w;
    const unsigned long long m1 = __ballot(valid && w.status == 1);
    if (m1) {
        const int src = __ffsll((long long)m1) - 1;
        *pe = __shfl(w.fe, src);
        *pt = (int64_t)__shfl((long long)w.ft, src);
        return 1;
    }
    if (__any(valid && w.status == 2)) return 2;
    {
        int longest = valid ? w.n : 0;
        for (int o = 32; o > 0; o >>= 1) longest = max(longest, __shfl_xor(longest, o));
        if (lane == 0 && longest > R.cnt[kTieLongest]) R.cnt[kTieLongest] = longest;
    }
    long long um = valid ? (long long)w.u : (long long)t;
    for (int o = 32; o > 0; o >>= 1) um = min(um, (long long)__shfl_xor(um, o));
    const int64_t umin = (int64_t)um;
    __threadfence();   // the walks wrote their states to global scratch
    double V = 0.0;
    if (umin >= 0) V = tie_exact(c, R, umin, lane);
    *umin_out = umin; *vmin_out = V;
    double tt = -INFINITY;
    if (valid) {
        double v;
        int sp;
        int64_t tau;
        if (umin >= 0) { v = V; sp = c.x[umin]; tau = umin + 1; }
        else { sp = (w.u < 0) ? (int)scr[w.n - 1] : (int)c.x[0]; v = tie_base(c, sp); tau = 1; }
        // eight steps per round trip: states, then their means / samples / log-probabilities, then the adds
        for (; tau <= t - 1; tau += 8) {
            int sc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int64_t tj = (tau + j) <= (t - 1) ? (tau + j) : (t - 1);
                const int64_t si = t - 1 - tj;
                const int xs = c.x[tj], ws = scr[si < w.n ? si : 0];
                sc[j] = (tj <= w.u) ? xs : ws;
            }
            double a[8], q[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int64_t tj = (tau + j) <= (t - 1) ? (tau + j) : (t - 1);
                const int spj = j == 0 ? sp : sc[j - 1];
                const double av = c.ctab[tie_lp_index(c.N, c.L, spj, sc[j])];
                const double dd = c.y[tj] - c.mean[sc[j] - 1];
                const double qv = c.A - (dd * dd) / c.den;
                const bool live = (tau + j) <= (t - 1);
                a[j] = live ? av : 0.0;
                q[j] = live ? qv : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) v = (v + a[j]) + q[j];
            sp = sc[7];
        }
        tt = addlp ? v + lpC : v;
    }
    double best = -INFINITY;
    int arg = -1;
    for (int i = 0; i < nl; i++) {
        const double ti = wave_bcast(tt, i);
        if (ti > best) { best = ti; arg = i; }
    }
    *winner = arg;
    return 0;
}

// replace the decoded path on (u, t-1] by lane `src`'s candidate path (u = where it met the old path)
__device__ void tie_flip(const TieChan &c, TieRun &R, int lane, int src, const TieWalk &mine, int64_t t)
{
    const int n = __shfl(mine.n, src);
    const int64_t u = (int64_t)__shfl((long long)mine.u, src);
    const int16_t *scr = R.walk + (int64_t)src * kTieWalk;
    for (int i = lane; i < n; i += 64) c.x[t - 1 - i] = scr[i];
    // blocks whose samples changed lose their increments; exact prefix values behind u stay
    const int64_t b0 = (u < 0 ? 0 : u) / kTieBlk, b1 = (t - 1) / kTieBlk;
    for (int64_t b = b0 + lane; b <= b1 && b < R.nblk; b += 64) R.tok[b] = 0;
    __threadfence();
    if (u < 0) {          // sample 0 changed: the fold starts from the new first state
        R.valid = 0;
        R.vcur = tie_base(c, c.x[0]);
        if (lane == 0) R.tv[0] = R.vcur;
    } else if (R.valid > b0) {
        R.valid = b0;
        R.vcur = R.tv[b0];
    }
}

__global__ __launch_bounds__(64) void kw_tie_resolve(WaveGeom g, const WaveConst *cst, const double *y,
                                                     const double *mean, const double *ctab_all,
                                                     const double *ysum, int16_t *x, uint32_t *psi,
                                                     const double *vend, int64_t *tie_cnt, int64_t *tie_list,
                                                     int16_t *tie_walk, const double *guess, const double *tc,
                                                     int32_t *tok, double *tv, int64_t nblk, int64_t *diag)
{
    __shared__ double inc[128];
    __shared__ int64_t stk_t[kTieStack];
    __shared__ int stk_e[kTieStack];
    const int ch = blockIdx.x, lane = threadIdx.x;
    int64_t *cnt = tie_cnt + ch * 8;
    if (cnt[kTieTrig] == 0) return;
    if (g.tie_debug & 2) {   // test aid: the behaviour without this file -- every flagged decision stays open
        const int64_t open = cnt[kTieListed] + cnt[kTieTail];
        if (lane == 0) {
            cnt[kTieOpen] = open;
            if (open) atomicAdd((unsigned long long *)&diag[7], (unsigned long long)open);
        }
        return;
    }
    const TieChan c = tie_chan(g, cst, y, mean, ctab_all, x, psi, ch);
    TieRun R;
    R.guess = guess + (int64_t)ch * nblk; R.tc = tc + (int64_t)ch * nblk * 2; R.tok = tok + (int64_t)ch * nblk;
    R.tv = tv + (int64_t)ch * (nblk + 1); R.nblk = nblk; R.valid = 0;
    R.walk = tie_walk + (int64_t)ch * kTieLanes * kTieWalk; R.cnt = cnt; R.inc = inc;
    R.vcur = tie_base(c, c.x[0]);
    if (lane == 0) R.tv[0] = R.vcur;
    const int N = c.N;
    int64_t *list = tie_list + (int64_t)ch * kTieCap;
    const int64_t listed = cnt[kTieListed];
    const int n = (int)(listed < kTieCap ? listed : kTieCap);
    int64_t open = listed > kTieCap ? listed - kTieCap : 0;   // beyond the list: cannot be settled here

    // the list is in time order (ordered compaction, kw_tie_collect)
    int64_t done = 0, flips = 0;
    for (int idx = 0; idx < n; idx++) {
        const int64_t t0 = list[idx] >> 5;
        const int e0 = (int)(list[idx] & 31);
        if (c.x[t0] != tie_entry_state(c, e0)) continue;              // no longer on the path
        if (!(tie_psi_get(c, e0, t0) >> (c.EB - 1))) continue;       // settled on the way to an earlier one
        int depth = 0;
        if (lane == 0) { stk_t[0] = t0; stk_e[0] = e0; }
        depth = 1;
        __syncthreads();
        int guard = 0;
        bool failed = false;
        while (depth > 0) {
            if (++guard > 4096) { failed = true; break; }
            const int64_t t = stk_t[depth - 1];
            const int e = stk_e[depth - 1];
            const int sj = tie_entry_state(c, e);
            // candidate `lane`: back-pointer code p = lane (0 silent, b = exit of ring b-1), list order
            const bool valid0 = lane <= N && !(e > 0 && lane == e);
            const int sC = tie_pred_state(c, lane <= N ? lane : 0);
            double lpC = -INFINITY;
            if (valid0) lpC = wpath_lp(c.N, c.L, c.ctab, sC, sj);
            const bool valid = valid0 && lpC > -INFINITY;
            int winner = -1, pe = 0;
            int64_t umin = 0, pt = 0;
            double vmin = 0.0;
            TieWalk mine;
            const int rc = tie_eval(c, R, lane, valid, sC, lpC, true, t, N + 1, &winner, &mine, &umin, &vmin, &pe, &pt);
            if (rc == 2) { failed = true; break; }
            if (rc == 1) {
                if (depth >= kTieStack) { failed = true; break; }
                __syncthreads();
                if (lane == 0) { stk_t[depth] = pt; stk_e[depth] = pe; }
                depth++;
                __syncthreads();
                continue;
            }
            // T2 starts as ones(Int16) (viterbi.jl:53): no finite candidate leaves the silent state
            const int pnew = winner < 0 ? 0 : winner;
            const int pold = (int)(tie_psi_get(c, e, t) & ((1u << (c.EB - 1)) - 1u));
            if (lane == 0) tie_psi_set(c, e, t, (uint32_t)pnew);
            __threadfence();
            done++;
            if (pnew != pold) {
                flips++;
                if (winner >= 0 && c.x[t] == sj) tie_flip(c, R, lane, pnew, mine, t);   // on the decoded path: P changes behind t
            }
            depth--;
            __syncthreads();
        }
        if (failed) { open++; break; }   // the prefix is no longer certain: later decisions stay open too
    }

    // final arg-max (viterbi.jl:90): end states within the threshold of the best, in state order
    if (cnt[kTieTail] != 0 && open == 0) {
        const int S = 1 + c.N * c.L;
        const double *rec = vend + ((int64_t)ch * g.nch + g.nch - 1) * (int64_t)S;
        const double thr = wave_thr(g, cst[ch], ysum, ch);
        double best = -INFINITY;
        for (int j = lane; j < S; j += 64) best = fmax(best, rec[j]);
        best = wave_max(best);
        // ordered compaction of the candidates into lanes 0..nc-1
        int nc = 0, myS = 1;
        bool over = false;
        for (int j0 = 0; j0 < S; j0 += 64) {
            const int j = j0 + lane;
            const bool in = j < S && (best - rec[j]) < thr;
            const unsigned long long m = __ballot(in);
            const int pos = nc + __popcll(m & ((1ull << lane) - 1ull));
            // lane `pos` takes state j + 1
            for (int k = 0; k < 64; k++) {
                const int pk = __shfl(in ? pos : -1, k);
                if (pk == lane) myS = j0 + k + 1;
            }
            nc += __popcll(m);
            if (nc > kTieLanes) { over = true; break; }
        }
        if (over || nc == 0) {
            open++;
        } else {
            int guard = 0;
            bool failed = false;
            for (;;) {
                if (++guard > 4096) { failed = true; break; }
                int winner = -1, pe = 0;
                int64_t umin = 0, pt = 0;
                double vmin = 0.0;
                TieWalk mine;
                const int rc = tie_eval(c, R, lane, lane < nc, myS, 0.0, false, c.T, nc, &winner, &mine, &umin, &vmin, &pe, &pt);
                if (rc == 2) { failed = true; break; }
                if (rc == 1) {
                    // settle the flagged junction on a candidate's way back first (with its own sub-decisions)
                    int depth = 1;
                    if (lane == 0) { stk_t[0] = pt; stk_e[0] = pe; }
                    __syncthreads();
                    while (depth > 0 && !failed) {
                        if (++guard > 4096) { failed = true; break; }
                        const int64_t t = stk_t[depth - 1];
                        const int e = stk_e[depth - 1];
                        const int sj = tie_entry_state(c, e);
                        const bool valid0 = lane <= N && !(e > 0 && lane == e);
                        const int sC = tie_pred_state(c, lane <= N ? lane : 0);
                        double lpC = -INFINITY;
                        if (valid0) lpC = wpath_lp(c.N, c.L, c.ctab, sC, sj);
                        const bool valid = valid0 && lpC > -INFINITY;
                        int w2 = -1, pe2 = 0;
                        int64_t um2 = 0, pt2 = 0;
                        double vm2 = 0.0;
                        TieWalk m2;
                        const int rc2 = tie_eval(c, R, lane, valid, sC, lpC, true, t, N + 1, &w2, &m2, &um2, &vm2, &pe2, &pt2);
                        if (rc2 == 2) { failed = true; break; }
                        if (rc2 == 1) {
                            if (depth >= kTieStack) { failed = true; break; }
                            __syncthreads();
                            if (lane == 0) { stk_t[depth] = pt2; stk_e[depth] = pe2; }
                            depth++;
                            __syncthreads();
                            continue;
                        }
                        const int pnew = w2 < 0 ? 0 : w2;
                        const int pold = (int)(tie_psi_get(c, e, t) & ((1u << (c.EB - 1)) - 1u));
                        if (lane == 0) tie_psi_set(c, e, t, (uint32_t)pnew);
                        __threadfence();
                        done++;
                        if (pnew != pold) flips++;   // off the decoded path by construction
                        depth--;
                        __syncthreads();
                    }
                    if (failed) break;
                    continue;
                }
                if (winner >= 0) {
                    const int sw = __shfl(myS, winner);
                    done++;
                    if (sw != (int)c.x[c.T - 1]) {
                        flips++;
                        tie_flip(c, R, lane, winner, mine, c.T);
                    }
                }
                break;
            }
            if (failed) open++;
        }
    }
    if (g.tie_debug & 1) (void)tie_exact(c, R, c.T - 1, lane);   // test aid: every block start of the exact prefix
    if (lane == 0) {
        cnt[kTieDone] = done;
        cnt[kTieFlips] = flips;
        cnt[kTieOpen] = open;
        if (open) atomicAdd((unsigned long long *)&diag[7], (unsigned long long)open);
    }
}

int wave_tie_resolve(WaveDev *r, const double *d_y, int16_t *d_x, hipStream_t st)
{
    const WaveGeom &g = r->g;
    const int64_t nblk = r->tie_nblk;
    { WPROF(r, "kw_tie_collect", st);
      const int64_t ntile = r->tie_ntile;
      const dim3 gt((unsigned)((ntile + 3) / 4), g.C);
      hipLaunchKernelGGL(kw_tie_collect<false>, gt, dim3(256), 0, st, g, d_x, r->psi, r->tie_cnt, r->tie_off, r->tie_list, ntile);
      hipLaunchKernelGGL(kw_tie_offsets, dim3(g.C), dim3(1024), 0, st, r->tie_cnt, r->tie_cnt, r->tie_off, ntile);
      hipLaunchKernelGGL(kw_tie_collect<true>, gt, dim3(256), 0, st, g, d_x, r->psi, r->tie_cnt, r->tie_off, r->tie_list, ntile); }
    { WPROF(r, "kw_tie_prefix", st);
      hipLaunchKernelGGL(kw_tie_bsum, dim3((unsigned)((nblk + 3) / 4), g.C), dim3(256), 0, st, g, r->d_cst, d_y, d_x,
                         r->d_mean, r->d_ctab, r->tie_cnt, r->tie_guess, nblk);
      hipLaunchKernelGGL(kw_tie_bscan, dim3(g.C), dim3(1024), 0, st, g, r->d_cst, d_y, d_x, r->d_mean, r->tie_cnt,
                         r->tie_guess, nblk);
      hipLaunchKernelGGL(kw_tie_btransfer, dim3((unsigned)((nblk + 63) / 64), g.C), dim3(64), 0, st, g, r->d_cst, d_y,
                         d_x, r->d_mean, r->d_ctab, r->tie_cnt, r->tie_guess, r->tie_c, r->tie_ok, nblk); }
    { WPROF(r, "kw_tie_resolve", st);
      hipLaunchKernelGGL(kw_tie_resolve, dim3(g.C), dim3(64), 0, st, g, r->d_cst, d_y, r->d_mean, r->d_ctab, r->ysum,
                         d_x, r->psi, r->vend, r->tie_cnt, r->tie_list, r->tie_walk, r->tie_guess, r->tie_c,
                         r->tie_ok, r->tie_v, nblk, r->diag); }
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

// sums over the channels of the last decode: [0] trigger, [1] flagged decisions on the path, [2] decisions
// re-decided exactly, [3] back-pointers that changed, [4] unresolved, [5] final arg-max flagged,
// [6] longest candidate walk (max), [7] prefix blocks run serially
int wave_tie_stats(WaveDev *r, hipStream_t st, int64_t out[8])
{
    HS_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> h((size_t)r->g.C * 8);
    HS_HIP(hipMemcpy(h.data(), r->tie_cnt, h.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out[i] = 0;
    for (int ch = 0; ch < r->g.C; ch++)
        for (int i = 0; i < 8; i++) {
            if (i == kTieLongest) out[i] = std::max(out[i], h[(size_t)ch * 8 + i]);
            else out[i] += h[(size_t)ch * 8 + i];
        }
    return HMMSORT_OK;
}

}  // namespace hmmsort
