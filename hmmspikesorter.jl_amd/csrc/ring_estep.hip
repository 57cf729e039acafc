// Ring engine, part 3: one Baum-Welch step = forward -> backward -> update
// (reference src/baumwelch.jl:25-51, :73-98, :205-309, :362-370) without ever materialising
// alpha, beta, gamma or xi.
//
// Log domain, same frame as the Viterbi part (the per-sample emission constant A is dropped).
// Per sample the serial recursions touch only the junction quantities
//   forward :  la0(t) = log alpha_t(silent),   lp_a(t') = log[alpha-mass entering ring a at t']
//                                                        + Rfull_a(t')      (stored in P)
//   backward:  lb0(t) = log beta_t(silent),    ly_a(t') = log beta of ring a's LAST state at
//                                                        t'+L-1 (stored in Q at the onset t')
// because a ring is a deterministic delay line: for every ring state
//   alpha_t(a,k) + beta_t(a,k) = lp_a(t') + ly_a(t'),  t' = t-k+1   (the posterior of "ring a
//   started at t'"), so gamma of all S states follows from N+1 numbers per sample and the M-step
//   sums become posterior-weighted spike-triggered sums over y:
//     sum_t gamma_t(a,k) * f(y_t) = sum_t' rho_a(t') * f(y[t'+k-1]).
// Each chain works in its own additive frame; Zc[c] = log sum_j alpha*beta evaluated once per
// chain ties forward and backward frames together (the reference's per-sample normaliser g,
// baumwelch.jl:217-223, is that same constant).
#include <cmath>
#include <type_traits>

#include "ring_common.h"

namespace hmmsort {

template <int N>
struct ChainIn {
    double y;
    double R[N];
    double X[N];
};

// ------------------------------------------------------------------------------------------
// forward chains (baumwelch.jl:25-51).  Same skeleton as k_vit_chain; max -> log-sum-exp.
// The N+1 exponentials exp(value - m) are shared by all N+1 junction sums.
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void k_fwd_chain(RingGeom g, EParams<N> ep,
                                                  const double *__restrict__ yT,
                                                  const double *__restrict__ Rf,
                                                  double *__restrict__ P, double *__restrict__ A0)
{
    constexpr int U = chain_unroll<N>();
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int B = g.B, H = g.H, L = g.L, ncol = g.ncol;
    const bool active = c < g.nch;
    const int64_t tc = (int64_t)c * B;
    const int nc = active ? (int)((g.T - tc) < B ? (g.T - tc) : B) : 0;
    const int s0 = (c == 0) ? 0 : -H;
    const int64_t planeR = (int64_t)B * ncol, planeP = (int64_t)(H + B) * ncol;
    const int cin = c > 0 ? c - 1 : 0;

    auto load = [&](ChainIn<N>(&d)[U], int sb) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = sb + u;
            const bool live = active && s >= s0 && s < nc;
            d[u].y = 0.0;
#pragma unroll
            for (int a = 0; a < N; a++) { d[u].R[a] = 0.0; d[u].X[a] = -INFINITY; }
            if (live) {
                const int64_t off = (s >= 0) ? (int64_t)s * ncol + c : (int64_t)(B + s) * ncol + cin;
                d[u].y = yT[off];
#pragma unroll
                for (int a = 0; a < N; a++) d[u].R[a] = Rf[a * planeR + off];
                if (s - L >= -H) {
                    const int64_t offp = (int64_t)(H + s - L) * ncol + c;
#pragma unroll
                    for (int a = 0; a < N; a++) d[u].X[a] = P[a * planeP + offp];
                }
            }
        }
    };

    ChainIn<N> cur[U], nxt[U];
    double la0 = 0.0;
    load(cur, -H);
    for (int sb = -H; sb < B; sb += U) {
        if (sb + U < B) load(nxt, sb + U);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = sb + u;
            const bool live = active && s >= s0 && s < nc;
            if (live) {
                double Pn[N];
                const double d = cur[u].y - ep.mean0;
                const double q0 = -(d * d) / ep.den;
                if (s == s0) {
                    if (c == 0) {  // baumwelch.jl:36: first column = emission only, every state
                        la0 = q0;
#pragma unroll
                        for (int a = 0; a < N; a++) Pn[a] = cur[u].R[a];
                    } else {       // warm-up start: silent, rings empty
                        la0 = 0.0;
#pragma unroll
                        for (int a = 0; a < N; a++) Pn[a] = -INFINITY;
                    }
                } else {
                    double m = la0;
#pragma unroll
                    for (int a = 0; a < N; a++) m = fmax(m, cur[u].X[a]);
                    const double e0 = exp(la0 - m);
                    double e[N];
#pragma unroll
                    for (int a = 0; a < N; a++) e[a] = exp(cur[u].X[a] - m);
                    double ssum = e0 * ep.p00;
#pragma unroll
                    for (int a = 0; a < N; a++) ssum += e[a] * ep.pend[a];
#pragma unroll
                    for (int a = 0; a < N; a++) {
                        double su = e0 * ep.p0[a];
#pragma unroll
                        for (int b = 0; b < N; b++)
                            if (b != a) su += e[b] * ep.px[b * N + a];
                        Pn[a] = (m + log(su)) + cur[u].R[a];
                    }
                    la0 = (m + log(ssum)) + q0;
                }
                const int64_t offp = (int64_t)(H + s) * ncol + c;
#pragma unroll
                for (int a = 0; a < N; a++) P[a * planeP + offp] = Pn[a];
                if (s >= -1) A0[(int64_t)(1 + s) * ncol + c] = la0;
            }
        }
        if (sb + U < B) {
#pragma unroll
            for (int u = 0; u < U; u++) cur[u] = nxt[u];
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward chains (baumwelch.jl:73-98).  Chain c runs from te = min(tc+nc+H, T)-1 down to tc.
// beta = 0 for every state at te: the reference's terminal condition when te is the last sample
// (:80), an arbitrary warm-up start otherwise.  ly_a(t') for rings that have not finished by te
// is 0 for the same reason.
// Q row (L + s') holds ly_a(tc + s'), s' in [-(L-1), B+H).
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void k_bwd_chain(RingGeom g, EParams<N> ep,
                                                  const double *__restrict__ yT,
                                                  const double *__restrict__ Rf,
                                                  double *__restrict__ Q, double *__restrict__ B0,
                                                  double *__restrict__ B0h)
{
    constexpr int U = chain_unroll<N>();
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int B = g.B, H = g.H, L = g.L, ncol = g.ncol;
    const bool active = c < g.nch;
    const int64_t tc = (int64_t)c * B;
    const int nc = active ? (int)((g.T - tc) < B ? (g.T - tc) : B) : 0;
    int64_t te = tc + nc + H;
    if (te > g.T) te = g.T;
    const int se = active ? (int)(te - 1 - tc) : -1;
    const int64_t planeR = (int64_t)B * ncol, planeQ = (int64_t)(L + B + H) * ncol;

    // inputs of step s (computing time t = tc+s from t+1): y, Rf and ly at time t+1
    auto load = [&](ChainIn<N>(&d)[U], int sb) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = sb - u;
            const bool live = active && s < se && s >= 0;
            d[u].y = 0.0;
#pragma unroll
            for (int a = 0; a < N; a++) { d[u].R[a] = 0.0; d[u].X[a] = 0.0; }
            if (live) {
                const int s1 = s + 1;
                const int64_t off = (s1 < B) ? (int64_t)s1 * ncol + c : (int64_t)(s1 - B) * ncol + c + 1;
                d[u].y = yT[off];
#pragma unroll
                for (int a = 0; a < N; a++) d[u].R[a] = Rf[a * planeR + off];
                if (s + L <= se) {  // the ring started at t+1 ends inside the processed range
                    const int64_t offq = (int64_t)(L + s1) * ncol + c;
#pragma unroll
                    for (int a = 0; a < N; a++) d[u].X[a] = Q[a * planeQ + offq];
                }
            }
        }
    };

    ChainIn<N> cur[U], nxt[U];
    double lb0 = 0.0;
    const int stop = B + H - 1;
    load(cur, stop);
    for (int sb = stop; sb >= 0; sb -= U) {
        if (sb - U >= 0) load(nxt, sb - U);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = sb - u;
            if (active && s <= se && s >= 0) {
                double Yn[N];
                if (s == se) {
                    lb0 = 0.0;
#pragma unroll
                    for (int a = 0; a < N; a++) Yn[a] = 0.0;
                    // onsets whose ring runs past te: ly = 0 (rows the sweep below never writes,
                    // read by the statistics kernels when te is the end of the data)
                    for (int i = 2; i <= L; i++) {
                        const int64_t o = (int64_t)(s + i) * ncol + c;
#pragma unroll
                        for (int a = 0; a < N; a++) Q[a * planeQ + o] = 0.0;
                    }
                } else {
                    const double d = cur[u].y - ep.mean0;
                    const double v0 = lb0 - (d * d) / ep.den;
                    double lw[N];
                    double m = v0;
#pragma unroll
                    for (int a = 0; a < N; a++) {
                        lw[a] = cur[u].R[a] + cur[u].X[a];
                        m = fmax(m, lw[a]);
                    }
                    const double E0 = exp(v0 - m);
                    double E[N];
#pragma unroll
                    for (int a = 0; a < N; a++) E[a] = exp(lw[a] - m);
                    double ssum = E0 * ep.p00;
#pragma unroll
                    for (int a = 0; a < N; a++) ssum += E[a] * ep.p0[a];
#pragma unroll
                    for (int a = 0; a < N; a++) {
                        double su = E0 * ep.pend[a];
#pragma unroll
                        for (int b = 0; b < N; b++)
                            if (b != a) su += E[b] * ep.px[a * N + b];
                        Yn[a] = m + log(su);
                    }
                    lb0 = m + log(ssum);
                }
                const int64_t offq = (int64_t)(s + 1) * ncol + c;  // onset index s-L+1 -> row s+1
#pragma unroll
                for (int a = 0; a < N; a++) Q[a * planeQ + offq] = Yn[a];
                if (s < nc) B0[(int64_t)s * ncol + c] = lb0;
                if (s == nc) B0h[c] = lb0;
            }
        }
        if (sb - U >= 0) {
#pragma unroll
            for (int u = 0; u < U; u++) cur[u] = nxt[u];
        }
    }
}

// ------------------------------------------------------------------------------------------
// per-chain normaliser: Zc = log sum over ALL states of alpha*beta at t* = tc + L - 1
//   silent: la0(t*) + lb0(t*);  ring state (a,k): lp_a(t') + ly_a(t'), t' = t*-k+1 in [tc, t*].
// ------------------------------------------------------------------------------------------
__global__ void k_znorm(RingGeom g, const double *__restrict__ P, const double *__restrict__ Q,
                        const double *__restrict__ A0, const double *__restrict__ B0,
                        double *__restrict__ Zc)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.nch) return;
    const int L = g.L, N = g.N, ncol = g.ncol;
    const int64_t planeP = (int64_t)(g.H + g.B) * ncol, planeQ = (int64_t)(L + g.B + g.H) * ncol;
    double m = A0[(int64_t)(1 + L - 1) * ncol + c] + B0[(int64_t)(L - 1) * ncol + c];
    double s = 1.0;
    for (int a = 0; a < N; a++)
        for (int i = 0; i < L; i++) {
            const double v = P[a * planeP + (int64_t)(g.H + i) * ncol + c] +
                             Q[a * planeQ + (int64_t)(L + i) * ncol + c];
            if (v > m) { s = s * exp(m - v) + 1.0; m = v; }
            else s += exp(v - m);
        }
    Zc[c] = m + log(s);
}

// ------------------------------------------------------------------------------------------
// posterior statistics (baumwelch.jl:216-305 fused).  Work item = 64 chain columns x a row range;
// tiles of TR rows: phase 1 (lane = column) turns lp+ly-Zc into rho and a per-(ring,row) bitmask
// of the non-zero columns; phase 2 (thread = ring state (a,k)) accumulates
//   G0 += rho, G1 += rho*y[t'+k-1], G2 += rho*y[t'+k-1]^2
// visiting only the non-zero entries (rho underflows to exactly 0 away from spikes).
// Pair group blockIdx.y covers ring states [256*y, 256*y+256).
// ------------------------------------------------------------------------------------------
struct StatsCfg {
    int TR;          // rows per tile
    int nitems;      // work items = (ncol/64) * rsplit
    int rsplit;      // row ranges per column group
    int rows_per;    // rows per work item
    int NLpad;       // N*L rounded up to 256
};

template <int N>
__global__ __launch_bounds__(256) void k_stats(RingGeom g, StatsCfg cfg, JParams<N> jp,
                                               const double *__restrict__ y,
                                               const double *__restrict__ Rf,
                                               const double *__restrict__ P,
                                               const double *__restrict__ Q,
                                               const double *__restrict__ A0,
                                               const double *__restrict__ B0,
                                               const double *__restrict__ Zc,
                                               double *__restrict__ partA,
                                               double *__restrict__ partS)
{
    extern __shared__ double lds[];
    const int TR = cfg.TR, L = g.L, B = g.B, H = g.H, ncol = g.ncol;
    const int WY = TR + L - 1;                    // y window per column
    double *rho = lds;                            // [N][TR][64]
    double *yt = lds + (size_t)N * TR * 64;       // [64][WY]
    unsigned long long *mask = (unsigned long long *)(yt + (size_t)64 * WY);  // [N][TR]
    int *vlen = (int *)(mask + N * TR);           // [64] valid samples from the tile start
    __shared__ double red[4];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pair = blockIdx.y * 256 + tid;      // ring state handled in phase 2
    const bool has_pair = pair < N * L;
    const int pa = has_pair ? pair / L : 0, pk = has_pair ? pair % L + 1 : 1;
    const bool scal = (blockIdx.y == 0);          // scalar sums are done by pair group 0 only
    const int64_t planeR = (int64_t)B * ncol, planeP = (int64_t)(H + B) * ncol,
                  planeQ = (int64_t)(L + B + H) * ncol;
    const int64_t T = g.T;

    double g0 = 0.0, g1 = 0.0, g2 = 0.0;
    double sx[N], s_all = 0.0, s_m = 0.0, s_y2 = 0.0;
#pragma unroll
    for (int a = 0; a < N; a++) sx[a] = 0.0;

    for (int item = blockIdx.x; item < cfg.nitems; item += gridDim.x) {
        const int c0 = (item / cfg.rsplit) * 64;
        const int rbeg = (item % cfg.rsplit) * cfg.rows_per;
        const int rend = rbeg + cfg.rows_per < B ? rbeg + cfg.rows_per : B;
        const int c = c0 + lane;
        const bool cact = c < g.nch;
        const double z = cact ? Zc[c] : 0.0;
        for (int r0 = rbeg; r0 < rend; r0 += TR) {
            __syncthreads();
            // y windows: column cl holds y[(c0+cl)*B + r0 + j], j in [0, WY)
            for (int i = tid; i < 64 * WY; i += 256) {
                const int cl = i / WY, j = i - cl * WY;
                const int64_t t = (int64_t)(c0 + cl) * B + r0 + j;
                yt[i] = (t < T) ? y[t] : 0.0;
            }
            if (tid < 64) {
                const int64_t rem = T - ((int64_t)(c0 + tid) * B + r0);
                vlen[tid] = rem < 0 ? 0 : (rem > WY ? WY : (int)rem);
            }
            __syncthreads();
            // phase 1: rows r0 + wv, r0 + wv + 4, ...
            for (int rr = wv; rr < TR; rr += 4) {
                const int s = r0 + rr;
                const int64_t t = (int64_t)c * B + s;
                const bool on = cact && s < rend && t < T;
                const int64_t offp = (int64_t)(H + s) * ncol + c, offq = (int64_t)(L + s) * ncol + c;
                double la_prev = 0.0;
                if (on && scal) {
                    const double a0 = A0[(int64_t)(1 + s) * ncol + c];
                    const double ga = exp((a0 + B0[(int64_t)s * ncol + c]) - z);  // gamma_t(silent)
                    const double yv = yt[lane * WY + rr];
                    s_all += ga;                       // baumwelch.jl:303 qq
                    if (t < T - 1) s_m += ga;          // :257 bb, t = 1..T-1
                    s_y2 += ga * (yv * yv);            // :302 with the new silent mean (= 0)
                    la_prev = A0[(int64_t)s * ncol + c];  // la0(t-1) in this chain's frame
                }
#pragma unroll
                for (int a = 0; a < N; a++) {
                    double rv = 0.0;
                    if (on) {
                        const double ly = Q[a * planeQ + offq];
                        rv = exp((P[a * planeP + offp] + ly) - z);
                        if (scal && t >= 1) {  // xi: silent at t-1 -> (a,1) at t   :240
                            const double rf = Rf[a * planeR + (int64_t)s * ncol + c];
                            sx[a] += exp((((la_prev + jp.c0[a]) + rf) + ly) - z);
                        }
                    }
                    rho[((size_t)a * TR + rr) * 64 + lane] = rv;
                    const unsigned long long bm = __ballot(rv != 0.0);
                    if (lane == 0) mask[a * TR + rr] = bm;
                }
            }
            __syncthreads();
            // phase 2
            if (has_pair) {
                for (int rr = 0; rr < TR; rr++) {
                    unsigned long long bm = mask[pa * TR + rr];
                    while (bm) {
                        const int cl = __builtin_ctzll(bm);
                        bm &= bm - 1;
                        const int j = rr + pk - 1;
                        if (j < vlen[cl]) {
                            const double rv = rho[((size_t)pa * TR + rr) * 64 + cl];
                            const double yv = yt[cl * WY + j];
                            g0 += rv;
                            g1 += rv * yv;
                            g2 += rv * (yv * yv);
                        }
                    }
                }
            }
        }
    }
    // per-block partials
    double *pa_out = partA + (size_t)blockIdx.x * 3 * cfg.NLpad;
    pa_out[pair] = g0;
    pa_out[cfg.NLpad + pair] = g1;
    pa_out[2 * cfg.NLpad + pair] = g2;
    if (scal) {
        double v[N + 3];
#pragma unroll
        for (int a = 0; a < N; a++) v[a] = sx[a];
        v[N] = s_all; v[N + 1] = s_m; v[N + 2] = s_y2;
#pragma unroll
        for (int i = 0; i < N + 3; i++) {
            double x = v[i];
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
            __syncthreads();
            if (lane == 0) red[wv] = x;
            __syncthreads();
            if (tid == 0) partS[(size_t)blockIdx.x * (N + 4) + i] = (red[0] + red[1]) + (red[2] + red[3]);
        }
        if (tid == 0) partS[(size_t)blockIdx.x * (N + 4) + N + 3] = 0.0;
    }
}

// virtual onsets t' = -j (rings already running at the first sample) and pp = gamma[:,1]
// (baumwelch.jl:263).  One block; appended as one more partial row.
__global__ __launch_bounds__(256) void k_stats_virtual(RingGeom g, int NLpad, int prow,
                                                       const double *__restrict__ y,
                                                       const double *__restrict__ P,
                                                       const double *__restrict__ Q,
                                                       const double *__restrict__ A0,
                                                       const double *__restrict__ B0,
                                                       const double *__restrict__ Zc,
                                                       double *__restrict__ partA,
                                                       double *__restrict__ pp)
{
    const int L = g.L, N = g.N, ncol = g.ncol;
    const int64_t planeP = (int64_t)(g.H + g.B) * ncol, planeQ = (int64_t)(L + g.B + g.H) * ncol;
    const double z = Zc[0];
    double *out = partA + (size_t)prow * 3 * NLpad;
    for (int pair = threadIdx.x; pair < NLpad; pair += blockDim.x) {
        double g0 = 0.0, g1 = 0.0, g2 = 0.0;
        if (pair < N * L) {
            const int a = pair / L, k = pair % L + 1;
            for (int j = 1; j <= L - 1; j++) {
                const int idx = -j + k - 1;
                if (idx < 0) continue;
                const double rv = exp((P[a * planeP + (int64_t)(g.H - j) * ncol] +
                                       Q[a * planeQ + (int64_t)(L - j) * ncol]) - z);
                const double yv = y[idx];
                g0 += rv; g1 += rv * yv; g2 += rv * (yv * yv);
            }
            // pp for state (a,k): the onset at t' = 1-k
            const int sp = 1 - k;
            pp[1 + pair] = (P[a * planeP + (int64_t)(g.H + sp) * ncol] +
                            Q[a * planeQ + (int64_t)(L + sp) * ncol]) - z;
        }
        out[pair] = g0; out[NLpad + pair] = g1; out[2 * NLpad + pair] = g2;
    }
    if (threadIdx.x == 0) pp[0] = (A0[(int64_t)1 * ncol] + B0[0]) - z;
}

// deterministic reduction of the partial rows: stats = [G0 | G1 | G2 | Xi | s_all | s_m | s_y2 | 0]
__global__ void k_stats_reduce(int NL, int NLpad, int N, int rowsA, int rowsS,
                               const double *__restrict__ partA, const double *__restrict__ partS,
                               double *__restrict__ stats)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = 3 * NL + N + 4;
    if (i >= total) return;
    double acc = 0.0;
    if (i < 3 * NL) {
        const int which = i / NL, pair = i % NL;
        for (int r = 0; r < rowsA; r++) acc += partA[(size_t)r * 3 * NLpad + which * NLpad + pair];
    } else {
        const int e = i - 3 * NL;
        for (int r = 0; r < rowsS; r++) acc += partS[(size_t)r * (N + 4) + e];
    }
    stats[i] = acc;
}

// M-step finish (baumwelch.jl:262-307) from the (possibly all-reduced) statistics.
// out = [mu (K x N col-major) | sigma | lp_new (N) | pp (S)]
__global__ __launch_bounds__(256) void k_mstep(int N, int L, const double *__restrict__ stats,
                                               const double *__restrict__ pp,
                                               double *__restrict__ out)
{
    __shared__ double red[8];
    const int NL = N * L, K = L + 1;
    const double *G0 = stats, *G1 = stats + NL, *G2 = stats + 2 * NL, *Xi = stats + 3 * NL;
    const double s_all = stats[3 * NL + N], s_m = stats[3 * NL + N + 1], s_y2 = stats[3 * NL + N + 2];
    double x2 = 0.0, qq = 0.0;
    for (int p = threadIdx.x; p < NL; p += blockDim.x) {
        const int a = p / L, k = p % L + 1;
        const double mu = G1[p] / G0[p];                 // :285  mu[j,l] /= gg[j,l]
        out[k + K * a] = mu;                             // state (a,k) uses row k+1 (1-based)
        x2 += (G2[p] - (2.0 * mu) * G1[p]) + (mu * mu) * G0[p];  // sum_t gamma (y-mu)^2
        qq += G0[p];
    }
    for (int a = threadIdx.x; a < N; a += blockDim.x) {
        out[K * a] = 0.0;                                // row 1 stays 0 (:268 fill!, never updated)
        out[K * N + 1 + a] = log(Xi[a]) - log(s_m);      // :264 xb[2:end]
    }
    for (int j = threadIdx.x; j < 1 + NL; j += blockDim.x) out[K * N + 1 + N + j] = pp[j];
    for (int o = 32; o > 0; o >>= 1) { x2 += __shfl_xor(x2, o); qq += __shfl_xor(qq, o); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = x2; red[4 + (threadIdx.x >> 6)] = qq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double X2 = ((red[0] + red[1]) + (red[2] + red[3])) + s_y2;
        const double QQ = ((red[4] + red[5]) + (red[6] + red[7])) + s_all;
        out[K * N] = sqrt(X2 / QQ);                      // :306-307
    }
}

template <int N>
static EParams<N> make_eparams(const RingDev *r)
{
    EParams<N> p;
    p.p00 = std::exp(r->ring.c00);
    for (int a = 0; a < N; a++) {
        p.p0[a] = std::exp(r->ring.c0[a]);
        p.pend[a] = std::exp(r->ring.cend[a]);
        for (int b = 0; b < N; b++) p.px[a * N + b] = (a == b) ? 0.0 : std::exp(r->ring.cx[a * N + b]);
    }
    p.mean0 = r->mean[0];
    p.den = r->den;
    return p;
}

template <int N>
static JParams<N> make_jparams_e(const RingDev *r)
{
    JParams<N> p;
    p.c00 = r->ring.c00;
    for (int a = 0; a < N; a++) {
        p.c0[a] = r->ring.c0[a];
        p.cend[a] = r->ring.cend[a];
        for (int b = 0; b < N; b++) p.cx[a * N + b] = r->ring.cx[a * N + b];
    }
    p.mean0 = r->mean[0];
    p.den = r->den;
    p.A = r->A;
    return p;
}

int ring_estep_launch(RingDev *r, const double *d_y, double *d_stats, hipStream_t st)
{
    const RingGeom &g = r->g;
    const int N = g.N, L = g.L, NL = N * L;
    int rc;
    HS_HIP(hipMemsetAsync(r->diag, 0, 8 * sizeof(int64_t), st));
    if ((rc = ring_launch_transpose_in(r, d_y, st))) return rc;
    if ((rc = ring_launch_prepass(r, st))) return rc;
    if ((rc = ring_launch_virtual(r, d_y, r->P, (int64_t)(g.H + g.B) * g.ncol, st))) return rc;
    // stats geometry
    StatsCfg cfg;
    cfg.NLpad = (NL + 255) / 256 * 256;
    const int ngroups = cfg.NLpad / 256;
    cfg.TR = 8;
    auto lds_bytes = [&](int TR) {
        return (size_t)N * TR * 64 * 8 + (size_t)64 * (TR + L - 1) * 8 + (size_t)N * TR * 8 + 64 * 4;
    };
    while (cfg.TR > 1 && lds_bytes(cfg.TR) > 150 * 1024) cfg.TR /= 2;
    HS_CHECK(lds_bytes(cfg.TR) <= 160 * 1024, HMMSORT_EUNSUP,
             "ring E-step: ring length %d does not fit the statistics tile", L);
    const int colgroups = g.ncol / 64;
    cfg.rsplit = 1;
    while (colgroups * cfg.rsplit < 1024 && cfg.rsplit < 8 && (g.B / (cfg.rsplit * 2)) >= 64)
        cfg.rsplit *= 2;
    cfg.rows_per = (g.B / cfg.rsplit + cfg.TR - 1) / cfg.TR * cfg.TR;
    cfg.nitems = colgroups * cfg.rsplit;
    const int gx = std::min(cfg.nitems, r->nparts - 1);
    rc = dispatch_N(N, [&](auto n) {
        constexpr int NN = decltype(n)::value;
        EParams<NN> ep = make_eparams<NN>(r);
        JParams<NN> jp = make_jparams_e<NN>(r);
        { PROF(r, "k_fwd_chain", st); hipLaunchKernelGGL((k_fwd_chain<NN>), dim3(g.ncol / 64), dim3(64), 0, st, g, ep, r->yT, r->Rf,
                           r->P, r->A0); }
        { PROF(r, "k_bwd_chain", st); hipLaunchKernelGGL((k_bwd_chain<NN>), dim3(g.ncol / 64), dim3(64), 0, st, g, ep, r->yT, r->Rf,
                           r->Q, r->B0, r->B0h); }
        { PROF(r, "k_znorm", st); hipLaunchKernelGGL(k_znorm, dim3((g.nch + 63) / 64), dim3(64), 0, st, g, r->P, r->Q, r->A0,
                           r->B0, r->Zc); }
        const size_t lds = lds_bytes(cfg.TR);
        if (lds > 64 * 1024)
            HS_HIP(hipFuncSetAttribute((const void *)k_stats<NN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        { PROF(r, "k_stats", st); hipLaunchKernelGGL((k_stats<NN>), dim3(gx, ngroups), dim3(256), lds, st, g, cfg, jp, d_y,
                           r->Rf, r->P, r->Q, r->A0, r->B0, r->Zc, r->partA, r->partS); }
        HS_HIP(hipGetLastError());
        return HMMSORT_OK;
    });
    if (rc) return rc;
    { PROF(r, "k_stats_virtual", st); hipLaunchKernelGGL(k_stats_virtual, dim3(1), dim3(256), 0, st, g, cfg.NLpad, gx, d_y, r->P, r->Q,
                       r->A0, r->B0, r->Zc, r->partA, r->pp); }
    const int total = 3 * NL + N + 4;
    { PROF(r, "k_stats_reduce", st); hipLaunchKernelGGL(k_stats_reduce, dim3((total + 255) / 256), dim3(256), 0, st, NL, cfg.NLpad, N,
                       gx + 1, gx, r->partA, r->partS, d_stats); }
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int ring_mstep_launch(RingDev *r, const double *d_stats, double *d_out, hipStream_t st)
{
    { PROF(r, "k_mstep", st); hipLaunchKernelGGL(k_mstep, dim3(1), dim3(256), 0, st, r->g.N, r->g.L, d_stats, r->pp, d_out); }
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

}  // namespace hmmsort
