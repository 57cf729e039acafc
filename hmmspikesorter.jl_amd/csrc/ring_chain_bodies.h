// Device bodies of the three serial junction sweeps (Viterbi, forward, backward), shared by the
// stand-alone kernels (ring_viterbi.hip, ring_estep.hip) and the fused launch (ring_fused.hip).
#pragma once
#include <cmath>

#include "fastmath.h"
#include "ring_common.h"

namespace hmmsort {

template <int N>
struct VitIn {
    double y;
    double R[N];
    double X[N];
};

// One lane = one chain; bx = index of the 64-chain column group.
template <int N>
__device__ __forceinline__ void vit_chain_body(int bx, const RingGeom &g, const JParams<N> &jp,
                                                  const double *__restrict__ yT,
                                                  const double *__restrict__ Rf,
                                                  double *__restrict__ P,
                                                  uint32_t *__restrict__ psi,
                                                  double *__restrict__ D0pre,
                                                  double *__restrict__ D0end)
{
    constexpr int U = chain_unroll<N>();
    constexpr int BITS = psi_bits_c(N), EPW = psi_epw_c(N), W = psi_words_c(N);
    const int c = bx * 64 + threadIdx.x;
    const int B = g.B, H = g.H, L = g.L, ncol = g.ncol;
    const bool active = c < g.nch;
    const int64_t tc = (int64_t)c * B;
    const int nc = active ? (int)((g.T - tc) < B ? (g.T - tc) : B) : 0;
    const int s0 = (c == 0) ? 0 : -H;
    const int64_t planeR = (int64_t)B * ncol, planeP = (int64_t)(H + B) * ncol;
    const int64_t planePsi = (int64_t)B * ncol;
    // Loads are unconditional (idle lanes read a clamped, valid address and discard the value):
    // straight-line loads let the compiler count vmcnt exactly, so a batch only waits for its
    // own data while the next batch's loads stay in flight.
    auto load = [&](VitIn<N>(&d)[U], int sb) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = sb + u;
            const bool live = active && s >= s0 && s < nc;
            const int sc = live ? s : 0;             // clamped step
            const int cc = active ? c : 0;           // clamped column
            const int64_t off = (sc >= 0) ? (int64_t)sc * ncol + cc : (int64_t)(B + sc) * ncol + (cc > 0 ? cc - 1 : 0);
            const bool hasx = live && (s - L >= -H);
            const int64_t offp = (int64_t)(hasx ? H + s - L : H) * ncol + cc;
            const double yv = yT[off];
            double rv[N], xv[N];
#pragma unroll
            for (int a = 0; a < N; a++) rv[a] = Rf[a * planeR + off];
#pragma unroll
            for (int a = 0; a < N; a++) xv[a] = P[a * planeP + offp];
            d[u].y = live ? yv : 0.0;
#pragma unroll
            for (int a = 0; a < N; a++) {
                d[u].R[a] = live ? rv[a] : 0.0;
                d[u].X[a] = hasx ? xv[a] : -INFINITY;
            }
        }
    };

    VitIn<N> bufA[U], bufB[U];
    double D0 = 0.0;
    const int sfirst = -H;  // wave-uniform loop start (chain 0 idles through the warm-up steps)
    auto run = [&](VitIn<N>(&cur)[U], int sb) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = sb + u;
            const bool live = active && s >= s0 && s < nc;
            if (live) {
                double Pn[N];
                uint32_t pw[W];
#pragma unroll
                for (int w = 0; w < W; w++) pw[w] = 0u;
                if (s == s0) {
                    // first sample of the chain: chain 0 = the reference's first column
                    // (viterbi.jl:55-63: emission only, T1[1,1] = 0); others = "silent, rings
                    // empty" warm-up start
                    if (c == 0) {
                        D0 = -jp.A;
#pragma unroll
                        for (int a = 0; a < N; a++) Pn[a] = cur[u].R[a];
                    } else {
                        D0 = 0.0;
#pragma unroll
                        for (int a = 0; a < N; a++) Pn[a] = -INFINITY;
                    }
                } else {
                    double best0 = D0 + jp.c00;
                    int p0 = 0;
#pragma unroll
                    for (int a = 0; a < N; a++) {
                        const double v = cur[u].X[a] + jp.cend[a];
                        if (v > best0) { best0 = v; p0 = a + 1; }
                    }
                    pw[0] = (uint32_t)p0;
#pragma unroll
                    for (int a = 0; a < N; a++) {
                        double ua = D0 + jp.c0[a];
                        int pa = 0;
#pragma unroll
                        for (int b = 0; b < N; b++) {
                            if (b == a) continue;
                            const double v = cur[u].X[b] + jp.cx[b * N + a];
                            if (v > ua) { ua = v; pa = b + 1; }
                        }
                        Pn[a] = ua + cur[u].R[a];
                        pw[(a + 1) / EPW] |= (uint32_t)pa << (((a + 1) % EPW) * BITS);
                    }
                    const double d = cur[u].y - jp.mean0;
                    D0 = best0 - (d * d) / jp.den;
                }
                const int64_t offp = (int64_t)(H + s) * ncol + c;
#pragma unroll
                for (int a = 0; a < N; a++) P[a * planeP + offp] = Pn[a];
                if (s >= 0) {
                    const int64_t o = (int64_t)s * ncol + c;
#pragma unroll
                    for (int w = 0; w < W; w++) psi[w * planePsi + o] = pw[w];
                } else if (s == -1) {
                    D0pre[c] = D0;  // delta(silent) one sample before the chain, warm-up frame
                }
                if (s == nc - 1) D0end[c] = D0;
            }
        }
    };
    // two batches per iteration, ping-pong: a batch's inputs are fetched while the previous
    // batch computes, with no register copies (H and B are multiples of 64, hence of 2U)
    load(bufA, sfirst);
    for (int sb = sfirst; sb < B; sb += 2 * U) {
        load(bufB, sb + U);
        run(bufA, sb);
        if (sb + 2 * U < B) load(bufA, sb + 2 * U);
        run(bufB, sb + U);
    }
}

template <int N>
struct ChainIn {
    double y;
    double R[N];
    double X[N];
};

// ------------------------------------------------------------------------------------------
// forward chains (baumwelch.jl:25-51).  Same skeleton as k_vit_chain; max -> log-sum-exp.
// The N+1 exponentials fexp(value - m) are shared by all N+1 junction sums.
// ------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void fwd_chain_body(int bx, const RingGeom &g, const EParams<N> &ep,
                                               const double *__restrict__ yT,
                                               const double *__restrict__ Rf,
                                               double *__restrict__ P, double *__restrict__ A0)
{
    constexpr int U = chain_unroll<N>();
    const int c = bx * 64 + threadIdx.x;
    const int B = g.B, H = g.H, L = g.L, ncol = g.ncol;
    const bool active = c < g.nch;
    const int64_t tc = (int64_t)c * B;
    const int nc = active ? (int)((g.T - tc) < B ? (g.T - tc) : B) : 0;
    const int s0 = (c == 0) ? 0 : -H;
    const int64_t planeR = (int64_t)B * ncol, planeP = (int64_t)(H + B) * ncol;

    auto load = [&](ChainIn<N>(&d)[U], int sb) {  // unconditional loads, see k_vit_chain
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = sb + u;
            const bool live = active && s >= s0 && s < nc;
            const int sc = live ? s : 0;
            const int cc = active ? c : 0;
            const int64_t off = (sc >= 0) ? (int64_t)sc * ncol + cc : (int64_t)(B + sc) * ncol + (cc > 0 ? cc - 1 : 0);
            const bool hasx = live && (s - L >= -H);
            const int64_t offp = (int64_t)(hasx ? H + s - L : H) * ncol + cc;
            const double yv = yT[off];
            double rv[N], xv[N];
#pragma unroll
            for (int a = 0; a < N; a++) rv[a] = Rf[a * planeR + off];
#pragma unroll
            for (int a = 0; a < N; a++) xv[a] = P[a * planeP + offp];
            d[u].y = live ? yv : 0.0;
#pragma unroll
            for (int a = 0; a < N; a++) {
                d[u].R[a] = live ? rv[a] : 0.0;
                d[u].X[a] = hasx ? xv[a] : -INFINITY;
            }
        }
    };

    ChainIn<N> bufA[U], bufB[U];
    double la0 = 0.0;
    auto run = [&](ChainIn<N>(&cur)[U], int sb) {
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
                    double ev[N + 1];  // exp(value - m), shared by all junction sums
                    ev[N] = la0 - m;
#pragma unroll
                    for (int a = 0; a < N; a++) ev[a] = cur[u].X[a] - m;
                    fexp_n<N + 1>(ev);
                    const double e0 = ev[N];
                    double sv[N + 1];  // the N+1 junction sums
                    sv[N] = e0 * ep.p00;
#pragma unroll
                    for (int a = 0; a < N; a++) sv[N] = __builtin_fma(ev[a], ep.pend[a], sv[N]);
#pragma unroll
                    for (int a = 0; a < N; a++) {
                        double su = e0 * ep.p0[a];
#pragma unroll
                        for (int b = 0; b < N; b++)
                            if (b != a) su = __builtin_fma(ev[b], ep.px[b * N + a], su);
                        sv[a] = su;
                    }
                    flog_n<N + 1>(sv);
#pragma unroll
                    for (int a = 0; a < N; a++) Pn[a] = (m + sv[a]) + cur[u].R[a];
                    la0 = (m + sv[N]) + q0;
                }
                const int64_t offp = (int64_t)(H + s) * ncol + c;
#pragma unroll
                for (int a = 0; a < N; a++) P[a * planeP + offp] = Pn[a];
                if (s >= -1) A0[(int64_t)(1 + s) * ncol + c] = la0;
            }
        }
    };
    load(bufA, -H);
    for (int sb = -H; sb < B; sb += 2 * U) {  // ping-pong, no register copies
        load(bufB, sb + U);
        run(bufA, sb);
        if (sb + 2 * U < B) load(bufA, sb + 2 * U);
        run(bufB, sb + U);
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
__device__ __forceinline__ void bwd_chain_body(int bx, const RingGeom &g, const EParams<N> &ep,
                                               const double *__restrict__ yT,
                                               const double *__restrict__ Rf,
                                               double *__restrict__ Q, double *__restrict__ B0,
                                               double *__restrict__ B0h)
{
    constexpr int U = chain_unroll<N>();
    const int c = bx * 64 + threadIdx.x;
    const int B = g.B, H = g.H, L = g.L, ncol = g.ncol;
    const bool active = c < g.nch;
    const int64_t tc = (int64_t)c * B;
    const int nc = active ? (int)((g.T - tc) < B ? (g.T - tc) : B) : 0;
    int64_t te = tc + nc + H;
    if (te > g.T) te = g.T;
    const int se = active ? (int)(te - 1 - tc) : -1;
    const int64_t planeR = (int64_t)B * ncol, planeQ = (int64_t)(L + B + H) * ncol;

    // inputs of step s (computing time t = tc+s from t+1): y, Rf and ly at time t+1
    auto load = [&](ChainIn<N>(&d)[U], int sb) {  // unconditional loads, see k_vit_chain
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = sb - u;
            const bool live = active && s < se && s >= 0;
            const int s1 = live ? s + 1 : 0;
            const int cc = active ? c : 0;
            const int64_t off = (s1 < B) ? (int64_t)s1 * ncol + cc : (int64_t)(s1 - B) * ncol + cc + 1;
            const bool hasq = live && (s + L <= se);  // the ring started at t+1 ends inside the range
            const int64_t offq = (int64_t)(hasq ? L + s1 : L) * ncol + cc;
            const double yv = yT[off];
            double rv[N], xv[N];
#pragma unroll
            for (int a = 0; a < N; a++) rv[a] = Rf[a * planeR + off];
#pragma unroll
            for (int a = 0; a < N; a++) xv[a] = Q[a * planeQ + offq];
            d[u].y = live ? yv : 0.0;
#pragma unroll
            for (int a = 0; a < N; a++) {
                d[u].R[a] = live ? rv[a] : 0.0;
                d[u].X[a] = hasq ? xv[a] : 0.0;
            }
        }
    };

    ChainIn<N> bufA[U], bufB[U];
    double lb0 = 0.0;
    const int stop = B + H - 1;
    auto run = [&](ChainIn<N>(&cur)[U], int sb) {
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
                    double ev[N + 1];
                    ev[N] = v0 - m;
#pragma unroll
                    for (int a = 0; a < N; a++) ev[a] = lw[a] - m;
                    fexp_n<N + 1>(ev);
                    const double E0 = ev[N];
                    double sv[N + 1];
                    sv[N] = E0 * ep.p00;
#pragma unroll
                    for (int a = 0; a < N; a++) sv[N] = __builtin_fma(ev[a], ep.p0[a], sv[N]);
#pragma unroll
                    for (int a = 0; a < N; a++) {
                        double su = E0 * ep.pend[a];
#pragma unroll
                        for (int b = 0; b < N; b++)
                            if (b != a) su = __builtin_fma(ev[b], ep.px[a * N + b], su);
                        sv[a] = su;
                    }
                    flog_n<N + 1>(sv);
#pragma unroll
                    for (int a = 0; a < N; a++) Yn[a] = m + sv[a];
                    lb0 = m + sv[N];
                }
                const int64_t offq = (int64_t)(s + 1) * ncol + c;  // onset index s-L+1 -> row s+1
#pragma unroll
                for (int a = 0; a < N; a++) Q[a * planeQ + offq] = Yn[a];
                if (s < nc) B0[(int64_t)s * ncol + c] = lb0;
                if (s == nc) B0h[c] = lb0;
            }
        }
    };
    load(bufA, stop);
    for (int sb = stop; sb >= 0; sb -= 2 * U) {  // ping-pong, no register copies
        load(bufB, sb - U);
        run(bufA, sb);
        if (sb - 2 * U >= 0) load(bufA, sb - 2 * U);
        run(bufB, sb - U);
    }
}

// ------------------------------------------------------------------------------------------
// k_post: per-chain normaliser + posteriors.  Lane = chain.
//   Zc = log sum over ALL states of alpha*beta at t* = tc + L - 1
//        silent: la0(t*) + lb0(t*);  ring state (a,k): lp_a(t') + ly_a(t'), t' in [tc, t*];
//   rho_a(t') = fexp(lp_a + ly_a - Zc)  -> rhoT (transposed layout, zero where there is no onset);
//   scalar sums: gamma_t(silent) (all t; t < T-1; times y^2), xi_a, sum_t' rho_a(t').
// ------------------------------------------------------------------------------------------

}  // namespace hmmsort
