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

#include "fastmath.h"
#include "ring_chain_bodies.h"
#include "ring_common.h"

namespace hmmsort {

// Forward and backward sweeps are independent until the statistics; one launch runs both, the
// forward chains on the even blocks and the backward chains on the odd ones, so 2 x (chains/64)
// wavefronts share the 1024 SIMDs instead of running half-empty one after the other.
template <int N>
__global__ __launch_bounds__(64) void k_fb_chain(RingGeom g, EParams<N> ep,
                                                 const double *__restrict__ yT,
                                                 const double *__restrict__ Rf,
                                                 double *__restrict__ P, double *__restrict__ A0,
                                                 double *__restrict__ Q, double *__restrict__ B0,
                                                 double *__restrict__ B0h)
{
    const int bx = blockIdx.x >> 1;
    if (blockIdx.x & 1) bwd_chain_body<N>(bx, g, ep, yT, Rf, Q, B0, B0h);
    else fwd_chain_body<N>(bx, g, ep, yT, Rf, P, A0);
}

constexpr int kZParts = 4;

// partial log-sum-exp of the ring-state terms lp_a(t') + ly_a(t') over window rows
// i = blockIdx.y, blockIdx.y + kZParts, ...  ->  Zp[part][0] = max, Zp[part][1] = sum exp(. - max)
template <int N>
__global__ __launch_bounds__(64) void k_znorm(RingGeom g, const double *__restrict__ P,
                                              const double *__restrict__ Q,
                                              double *__restrict__ Zp)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= g.nch) return;
    const int H = g.H, L = g.L, ncol = g.ncol;
    const int64_t planeP = (int64_t)(H + g.B) * ncol, planeQ = (int64_t)(L + g.B + H) * ncol;
    double m = -INFINITY, sm = 0.0;
#pragma unroll 4
    for (int i = blockIdx.y; i < L; i += kZParts) {
        double v[N];
#pragma unroll
        for (int a = 0; a < N; a++)
            v[a] = P[a * planeP + (int64_t)(H + i) * ncol + c] + Q[a * planeQ + (int64_t)(L + i) * ncol + c];
        double mi = v[0];
#pragma unroll
        for (int a = 1; a < N; a++) mi = fmax(mi, v[a]);
        if (mi > m) { sm *= fexp(m - mi); m = mi; }
#pragma unroll
        for (int a = 0; a < N; a++) sm += fexp(v[a] - m);
    }
    Zp[((int64_t)blockIdx.y * 2 + 0) * ncol + c] = m;
    Zp[((int64_t)blockIdx.y * 2 + 1) * ncol + c] = sm;
}

// Zc = log( exp(la0(t*) + lb0(t*)) + sum of the partials ),  t* = tc + L - 1
__device__ __forceinline__ double z_combine(const RingGeom &g, const double *__restrict__ Zp,
                                            const double *__restrict__ A0,
                                            const double *__restrict__ B0, int c)
{
    const int L = g.L, ncol = g.ncol;
    double m = A0[(int64_t)L * ncol + c] + B0[(int64_t)(L - 1) * ncol + c];
    double pm[kZParts], ps[kZParts];
#pragma unroll
    for (int j = 0; j < kZParts; j++) {
        pm[j] = Zp[((int64_t)j * 2 + 0) * ncol + c];
        ps[j] = Zp[((int64_t)j * 2 + 1) * ncol + c];
    }
    const double v0 = m;
#pragma unroll
    for (int j = 0; j < kZParts; j++) m = fmax(m, pm[j]);
    double sm = fexp(v0 - m);
#pragma unroll
    for (int j = 0; j < kZParts; j++) sm += ps[j] * fexp(pm[j] - m);
    return m + flog(sm);
}

// grid = (column groups, row ranges of RP rows); one wave each
constexpr int kPostRows = 32;

template <int N>
__global__ __launch_bounds__(64) void k_post(RingGeom g, JParams<N> jp,
                                             const double *__restrict__ yT,
                                             const double *__restrict__ Rf,
                                             const double *__restrict__ P,
                                             const double *__restrict__ Q,
                                             const double *__restrict__ A0,
                                             const double *__restrict__ B0,
                                             const double *__restrict__ Zp, double *__restrict__ Zc,
                                             double *__restrict__ rhoT, double *__restrict__ partS)
{
    const int lane = threadIdx.x;
    const int c = blockIdx.x * 64 + lane;
    const int B = g.B, H = g.H, L = g.L, ncol = g.ncol;
    const bool active = c < g.nch;
    const int64_t tc = (int64_t)c * B;
    const int nc = active ? (int)((g.T - tc) < B ? (g.T - tc) : B) : 0;
    const int64_t planeR = (int64_t)B * ncol, planeP = (int64_t)(H + B) * ncol,
                  planeQ = (int64_t)(L + B + H) * ncol;
    const double z = active ? z_combine(g, Zp, A0, B0, c) : 0.0;
    if (active && blockIdx.y == 0) Zc[c] = z;
    const int sbeg = blockIdx.y * kPostRows;
    const int send = sbeg + kPostRows < B ? sbeg + kPostRows : B;
    double sx[N], ra[N], s_all = 0.0, s_m = 0.0, s_y2 = 0.0;
#pragma unroll
    for (int a = 0; a < N; a++) { sx[a] = 0.0; ra[a] = 0.0; }
#pragma unroll 4
    for (int s = sbeg; s < send; s++) {
        const int64_t off = (int64_t)s * ncol + c;
        double rv[N];
#pragma unroll
        for (int a = 0; a < N; a++) rv[a] = 0.0;
        if (s < nc && tc + s >= g.own_lo && tc + s < g.own_hi) {  // owned samples / onsets only
            const int64_t t = tc + s;
            const int64_t offp = (int64_t)(H + s) * ncol + c, offq = (int64_t)(L + s) * ncol + c;
            const double a0 = A0[(int64_t)(1 + s) * ncol + c];
            const double la_prev = A0[off];                   // la0(t-1) in this chain's frame
            double ex[2 * N + 1];
            ex[2 * N] = (a0 + B0[off]) - z;                   // gamma_t(silent)
#pragma unroll
            for (int a = 0; a < N; a++) {
                const double ly = Q[a * planeQ + offq];
                ex[a] = (P[a * planeP + offp] + ly) - z;      // rho_a(t)
                // xi: silent(t-1) -> (a,1)(t)   baumwelch.jl:240
                ex[N + a] = t >= 1 ? (((la_prev + jp.c0[a]) + Rf[a * planeR + off]) + ly) - z : -INFINITY;
            }
            fexp_n<2 * N + 1>(ex);
            const double ga = ex[2 * N];
            const double yv = yT[off];
            s_all += ga;                                      // baumwelch.jl:303 qq
            if (!g.last || t < g.T - 1) s_m += ga;            // :257 bb, t = 1..T-1 of the recording
            s_y2 += ga * (yv * yv);                           // :302 with the new silent mean (= 0)
            // onsets whose ring runs past the end of the recording (t > T-L) are summed per phase
            // in k_stats_edges: G0(a,k) must be a sum of positive terms, never "total - tail"
            const bool bulk = !(g.last && t > g.T - L);
#pragma unroll
            for (int a = 0; a < N; a++) {
                rv[a] = ex[a];
                ra[a] += bulk ? ex[a] : 0.0;
                sx[a] += ex[N + a];
            }
        }
#pragma unroll
        for (int a = 0; a < N; a++) rhoT[a * planeR + off] = rv[a];
    }
    // wave reduction -> partS[block][2N+3] = sx | ra | s_all s_m s_y2
    double v[2 * N + 3];
#pragma unroll
    for (int a = 0; a < N; a++) { v[a] = sx[a]; v[N + a] = ra[a]; }
    v[2 * N] = s_all; v[2 * N + 1] = s_m; v[2 * N + 2] = s_y2;
    const size_t prow = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
    for (int i = 0; i < 2 * N + 3; i++) {
        double x = v[i];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if (lane == 0) partS[prow * (2 * N + 3) + i] = x;
    }
}

// ------------------------------------------------------------------------------------------
// k_gsum: spike-triggered sums  G1(a,k) = sum_t' rho_a(t') y[t'+k-1],  G2 likewise with y^2
// (baumwelch.jl:270-282, :297-305).  Dense, register tiled: lane = chain, one wave = 64 chains x
// all N rings x KB consecutive phases k; per onset row one coalesced load of y (sliding window in
// registers) and N coalesced loads of rho feed 2*N*KB fma.  y beyond the end of the data is 0.
// ------------------------------------------------------------------------------------------
template <int N> constexpr int gsum_kb() { return N <= 4 ? 8 : (N <= 8 ? 4 : 2); }

template <int N>
__global__ __launch_bounds__(64) void k_gsum(RingGeom g, const double *__restrict__ yT,
                                             const double *__restrict__ rhoT,
                                             double *__restrict__ partG)
{
    constexpr int KB = gsum_kb<N>();
    const int lane = threadIdx.x;
    // 1-D grid, XCD-aware: the nkb waves that re-read the same 64 chain columns get consecutive
    // slots of ONE XCD (blocks b and b+8 share an XCD), so rho/y are fetched from HBM once and
    // served from that XCD's L2 afterwards.
    const int nkb = (g.L + KB - 1) / KB;
    const int grp = blockIdx.x / (8 * nkb), rem = blockIdx.x % (8 * nkb);
    const int cgi = grp * 8 + rem % 8, kbi = rem / 8;
    if (cgi * 64 >= g.ncol) return;
    const int c = cgi * 64 + lane;
    const int B = g.B, L = g.L, ncol = g.ncol;
    const int k0 = kbi * KB + 1;                   // phases k0 .. k0+KB-1
    const int64_t tb = (int64_t)c * B;
    const int64_t planeR = (int64_t)B * ncol;
    const bool active = c < g.nch;
    auto Y = [&](int r) -> double {                // y[tb + r], 0 past the end (unconditional load)
        const bool ok = active && tb + r < g.T;
        const int rr = ok ? r : 0;
        const int cc = active ? c : 0;
        const int64_t o = (rr < B) ? (int64_t)rr * ncol + cc : (int64_t)(rr - B) * ncol + cc + 1;
        const double v = yT[o];
        return ok ? v : 0.0;
    };
    double g1[N][KB], g2[N][KB], w[KB];
#pragma unroll
    for (int a = 0; a < N; a++)
#pragma unroll
        for (int j = 0; j < KB; j++) { g1[a][j] = 0.0; g2[a][j] = 0.0; }
    // window for onset row s: w[(s + j) % KB] = y[tb + s + k0 - 1 + j]
#pragma unroll
    for (int j = 0; j < KB - 1; j++) w[j] = Y(k0 - 1 + j);
    double rvA[KB][N], yA[KB], rvB[KB][N], yB[KB];
    auto loadb = [&](double(&rv)[KB][N], double(&yy)[KB], int sb) {
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const int s = sb + u;
            yy[u] = Y(s + k0 + KB - 2);
#pragma unroll
            for (int a = 0; a < N; a++) rv[u][a] = rhoT[a * planeR + (int64_t)s * ncol + c];
        }
    };
    auto run = [&](double(&rv)[KB][N], double(&yy)[KB]) {
#pragma unroll
        for (int u = 0; u < KB; u++) {
            w[(u + KB - 1) % KB] = yy[u];
#pragma unroll
            for (int j = 0; j < KB; j++) {
                const double yv = w[(u + j) % KB];
                const double y2 = yv * yv;
#pragma unroll
                for (int a = 0; a < N; a++) {
                    g1[a][j] = __builtin_fma(rv[u][a], yv, g1[a][j]);
                    g2[a][j] = __builtin_fma(rv[u][a], y2, g2[a][j]);
                }
            }
        }
    };
    loadb(rvA, yA, 0);
    for (int sb = 0; sb < B; sb += 2 * KB) {  // B is a multiple of 64 >= 2*KB
        loadb(rvB, yB, sb + KB);
        run(rvA, yA);
        if (sb + 2 * KB < B) loadb(rvA, yA, sb + 2 * KB);
        run(rvB, yB);
    }
    const int NL = N * L;
    double *out = partG + (size_t)cgi * 2 * NL;
#pragma unroll
    for (int a = 0; a < N; a++)
#pragma unroll
        for (int j = 0; j < KB; j++) {
            double x1 = g1[a][j], x2 = g2[a][j];
            for (int o = 32; o > 0; o >>= 1) { x1 += __shfl_xor(x1, o); x2 += __shfl_xor(x2, o); }
            const int k = k0 + j;
            if (lane == 0 && k <= L) {
                out[a * L + (k - 1)] = x1;
                out[NL + a * L + (k - 1)] = x2;
            }
        }
}

// ------------------------------------------------------------------------------------------
// Boundary certificate of the warm-ups (diag[3..6]).  Lane = boundary between chains c-1 and c.
// Forward: the state chain c reached at the end of its warm-up (la0 at tc-1 and the L onsets
// still inside their rings) against what chain c-1 computed for the same quantities; backward:
// the state chain c-1 reached at tc coming down from its warm-up (lb0 at tc and the ring-end
// betas Y_a(tc..tc+L-2)) against chain c's own.  Both pairs may differ by a frame constant D,
// taken at the entry with the largest posterior weight; the error is the posterior-weighted
// relative mismatch  sum_e w_e * |exp((x_e - x'_e) - D) - 1|  (entries nobody can reach have
// weight ~0 and do not count).
// ------------------------------------------------------------------------------------------
constexpr int kChkParts = 8;

// block = 64 boundaries x kChkParts entry subsets (512 threads); partial results meet in LDS
template <int N>
__global__ __launch_bounds__(64 * kChkParts) void k_fb_check(RingGeom g, double tol,
                                                            const double *__restrict__ P,
                                                            const double *__restrict__ Q,
                                                            const double *__restrict__ A0,
                                                            const double *__restrict__ B0,
                                                            const double *__restrict__ B0h,
                                                            const double *__restrict__ Zc,
                                                            const double *__restrict__ rhoT,
                                                            int64_t *__restrict__ diag)
{
    __shared__ double shw[kChkParts][64], shd[kChkParts][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;  // boundary at tc = c*B, c >= 1
    const bool on = c >= 1 && c < g.nch;
    const int B = g.B, H = g.H, L = g.L, ncol = g.ncol;
    const int64_t planeR = (int64_t)B * ncol, planeP = (int64_t)(H + B) * ncol,
                  planeQ = (int64_t)(L + B + H) * ncol;
    const int cp = c - 1;
    for (int dir = 0; dir < 2; dir++) {
        // entry e of ring a: forward j = e+1 (onset tc-j), backward i = e (ring-end beta at tc+i)
        const int ne = dir == 0 ? L : L - 1;
        auto wofs = [&](int a, int e) {   // posterior weight of the entry (rho of chain c-1)
            const int row = dir == 0 ? B - (e + 1) : B + e - L + 1;
            return a * planeR + (int64_t)row * ncol + cp;
        };
        auto diff = [&](int a, int e) {   // this chain's warm-up value minus the neighbour's
            double hv, mv;
            if (dir == 0) {
                hv = P[a * planeP + (int64_t)(H - (e + 1)) * ncol + c];
                mv = P[a * planeP + (int64_t)(H + B - (e + 1)) * ncol + cp];
            } else {
                hv = Q[a * planeQ + (int64_t)(B + e + 1) * ncol + cp];
                mv = Q[a * planeQ + (int64_t)(e + 1) * ncol + c];
            }
            return (hv == mv) ? 0.0 : hv - mv;   // also covers -inf vs -inf
        };
        double w0 = 0.0, d0 = 0.0;
        if (on) {
            if (dir == 0) {
                const double m0 = A0[(int64_t)B * ncol + cp];
                d0 = A0[c] - m0;
                w0 = fexp((m0 + B0[(int64_t)(B - 1) * ncol + cp]) - Zc[cp]);
            } else {
                const double m0 = B0[c];
                d0 = B0h[cp] - m0;
                w0 = fexp((A0[(int64_t)1 * ncol + c] + m0) - Zc[c]);
            }
        }
        // phase 1: entry with the largest weight -> frame constant D
        double wb = part == 0 ? w0 : -1.0, db = d0;
        if (on)
            for (int a = 0; a < N; a++)
                for (int e = part; e < ne; e += kChkParts) {
                    const double w = rhoT[wofs(a, e)];
                    if (w > wb) { wb = w; db = diff(a, e); }
                }
        shw[part][lane] = wb; shd[part][lane] = db;
        __syncthreads();
        double D = shd[0][lane], wbest = shw[0][lane];
#pragma unroll
        for (int q = 1; q < kChkParts; q++)
            if (shw[q][lane] > wbest) { wbest = shw[q][lane]; D = shd[q][lane]; }
        __syncthreads();
        // phase 2: posterior-weighted relative mismatch
        double err = part == 0 ? w0 * fabs(fexp(fmin(d0 - D, 700.0)) - 1.0) : 0.0;
        if (on)
            for (int a = 0; a < N; a++)
#pragma unroll 2
                for (int e = part; e < ne; e += kChkParts) {
                    const double w = rhoT[wofs(a, e)];
                    if (w > 0.0) err += w * fabs(fexp(fmin(diff(a, e) - D, 700.0)) - 1.0);
                }
        shw[part][lane] = err;
        __syncthreads();
        if (part == 0 && on) {
            double tot = 0.0;
#pragma unroll
            for (int q = 0; q < kChkParts; q++) tot += shw[q][lane];
            if (!(tot <= tol)) atomicAdd((unsigned long long *)&diag[3 + 2 * dir], 1ull);
            if (tot == tot)
                atomicMax((unsigned long long *)&diag[4 + 2 * dir], (unsigned long long)__double_as_longlong(tot));
        }
        __syncthreads();
    }
}

// LDS-tiled variant of k_gsum: one workgroup = 64 chain columns x 8 consecutive phase groups
// (blockIdx.y selects which 8; 512 threads keep 256 VGPRs per lane), wave w = phase group
// 8*blockIdx.y + w (surplus waves only help staging).
// A tile of TR onset rows of rho (all rings) and the TR + nkb*KB - 1 rows of y they touch are
// staged ONCE in LDS (coalesced global rows -> [row][lane], conflict-free reads) and consumed by
// all phase groups, instead of every phase group streaming them from L2.  Staging goes through
// registers so that all global loads of a tile are issued back to back.
template <int N>
__global__ __launch_bounds__(512) void k_gsum_lds(RingGeom g, int nkb, const double *__restrict__ yT,
                                                  const double *__restrict__ rhoT,
                                                  double *__restrict__ partG)
{
    constexpr int KB = gsum_kb<N>(), TR = 16;
    constexpr int NRH = N * TR * 64 / 512;          // rho elements per thread and tile
    constexpr int NYM = (TR + 8 * KB - 1 + 7) / 8;  // upper bound of y elements per thread and tile
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, kbi = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64;
    const int B = g.B, L = g.L, ncol = g.ncol;
    const int kb0 = blockIdx.y * 8;           // first phase group of this workgroup
    const int nloc = nkb - kb0 < 8 ? nkb - kb0 : 8;
    const int YR = TR + nloc * KB - 1;
    const int yoff = kb0 * KB;                // y rows of this workgroup start at s0 + yoff
    double *lrho = lds;                       // [N][TR][64]
    double *ly = lds + (size_t)N * TR * 64;   // [YR][64]
    const int64_t planeR = (int64_t)B * ncol;
    const int kl = kbi * KB + 1;              // first phase relative to the staged y window
    const int k0 = yoff + kl;
    double g1[N][KB], g2[N][KB], w[KB];
#pragma unroll
    for (int a = 0; a < N; a++)
#pragma unroll
        for (int j = 0; j < KB; j++) { g1[a][j] = 0.0; g2[a][j] = 0.0; }
    for (int s0 = 0; s0 < B; s0 += TR) {
        double tr_[NRH], ty_[NYM];
#pragma unroll
        for (int q = 0; q < NRH; q++) {
            const int i = threadIdx.x + q * 512;
            const int ln = i & 63, rest = i >> 6, u = rest % TR, a = rest / TR;
            tr_[q] = rhoT[a * planeR + (int64_t)(s0 + u) * ncol + c0 + ln];
        }
#pragma unroll
        for (int q = 0; q < NYM; q++) {
            const int i = threadIdx.x + q * 512;
            const int ln = i & 63, r = s0 + yoff + (i >> 6);
            const int cc = c0 + ln;
            const bool ok = i < YR * 64 && cc < g.nch && (int64_t)cc * B + r < g.T;
            const int rr = ok ? r : 0, cq = ok ? cc : 0;
            const int64_t o = (rr < B) ? (int64_t)rr * ncol + cq : (int64_t)(rr - B) * ncol + cq + 1;
            const double v = yT[o];
            ty_[q] = ok ? v : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NRH; q++) lrho[threadIdx.x + q * 512] = tr_[q];
#pragma unroll
        for (int q = 0; q < NYM; q++) {
            const int i = threadIdx.x + q * 512;
            if (i < YR * 64) ly[i] = ty_[q];
        }
        __syncthreads();
        if (kbi >= nloc) continue;  // surplus waves only help staging (wave-uniform)
        // window for onset row u: w[(u + j) % KB] = y[row s0 + u + k0 - 1 + j]
#pragma unroll
        for (int j = 0; j < KB - 1; j++) w[j] = ly[(kl - 1 + j) * 64 + lane];
#pragma unroll
        for (int ub = 0; ub < TR; ub += KB) {
#pragma unroll
            for (int uu = 0; uu < KB; uu++) {
                const int u = ub + uu;
                w[(uu + KB - 1) % KB] = ly[(u + kl + KB - 2) * 64 + lane];
                double rv[N];
#pragma unroll
                for (int a = 0; a < N; a++) rv[a] = lrho[(a * TR + u) * 64 + lane];
#pragma unroll
                for (int j = 0; j < KB; j++) {
                    const double yv = w[(uu + j) % KB];
                    const double y2 = yv * yv;
#pragma unroll
                    for (int a = 0; a < N; a++) {
                        g1[a][j] = __builtin_fma(rv[a], yv, g1[a][j]);
                        g2[a][j] = __builtin_fma(rv[a], y2, g2[a][j]);
                    }
                }
            }
        }
    }
    if (kbi >= nloc) return;
    const int NL = N * L;
    double *out = partG + (size_t)blockIdx.x * 2 * NL;
#pragma unroll
    for (int a = 0; a < N; a++)
#pragma unroll
        for (int j = 0; j < KB; j++) {
            double x1 = g1[a][j], x2 = g2[a][j];
            for (int o = 32; o > 0; o >>= 1) { x1 += __shfl_xor(x1, o); x2 += __shfl_xor(x2, o); }
            const int k = k0 + j;
            if (lane == 0 && k <= L) {
                out[a * L + (k - 1)] = x1;
                out[NL + a * L + (k - 1)] = x2;
            }
        }
}

// Matrix-core variant of k_gsum for many long rings (BASELINE config 5, where the statistics are
// the second largest kernel): the spike-triggered sums are a matrix product over time,
//     G1[k][a] = sum_t y(t+k-1) * rho_a(t),   G2[k][a] = sum_t y(t+k-1)^2 * rho_a(t),
// i.e. C[16 lags x 16 rings] += A[16 lags x 4 samples] * B[4 samples x 16 rings] per
// v_mfma_f64_16x16x4_f64, with A a Toeplitz window of y (lane l: y[t0 + (l>>4) + lag0 + (l&15)])
// and B the posteriors (lane l: rho_{l&15}(t0 + (l>>4)); rings N..15 are zero padding).
// One workgroup = the same 64 chain columns as k_gsum_lds (so the partials land in the same
// layout), swept as 8 groups of 8 adjacent columns; a tile of 32 onset rows of rho and the rows of
// y it touches are staged through LDS ([col][row][ring] resp. [col][row]); wave w owns lag tiles
// w*tpw .. w*tpw+tpw-1 (two accumulator tiles each: y and y^2) for every column and row.
// C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg.
typedef double gs_d4 __attribute__((ext_vector_type(4)));
constexpr int kGmTR = 32, kGmCW = 8;

template <int N, int TPW>
__global__ __launch_bounds__(256) void k_gsum_mfma(RingGeom g, const double *__restrict__ yT,
                                                   const double *__restrict__ rhoT,
                                                   double *__restrict__ partG)
{
    constexpr int TR = kGmTR, CW = kGmCW, RS = TR * 16 + 2;
    extern __shared__ double lds[];
    const int B = g.B, L = g.L, ncol = g.ncol;
    constexpr int tpw = TPW, ntile = 4 * TPW;  // lag tiles per wave / per workgroup, compile time;
                                               // tiles past L compute lags nobody stores
    constexpr int YR = TR + ntile * 16 + 1;    // staged y rows per column (odd stride)
    double *lr = lds;                      // [CW][RS]: rho, ring fastest
    double *ly = lds + CW * RS;            // [CW][YR]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t planeR = (int64_t)B * ncol;
    gs_d4 c1[TPW], c2[TPW];
#pragma unroll
    for (int q = 0; q < TPW; q++) { c1[q] = gs_d4{0.0, 0.0, 0.0, 0.0}; c2[q] = gs_d4{0.0, 0.0, 0.0, 0.0}; }
    const int lk = lane >> 4, li = lane & 15;
    for (int sub = 0; sub < 64 / CW; sub++) {
        const int c0 = blockIdx.x * 64 + sub * CW;
        for (int s0 = 0; s0 < B; s0 += TR) {
            constexpr int NRH = 16 * TR * CW / 256, NYM = (CW * (YR - 1) + 255) / 256;
            double tr_[NRH], ty_[NYM];
#pragma unroll
            for (int k = 0; k < NRH; k++) {
                const int i = tid + k * 256;
                const int cc = i % CW, rest = i / CW, u = rest % TR, a = rest / TR;
                const double v = rhoT[a < N ? a * planeR + (int64_t)(s0 + u) * ncol + c0 + cc : 0];
                tr_[k] = a < N ? v : 0.0;
            }
#pragma unroll
            for (int k = 0; k < NYM; k++) {
                const int i = tid + k * 256;
                const int cc = i % CW, rr = i / CW, r = s0 + rr, col = c0 + cc;
                const bool ok = i < CW * (YR - 1) && col < g.nch && (int64_t)col * B + r < g.T;
                const int rq = ok ? r : 0, cq = ok ? col : 0;
                const int64_t o = (rq < B) ? (int64_t)rq * ncol + cq : (int64_t)(rq - B) * ncol + cq + 1;
                const double v = yT[o];
                ty_[k] = ok ? v : 0.0;
            }
            __syncthreads();  // the previous tile has been consumed
#pragma unroll
            for (int k = 0; k < NRH; k++) {
                const int i = tid + k * 256;
                const int cc = i % CW, rest = i / CW, u = rest % TR, a = rest / TR;
                lr[cc * RS + u * 16 + a] = tr_[k];
            }
#pragma unroll
            for (int k = 0; k < NYM; k++) {
                const int i = tid + k * 256;
                if (i < CW * (YR - 1)) ly[(i % CW) * YR + i / CW] = ty_[k];
            }
            __syncthreads();
            for (int cc = 0; cc < CW; cc++) {
                const double *lrc = lr + cc * RS + lk * 16 + li;
                const double *lyc = ly + cc * YR + lk + li + wv * tpw * 16;
#pragma unroll 2
                for (int ks = 0; ks < TR / 4; ks++) {
                    const double b = lrc[ks * 64];
#pragma unroll
                    for (int q = 0; q < TPW; q++) {
                        const double a = lyc[4 * ks + q * 16];
                        c1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1[q], 0, 0, 0);
                        c2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a * a, b, c2[q], 0, 0, 0);
                    }
                }
            }
        }
    }
    const int NL = N * L;
    double *out = partG + (size_t)blockIdx.x * 2 * NL;
#pragma unroll
    for (int q = 0; q < TPW; q++) {
        const int tile = wv * tpw + q;
        if (li < N) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int lag = tile * 16 + lk + 4 * r;  // k - 1
                if (lag < L) {
                    out[li * L + lag] = c1[q][r];
                    out[NL + li * L + lag] = c2[q][r];
                }
            }
        }
    }
}

// Matrix-core statistics for FEW rings (N <= 8; the headline model has 4): the 16 columns of the
// B operand hold NS = 16/NP copies of the NP (= N rounded up to a power of two) rings, copy s
// delayed by 16*s samples,
//     B[kk][a + NP*s] = rho_a(tau + kk - 16 s),   A[i][kk] = y(tau + kk + i + LPT*q)
//     C[i][a + NP*s] += ...  =  sum over onsets t = tau + kk - 16 s of rho_a(t) * y(t + 16 s + i + LPT*q),
// so one v_mfma_f64_16x16x4_f64 accumulates LPT = 16*NS consecutive lags of every ring with no
// padded column (N = 4, L = 59: all 59 lags of all 4 rings in ONE accumulator tile, 92 % useful).
// tau sweeps rows 0 .. B-1+16(NS-1) of a chain column; rho rows outside [0, B) are staged as zeros.
// Workgroup = the 64 columns of k_gsum_lds (same partial layout) as 8 groups of 8 adjacent columns;
// a tile of TR tau-rows is staged through LDS (rho [col][row][ring], y [col][row]) by all 4 waves,
// wave w then sweeps columns 2w, 2w+1.  The four waves' accumulators are summed through LDS.
constexpr int kGxTR = 64;
template <int N, int NT>
__global__ __launch_bounds__(256) void k_gsum_mx(RingGeom g, const double *__restrict__ yT,
                                                 const double *__restrict__ rhoT,
                                                 double *__restrict__ partG)
{
    constexpr int NP = N <= 1 ? 1 : (N <= 2 ? 2 : (N <= 4 ? 4 : 8));
    constexpr int NS = 16 / NP, LPT = 16 * NS, HS = 16 * (NS - 1);
    constexpr int TR = kGxTR, CW = 8, RR = TR + HS;  // staged rho rows per tile
    extern __shared__ double lds[];
    constexpr int ntile = NT;                      // accumulator tiles of LPT lags, compile time:
                                                   // a run-time count makes the compiler shuffle
                                                   // all tiles through AGPRs around every MFMA
    const int B = g.B, L = g.L, ncol = g.ncol;
    constexpr int YR = TR + 19 + LPT * (NT - 1);   // staged y rows per column (odd)
    constexpr int RS = RR * NP + 2;                // rho column stride
    double *lr = lds;                              // [CW][RS]
    double *ly = lds + CW * RS;                    // [CW][YR]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lk = lane >> 4, lj = lane & 15, la = lj % NP, lsft = lj / NP;
    const int64_t planeR = (int64_t)B * ncol;
    gs_d4 c1[NT], c2[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) { c1[q] = gs_d4{0.0, 0.0, 0.0, 0.0}; c2[q] = gs_d4{0.0, 0.0, 0.0, 0.0}; }
    for (int sub = 0; sub < 64 / CW; sub++) {
        const int c0 = blockIdx.x * 64 + sub * CW;
        for (int s0 = 0; s0 < B + HS; s0 += TR) {
            // all global loads of the tile are issued back to back into registers, then stored
            constexpr int NRH = (NP * RR * CW + 255) / 256, NYM = (CW * (YR - 1) + 255) / 256;
            double tr_[NRH], ty_[NYM];
#pragma unroll
            for (int k = 0; k < NRH; k++) {   // rho rows s0-HS .. s0+TR-1
                const int i = tid + k * 256;
                const int cc = i % CW, rest = i / CW, u = rest % RR, a = rest / RR;
                const int row = s0 - HS + u;
                const bool ok = i < NP * RR * CW && a < N && row >= 0 && row < B;
                const double v = rhoT[ok ? a * planeR + (int64_t)row * ncol + c0 + cc : 0];
                tr_[k] = ok ? v : 0.0;
            }
#pragma unroll
            for (int k = 0; k < NYM; k++) {   // y rows s0 .. s0+YR-2
                const int i = tid + k * 256;
                const int cc = i % CW, rr = i / CW, r = s0 + rr, col = c0 + cc;
                const bool ok = i < CW * (YR - 1) && col < g.nch && (int64_t)col * B + r < g.T;
                const int rq = ok ? r : 0, cq = ok ? col : 0;
                const int64_t o = (rq < B) ? (int64_t)rq * ncol + cq
                                           : (rq < 2 * B ? (int64_t)(rq - B) * ncol + cq + 1
                                                         : (int64_t)(rq - 2 * B) * ncol + cq + 2);
                const double v = yT[o];
                ty_[k] = ok ? v : 0.0;
            }
            __syncthreads();  // the previous tile has been consumed
#pragma unroll
            for (int k = 0; k < NRH; k++) {
                const int i = tid + k * 256;
                const int cc = i % CW, rest = i / CW, u = rest % RR, a = rest / RR;
                if (i < NP * RR * CW) lr[cc * RS + u * NP + a] = tr_[k];
            }
#pragma unroll
            for (int k = 0; k < NYM; k++) {
                const int i = tid + k * 256;
                if (i < CW * (YR - 1)) ly[(i % CW) * YR + i / CW] = ty_[k];
            }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < CW / 4; h++) {
                const int cc = wv * (CW / 4) + h;
                const double *lrc = lr + cc * RS + (lk - 16 * lsft + HS) * NP + la;
                const double *lyc = ly + cc * YR + lk + lj;
#pragma unroll 4
                for (int ts = 0; ts < TR / 4; ts++) {
                    const double b = lrc[ts * 4 * NP];
#pragma unroll
                    for (int q = 0; q < NT; q++) {
                        const double a = lyc[ts * 4 + q * LPT];
                        c1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1[q], 0, 0, 0);
                        c2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a * a, b, c2[q], 0, 0, 0);
                    }
                }
            }
        }
    }
    // sum the four waves' tiles: LDS [wave][ntile][2][4 regs][64 lanes]
    __syncthreads();
    double *red = lds;
#pragma unroll
    for (int q = 0; q < NT; q++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            red[(((wv * ntile + q) * 2 + 0) * 4 + r) * 64 + lane] = c1[q][r];
            red[(((wv * ntile + q) * 2 + 1) * 4 + r) * 64 + lane] = c2[q][r];
        }
    }
    __syncthreads();
    const int NL = N * L;
    double *out = partG + (size_t)blockIdx.x * 2 * NL;
    for (int e = tid; e < ntile * 2 * 4 * 64; e += 256) {
        const int ln = e & 63, r = (e >> 6) & 3, which = (e >> 8) & 1, q = e >> 9;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 4; w++) v += red[(((w * ntile + q) * 2 + which) * 4 + r) * 64 + ln];
        const int j = ln & 15, a = j % NP, sft = j / NP;
        const int lag = q * LPT + 16 * sft + (ln >> 4) + 4 * r;
        if (a < N && lag < L) out[which * NL + a * L + lag] = v;
    }
}

// virtual onsets t' = -j (rings already running at the first sample), the end-of-data correction
// of G0 and pp = gamma[:,1] (baumwelch.jl:263).  One thread per ring state.
//   extra[0..NL)    = G0 contribution of virtual onsets  +  sum of rho over the onsets of the last
//                     L-1 samples that still reach phase k (k_post leaves those out of its total)
//   extra[NL..2NL)  = G1 of virtual onsets,  extra[2NL..3NL) = G2 of virtual onsets
__global__ __launch_bounds__(64) void k_stats_edges(RingGeom g, const double *__restrict__ y,
                                                     const double *__restrict__ P,
                                                     const double *__restrict__ Q,
                                                     const double *__restrict__ A0,
                                                     const double *__restrict__ B0,
                                                     const double *__restrict__ Zc,
                                                     const double *__restrict__ rhoT,
                                                     double *__restrict__ extra,
                                                     double *__restrict__ pp)
{
    const int L = g.L, N = g.N, ncol = g.ncol, NL = N * L;
    const int64_t planeP = (int64_t)(g.H + g.B) * ncol, planeQ = (int64_t)(L + g.B + g.H) * ncol,
                  planeR = (int64_t)g.B * ncol;
    const double z = Zc[0];
    for (int pair = blockIdx.x * blockDim.x + threadIdx.x; pair < NL; pair += gridDim.x * blockDim.x) {
        const int a = pair / L, k = pair % L + 1;
        double g0 = 0.0, g1 = 0.0, g2 = 0.0;
        for (int j = 1; j <= L - 1 && g.first; j++) {  // virtual onsets exist at the recording start only
            const int idx = -j + k - 1;
            if (idx < 0) continue;
            const double rv = fexp((P[a * planeP + (int64_t)(g.H - j) * ncol] +
                                   Q[a * planeQ + (int64_t)(L - j) * ncol]) - z);
            const double yv = y[idx];
            g0 += rv; g1 += rv * yv; g2 += rv * (yv * yv);
        }
        // truncated rings at the end of the recording: onsets t' in (T-L, T-k] still cover phase k
        double tail = 0.0;
        if (g.last)
            for (int64_t t = g.T - L + 1; t <= g.T - k; t++)
                if (t >= 0) tail += rhoT[a * planeR + (t % g.B) * ncol + (t / g.B)];
        extra[pair] = g0 + tail;
        extra[NL + pair] = g1;
        extra[2 * NL + pair] = g2;
        const int sp = 1 - k;  // pp for state (a,k): the onset at t' = 1-k
        pp[1 + pair] = (P[a * planeP + (int64_t)(g.H + sp) * ncol] +
                        Q[a * planeQ + (int64_t)(L + sp) * ncol]) - z;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) pp[0] = (A0[(int64_t)1 * ncol] + B0[0]) - z;
}

// deterministic final assembly: stats = [G0 | G1 | G2 | Xi | s_all | s_m | s_y2 | 0];
// one wave per output element, fixed summation order
__global__ __launch_bounds__(64) void k_stats_final(int N, int L, int rowsG, int rowsS,
                                                    const double *__restrict__ partG,
                                                    const double *__restrict__ partS,
                                                    const double *__restrict__ extra,
                                                    double *__restrict__ stats)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    const int NL = N * L, ws = 2 * N + 3;
    double acc = 0.0;
    if (i < NL) {                       // G0(a,k) = sum_t' rho_a(t') + edge corrections
        const int a = i / L;
        for (int r = lane; r < rowsS; r += 64) acc += partS[(size_t)r * ws + N + a];
    } else if (i < 3 * NL) {            // G1, G2
        const int e = i - NL;
        for (int r = lane; r < rowsG; r += 64) acc += partG[(size_t)r * 2 * NL + e];
    } else if (i < 3 * NL + N) {        // Xi
        for (int r = lane; r < rowsS; r += 64) acc += partS[(size_t)r * ws + (i - 3 * NL)];
    } else if (i < 3 * NL + N + 3) {    // s_all, s_m, s_y2
        for (int r = lane; r < rowsS; r += 64) acc += partS[(size_t)r * ws + 2 * N + (i - 3 * NL - N)];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) stats[i] = acc + (i < 3 * NL ? extra[i] : 0.0);
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
        out[K * N + 1 + a] = flog(Xi[a]) - flog(s_m);      // :264 xb[2:end]
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

int ring_estep_launch(RingDev *r, const double *d_y, double *d_stats, hipStream_t st)
{
    const RingGeom &g = r->g;
    int rc;
    HS_HIP(hipMemsetAsync(r->diag, 0, 8 * sizeof(int64_t), st));
    if ((rc = ring_prepare(r, d_y, st))) return rc;
    if ((rc = ring_launch_virtual(r, d_y, r->P, (int64_t)(g.H + g.B) * g.ncol, st))) return rc;
    const int colgroups = g.ncol / 64;
    rc = dispatch_N(g.N, [&](auto n) {
        constexpr int NN = decltype(n)::value;
        EParams<NN> ep = make_eparams<NN>(r);
        { PROF(r, "k_fb_chain", st); hipLaunchKernelGGL((k_fb_chain<NN>), dim3(2 * colgroups), dim3(64), 0, st, g, ep, r->yT, r->Rf,
                           r->P, r->A0, r->Q, r->B0, r->B0h); }
        HS_HIP(hipGetLastError());
        return HMMSORT_OK;
    });
    if (rc) return rc;
    return ring_estep_post(r, d_y, d_stats, st);
}

// everything after the forward/backward sweeps: normaliser, posteriors, statistics, certificate
int ring_estep_post(RingDev *r, const double *d_y, double *d_stats, hipStream_t st)
{
    const RingGeom &g = r->g;
    const int N = g.N, L = g.L, NL = N * L;
    int rc;
    const int colgroups = g.ncol / 64;
    rc = dispatch_N(N, [&](auto n) {
        constexpr int NN = decltype(n)::value;
        constexpr int KB = gsum_kb<NN>();
        JParams<NN> jp = make_jparams<NN>(r);

        { PROF(r, "k_znorm", st); hipLaunchKernelGGL((k_znorm<NN>), dim3(colgroups, kZParts), dim3(64), 0, st, g, r->P, r->Q,
                           r->Zp); }
        { PROF(r, "k_post", st); hipLaunchKernelGGL((k_post<NN>), dim3(colgroups, (g.B + kPostRows - 1) / kPostRows), dim3(64), 0, st, g, jp, r->yT, r->Rf, r->P,
                           r->Q, r->A0, r->B0, r->Zp, r->Zc, r->rhoT, r->partS); }
        // the boundary certificate only needs the posteriors: it runs on the plan's internal
        // stream beside the statistics kernels and is joined at the end of this function
        HS_HIP(hipEventRecord(r->ev_post, st));
        HS_HIP(hipStreamWaitEvent(r->side, r->ev_post, 0));
        // the edge terms (virtual onsets, end-of-data tails, pp: one thread per ring state, latency
        // bound) need the posteriors only as well: first on the side stream, then the certificate
        { PROF(r, "k_stats_edges", r->side); hipLaunchKernelGGL(k_stats_edges, dim3((NL + 63) / 64), dim3(64), 0, r->side, g, d_y, r->P, r->Q, r->A0, r->B0,
                           r->Zc, r->rhoT, r->extra, r->pp); }
        HS_HIP(hipEventRecord(r->ev_edges, r->side));
        { PROF(r, "k_fb_check", r->side); hipLaunchKernelGGL((k_fb_check<NN>), dim3(colgroups), dim3(64 * kChkParts), 0, r->side, g, 1e-9, r->P, r->Q,
                           r->A0, r->B0, r->B0h, r->Zc, r->rhoT, r->diag); }
        HS_HIP(hipEventRecord(r->ev_chk, r->side));
        const int ntile = (L + 15) / 16;
        constexpr int NPx = NN <= 1 ? 1 : (NN <= 2 ? 2 : (NN <= 4 ? 4 : 8));
        constexpr int LPTx = 16 * (16 / NPx), HSx = LPTx - 16;
        const int ntx = (L + LPTx - 1) / LPTx;
        // measured at 10 M samples against k_gsum_lds: N=4,L=59 0.39 vs 0.55 ms; N=2,L=59 0.41 vs 0.22;
        // N=8,L=127 1.80 vs 1.56 -> used for 3-4 rings (the headline model)
        if (NPx == 4 && ntx <= 4 && g.B % 64 == 0 && 2 * g.B >= kGxTR + 19 + LPTx * ntx + HSx) {
            const size_t l1 = ((size_t)8 * ((kGxTR + HSx) * NPx + 2) + (size_t)8 * (kGxTR + 19 + LPTx * (ntx - 1))) * 8;
            const size_t l2 = (size_t)4 * ntx * 2 * 4 * 64 * 8;
            const size_t lds = l1 > l2 ? l1 : l2;
            constexpr int NM = NN <= 8 ? NN : 8;
            auto go = [&](auto kern) -> int {
                if (lds > 64 * 1024)
                    HS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds));
                PROF(r, "k_gsum", st);
                hipLaunchKernelGGL(kern, dim3(colgroups), dim3(256), lds, st, g, r->yT, r->rhoT, r->partA);
                return HMMSORT_OK;
            };
            int rc2 = ntx == 1 ? go(k_gsum_mx<NM, 1>) : ntx == 2 ? go(k_gsum_mx<NM, 2>)
                    : ntx == 3 ? go(k_gsum_mx<NM, 3>) : go(k_gsum_mx<NM, 4>);
            if (rc2) return rc2;
        } else
        // fp64 MFMA issues at the vector-FMA rate on gfx950, so the matrix-core kernel (rings padded
        // to 16 columns) only wins where the vector kernel has to re-read rho for many phase groups:
        // measured at 10-40 M samples: N=16,L=255 21 vs 51 ms; N=12,L=127 4.6 vs 3.0; N=8,L=127 4.3 vs 1.6
        if (NN >= 9 && L > 160 && ntile <= 16 && g.B % kGmTR == 0 && g.B >= kGmTR + ((ntile + 3) / 4) * 64) {
            const int tpw = (ntile + 3) / 4;
            const size_t lds = ((size_t)kGmCW * (kGmTR * 16 + 2) + (size_t)kGmCW * (kGmTR + 4 * tpw * 16 + 1)) *
                               sizeof(double);
            PROF(r, "k_gsum", st);
            if (tpw == 1) hipLaunchKernelGGL((k_gsum_mfma<NN, 1>), dim3(colgroups), dim3(256), lds, st, g, r->yT, r->rhoT, r->partA);
            else if (tpw == 2) hipLaunchKernelGGL((k_gsum_mfma<NN, 2>), dim3(colgroups), dim3(256), lds, st, g, r->yT, r->rhoT, r->partA);
            else if (tpw == 3) hipLaunchKernelGGL((k_gsum_mfma<NN, 3>), dim3(colgroups), dim3(256), lds, st, g, r->yT, r->rhoT, r->partA);
            else hipLaunchKernelGGL((k_gsum_mfma<NN, 4>), dim3(colgroups), dim3(256), lds, st, g, r->yT, r->rhoT, r->partA);
        } else {
            const int nkb = (L + KB - 1) / KB;
            const size_t lds = ((size_t)NN * 16 * 64 + (size_t)(16 + 8 * KB - 1) * 64) * sizeof(double);
            if (NN * 16 * 64 % 512 == 0 && lds <= 150 * 1024 && g.B % 16 == 0 && g.B >= L + 16 + 8 * KB) {
                if (lds > 64 * 1024)
                    HS_HIP(hipFuncSetAttribute((const void *)k_gsum_lds<NN>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                PROF(r, "k_gsum", st);
                hipLaunchKernelGGL((k_gsum_lds<NN>), dim3(colgroups, (nkb + 7) / 8), dim3(512), lds, st, g, nkb, r->yT,
                                   r->rhoT, r->partA);
            } else {
                { PROF(r, "k_gsum", st); hipLaunchKernelGGL((k_gsum<NN>), dim3(((colgroups + 7) / 8) * 8 * ((L + KB - 1) / KB)), dim3(64), 0, st, g, r->yT,
                           r->rhoT, r->partA); }
            }
        }

        HS_HIP(hipGetLastError());
        return HMMSORT_OK;
    });
    if (rc) return rc;
    HS_HIP(hipStreamWaitEvent(st, r->ev_edges, 0));
    const int total = 3 * NL + N + 4;
    { PROF(r, "k_stats_final", st); hipLaunchKernelGGL(k_stats_final, dim3(total), dim3(64), 0, st, N, L, colgroups,
                       colgroups * ((g.B + kPostRows - 1) / kPostRows), r->partA, r->partS, r->extra, d_stats); }
    HS_HIP(hipGetLastError());
    HS_HIP(hipStreamWaitEvent(st, r->ev_chk, 0));
    return HMMSORT_OK;
}

int ring_mstep_launch(RingDev *r, const double *d_stats, double *d_out, hipStream_t st)
{
    { PROF(r, "k_mstep", st); hipLaunchKernelGGL(k_mstep, dim3(1), dim3(256), 0, st, r->g.N, r->g.L, d_stats, r->pp, d_out); }
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

}  // namespace hmmsort
