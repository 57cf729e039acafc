// Wave engine, part 3: one Baum-Welch step = forward -> backward -> update
// (reference src/baumwelch.jl:25-51, :73-98, :205-309, :362-370) without ever materialising alpha,
// beta, gamma or xi, one wavefront per chain.  Specification: tests/wave_model.py (fwd_chain,
// bwd_chain, estep), checked against the CPU oracle.
//
// Scaled representation (no logarithm inside the sweeps, N+1 exponentials per sample):
//   exp(la0(t)) = x_t exp(M_t)          silent forward value; M_t = max-plus envelope (lane scan),
//                                       x_t = al_t x_{t-1} + be_t (linear lane scan)
//   onset mass of ring a at t:          lp_a(t) = fref_t + sc_a + log fv_a(t) + R_a(t)
//   exp(lb0(t)) = xb_t exp(Mb_t)        silent backward value
//   beta of ring a's last state at t:   Yn_a(t) = Mb_t + log yn_a(t)
// sc_a = scale of the entry transitions into ring a (floored at -700): entry probabilities far below
// the double range stay in exponents, so the re-estimated lp of a vanishing template is finite as in
// the reference's log-domain folds (baumwelch.jl:254-261).
// The backward sweep is fused with update(): at step t it has the posteriors
//   gamma_t(silent) = xb_t g,                      g  = exp(la0(t) + Mb_t - z)
//   xi_a(t+1)       = CP0_a w_a g                  (silent -> ring a), w_a = yn-weighted start term
//   rho_a(t+1)      = fv_a(t+1) w_a g2,            g2 = exp(fref_{t+1} + Mb_t - z)
// (rho_a(t') = posterior that ring a started at t' = gamma of all L states of that ring pass), z =
// the chain's normaliser log sum_j alpha(j) beta(j), evaluated once where the sweep enters the
// chain's own samples.  The M-step sums are posterior-weighted spike-triggered sums of y (kw_gsum*).
#include <cmath>
#include <type_traits>

#include "fastmath.h"
#include "wave_common.h"

namespace hmmsort {

constexpr double kLn2 = 6.93147180559945286227e-01;
typedef double wg_d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double scale_of(double v, double s)  // s + log2-exponent of v, -inf for v <= 0
{
    return v > 0.0 ? s + (double)ilogb(v) * kLn2 : -INFINITY;
}

template <int N>
struct FIn {
    double y;
    double R[N];
};

// sum over b != a of t_b * c[b] for every a, O(N): prefix/suffix sums (no subtraction, no cancellation)
template <int N>
__device__ __forceinline__ void excl_sums(const double (&t)[N], double (&out)[N])
{
    double pre = 0.0;
#pragma unroll
    for (int a = 0; a < N; a++) { out[a] = pre; pre += t[a]; }
    double suf = 0.0;
#pragma unroll
    for (int a = N - 1; a >= 0; a--) { out[a] += suf; suf += t[a]; }
}

// ------------------------------------------------------------------------------------------
// forward chains (baumwelch.jl:25-51)
// LDS: constants sc0 | mean0 | den | P00 | einv | sc[N] | CP0[N] | PEND[N] | CPXT[N*N] (UC: CPXin[N]),
// then the delay line (v_a, s_a) of the last L + W onsets.
// ------------------------------------------------------------------------------------------
template <int N, bool UC>
__global__ __launch_bounds__(64, N <= 4 ? 3 : (N <= 8 ? 2 : 1)) void kw_fwd(WaveGeom g, const WaveConst *__restrict__ cst,
                                                            const double *__restrict__ y,
                                                            const double *__restrict__ Rf,
                                                            const double *__restrict__ virt,
                                                            double *__restrict__ FA0, double *__restrict__ FV,
                                                            double *__restrict__ FREF, double *__restrict__ fpre,
                                                            double *__restrict__ trash)
{
    constexpr int D = wave_depth<N>();
    constexpr int KSC = 5, KCP0 = 5 + N, KPEND = 5 + 2 * N, KCPX = 5 + 3 * N, KSIZE = 5 + 3 * N + N * N;
    extern __shared__ double lds[];
    const int L = g.L, W = g.W, RB = g.RB, B = g.B;
    // onset t': (v_a, s_a), X_a = s_a + log v_a; row stride RB+1 (spare slot).  LOGDL (more than 8 rings: two lines
    // per ring would leave one wave per CU at L = 255): ONE line per ring holding X_a itself -- one logarithm per
    // ring and sample when the entry is stored, the exponential at the exit is the one the sweep takes anyway
    constexpr bool LOGDL = N > 8;
    double *KC = lds, *DLv = lds + KSIZE, *DLs = LOGDL ? DLv : lds + KSIZE + N * (RB + 1);
    const int lane = threadIdx.x;
    const int cg = blockIdx.x, ch = cg / g.nch, c = cg % g.nch;
    const int64_t T = g.T;
    const int64_t tc = (int64_t)c * B;
    const int nc = (int)((T - tc) < B ? (T - tc) : B);
    const int64_t tend = tc + nc;
    const double *yc = y + (int64_t)ch * T;
    const double *Rc = Rf + (int64_t)ch * N * T;
    double *FAc = FA0 + (int64_t)ch * T, *FRc = FREF + (int64_t)ch * T, *FVc = FV + (int64_t)ch * N * T;
    const int64_t FR = 1 + (int64_t)L * (N + 1);
    double *rec = fpre + cg * FR;

    if (LOGDL) for (int i = lane; i < N * (RB + 1); i += 64) DLv[i] = -INFINITY;
    else for (int i = lane; i < 2 * N * (RB + 1); i += 64) DLv[i] = 0.0;
    {
        const WaveConst &Kg = cst[ch];
        if (lane == 0) { KC[0] = Kg.sc0; KC[1] = Kg.mean0; KC[2] = 1.0 / Kg.den; KC[3] = Kg.P00; KC[4] = fexp(-Kg.sc0); }
        if (lane < N) { KC[KSC + lane] = Kg.sc[lane]; KC[KCP0 + lane] = Kg.CP0[lane]; KC[KPEND + lane] = Kg.PEND[lane]; }
        if (UC) { if (lane < N) KC[KCPX + lane] = Kg.CPXin[lane]; }
        else for (int i = lane; i < N * N; i += 64) KC[KCPX + i] = Kg.CPXT[i];
    }
    __syncthreads();
    int64_t tinit;
    double M, x = 1.0;
    if (c == 0) {  // baumwelch.jl:36: first column = emission only, every state
        tinit = 0;
        for (int i = lane; i < N * L; i += 64) {
            const int a = i / L, j = i % L + 1;
            if (j < L) {
                if (!LOGDL) DLv[a * (RB + 1) + (L - j)] = 1.0;
                DLs[a * (RB + 1) + (L - j)] = virt[((int64_t)ch * N + a) * (L + 1) + j];
            }
        }
        const double d0 = yc[0] - KC[1];
        const double q00 = -((d0 * d0) * KC[2]);
        M = (double)(float)q00;                           // float-representable scale, remainder in x
        x = fexp(q00 - M);
        if (lane < N) {
            const double f0 = fexp(-KC[KSC + lane]);
            if (LOGDL) DLs[lane * (RB + 1) + L] = Rc[(int64_t)lane * T];   // sc + R + log(exp(-sc))
            else {
                DLv[lane * (RB + 1) + L] = f0;
                DLs[lane * (RB + 1) + L] = KC[KSC + lane] + Rc[(int64_t)lane * T];
            }
            FVc[(int64_t)lane * T] = f0;
        }
        if (lane == 0) { FAc[0] = q00; FRc[0] = 0.0; }
    } else {
        tinit = tc - g.Hw;
        M = 0.0;
    }
    __syncthreads();

    const int n_total = (int)(tend - 1 - tinit);
    auto load = [&](FIn<N> &d, int off) {
        const int nact = n_total - off < W ? n_total - off : W;
        const int li = lane < nact ? lane : 0;
        int64_t tb = tinit + 1 + off;
        tb = tb < T ? tb : T - 1;
        d.y = (yc + tb)[li];
#pragma unroll
        for (int a = 0; a < N; a++) d.R[a] = (Rc + (int64_t)a * T + tb)[li];
    };
    int rs = (1 + lane) % RB, ws = (L + 1 + lane) % RB;
    // MODE 0: warm-up super-step, no global stores; 1: owned super-step, every lane stores (idle lanes of
    // the last partial step into a trash line); 2: the last super-steps of the warm-up, which also leave
    // the boundary record (conditional stores).  Modes 0 and 1 have straight-line global accesses only,
    // so hipcc counts vmcnt exactly and the input pipeline stays D super-steps deep.
    auto run = [&](const FIn<N> &d, int off, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const int nact = n_total - off < W ? n_total - off : W;
        const bool live = lane < nact;
        const int64_t tb = tinit + 1 + off;
        const int64_t t = tb + lane;
        const double sc0 = KC[0];
        double v[N], E[N + 1];
        double e = -INFINITY;
#pragma unroll
        for (int a = 0; a < N; a++) {
            if (LOGDL) {
                v[a] = 1.0;
                E[a] = live ? DLs[a * (RB + 1) + rs] + sc0 : -INFINITY;   // log of the exit of ring a
                e = fmax(e, E[a]);
            } else {
                v[a] = live ? DLv[a * (RB + 1) + rs] : 0.0;
                E[a] = DLs[a * (RB + 1) + rs] + sc0;            // scale of the exit of ring a
                e = fmax(e, scale_of(v[a], E[a]));
            }
        }
        const double dd = d.y - KC[1];
        const double q0 = -((dd * dd) * KC[2]);           // KC[2] = 1/den: 1 ulp from the reference's division (bar here: 1e-6)
        float fa = live ? (float)(sc0 + q0) : 0.0f;
        float fb = live ? (float)(e + q0) : -INFINITY;
        scan_maxplus_f32(fa, fb);                         // the envelope is a scale: single precision
        const float Mf = (float)M;                        // exact: the carry is kept float-representable
        const float Mtf = fmaxf(Mf + fa, fb);
        const double Mt = (double)Mtf;
        const double Mprev = (double)lane_prevf(Mtf, Mf);  // cross-lane moves must run with all lanes enabled
        const double ref = Mt - q0;
        E[N] = (Mprev + sc0) - ref;
#pragma unroll
        for (int a = 0; a < N; a++) E[a] = v[a] > 0.0 ? fmin(E[a] - ref, 700.0) : -INFINITY;  // 0 * exp(big) = NaN; -inf stays
        fexp_n<N + 1>(E);
        const double E0 = E[N];
        double Ea[N];
        double al = E0 * KC[3], be = 0.0;
#pragma unroll
        for (int a = 0; a < N; a++) {
            Ea[a] = LOGDL ? E[a] : v[a] * E[a];
            be = __builtin_fma(Ea[a], KC[KPEND + a], be);
        }
        al = live ? al : 1.0;
        be = live ? be : 0.0;
        scan_linear(al, be);
        const double xt = __builtin_fma(al, x, be);
        const double xprev = lane_prev(xt, x);
        const double einv = KC[4];
        const double base = (xprev * E0) * einv;          // exp(la0(t-1) - ref)
        double u[N];
        if (UC) {
            double tb_[N], ex[N];
#pragma unroll
            for (int a = 0; a < N; a++) tb_[a] = Ea[a] * einv;
            excl_sums<N>(tb_, ex);
#pragma unroll
            for (int a = 0; a < N; a++) u[a] = __builtin_fma(ex[a], KC[KCPX + a], base * KC[KCP0 + a]);
        } else {
#pragma unroll
            for (int a = 0; a < N; a++) {
                double su = base * KC[KCP0 + a];
#pragma unroll
                for (int b = 0; b < N; b++)
                    if (b != a) su = __builtin_fma(Ea[b] * einv, KC[KCPX + a * N + b], su);
                u[a] = su;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int wsl = live ? ws : RB;   // idle lanes write the spare slot behind the ring
        if (LOGDL) {
            double lu[N];
#pragma unroll
            for (int a = 0; a < N; a++) lu[a] = u[a];
            flog_n<N>(lu);
#pragma unroll
            for (int a = 0; a < N; a++) DLs[a * (RB + 1) + wsl] = ((ref + KC[KSC + a]) + d.R[a]) + (u[a] > 0.0 ? lu[a] : -INFINITY);
        } else {
#pragma unroll
            for (int a = 0; a < N; a++) {
                DLv[a * (RB + 1) + wsl] = u[a];
                DLs[a * (RB + 1) + wsl] = (ref + KC[KSC + a]) + d.R[a];
            }
        }
        if (MODE == 1) {
            const double la0 = Mt + flog(xt);
            *(live ? FAc + tb + lane : trash + lane) = la0;
            *(live ? FRc + tb + lane : trash + 64 + lane) = ref;
#pragma unroll
            for (int a = 0; a < N; a++) *(live ? FVc + (int64_t)a * T + tb + lane : trash + 64 * (2 + a) + lane) = u[a];
        }
        if (MODE == 2) {
            if (live && t >= tc - L && t < tc) {  // warm-up copy of the boundary state (certificate)
                const int64_t jj = t - (tc - L);
#pragma unroll
                for (int a = 0; a < N; a++) rec[1 + jj * (N + 1) + a] = u[a];
                rec[1 + jj * (N + 1) + N] = ref;
                if (t == tc - 1) rec[0] = Mt + flog(xt);
            }
        }
        // carry from the last LIVE lane: an idle lane's prefix is the same sum in another association, and
        // (x, M) must be the pair of one lane
        x = wave_bcast(xt, nact - 1);
        M = wave_bcast(Mt, nact - 1);
        const int ee = x > 0.0 ? ilogb(x) : 0;
        if (ee >= 32 || ee <= -32) {   // renormalise the carry: x exp(M) unchanged, M stays a float value
            const double Mx = M + (double)ee * kLn2, Mn = (double)(float)Mx;
            x = ldexp(x, -ee) * fexp(Mx - Mn);
            M = Mn;
        }
        rs += W; rs = rs >= RB ? rs - RB : rs;
        ws += W; ws = ws >= RB ? ws - RB : ws;
    };
    // The main loop is branch-free (whole groups of D super-steps, loads clamped past the end), so that
    // hipcc can count vmcnt across the back edge; the last < D super-steps run from the buffers the main
    // loop has already filled.
    auto sweep = [&](int from, int to, auto mode_tag) {
        FIn<N> buf[D];
#pragma unroll
        for (int i = 0; i < D; i++) load(buf[i], from + i * W);
        int off = from;
        for (; off + D * W <= to; off += D * W) {
#pragma unroll
            for (int i = 0; i < D; i++) {
                run(buf[i], off + i * W, mode_tag);
                load(buf[i], off + (i + D) * W);
            }
        }
#pragma unroll
        for (int i = 0; i < D; i++)
            if (off + i * W < to) run(buf[i], off + i * W, mode_tag);
    };
    const int n_warm = c == 0 ? 0 : g.Hw - 1;                     // multiple of W
    const int n_rec = ((L + W - 1) / W) * W < n_warm ? ((L + W - 1) / W) * W : n_warm;
    if (n_warm - n_rec > 0) sweep(0, n_warm - n_rec, std::integral_constant<int, 0>());
    if (n_rec > 0) sweep(n_warm - n_rec, n_warm, std::integral_constant<int, 2>());
    sweep(n_warm, n_total, std::integral_constant<int, 1>());
}

// ------------------------------------------------------------------------------------------
// backward chains (baumwelch.jl:73-98) + posteriors of update() (:205-309)
// Chain c starts at te-1, te = min(tend + He, T), with beta = 0 for every state (the reference's
// terminal condition when te is the end of the data, an arbitrary warm-up start otherwise), sweeps
// down to tstar = tend-1 (warm-up), fixes the normaliser z there, and sweeps its own samples down to
// tc-1 (the step at tc-1 yields the posteriors of the onsets at tc).
// Delay line: Yn(tau) at slot (te-1-tau) mod RB; slots never written hold (1, 0) = "beta = 0".
// LDS: constants sc0 | mean0 | den | P00 | sc[N] | CP0[N] | PEND[N] | CPX[N*N] (UC: CPXin[N]).
// ------------------------------------------------------------------------------------------
template <int N>
struct BIn {
    double y1, y0;    // y(t+1), y(t)
    double w2;        // W2(t+1)
    double R[N];      // R_a(t+1)
    double fv[N];     // fv_a(t+1)
    double fref, la0; // fref(t+1), la0(t)
};

// y ring of the fused statistics: the product reads back to the largest lag (+ 18) behind tau, tau trails the
// write head by up to 72 steps and a step writes 64 entries ahead: 64 lags -> 202 entries, 128 lags -> 266
constexpr int bwd_yring(int ft) { return ft <= 2 ? 256 : 512; }

// FT > 0 (fused statistics; 3-4 rings of at most 64 states: FT = 1; 5-8 rings of at most 64 / 128 states: FT = 2 / 4):
// the spike-triggered sums G1[a][lag] = sum_t' rho_a(t') y(t' + lag) are accumulated by the sweep itself on the
// matrix cores, so rho is not read back by a statistics kernel.  The sweep runs backward in time; with
// sigma = step index (rising while t' falls) the product pairs rho(sigma) with y(sigma - lag), i.e. with samples
// the sweep has already seen.  Per v_mfma_f64_16x16x4_f64 and accumulator tile q:
//   A_q[i][kk] = Y(tau + kk - D_q - i),  B[kk][a + NP s'] = rho_a(tau + kk - 16 s')   =>  lag = D_q - 16 s' + i,
// NS = 16 / NP delayed copies of the NP (4 or 8) ring columns fill the 16 columns, D_q = 16 NS q + 16 (NS - 1):
// tile q holds the 16 NS lags from 16 NS q of every ring, FT tiles hold them all (B is read once per FT MFMAs).
// rho and y of the last few hundred steps live in two LDS rings of the wave; a super-step of W steps feeds
// FT W/4 MFMAs (the remainder waits for the next one), and after the last step the copies are drained with zeros.
#ifndef HS_BWD_W4
#define HS_BWD_W4 2   // waves per SIMD asked of the compiler for up to 4 rings (3, with a one-deep input pipeline and chains of 3 264 samples: 18 spilled registers, 0.382 against 0.369 ms)
#endif
template <int N, bool UC, int FT = 0>
__global__ __launch_bounds__(64, N <= 4 ? HS_BWD_W4 : 1) void kw_bwd(WaveGeom g, const WaveConst *__restrict__ cst,
                                                            const double *__restrict__ y,
                                                            const double *__restrict__ Rf,
                                                            const double *__restrict__ FA0,
                                                            const double *__restrict__ FV,
                                                            const double *__restrict__ FREF,
                                                            const double *__restrict__ fpre,
                                                            const double *__restrict__ W2,
                                                            double *__restrict__ rho, double *__restrict__ partS,
                                                            double *__restrict__ Zc, double *__restrict__ bpre,
                                                            double *__restrict__ bown, double *__restrict__ yhead,
                                                            double *__restrict__ trash, double *__restrict__ partG)
{
    constexpr bool FUSE = FT > 0;
    static_assert(!FUSE || (N >= 3 && N <= 8), "the fused statistics are written for delayed copies of 3-8 rings");
    static_assert(!FUSE || N > 4 || FT == 1, "3-4 rings: one tile of 64 lags");
    constexpr int NPc = N <= 4 ? 4 : 8, NSc = 16 / NPc, LPT = 16 * NSc;   // ring columns per copy, copies, lags per tile
    constexpr int RPAD = NPc == 4 ? 1 : 2;                   // padding rows per 16 rows of the rho ring (bank spread of the copies)
    constexpr int RGROWS = 128 + 8 * RPAD;
    constexpr int YM = bwd_yring(FT);                        // y ring: largest lag + two step widths + slack
    constexpr int NACC = FT <= 1 ? 2 : FT;                   // FT = 1: two tiles alternate (independent MFMA chains)
#ifndef HS_BWD_D4
#define HS_BWD_D4 2
#endif
    constexpr int D = N <= 4 ? HS_BWD_D4 : (N <= 8 ? 2 : 1);   // input pipeline depth in super-steps (N <= 4: 3 leaves 4 spilled registers and runs 2 % slower, 4 spills heavily: 0.70 ms)
    constexpr int KSC = 4, KCP0 = 4 + N, KPEND = 4 + 2 * N, KCPX = 4 + 3 * N, KSIZE = 4 + 3 * N + N * N;
    extern __shared__ double lds[];
    const int L = g.L, W = g.W, RB = g.RB, B = g.B;
    double *KC = lds, *DLv = lds + KSIZE, *DLs = lds + KSIZE + N * (RB + 1);   // row stride RB+1: spare slot
    const int lane = threadIdx.x;
    const int cg = blockIdx.x, ch = cg / g.nch, c = cg % g.nch;
    const int64_t T = g.T;
    const int64_t tc = (int64_t)c * B;
    const int nc = (int)((T - tc) < B ? (T - tc) : B);
    const int64_t tend = tc + nc, tstar = tend - 1;
    const int64_t te = (tend + g.He) < T ? (tend + g.He) : T;
    const double *yc = y + (int64_t)ch * T;
    const double *Rc = Rf + (int64_t)ch * N * T;
    const double *FAc = FA0 + (int64_t)ch * T, *FRc = FREF + (int64_t)ch * T, *FVc = FV + (int64_t)ch * N * T;
    double *rhoc = rho + (int64_t)ch * N * T;
    const double *W2c = W2 + (int64_t)ch * T;
    const int64_t FR = 1 + (int64_t)L * (N + 1);
    const double la0pre = c > 0 ? fpre[cg * FR] : 0.0;
    double *recp = bpre + cg * FR, *reco = bown + cg * FR;
    double *yh = yhead + (int64_t)ch * (N * L + 2);

    for (int i = lane; i < N * (RB + 1); i += 64) DLv[i] = 1.0;
    for (int i = lane; i < RB + 1; i += 64) DLs[i] = 0.0;
    // fused statistics: y ring (YM steps) | rho ring (128 steps x NPc ring columns, RPAD padding rows per 16)
    double *YY = DLs + (RB + 1), *RG = YY + YM;
    const int lk = lane >> 4, lj = lane & 15, gla = lj % NPc, glsft = lj / NPc;
    wg_d4 cacc[NACC > 0 ? NACC : 1];
#pragma unroll
    for (int q = 0; q < NACC; q++) cacc[q] = wg_d4{0.0, 0.0, 0.0, 0.0};
    int gtau = -1;                                       // next tau of the product, -1: no owned step yet
    auto rgrow = [&](int r) { return (r + RPAD * (r >> 4)) * NPc; };
    // products for tau = gtau, gtau + 4, ... while tau + 4 <= upto.  FT = 1: two at a time on two accumulator
    // tiles (independent MFMA chains); FT > 1: the FT tiles of one tau are independent of each other.  Reading the
    // operands of a whole super-step first and issuing 16 MFMAs back to back was slower (0.427 vs 0.400 ms: 64
    // more live VGPRs).
    auto gmfma = [&](int upto) {
        if constexpr (FT <= 1) {
            constexpr int D0 = 16 * (NSc - 1);
            for (; gtau + 8 <= upto; gtau += 8) {
                const int r0 = (gtau + lk - 16 * glsft) & 127, r1 = (gtau + 4 + lk - 16 * glsft) & 127;
                const double b0 = RG[rgrow(r0) + gla], b1 = RG[rgrow(r1) + gla];
                const double a0 = YY[(gtau + lk - D0 - lj) & (YM - 1)], a1 = YY[(gtau + 4 + lk - D0 - lj) & (YM - 1)];
                cacc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, cacc[0], 0, 0, 0);
                cacc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, cacc[1], 0, 0, 0);
            }
            if (gtau + 4 <= upto) {
                const int r0 = (gtau + lk - 16 * glsft) & 127;
                const double b0 = RG[rgrow(r0) + gla];
                const double a0 = YY[(gtau + lk - D0 - lj) & (YM - 1)];
                cacc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, cacc[0], 0, 0, 0);
                gtau += 4;
            }
        } else {
            for (; gtau + 4 <= upto; gtau += 4) {
                const int r0 = (gtau + lk - 16 * glsft) & 127;
                const double b0 = RG[rgrow(r0) + gla];
                double av[FT];
#pragma unroll
                for (int q = 0; q < FT; q++) av[q] = YY[(gtau + lk - (LPT * q + 16 * (NSc - 1)) - lj) & (YM - 1)];
#pragma unroll
                for (int q = 0; q < FT; q++) cacc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], b0, cacc[q], 0, 0, 0);
            }
        }
    };
    if (FUSE) {
        for (int i = lane; i < YM + RGROWS * NPc; i += 64) YY[i] = 0.0;
    }
    {
        const WaveConst &Kg = cst[ch];
        if (lane == 0) { KC[0] = Kg.sc0; KC[1] = Kg.mean0; KC[2] = 1.0 / Kg.den; KC[3] = Kg.P00; }
        if (lane < N) { KC[KSC + lane] = Kg.sc[lane]; KC[KCP0 + lane] = Kg.CP0[lane]; KC[KPEND + lane] = Kg.PEND[lane]; }
        if (UC) { if (lane < N) KC[KCPX + lane] = Kg.CPXin[lane]; }
        else for (int i = lane; i < N * N; i += 64) KC[KCPX + i] = Kg.CPX[i];
    }
    __syncthreads();

    double Mb = 0.0, xb = 1.0, z = 0.0;
    double sx[N], ra[N], s2[N], s_all = 0.0, s_m = 0.0, s_y2 = 0.0;
#pragma unroll
    for (int a = 0; a < N; a++) { sx[a] = 0.0; ra[a] = 0.0; s2[a] = 0.0; }

    const int n_warm = (int)(te - tend);                 // steps t = te-2 .. tstar
    const int n_total = n_warm + nc;                     // ... then tstar-1 .. tc-1
    // super-step schedule: the first warm-up step may be partial so that the warm-up ends exactly at
    // tstar; then full steps; the last owned step may be partial.  start(k) = steps done before step k.
    const int w0 = n_warm % W ? n_warm % W : W;
    auto start = [&](int k) { return k == 0 ? 0 : w0 + (k - 1) * W; };
    auto width = [&](int done) {
        if (done < n_warm) return done == 0 ? w0 : W;
        return n_total - done < W ? n_total - done : W;
    };
    auto load = [&](BIn<N> &d, int done) {
        const int nact = done < n_total ? width(done) : 0;
        const int li = lane < nact ? lane : 0;
        int64_t tb = te - 2 - done;                      // time of lane 0; lane j handles tb - j
        tb = tb < -1 ? -1 : tb;
        const int64_t t = tb - li;
        const int64_t t1 = t + 1 < T ? t + 1 : T - 1;
        const int64_t tz = t < 0 ? 0 : t;
        d.y1 = yc[t1];
        d.y0 = yc[tz];
        d.w2 = W2c[t1];
        d.fref = FRc[t1];
        d.la0 = FAc[tz];
#pragma unroll
        for (int a = 0; a < N; a++) {
            d.R[a] = Rc[(int64_t)a * T + t1];
            d.fv[a] = FVc[(int64_t)a * T + t1];
        }
    };
    // MODE 0: warm-up super-step (recursion only, no global stores); 1: owned super-step (posteriors; every
    // lane stores rho, idle lanes of a partial step into a trash line); 2: the last super-steps of the
    // warm-up and of the chain's own samples, which also leave the boundary records and the head of the
    // recording (conditional stores).  Modes 0 and 1 keep global accesses straight-line (exact vmcnt).
    auto run = [&](const BIn<N> &d, int done, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const int nact = width(done);
        const bool live = lane < nact;
        const int64_t tb = te - 2 - done;                // time of lane 0
        const int64_t t = tb - lane;                     // step index i = done + 1 + lane, t = te-1-i
        const bool owned = MODE == 1 || MODE == 3 || (MODE == 2 && done >= n_warm);   // wave-uniform
        if (FUSE) YY[(done + 1 + lane) & (YM - 1)] = d.y1;    // y of this lane's onset time t+1 (idle lanes: steps still to come)
        int ws = (done + 1) % RB + lane;                 // (done+1) % RB is wave-uniform
        ws = ws >= RB ? ws - RB : ws;
        int rs = ws - L;
        rs = rs < 0 ? rs + RB : rs;
        const double sc0 = KC[0];
        double vb[N], E[N + 1];
        const double sbv = DLs[rs];
        double e = -INFINITY;
#pragma unroll
        for (int a = 0; a < N; a++) {
            vb[a] = live ? DLv[a * (RB + 1) + rs] : 0.0;
            E[a] = (sbv + d.R[a]) + KC[KSC + a];         // sw_a
            e = fmax(e, scale_of(vb[a], E[a]));
        }
        const double dd = d.y1 - KC[1];
        const double q1 = -((dd * dd) * KC[2]);           // KC[2] = 1/den
        float fa = live ? (float)(q1 + sc0) : 0.0f;
        float fb = live ? (float)e : -INFINITY;
        scan_maxplus_f32(fa, fb);                         // the envelope is a scale: single precision
        const float Mbf = (float)Mb;                      // exact: the carry is kept float-representable
        const float Mtf = fmaxf(Mbf + fa, fb);
        const double Mt = (double)Mtf;
        const double Mnext = (double)lane_prevf(Mtf, Mbf);
        E[N] = ((Mnext + q1) + sc0) - Mt;
#pragma unroll
        for (int a = 0; a < N; a++) E[a] = vb[a] > 0.0 ? fmin(E[a] - Mt, 700.0) : -INFINITY;
        fexp_n<N + 1>(E);
        const double E0 = E[N];
        double wa[N];
        double al = E0 * KC[3], be = 0.0;
#pragma unroll
        for (int a = 0; a < N; a++) {
            wa[a] = vb[a] * E[a];
            be = __builtin_fma(wa[a], KC[KCP0 + a], be);
        }
        al = live ? al : 1.0;
        be = live ? be : 0.0;
        scan_linear(al, be);
        const double xt = __builtin_fma(al, xb, be);
        const double xnext = lane_prev(xt, xb);
        const double base = xnext * E0;
        double yn[N];
        if (UC) {   // ring a's end -> ring b's first state: CPX[a][b] = q_b for every a != b
            double tq[N], ex[N];
#pragma unroll
            for (int b = 0; b < N; b++) tq[b] = wa[b] * KC[KCPX + b];
            excl_sums<N>(tq, ex);
#pragma unroll
            for (int a = 0; a < N; a++) yn[a] = __builtin_fma(base, KC[KPEND + a], ex[a]);
        } else {
#pragma unroll
            for (int a = 0; a < N; a++) {
                double su = base * KC[KPEND + a];
#pragma unroll
                for (int b = 0; b < N; b++)
                    if (b != a) su = __builtin_fma(wa[b], KC[KCPX + a * N + b], su);
                yn[a] = su;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int wsl = live ? ws : RB;   // idle lanes write the spare slot behind the ring
#pragma unroll
        for (int a = 0; a < N; a++) DLv[a * (RB + 1) + wsl] = yn[a];
        DLs[wsl] = Mt;
        if (MODE == 2 && live) {
            // boundary records for the certificate: values at the first L samples of the next
            // chain (warm-up) / of this chain (own sweep)
            double *rr = nullptr;
            int64_t jj = 0;
            if (!owned && t >= tend && t < tend + L) { rr = recp; jj = t - tend; }
            if (owned && t >= tc && t < tc + L) { rr = reco; jj = t - tc; }
            if (rr) {
#pragma unroll
                for (int a = 0; a < N; a++) rr[1 + jj * (N + 1) + a] = yn[a];
                rr[1 + jj * (N + 1) + N] = Mt;
                if (jj == 0) rr[0] = Mt + flog(xt);
            }
        }
        if (owned) {
            const double la = t >= tc ? d.la0 : la0pre;
            double gg[2] = {fmin((la + Mt) - z, 700.0), fmin((d.fref + Mt) - z, 700.0)};
            fexp_n<2>(gg);
            const int64_t t1 = t + 1;
            if (live && t >= tc && t >= g.own_lo && t < g.own_hi) {
                const double ga = xt * gg[0];                    // gamma_t(silent)
                s_all += ga;                                     // baumwelch.jl:303 qq
                if (!g.last || t < T - 1) s_m += ga;             // :257 bb, t = 1..T-1 of the recording
                s_y2 = __builtin_fma(ga, d.y0 * d.y0, s_y2);     // :302 with the new silent mean (= 0)
            }
            const bool own1 = live && t1 >= g.own_lo && t1 < g.own_hi;
            const bool bulk = !(g.last && t1 > T - L);           // truncated rings: summed per phase in kw_stats_final
#pragma unroll
            for (int a = 0; a < N; a++) {
                const double rv = own1 ? (d.fv[a] * wa[a]) * gg[1] : 0.0;
                if (MODE != 3) *(live ? rhoc + (int64_t)a * T + t1 : trash + 64 * a + lane) = rv;
                ra[a] += bulk ? rv : 0.0;
                s2[a] = __builtin_fma(rv, d.w2, s2[a]);          // sum_k G2(a,k) (baumwelch.jl:302)
                sx[a] += (own1 && t >= 0) ? wa[a] * gg[0] : 0.0;  // xi'_a(t+1): silent(t) -> (a,1)(t+1), :240
                if (FUSE) RG[rgrow((done + 1 + lane) & 127) + a] = rv;   // idle lanes: zeros at steps still to come
            }
            if (FUSE) {
#pragma unroll
                for (int a = N; a < NPc; a++) RG[rgrow((done + 1 + lane) & 127) + a] = 0.0;
                if (gtau < 0) gtau = (done + 1) & ~3;            // wave-uniform
                gmfma(done + 1 + nact);
            }
            if (MODE == 2 && c == 0 && t < L && live && t >= 0) {   // head of the recording: virtual onsets, pp
                double lg[N + 1];
#pragma unroll
                for (int a = 0; a < N; a++) lg[a] = yn[a];
                lg[N] = xt;
                flog_n<N + 1>(lg);
#pragma unroll
                for (int a = 0; a < N; a++) yh[a * L + t] = Mt + lg[a];
                if (t == 0) yh[N * L] = Mt + lg[N];
            }
        }
        xb = wave_bcast(xt, nact - 1);                   // the pair of the last live lane
        Mb = wave_bcast(Mt, nact - 1);
        const int ee = xb > 0.0 ? ilogb(xb) : 0;
        if (ee >= 32 || ee <= -32) {
            const double Mx = Mb + (double)ee * kLn2, Mn = (double)(float)Mx;
            xb = ldexp(xb, -ee) * fexp(Mx - Mn);
            Mb = Mn;
        }
    };
    // normaliser at tstar (all lanes return the same z) and gamma_tstar(silent)
    auto znorm = [&]() {
        __syncthreads();
        const double la = FAc[tstar];
        double zmax = la + Mb + (double)ilogb(xb) * kLn2;
        double sc_[N], mant[N];   // one onset per lane and pass
        for (int i0 = 0; i0 < L; i0 += 64) {
            const int i = i0 + lane;
            const bool on = i < L;
            const int64_t tp = tstar - (on ? i : 0);
            int slot = (int)((te - 1 - (tp + L - 1)) % RB);
            slot = slot < 0 ? slot + RB : slot;
            const double sbv = DLs[slot], fr = FRc[tp];
#pragma unroll
            for (int a = 0; a < N; a++) {
                const double m = FVc[(int64_t)a * T + tp] * DLv[a * (RB + 1) + slot];
                const double s = ((fr + KC[KSC + a]) + Rc[(int64_t)a * T + tp]) + sbv;
                if (on) zmax = fmax(zmax, scale_of(m, s));
            }
        }
        zmax = wave_max(zmax);
        double zs = 0.0;
        for (int i0 = 0; i0 < L; i0 += 64) {
            const int i = i0 + lane;
            const bool on = i < L;
            const int64_t tp = tstar - (on ? i : 0);
            int slot = (int)((te - 1 - (tp + L - 1)) % RB);
            slot = slot < 0 ? slot + RB : slot;
            const double sbv = DLs[slot], fr = FRc[tp];
#pragma unroll
            for (int a = 0; a < N; a++) {
                mant[a] = FVc[(int64_t)a * T + tp] * DLv[a * (RB + 1) + slot];
                sc_[a] = mant[a] > 0.0 ? fmin((((fr + KC[KSC + a]) + Rc[(int64_t)a * T + tp]) + sbv) - zmax, 700.0) : -INFINITY;
            }
            fexp_n<N>(sc_);
#pragma unroll
            for (int a = 0; a < N; a++) zs += on ? mant[a] * sc_[a] : 0.0;
        }
        zs = wave_sum(zs);
        zs += xb * fexp((la + Mb) - zmax);
        z = zmax + flog(zs);
        if (lane == 0) {
            Zc[cg] = z;
            if (tstar >= g.own_lo && tstar < g.own_hi) {
                const double ga = xb * fexp((la + Mb) - z);
                s_all += ga;
                if (!g.last || tstar < T - 1) s_m += ga;
                const double yv = yc[tstar];
                s_y2 = __builtin_fma(ga, yv * yv, s_y2);
            }
        }
        __syncthreads();
    };

    const int nsteps = n_total <= 0 ? 0 : 1 + (n_total - w0 + W - 1) / W;   // w0 first, then W each
    const int kwarm = n_warm == 0 ? 0 : 1 + (n_warm - w0) / W;               // warm-up steps
    const int rk = (L + W - 1) / W + 1;                                      // steps that touch a record window
    // branch-free main loop over whole groups of D steps (see kw_fwd), remainder from the filled buffers
    auto sweep = [&](int k0, int k1, auto mode_tag) {
        BIn<N> buf[D];
#pragma unroll
        for (int i = 0; i < D; i++) load(buf[i], start(k0 + i));
        int k = k0;
        for (; k + D <= k1; k += D) {
#pragma unroll
            for (int i = 0; i < D; i++) {
                run(buf[i], start(k + i), mode_tag);
                load(buf[i], start(k + i + D));
            }
        }
#pragma unroll
        for (int i = 0; i < D; i++)
            if (k + i < k1) run(buf[i], start(k + i), mode_tag);
    };
    const int kw1 = kwarm - rk > 0 ? kwarm - rk : 0;
    if (kw1 > 0) sweep(0, kw1, std::integral_constant<int, 0>());
    if (kwarm > kw1) sweep(kw1, kwarm, std::integral_constant<int, 2>());
    znorm();
    const int ko1 = nsteps - rk > kwarm ? nsteps - rk : kwarm;
    if (FUSE) {
        // rho itself is only read back near the chain ends (certificate windows, end-of-data terms of kw_stats_final):
        // MODE 3 = MODE 1 without the rho stores for the bulk of the chain
        const int km = kwarm + rk < ko1 ? kwarm + rk : ko1;
        if (km > kwarm) sweep(kwarm, km, std::integral_constant<int, 1>());
        if (ko1 > km) sweep(km, ko1, std::integral_constant<int, 3>());
    } else if (ko1 > kwarm) sweep(kwarm, ko1, std::integral_constant<int, 1>());
    if (nsteps > ko1) sweep(ko1, nsteps, std::integral_constant<int, 2>());
    if (FUSE) {
        // drain: the delayed copies of the last steps still meet their y; rho of steps that never come is zero
        const int sig_end = n_total + 1;
        if (gtau >= 0) {
#pragma unroll
            for (int a = 0; a < NPc; a++) RG[rgrow((sig_end + lane) & 127) + a] = 0.0;
            gmfma(sig_end + 16 * (NSc - 1) + 4);               // tau up to sig_end + 16 (NS - 1): the last delayed copy
        }
        double *pg = partG + (int64_t)cg * N * L;
#pragma unroll
        for (int q = 0; q < (FT <= 1 ? 1 : FT); q++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int lag = LPT * q + 16 * (NSc - 1 - glsft) + lk + 4 * r;
                const double v = FT <= 1 ? cacc[0][r] + cacc[1][r] : cacc[q][r];
                if (gla < N && lag < L) pg[gla * L + lag] = v;
            }
    }
    // per-chain partial sums -> partS[cg][3N+3] = sx | ra | s2 | s_all s_m s_y2
    double *ps = partS + (int64_t)cg * (3 * N + 3);
#pragma unroll
    for (int a = 0; a < N; a++) {
        const double v1 = wave_sum(sx[a]), v2 = wave_sum(ra[a]), v3 = wave_sum(s2[a]);
        if (lane == 0) { ps[a] = v1; ps[N + a] = v2; ps[2 * N + a] = v3; }
    }
    {
        const double v1 = wave_sum(s_all), v2 = wave_sum(s_m), v3 = wave_sum(s_y2);
        if (lane == 0) { ps[3 * N] = v1; ps[3 * N + 1] = v2; ps[3 * N + 2] = v3; }
    }
}

// ------------------------------------------------------------------------------------------
// Boundary certificate of the warm-ups (diag[3..6]).  One wavefront per boundary between chains
// c-1 and c.  Forward: the state chain c reached at the end of its warm-up (la0 at tc-1 and the L
// onsets still inside their rings) against what chain c-1 computed for the same quantities;
// backward: the state chain c-1 reached at tc coming down from its warm-up (lb0 at tc and the
// ring-end betas Yn_a(tc..tc+L-2)) against chain c's own.  Both pairs may differ by a frame constant
// D, taken at the entry with the largest posterior weight; the error is the posterior-weighted
// relative mismatch sum_e w_e |exp((x_e - x'_e) - D) - 1| (tolerance 1e-9).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void kw_fb_check(WaveGeom g, double tol, const double *__restrict__ FA0,
                                                  const double *__restrict__ FV, const double *__restrict__ FREF,
                                                  const double *__restrict__ fpre, const double *__restrict__ bpre,
                                                  const double *__restrict__ bown, const double *__restrict__ rho,
                                                  int64_t *__restrict__ diag, double *__restrict__ dbg)
{
    const int lane = threadIdx.x;
    const int cg = blockIdx.x, ch = cg / g.nch, c = cg % g.nch;
    if (c == 0) return;
    const int N = g.N, L = g.L;
    const int64_t T = g.T, tc = (int64_t)c * g.B;
    const int64_t FR = 1 + (int64_t)L * (N + 1);
    const double *FAc = FA0 + (int64_t)ch * T, *FRc = FREF + (int64_t)ch * T, *FVc = FV + (int64_t)ch * N * T;
    const double *rhoc = rho + (int64_t)ch * N * T;
    for (int dir = 0; dir < 2; dir++) {
        const int ne = dir == 0 ? L : L - 1;
        // entry (a, e): forward onset t' = tc-1-e; backward ring-end beta at tc+e (onset tc+e-L+1)
        auto weight = [&](int a, int e) {
            const int64_t tp = dir == 0 ? tc - 1 - e : tc + e - L + 1;
            return tp >= 0 ? rhoc[(int64_t)a * T + tp] : 0.0;
        };
        auto diff = [&](int a, int e) {
            double hv, hs, mv, ms;
            if (dir == 0) {
                const int64_t tp = tc - 1 - e, jj = tp - (tc - L);
                hv = fpre[cg * FR + 1 + jj * (N + 1) + a]; hs = fpre[cg * FR + 1 + jj * (N + 1) + N];
                mv = FVc[(int64_t)a * T + tp]; ms = FRc[tp];
            } else {
                hv = bpre[(cg - 1) * FR + 1 + (int64_t)e * (N + 1) + a]; hs = bpre[(cg - 1) * FR + 1 + (int64_t)e * (N + 1) + N];
                mv = bown[cg * FR + 1 + (int64_t)e * (N + 1) + a]; ms = bown[cg * FR + 1 + (int64_t)e * (N + 1) + N];
            }
            if (hv == mv && hs == ms) return 0.0;
            if (!(hv > 0.0) && !(mv > 0.0)) return 0.0;      // both impossible
            return (hs - ms) + (flog(hv) - flog(mv));
        };
        // silent entry: weight = 1 - ring mass at the boundary sample
        double ringmass = 0.0;
        const int64_t tb = dir == 0 ? tc - 1 : tc;
        for (int idx = lane; idx < N * L; idx += 64) {
            const int a = idx / L, k = idx % L;
            const int64_t tp = tb - k;
            if (tp >= 0) ringmass += rhoc[(int64_t)a * T + tp];
        }
        ringmass = wave_sum(ringmass);
        const double w0 = fmax(1.0 - ringmass, 0.0);
        const double d0 = dir == 0 ? fpre[cg * FR] - FAc[tc - 1] : bpre[(cg - 1) * FR] - bown[cg * FR];
        double wb = lane == 0 ? w0 : -1.0, db = d0;
        int bi = -1;
        for (int idx = lane; idx < N * ne; idx += 64) {
            const int a = idx / ne, e = idx % ne;
            const double w = weight(a, e);
            if (w > wb) { wb = w; db = diff(a, e); bi = idx; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double ow = __shfl_xor(wb, o), od = __shfl_xor(db, o);
            const int oi = __shfl_xor(bi, o);
            if (ow > wb) { wb = ow; db = od; bi = oi; }
        }
        bi = __builtin_amdgcn_readfirstlane(bi);
        const double D = wave_bcast(db, 0);
        double err = lane == 0 ? w0 * fabs(fexp(fmin(d0 - D, 700.0)) - 1.0) : 0.0;
        for (int idx = lane; idx < N * ne; idx += 64) {
            const int a = idx / ne, e = idx % ne;
            const double w = weight(a, e);
            if (w > 0.0) err += w * fabs(fexp(fmin(diff(a, e) - D, 700.0)) - 1.0);
        }
        err = wave_sum(err);
        if (lane == 0) {
            if (!(err <= tol) && dbg && atomicAdd((unsigned long long *)&diag[3 + 2 * dir], 0ull) == 0ull) {
                dbg[0] = cg; dbg[1] = dir; dbg[2] = D; dbg[3] = w0; dbg[4] = d0; dbg[5] = err; dbg[6] = ringmass;
                dbg[7] = wb; dbg[8] = bi; dbg[9] = tc; dbg[10] = ne;
            }
            if (!(err <= tol)) atomicAdd((unsigned long long *)&diag[3 + 2 * dir], 1ull);
            if (err == err)
                atomicMax((unsigned long long *)&diag[4 + 2 * dir], (unsigned long long)__double_as_longlong(err));
        }
    }
}

// ------------------------------------------------------------------------------------------
// spike-triggered sums  G1(a,k) = sum_t' rho_a(t') y[t'+k-1],  G2 likewise with y^2
// (baumwelch.jl:270-282, :297-305); y beyond the end of the data is 0.
// Generic vector kernel (any N, L): block = 4096 onsets, sub-tiles of 256 staged in LDS, one
// thread per (ring, phase) pair.
// ------------------------------------------------------------------------------------------
constexpr int kGsTile = 4096, kGsSub = 256;

__global__ __launch_bounds__(256) void kw_gsum_generic(WaveGeom g, const double *__restrict__ y,
                                                       const double *__restrict__ rho, int nparts,
                                                       double *__restrict__ partG)
{
    extern __shared__ double lds[];
    const int N = g.N, L = g.L, NL = N * L, ch = blockIdx.y;
    double *lr = lds;                 // [N][kGsSub]
    double *ly = lds + N * kGsSub;    // [kGsSub + L]
    const int64_t T = g.T, t0 = (int64_t)blockIdx.x * kGsTile;
    const double *yc = y + (int64_t)ch * T, *rc = rho + (int64_t)ch * N * T;
    const int p = threadIdx.x + 256 * blockIdx.z;   // this thread's (ring, phase) pair
    const int a = p < NL ? p / L : 0, k = p < NL ? p % L : 0;
    double s1 = 0.0;
    for (int s0 = 0; s0 < kGsTile; s0 += kGsSub) {
        const int64_t tb = t0 + s0;
        if (tb >= T) break;
        __syncthreads();
        for (int i = threadIdx.x; i < N * kGsSub; i += 256) {
            const int aa = i / kGsSub, u = i % kGsSub;
            const int64_t t = tb + u;
            lr[i] = t < T ? rc[(int64_t)aa * T + t] : 0.0;
        }
        for (int i = threadIdx.x; i < kGsSub + L; i += 256) {
            const int64_t t = tb + i;
            ly[i] = t < T ? yc[t] : 0.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int u = 0; u < kGsSub; u++) {
            const double r = lr[a * kGsSub + u], yv = ly[u + k];
            s1 = __builtin_fma(r, yv, s1);
        }
    }
    double *out = partG + ((int64_t)ch * nparts + blockIdx.x) * NL;
    if (p < NL) out[p] = s1;
}

// Matrix-core statistics for few rings (N <= 8): the spike-triggered sums are a matrix product over
// time, C[16 lags x 16 columns] += A[16 x 4 samples] B[4 x 16] per v_mfma_f64_16x16x4_f64 with A a
// Toeplitz window of y and B the posteriors.  The 16 B-columns hold NS = 16/NP delayed copies of the NP
// rings (copy s delayed by 16 s samples: B[kk][a + NP s] = rho_a(tau + kk - 16 s)), so one MFMA
// accumulates LPT = 16 NS consecutive lags of every ring with no padded column (N = 4, L = 59: all 59
// lags of all 4 rings in ONE accumulator tile).  Natural layout: a "column" is a block of kGxBv
// consecutive samples (rho outside the column staged as zeros, y read on into the next column); a
// workgroup sweeps kGxSubs groups of 8 columns, wave w takes columns 2w, 2w+1 of a group.  Tiles of TR
// rows go global -> registers -> LDS; the next tile's global loads are in flight during the MFMAs.
// LDS rows of rho are padded by one row per 16 (the 4 delayed copies would otherwise hit one bank).
constexpr int kGxTR = 64, kGxBv = 512, kGxSubs = 1;

// blockIdx.z = lag group: NT accumulator tiles of LPT lags each, starting at lag z NT LPT (models with more
// than 4 LPT lags sweep rho once per group).
template <int N, int NT>
__global__ __launch_bounds__(256) void kw_gsum_mx(WaveGeom g, const double *__restrict__ y,
                                                  const double *__restrict__ rho, int nparts,
                                                  double *__restrict__ partG)
{
    constexpr int NP = N <= 1 ? 1 : (N <= 2 ? 2 : (N <= 4 ? 4 : (N <= 8 ? 8 : 16)));
    constexpr int NS = 16 / NP, LPT = 16 * NS, HS = 16 * (NS - 1);
    constexpr int TR = kGxTR, CW = 8, RR = TR + HS, Bv = kGxBv;
    constexpr int RRP = RR + RR / 16 + 1;          // padded rho rows per column
    constexpr int YR = TR + 19 + LPT * (NT - 1);   // staged y rows per column (odd)
    constexpr int RS = RRP * NP + 2;               // rho column stride
    constexpr int TPS = (Bv + HS + TR - 1) / TR;   // tiles per column group
    // staging indices are split with shifts only: rows per column rounded up to powers of two (the surplus
    // rows are never loaded); integer divisions by 112 or 82 per element cost more than the MFMAs
    constexpr int RRL = RR <= 64 ? 64 : (RR <= 128 ? 128 : (RR <= 256 ? 256 : 512)), YRL = (YR - 1) <= 128 ? 128 : ((YR - 1) <= 256 ? 256 : 512);
    constexpr int NRH = NP * RRL * CW / 256, NYM = CW * YRL / 256;
    extern __shared__ double lds[];
    const int L = g.L, ch = blockIdx.y, lag0 = blockIdx.z * NT * LPT;
    const int64_t T = g.T;
    double *lr = lds;                              // [CW][RS]
    double *ly = lds + CW * RS;                    // [CW][YR]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lk = lane >> 4, lj = lane & 15, la = lj % NP, lsft = lj / NP;
    const double *yc = y + (int64_t)ch * T, *rc = rho + (int64_t)ch * N * T;
    wg_d4 c1[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) c1[q] = wg_d4{0.0, 0.0, 0.0, 0.0};
    int ntiles = 0;
    for (int sub = 0; sub < kGxSubs; sub++)
        if (((int64_t)blockIdx.x * kGxSubs + sub) * CW * Bv < T) ntiles += TPS;
    double tr_[NRH], ty_[NYM];
    auto load_regs = [&](int ti) {   // ring index fastest: conflict-free LDS stores, 128-byte global segments
        const int sub = ti / TPS, s0 = (ti % TPS) * TR;
        const int64_t col0 = ((int64_t)blockIdx.x * kGxSubs + sub) * CW;
#pragma unroll
        for (int k = 0; k < NRH; k++) {
            const int i = tid + k * 256;
            const int a = i % NP, u = (i / NP) % RRL, cc = i / (NP * RRL);
            const int row = s0 - HS + u;
            const int64_t t = (col0 + cc) * Bv + row;
            const bool ok = u < RR && a < N && row >= 0 && row < Bv && t < T;
            const double v = rc[ok ? (int64_t)a * T + t : 0];
            tr_[k] = ok ? v : 0.0;
        }
#pragma unroll
        for (int k = 0; k < NYM; k++) {
            const int i = tid + k * 256;
            const int rr = i % YRL, cc = i / YRL;
            const int64_t t = (col0 + cc) * Bv + s0 + rr + lag0;
            const bool ok = rr < YR - 1 && t < T;
            const double v = yc[ok ? t : 0];
            ty_[k] = ok ? v : 0.0;
        }
    };
    if (ntiles > 0) load_regs(0);
    for (int ti = 0; ti < ntiles; ti++) {
        __syncthreads();  // the previous tile has been consumed
#pragma unroll
        for (int k = 0; k < NRH; k++) {
            const int i = tid + k * 256;
            const int a = i % NP, u = (i / NP) % RRL, cc = i / (NP * RRL);
            if (u < RR) lr[cc * RS + (u + u / 16) * NP + a] = tr_[k];
        }
#pragma unroll
        for (int k = 0; k < NYM; k++) {
            const int i = tid + k * 256;
            const int rr = i % YRL, cc = i / YRL;
            if (rr < YR - 1) ly[cc * YR + rr] = ty_[k];
        }
        __syncthreads();
        if (ti + 1 < ntiles) load_regs(ti + 1);   // in flight during the MFMAs
#pragma unroll
        for (int h = 0; h < CW / 4; h++) {
            const int cc = wv * (CW / 4) + h;
            const double *lyc = ly + cc * YR + lk + lj;
#pragma unroll 4
            for (int ts = 0; ts < TR / 4; ts++) {
                const int u = lk - 16 * lsft + HS + 4 * ts;          // staged row of rho_a(tau + kk - 16 s)
                const double b = lr[cc * RS + (u + u / 16) * NP + la];
#pragma unroll
                for (int q = 0; q < NT; q++) {
                    const double a = lyc[ts * 4 + q * LPT];
                    c1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1[q], 0, 0, 0);
                }
            }
        }
    }
    // sum the four waves' tiles, one accumulator tile at a time: LDS [wave][4 regs][64 lanes] (all NT tiles at once
    // would take 8 KB x NT of LDS: with 8 tiles that halved the occupancy, 16 would not fit beside the staging)
    const int NL = N * L;
    double *out = partG + ((int64_t)ch * nparts + blockIdx.x) * NL;
    double *red = lds;
#pragma unroll
    for (int q = 0; q < NT; q++) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) red[(wv * 4 + r) * 64 + lane] = c1[q][r];
        __syncthreads();
        {
            const int e = tid;                       // 256 threads = 4 regs x 64 lanes
            const int ln = e & 63, r = e >> 6;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 4; w++) v += red[(w * 4 + r) * 64 + ln];
            const int j = ln & 15, a = j % NP, sft = j / NP;
            const int lag = lag0 + q * LPT + 16 * sft + (ln >> 4) + 4 * r;
            if (a < N && lag < L) out[a * L + lag] = v;
        }
    }
}

// deterministic final assembly: stats[ch] = [G0 | G1 | G2 | Xi' | s_all | s_m | s_y2 | 0].
// G2 only enters the M-step through its sum over all ring states (sigma, baumwelch.jl:297-307), so the
// real onsets' share sum_k G2(a,k) = sum_t' rho_a(t') W2(t') is kept in entry (a, 1) and the other
// entries only hold the virtual onsets' terms.
__global__ __launch_bounds__(64) void kw_stats_final(WaveGeom g, int rowsG, const double *__restrict__ partG,
                                                     const double *__restrict__ partS, const double *__restrict__ y,
                                                     const double *__restrict__ Rf, const double *__restrict__ virt,
                                                     const double *__restrict__ FA0, const double *__restrict__ rho,
                                                     const double *__restrict__ Zc, const double *__restrict__ yhead,
                                                     double *__restrict__ pp, double *__restrict__ stats)
{
    const int N = g.N, L = g.L, i = blockIdx.x, ch = blockIdx.y, lane = threadIdx.x;
    const int NL = N * L, ws = 3 * N + 3, total = 3 * NL + N + 4, S = 1 + NL;
    const double *pS = partS + (int64_t)ch * g.nch * ws;
    const double *pG = partG + (int64_t)ch * rowsG * NL;
    // one column of a row-major partial table, rows strided over the lanes; eight loads in flight per lane (the
    // kernel sits between the backward sweep and the M-step on the critical path: 0.05 -> 0.02 ms)
    auto colsum = [&](const double *p, int rows, size_t stride) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
        int r = lane;
        for (; r + 448 < rows; r += 512) {
            a0 += p[(size_t)r * stride];         a1 += p[(size_t)(r + 64) * stride];
            a2 += p[(size_t)(r + 128) * stride]; a3 += p[(size_t)(r + 192) * stride];
            a4 += p[(size_t)(r + 256) * stride]; a5 += p[(size_t)(r + 320) * stride];
            a6 += p[(size_t)(r + 384) * stride]; a7 += p[(size_t)(r + 448) * stride];
        }
        for (; r < rows; r += 64) a0 += p[(size_t)r * stride];
        return ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    };
    double acc = 0.0;
    if (i < NL) {
        acc = colsum(pS + N + i / L, g.nch, ws);
    } else if (i < 2 * NL) {
        acc = colsum(pG + (i - NL), rowsG, NL);
    } else if (i < 3 * NL) {
        const int e = i - 2 * NL;
        if (e % L == 0) acc = colsum(pS + 2 * N + e / L, g.nch, ws);
    } else if (i < 3 * NL + N) {
        acc = colsum(pS + (i - 3 * NL), g.nch, ws);
    } else if (i < 3 * NL + N + 3) {
        acc = colsum(pS + 3 * N + (i - 3 * NL - N), g.nch, ws);
    }
    // edge terms of this entry (the block's lanes share the loop a separate kernel used to run serially per entry,
    // on the critical path between the backward sweep and the M-step): virtual onsets t' = -j at the start of the
    // recording, the end-of-data correction of G0, and pp = gamma[:,1] (baumwelch.jl:263)
    if (i < 3 * NL) {
        const int64_t T = g.T;
        const int pair = i % NL, which = i / NL, a = pair / L, k = pair % L + 1;
        const double *yc = y + (int64_t)ch * T, *rc = rho + (int64_t)ch * N * T;
        const double *yh = yhead + (int64_t)ch * (NL + 2);
        const double *V = virt + ((int64_t)ch * N + a) * (L + 1);
        const double z = Zc[(int64_t)ch * g.nch];
        if (g.first)
            for (int j = 1 + lane; j <= L - 1; j += 64) {
                const int idx = -j + k - 1;
                if (idx < 0) continue;
                const double rv = fexp((V[j] + yh[a * L + (L - 1 - j)]) - z);
                const double yv = yc[idx];
                acc += which == 0 ? rv : (which == 1 ? rv * yv : rv * (yv * yv));
            }
        if (which == 0) {
            if (g.last)
                for (int64_t t = T - L + 1 + lane; t <= T - k; t += 64)
                    if (t >= 0) acc += rc[(int64_t)a * T + t];
            if (lane == 0) {
                const double lpv = k == 1 ? Rf[((int64_t)ch * N + a) * T] : V[k - 1];
                pp[(int64_t)ch * S + 1 + pair] = (lpv + yh[a * L + (L - k)]) - z;
                if (pair == 0) pp[(int64_t)ch * S] = (FA0[(int64_t)ch * T] + yh[NL]) - z;
            }
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) stats[(int64_t)ch * total + i] = acc;
}

// M-step finish (baumwelch.jl:262-307) from the (possibly all-reduced) statistics.
// out[ch] = [mu (K x N col-major) | sigma | lp_new (N) | pp (S)]
__global__ __launch_bounds__(256) void kw_mstep(WaveGeom g, const WaveConst *__restrict__ cst,
                                                const double *__restrict__ stats_all,
                                                const double *__restrict__ pp_all, double *__restrict__ out_all)
{
    __shared__ double red[8];
    const int N = g.N, L = g.L, NL = N * L, K = L + 1, ch = blockIdx.x, S = 1 + NL;
    const double *stats = stats_all + (int64_t)ch * (3 * NL + N + 4);
    const double *pp = pp_all + (int64_t)ch * S;
    double *out = out_all + (int64_t)ch * (K * N + 1 + N + S);
    const double *G0 = stats, *G1 = stats + NL, *G2 = stats + 2 * NL, *Xi = stats + 3 * NL;
    const double s_all = stats[3 * NL + N], s_m = stats[3 * NL + N + 1], s_y2 = stats[3 * NL + N + 2];
    double x2 = 0.0, qq = 0.0;
    for (int p = threadIdx.x; p < NL; p += blockDim.x) {
        const int a = p / L, k = p % L + 1;
        const double mu = G1[p] / G0[p];                 // :285  mu[j,l] /= gg[j,l]
        out[k + K * a] = mu;
        x2 += (G2[p] - (2.0 * mu) * G1[p]) + (mu * mu) * G0[p];
        qq += G0[p];
    }
    for (int a = threadIdx.x; a < N; a += blockDim.x) {
        out[K * a] = 0.0;                                // row 1 stays 0 (:268 fill!, never updated)
        // :264 xb[2:end]; Xi' excludes the coefficient exp(c0_a - sc_a), added back in the log domain
        out[K * N + 1 + a] = (cst[ch].xishift[a] + flog(Xi[a])) - flog(s_m);
    }
    for (int j = threadIdx.x; j < S; j += blockDim.x) out[K * N + 1 + N + j] = pp[j];
    x2 = wave_sum(x2); qq = wave_sum(qq);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = x2; red[4 + (threadIdx.x >> 6)] = qq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double X2 = ((red[0] + red[1]) + (red[2] + red[3])) + s_y2;
        const double QQ = ((red[4] + red[5]) + (red[6] + red[7])) + s_all;
        out[K * N] = sqrt(X2 / QQ);                      // :306-307
    }
}

template <typename Kern>
static int wave_lds_attr2(Kern kern, size_t lds)
{
    if (lds > 64 * 1024)
        HS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return HMMSORT_OK;
}

static int wave_estep_sweeps(WaveDev *r, const double *d_y, double *d_stats, hipStream_t st)
{
    const WaveGeom &g = r->g;
    const int N = g.N, L = g.L, NL = N * L;
    const int nchT = g.C * g.nch;
    int rowsG = 0;
    int rc = dispatch_N(N, [&](auto n) {
        constexpr int NN = decltype(n)::value;
        const size_t ldsf = ((size_t)(NN > 8 ? 1 : 2) * NN * (g.RB + 1) + 5 + 3 * NN + NN * NN) * sizeof(double);
        const bool generic = getenv("HMMSORT_GSUM_GENERIC") != nullptr;
        // 3-4 rings of at most 64 states, 5-8 rings of at most 128 (uniform exit -> entry values): the backward
        // sweep accumulates G1 itself (matrix cores, LDS rings)
        constexpr bool kFuse4 = NN == 3 || NN == 4, kFuse8 = NN >= 5 && NN <= 8;
        const bool sep = generic || getenv("HMMSORT_GSUM_SEPARATE") != nullptr;
        const int ft = sep ? 0 : (kFuse4 && L <= 64) ? 1 : (kFuse8 && r->uniform_cx && L <= 64) ? 2
                       : (kFuse8 && r->uniform_cx && L <= 128) ? 4 : 0;
        const bool fuse = ft > 0;
        const int ym = bwd_yring(ft);
        const size_t ldsb = ((size_t)(NN + 1) * (g.RB + 1) + 4 + 3 * NN + NN * NN +
                             (fuse ? ym + (128 + 8 * (NN <= 4 ? 1 : 2)) * (NN <= 4 ? 4 : 8) : 0)) * sizeof(double);
        auto kf = kw_fwd<NN, true>;
        auto kb = kw_bwd<NN, true>;
        if constexpr (NN <= 8) {   // per-source exit -> entry values: up to 8 rings (wave_supported)
            if (!r->uniform_cx) { kf = kw_fwd<NN, false>; kb = kw_bwd<NN, false>; }
        }
        if constexpr (kFuse4) {
            if (ft == 1) kb = r->uniform_cx ? kw_bwd<NN, true, 1> : kw_bwd<NN, false, 1>;
        }
        if constexpr (kFuse8) {
            if (ft == 2) kb = kw_bwd<NN, true, 2>;
            if (ft == 4) kb = kw_bwd<NN, true, 4>;
        }
        int rc2;
        if ((rc2 = wave_lds_attr2(kf, ldsf)) || (rc2 = wave_lds_attr2(kb, ldsb))) return rc2;
        { WPROF(r, "kw_fwd", st);
          hipLaunchKernelGGL(kf, dim3(nchT), dim3(64), ldsf, st, g, r->d_cst, d_y, r->Rf, r->virt, r->FA0,
                             r->FV, r->FREF, r->fpre, r->trash); }
        { WPROF(r, "kw_bwd", st);
          hipLaunchKernelGGL(kb, dim3(nchT), dim3(64), ldsb, st, g, r->d_cst, d_y, r->Rf, r->FA0, r->FV,
                             r->FREF, r->fpre, r->W2, r->rho, r->partS, r->Zc, r->bpre, r->bown, r->yhead, r->trash, r->partG); }
        HS_HIP(hipGetLastError());
        // the certificate runs beside the statistics / final assembly
        HS_HIP(hipEventRecord(r->ev_a, st));
        HS_HIP(hipStreamWaitEvent(r->side, r->ev_a, 0));
        if (g.nch > 1) {
            WPROF(r, "kw_fb_check", r->side);
            hipLaunchKernelGGL(kw_fb_check, dim3(nchT), dim3(64), 0, r->side, g, 1e-9, r->FA0, r->FV, r->FREF, r->fpre,
                               r->bpre, r->bown, r->rho, r->diag, r->dbg);
        }
        HS_HIP(hipEventRecord(r->ev_c, r->side));   // the certificate only has to be done when the call's work is
        constexpr int NPx = NN <= 1 ? 1 : (NN <= 2 ? 2 : (NN <= 4 ? 4 : (NN <= 8 ? 8 : 16)));
        constexpr int LPTx = 16 * (16 / NPx), HSx = LPTx - 16;
        const int ntx = (L + LPTx - 1) / LPTx;
        if (fuse) {
            rowsG = g.nch;                      // partG[ch][chain][N L], written by kw_bwd
        } else if (!generic) {
            // up to 4 accumulator tiles per workgroup for N <= 8 (longer rings take one pass over rho per group of 4);
            // more than 8 rings (16 lags per tile): up to 16 tiles, so that rings of up to 255 states need ONE pass
            // over rho (128 B of posteriors per sample and pass at N = 16); HMMSORT_GSUM_TILES = 4 / 8 / 16: tuning aid
            // (measured at N = 16, L = 255, 40 M samples: 4 tiles and four passes over rho 12.3 ms and 22 GB of HBM
            // traffic, 16 tiles and one pass 10.2 ms)
            static const int nt_big = getenv("HMMSORT_GSUM_TILES") ? atoi(getenv("HMMSORT_GSUM_TILES")) : 16;
            const int want = nt_big >= 16 ? 16 : (nt_big >= 8 ? 8 : 4);
            const int ntk = NN > 8 ? (ntx <= 4 ? 4 : (ntx <= 8 ? (want < 8 ? want : 8) : want)) : (ntx < 4 ? ntx : 4);
            const int ngrp = (ntx + ntk - 1) / ntk;
            rowsG = (int)((g.T + kGxSubs * 8 * (int64_t)kGxBv - 1) / (kGxSubs * 8 * (int64_t)kGxBv));
            constexpr int RRx = kGxTR + HSx, RRPx = RRx + RRx / 16 + 1;
            const size_t l1 = ((size_t)8 * (RRPx * NPx + 2) + (size_t)8 * (kGxTR + 19 + LPTx * (ntk - 1))) * 8;
            const size_t l2 = (size_t)4 * 4 * 64 * 8;
            const size_t lds = l1 > l2 ? l1 : l2;
            auto go = [&](auto kern) -> int {
                int rc3 = wave_lds_attr2(kern, lds);
                if (rc3) return rc3;
                WPROF(r, "kw_gsum", st);
                hipLaunchKernelGGL(kern, dim3(rowsG, g.C, ngrp), dim3(256), lds, st, g, d_y, r->rho, rowsG, r->partG);
                return HMMSORT_OK;
            };
            if constexpr (NN > 8) rc2 = ntk == 4 ? go(kw_gsum_mx<NN, 4>) : ntk == 8 ? go(kw_gsum_mx<NN, 8>) : go(kw_gsum_mx<NN, 16>);
            else rc2 = ntk == 1 ? go(kw_gsum_mx<NN, 1>) : ntk == 2 ? go(kw_gsum_mx<NN, 2>)
                       : ntk == 3 ? go(kw_gsum_mx<NN, 3>) : go(kw_gsum_mx<NN, 4>);
            if (rc2) return rc2;
        } else {
            rowsG = r->gparts;
            const size_t lds = ((size_t)NN * kGsSub + kGsSub + L) * sizeof(double);
            WPROF(r, "kw_gsum", st);
            hipLaunchKernelGGL(kw_gsum_generic, dim3(rowsG, g.C, (NL + 255) / 256), dim3(256), lds, st, g, d_y, r->rho, rowsG, r->partG);
        }
        HS_HIP(hipGetLastError());
        return HMMSORT_OK;
    });
    if (rc) return rc;
    const int total = 3 * NL + N + 4;
    { WPROF(r, "kw_stats_final", st);
      hipLaunchKernelGGL(kw_stats_final, dim3(total, g.C), dim3(64), 0, st, g, rowsG, r->partG, r->partS, d_y, r->Rf,
                         r->virt, r->FA0, r->rho, r->Zc, r->yhead, r->pp, d_stats); }
    HS_HIP(hipStreamWaitEvent(st, r->ev_c, 0));
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int wave_estep(WaveDev *r, const double *d_y, double *d_stats, hipStream_t st)
{
    return wave_graphed(r, 2, d_y, d_stats, nullptr, nullptr, st, [&](hipStream_t s) -> int {
        int rc;
        HS_HIP(hipMemsetAsync(r->diag, 0, 8 * sizeof(int64_t), s));
        if ((rc = wave_prepare(r, d_y, s))) return rc;
        return wave_estep_sweeps(r, d_y, d_stats, s);
    });
}

int wave_mstep(WaveDev *r, const double *d_stats, double *d_out, hipStream_t st)
{
    { WPROF(r, "kw_mstep", st);
      hipLaunchKernelGGL(kw_mstep, dim3(r->g.C), dim3(256), 0, st, r->g, r->d_cst, d_stats, r->pp, d_out); }
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

// one decode + one E-step of the same signal and model: the Viterbi sweep with its post-processing
// runs on the plan's second internal stream beside the forward/backward sweeps
int wave_decode_estep(WaveDev *r, const double *d_y, int16_t *d_x, double *d_ll, double *d_stats, hipStream_t st)
{
    return wave_graphed(r, 3, d_y, d_x, d_ll, d_stats, st, [&](hipStream_t s) -> int {
        int rc;
        HS_HIP(hipMemsetAsync(r->diag, 0, 8 * sizeof(int64_t), s));
        if ((rc = wave_prepare(r, d_y, s))) return rc;
        HS_HIP(hipEventRecord(r->ev_fork, s));
        HS_HIP(hipStreamWaitEvent(r->side2, r->ev_fork, 0));
        // the E-step chain is the critical path: its launches are enqueued first (the dozen small launches of the
        // decode would otherwise hold the forward sweep back by their host-side enqueue time, ~0.1 ms)
        if ((rc = wave_estep_sweeps(r, d_y, d_stats, s))) return rc;
        if ((rc = wave_viterbi_sweep(r, d_y, r->side2))) return rc;
        if ((rc = wave_viterbi_post(r, d_y, d_x, d_ll, r->side2))) return rc;
        HS_HIP(hipEventRecord(r->ev_join, r->side2));
        HS_HIP(hipStreamWaitEvent(s, r->ev_join, 0));
        return HMMSORT_OK;
    });
}

}  // namespace hmmsort
