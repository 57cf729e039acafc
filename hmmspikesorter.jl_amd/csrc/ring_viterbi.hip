// Ring engine, part 2: time-parallel Viterbi (reference src/viterbi.jl:44-98).
//
// Per sample t (0-based) the serial recursion keeps, per chain (= per lane):
//   D0          = delta_t(silent)
//   P_a(t')     = delta of ring a's LAST state at time t'+L-1 for the onset at t'
//               = U_a(t') + Rfull_a(t'),  U_a(t') = best predecessor score of state (a,1) at t'
// and decides, for the N+1 junction states, the back-pointer psi in {0 = silent, b = ring b's
// last state}.  Candidates are scanned in source-state order (silent first, then rings
// ascending) with a strict '>' -- the reference's tie rule (viterbi.jl:74-84).
// Everything is relative to the frame delta'_t = delta_t - A*(t+1) (A = emission constant).
//
// Exactness: inside one chain the arithmetic differs from the reference's only by rounding at
// the 1e-13 level (ring scores are pre-summed; each chain carries its own small offset instead
// of the reference's O(T) cumulative sum, whose own rounding unit is larger).  A decoded path can
// therefore differ from the reference only where two alternatives tie within rounding noise of
// the reference itself.  hmmsort_set_option("engine", STRICT) gives the op-for-op engine.
#include <algorithm>
#include <cmath>
#include <type_traits>

#include "ring_chain_bodies.h"
#include "ring_common.h"

namespace hmmsort {

// One lane = one chain.  grid = ncol/64 blocks of 64 threads.
template <int N>
__global__ __launch_bounds__(64) void k_vit_chain(RingGeom g, JParams<N> jp,
                                                  const double *__restrict__ yT,
                                                  const double *__restrict__ Rf,
                                                  double *__restrict__ P,
                                                  uint32_t *__restrict__ psi,
                                                  double *__restrict__ D0pre,
                                                  double *__restrict__ D0end)
{
    vit_chain_body<N>(blockIdx.x, g, jp, yT, Rf, P, psi, D0pre, D0end);
}

// Final state = argmax over all S states at the last sample, first maximum in state order
// (viterbi.jl:90).  delta(a,k) at T-1 is P_a(T-k) (truncated ring score).  One wave.
__global__ __launch_bounds__(64) void k_vit_tail(RingGeom g, const double *__restrict__ P,
                                                 const double *__restrict__ D0end,
                                                 int32_t *__restrict__ final_state)
{
    const int lane = threadIdx.x;
    const int c = g.nch - 1;
    const int64_t tc = (int64_t)c * g.B;
    const int64_t planeP = (int64_t)(g.H + g.B) * g.ncol;
    const int S = 1 + g.N * g.L;
    double best = -INFINITY;
    int bi = S;  // sentinel
    for (int j = lane; j < S; j += 64) {
        double v;
        if (j == 0) {
            v = D0end[c];
        } else {
            const int a = (j - 1) / g.L, k = (j - 1) % g.L + 1;
            const int64_t s = (g.T - k) - tc;  // onset time T-k relative to the last chain
            v = P[a * planeP + (int64_t)(g.H + s) * g.ncol + c];
        }
        if (v > best) { best = v; bi = j; }  // ascending j per lane: keeps the first maximum
    }
    // wave argmax, lowest index on ties
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) final_state[0] = (bi >= S) ? 0 : bi;
}

__device__ __forceinline__ int psi_get(const RingGeom &g, const uint32_t *__restrict__ psi,
                                       int64_t plane, int64_t off, int e)
{
    const uint32_t w = psi[(int64_t)(e / g.epw) * plane + off];
    return (int)((w >> ((e % g.epw) * g.bits)) & ((1u << g.bits) - 1u));
}

// one backward step of the path: state (a,k) at time t (a = -1: silent) -> state at t-1, given
// the offset of sample t in the transposed psi array
__device__ __forceinline__ void walk_step(const RingGeom &g, const uint32_t *__restrict__ psi,
                                          int64_t off, int &a, int &k)
{
    if (a >= 0 && k > 1) { k--; return; }  // ring interior: single predecessor
    const int p = psi_get(g, psi, (int64_t)g.B * g.ncol, off, a + 1);
    if (p == 0) { a = -1; k = 0; }
    else { a = p - 1; k = g.L; }
}

// Backtrace (viterbi.jl:93-94).  Chain c starts its walk H samples after its own end (from the
// silent state; from the true final state when that point is the end of the data); by the time
// the walk enters the chain it has merged with the true path.  bstate[c] records the walk's
// state on the first sample of chain c+1 for the stitch check.  The psi words of a step do not
// depend on the walk's state, so they are fetched a batch ahead.
template <int N>
__global__ __launch_bounds__(64) void k_vit_backtrace(RingGeom g, const uint32_t *__restrict__ psi,
                                                      const int32_t *__restrict__ final_state,
                                                      int16_t *__restrict__ xT,
                                                      int32_t *__restrict__ bstate)
{
    constexpr int BITS = psi_bits_c(N), EPW = psi_epw_c(N), W = psi_words_c(N), UB = 16;
    const int c = blockIdx.x * 64 + threadIdx.x;
    const bool active = c < g.nch;
    const int B = g.B, L = g.L, ncol = g.ncol;
    const int64_t tc = (int64_t)c * B;
    const int nc = active ? (int)((g.T - tc) < B ? (g.T - tc) : B) : 0;
    int64_t te = tc + nc + g.H;
    if (te > g.T) te = g.T;
    const int se = (int)(te - 1 - tc);  // first (highest) step of this lane's walk
    int a = -1, k = 0;
    if (active && te == g.T) {
        const int fs = final_state[0];
        if (fs > 0) { a = (fs - 1) / L; k = (fs - 1) % L + 1; }
    }
    const int64_t plane = (int64_t)B * ncol;
    auto load = [&](uint32_t(&d)[UB][W], int sb) {  // steps sb, sb-1, ..., sb-UB+1 (unconditional)
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int s = sb - u;
            const bool need = active && s <= se && s >= 0 && (tc + s) >= 1;
            const int sc = need ? s : 0;
            const int cc = active ? c : 0;
            const int64_t off = (sc < B) ? (int64_t)sc * ncol + cc : (int64_t)(sc - B) * ncol + cc + 1;
#pragma unroll
            for (int w = 0; w < W; w++) {
                const uint32_t v = psi[w * plane + off];
                d[u][w] = need ? v : 0u;
            }
        }
    };
    uint32_t bufA[UB][W], bufB[UB][W];
    const int stop = B + g.H - 1;  // wave-uniform start (H <= B, so s - B < B)
    auto run = [&](uint32_t(&cur)[UB][W], int sb) {
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int s = sb - u;
            const bool live = active && s <= se && s >= 0;
            const int id = (a < 0) ? 1 : 2 + a * L + (k - 1);
            if (live && s <= nc) {
                if (s < nc) xT[(int64_t)s * ncol + c] = (int16_t)id;
                else bstate[c] = id;
            }
            // predecessor (branch-free): ring interior -> k-1; junction -> psi entry a+1
            const int e = a + 1;
            uint32_t wsel = cur[u][0];
#pragma unroll
            for (int w = 1; w < W; w++) wsel = (e / EPW == w) ? cur[u][w] : wsel;
            const int p = (int)((wsel >> ((e % EPW) * BITS)) & ((1u << BITS) - 1u));
            const bool interior = (a >= 0) && (k > 1);
            const int na = interior ? a : p - 1;          // p == 0 -> -1 (silent)
            const int nk = interior ? k - 1 : (p == 0 ? 0 : L);
            const bool step = live && (tc + s >= 1);
            a = step ? na : a;
            k = step ? nk : k;
        }
    };
    load(bufA, stop);
    for (int sb = stop; sb >= 0; sb -= 2 * UB) {
        load(bufB, sb - UB);
        run(bufA, sb);
        if (sb - 2 * UB >= 0) load(bufA, sb - 2 * UB);
        run(bufB, sb - UB);
    }
}

// Stitch check: the state chain c's walk had on the first sample of chain c+1 must equal what
// chain c+1 emitted there; otherwise chain c is queued for a re-walk.
__global__ void k_stitch_check(RingGeom g, const int16_t *__restrict__ xT,
                               const int32_t *__restrict__ bstate, int32_t *__restrict__ redo)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.nch - 1) return;
    if (bstate[c] != (int)xT[c + 1]) {  // row 0, column c+1
        const int slot = atomicAdd(&redo[0], 1);
        if (slot < g.ncol) redo[1 + slot] = c;
    }
}

// Serial repair of queued chains (rare): re-walk chain c from the state chain c+1 emitted on its
// first sample, then cascade downwards while the junction with the previous chain disagrees.
__global__ void k_stitch_fix(RingGeom g, const uint32_t *__restrict__ psi, int16_t *__restrict__ xT,
                             int32_t *__restrict__ bstate, const int32_t *__restrict__ redo,
                             int64_t *__restrict__ diag)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n = redo[0] < g.ncol ? redo[0] : g.ncol;
    const int B = g.B, L = g.L;
    int64_t fixes = 0;
    for (int i = 0; i < n; i++) {
        int c = redo[1 + i];
        while (c >= 0) {
            const int want = xT[c + 1];
            if (bstate[c] == want) break;
            bstate[c] = want;
            fixes++;
            int a = -1, k = 0;
            if (want > 1) { a = (want - 2) / L; k = (want - 2) % L + 1; }
            const int64_t tc = (int64_t)c * B;
            int64_t t = tc + B;  // first sample of chain c+1 (chain c < nch-1 is full length)
            walk_step(g, psi, (int64_t)c + 1, a, k);  // row 0, column c+1
            for (t = t - 1; t >= tc; t--) {
                const int id = (a < 0) ? 1 : 2 + a * L + (k - 1);
                xT[(t - tc) * g.ncol + c] = (int16_t)id;
                if (t == 0) break;
                if (t > tc) walk_step(g, psi, (t - tc) * g.ncol + c, a, k);
            }
            c--;  // did the first sample of chain c change?  then chain c-1 must be re-checked
        }
    }
    diag[1] += fixes;
}

// Boundary certificate of the Viterbi warm-up.  Chain c's warm-up and chain c-1's own sweep both
// computed delta(silent) at tc-1 and the L onsets per ring still inside their rings.  Back-pointer
// decisions depend only on DIFFERENCES between those entries, so if every RELEVANT entry of the
// warm-up equals its counterpart up to one common constant, all decisions of chain c are those of
// a sequential sweep.  An in-flight onset is provably irrelevant when, in both frames, it loses
// every competition at the moment it leaves its ring to the path that simply stays silent from
// tc-1 on:  P_a(t') + kappa_a < delta(silent, tc-1) + sum_{t=tc..te}(c00 + q_t(silent)) - 1e-6,
// te = t'+L-1, kappa_a = max_j [lp((a,L)->j) - lp(silent->j)].  (Every entry is finite: an
// onset's score is at least delta(silent)+lp.)  Block = 64 boundaries x 8 entry subsets; flagged
// boundaries are counted in diag[0], the largest relevant spread goes to diag[2].
constexpr int kVChkParts = 8;

struct KappaArg { double c00, mean0, den; double kappa[kRingMaxN]; };

__global__ __launch_bounds__(64 * kVChkParts) void k_halo_check(RingGeom g, KappaArg ka, double tol,
                                                               const double *__restrict__ yT,
                                                               const double *__restrict__ P,
                                                               const double *__restrict__ D0pre,
                                                               const double *__restrict__ D0end,
                                                               int64_t *__restrict__ diag)
{
    extern __shared__ double pre[];  // [L][64] prefix sums of (c00 + q_t(silent)), t = tc..tc+i
    __shared__ double shlo[kVChkParts][64], shhi[kVChkParts][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool on = c >= 1 && c < g.nch;
    const int B = g.B, H = g.H, L = g.L, N = g.N, ncol = g.ncol;
    const int64_t planeP = (int64_t)(H + B) * ncol;
    const int cc = on ? c : 1;
    for (int i = part; i < L; i += kVChkParts) {
        const int64_t t = (int64_t)cc * B + i;
        double v = 0.0;
        if (t < g.T) { const double d = yT[(int64_t)i * ncol + cc] - ka.mean0; v = ka.c00 - (d * d) / ka.den; }
        pre[i * 64 + lane] = v;
    }
    __syncthreads();
    if (part == 0) {
        double acc = 0.0;
        for (int i = 0; i < L; i++) { acc += pre[i * 64 + lane]; pre[i * 64 + lane] = acc; }
    }
    __syncthreads();
    double lo = INFINITY, hi = -INFINITY;
    if (on) {
        const double dh = D0pre[c], dm = D0end[c - 1];
        if (part == 0) { const double d = dh - dm; lo = d; hi = d; }
        for (int a = 0; a < N; a++)
#pragma unroll 4
            for (int j = 1 + part; j <= L; j += kVChkParts) {
                const double hv = P[a * planeP + (int64_t)(H - j) * ncol + c];
                const double mv = P[a * planeP + (int64_t)(H + B - j) * ncol + c - 1];
                // silent-stay path up to the exit time te = tc + (L-1-j)
                const double sil = (j < L) ? pre[(L - 1 - j) * 64 + lane] : 0.0;
                const bool dead = (hv + ka.kappa[a] < dh + sil - 1e-6) && (mv + ka.kappa[a] < dm + sil - 1e-6);
                if (!dead) {
                    const double d = (hv == mv) ? 0.0 : hv - mv;
                    lo = fmin(lo, d); hi = fmax(hi, d);
                    if (d != d) hi = INFINITY;  // NaN -> flagged
                }
            }
    }
    shlo[part][lane] = lo; shhi[part][lane] = hi;
    __syncthreads();
    if (part == 0 && on) {
#pragma unroll
        for (int q = 1; q < kVChkParts; q++) { lo = fmin(lo, shlo[q][lane]); hi = fmax(hi, shhi[q][lane]); }
        const double spread = hi - lo;
        if (!(spread <= tol)) atomicAdd((unsigned long long *)&diag[0], 1ull);
        if (spread == spread && spread < INFINITY)
            atomicMax((unsigned long long *)&diag[2], (unsigned long long)__double_as_longlong(spread));
    }
}

// xT[s*ncol + c] -> x[c*B + s]
__global__ __launch_bounds__(256) void k_transpose_x(const int16_t *__restrict__ xT, int64_t T,
                                                     int B, int ncol, int16_t *__restrict__ x)
{
    __shared__ int16_t tile[64][66];
    const int s0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int sl = ty + 4 * i;
        tile[sl][tx] = xT[(int64_t)(s0 + sl) * ncol + c0 + tx];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int cl = ty + 4 * i;
        const int64_t t = (int64_t)(c0 + cl) * B + s0 + tx;
        if (t < T) x[t] = tile[tx][cl];
    }
}

// ll = sum_{t=1..T-1} T1[x_t, t]  (viterbi.jl:92-96), computed without the trellis:
//   T1[x_t,t] = T1[x_0,0] + sum_{u=1..t} inc_u,  inc_u = lp(x_{u-1}->x_u) + q_u(x_u)
//   => ll = (T-1)*T1[x_0,0] + sum_{u=1..T-1} (T-u)*inc_u        (a plain parallel reduction).
// ctab: c00 | c0[N] | cend[N] | cx[N*N] | cint[N*L]
__device__ __forceinline__ double path_lp(const RingGeom &g, const double *__restrict__ ctab,
                                          int xp, int xc)
{
    const int N = g.N, L = g.L;
    if (xp == 1) return xc == 1 ? ctab[0] : ctab[1 + (xc - 2) / L];
    const int a = (xp - 2) / L, k = (xp - 2) % L + 1;
    if (k < L) return ctab[1 + 2 * N + N * N + a * L + k];
    if (xc == 1) return ctab[1 + N + a];
    return ctab[1 + 2 * N + a * N + (xc - 2) / L];
}

__global__ __launch_bounds__(256) void k_ll_partial(RingGeom g, const double *__restrict__ y,
                                                    const int16_t *__restrict__ x,
                                                    const double *__restrict__ mean,
                                                    const double *__restrict__ ctab, double A,
                                                    double den, double *__restrict__ part)
{
    __shared__ double red[4];
    const int64_t T = g.T;
    double acc = 0.0;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; u < T;
         u += (int64_t)gridDim.x * blockDim.x) {
        const int xp = x[u - 1], xc = x[u];
        const double d = y[u] - mean[xc - 1];
        const double inc = path_lp(g, ctab, xp, xc) + (A - (d * d) / den);
        acc += (double)(T - u) * inc;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int x0 = x[0];
        if (x0 != 1) {  // T1[1,1] = 0 for the silent state (viterbi.jl:63)
            const double d = y[0] - mean[x0 - 1];
            acc += (double)(T - 1) * (A - (d * d) / den);
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_sum_partials(const double *__restrict__ part, int n,
                                                      double *__restrict__ out)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += part[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = (red[0] + red[1]) + (red[2] + red[3]);
}

int ring_viterbi_launch(RingDev *r, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st)
{
    const RingGeom &g = r->g;
    int rc;
    HS_HIP(hipMemsetAsync(r->diag, 0, 8 * sizeof(int64_t), st));
    if ((rc = ring_prepare(r, d_y, st))) return rc;
    if ((rc = ring_launch_virtual(r, d_y, r->Pv, (int64_t)(g.H + g.B) * g.ncol, st))) return rc;
    rc = dispatch_N(g.N, [&](auto n) {
        constexpr int N = decltype(n)::value;
        JParams<N> jp = make_jparams<N>(r);
        { PROF(r, "k_vit_chain", st); hipLaunchKernelGGL((k_vit_chain<N>), dim3(g.ncol / 64), dim3(64), 0, st, g, jp, r->yT,
                           r->Rf, r->Pv, r->psi, r->D0pre, r->D0end); }
        HS_HIP(hipGetLastError());
        return HMMSORT_OK;
    });
    if (rc) return rc;
    return ring_viterbi_post(r, d_y, d_x, d_ll, st);
}

// The first decoded state, exactly as the reference finds it: x[0] = psi_1(x[1]) and psi_1 only sees
// the first trellis column, which is plain emission (viterbi.jl:55-63): T1[s,0] = funcl(y[0], mean_s),
// T1[1,0] = 0.  Template tails are ~1e-16, so the "ring in its last phase at sample 0" candidates
// differ by a few ulps only, and with sigma < 0.4 (funcl > 0 there) one of them usually wins: the
// reference's choice then hangs on the rounding of T1 + lp in ITS frame, which the ring engine's
// per-sample constant shift cannot reproduce.  Re-deciding this one sample with the reference's
// own operations (strict '>', list order = ascending source) makes it exact.
__global__ void k_first_state(RingGeom g, const double *__restrict__ y, const double *__restrict__ mean,
                              const double *__restrict__ ctab, double A, double den,
                              int16_t *__restrict__ x)
{
    if (threadIdx.x != 0 || blockIdx.x != 0 || g.T < 2) return;
    const int N = g.N, L = g.L;
    const double *c0 = ctab + 1, *cend = ctab + 1 + N, *cx = ctab + 1 + 2 * N, *cint = ctab + 1 + 2 * N + N * N;
    const double y0 = y[0];
    auto T1 = [&](int a, int k) {  // funcl, utils.jl:4, as the strict engine writes it
        const double dd = y0 - mean[1 + a * L + (k - 1)];
        return A - (dd * dd) / den;
    };
    double best = -INFINITY;
    int arg = 1;
    auto cand = [&](int state, double t1, double lp) {
        const double tt = t1 + lp;
        if (tt > best) { best = tt; arg = state; }
    };
    const int x1 = x[1];
    if (x1 == 1) {
        cand(1, 0.0, ctab[0]);
        for (int a = 0; a < N; a++) cand(1 + a * L + L, T1(a, L), cend[a]);
    } else {
        const int b = (x1 - 2) / L, k = (x1 - 2) % L + 1;
        if (k == 1) {
            cand(1, 0.0, c0[b]);
            for (int a = 0; a < N; a++)
                if (a != b) cand(1 + a * L + L, T1(a, L), cx[a * N + b]);
        } else {
            cand(1 + b * L + (k - 1), T1(b, k - 1), cint[b * L + (k - 1)]);
        }
    }
    x[0] = (int16_t)arg;
}

// everything after the chain sweep: final state, backtrace + stitch, boundary certificate, x, ll
int ring_viterbi_post(RingDev *r, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st)
{
    const RingGeom &g = r->g;
    int rc;
    HS_HIP(hipMemsetAsync(r->redo, 0, sizeof(int32_t), st));
    { PROF(r, "k_vit_tail", st); hipLaunchKernelGGL(k_vit_tail, dim3(1), dim3(64), 0, st, g, r->Pv, r->D0end, r->final_state); }
    rc = dispatch_N(g.N, [&](auto n) {
        constexpr int N = decltype(n)::value;
        { PROF(r, "k_vit_backtrace", st); hipLaunchKernelGGL((k_vit_backtrace<N>), dim3(g.ncol / 64), dim3(64), 0, st, g, r->psi,
                           r->final_state, r->xT, r->bstate); }
        HS_HIP(hipGetLastError());
        return HMMSORT_OK;
    });
    if (rc) return rc;
    { PROF(r, "k_stitch_check", st); hipLaunchKernelGGL(k_stitch_check, dim3((g.nch + 255) / 256), dim3(256), 0, st, g, r->xT,
                       r->bstate, r->redo); }
    { PROF(r, "k_stitch_fix", st); hipLaunchKernelGGL(k_stitch_fix, dim3(1), dim3(64), 0, st, g, r->psi, r->xT, r->bstate, r->redo,
                       r->diag); }
    {
        KappaArg ka;
        ka.c00 = r->ring.c00; ka.mean0 = r->mean[0]; ka.den = r->den;
        for (int a = 0; a < g.N; a++) {
            double k = r->ring.cend[a] - r->ring.c00;
            for (int b = 0; b < g.N; b++)
                if (b != a) k = std::max(k, r->ring.cx[a * g.N + b] - r->ring.c0[b]);
            ka.kappa[a] = k;
        }
        PROF(r, "k_halo_check", st);
        hipLaunchKernelGGL(k_halo_check, dim3(g.ncol / 64), dim3(64 * kVChkParts), (size_t)g.L * 64 * sizeof(double), st,
                           g, ka, 1e-6, r->yT, r->Pv, r->D0pre, r->D0end, r->diag);
        HS_HIP(hipGetLastError());  // a certificate that did not launch must not read as one that passed
    }
    { PROF(r, "k_transpose_x", st); hipLaunchKernelGGL(k_transpose_x, dim3(g.B / 64, g.ncol / 64), dim3(256), 0, st, r->xT, g.T,
                       g.B, g.ncol, d_x); }
    { PROF(r, "k_first_state", st); hipLaunchKernelGGL(k_first_state, dim3(1), dim3(64), 0, st, g, d_y, r->d_mean, r->d_ctab,
                       r->A, r->den, d_x); }
    { PROF(r, "k_ll_partial", st); hipLaunchKernelGGL(k_ll_partial, dim3(r->nparts), dim3(256), 0, st, g, d_y, d_x, r->d_mean,
                       r->d_ctab, r->A, r->den, r->part); }
    { PROF(r, "k_sum_partials", st); hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, r->part, r->nparts, d_ll); }
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

}  // namespace hmmsort
