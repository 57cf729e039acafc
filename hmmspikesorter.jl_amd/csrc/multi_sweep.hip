// Blocked engine, structure-exploiting sweep for overlap models of THREE to FIVE templates (reference
// types.jl:65-127 with allow_overlaps = true; N = 4, K = 60 is the largest model the reference's CLI builds,
// hmmsort.jl:50-54: 21 123 states).  pair_sweep.hip is the two-template case; this file generalises its idea.
//
// isvalid_transition (types.jl:94-113) makes the overlap model a product of per-neuron chains (silent: lpz per
// step, start: lp_i, a ring of L = K-1 deterministic steps) restricted to the states generate_states builds: at
// most two neurons active.  With Z = silent, A_i(k) = only neuron i active at phase k, P(i:k1, j:k2) = both:
//   * a pair state with both phases >= 2 has ONE predecessor, P(i:k1-1, j:k2-1), at the constant (N-2) lpz: a pair
//     run is a pure delay whose emissions separate, (y-m0-a-b)^2 = (y-m0-a)^2 + (y-m0-b)^2 - (y-m0)^2 + 2ab
//     (a, b = deviations from the silent mean), so its total gain is a difference of the two neurons' prefix gains
//     minus a table;
//   * decisions are taken by Z, the singles and the pair ENTRIES only:
//       A_i(k)      <- A_i(k-1) | P(i:k-1, l:L), l != i             ("continue alone" | "partner l just ended")
//       P(i:k, y:1) <- A_i(k-1) | P(i:k-1, l:L), l != i, y          (y starts beside the continuing i)
//       Z, A_i(1), P(i:1, j:1) <- Z | A_l(L) | P(l:L, l':L)          (the junction: everything that just ended)
// One workgroup sweeps one block of the blocked engine with N + 1 wavefronts and one barrier per sample:
//   * wavefront i < N is neuron i's track, LANES ARE PHASES: A_i and the prefix gains shift by one lane per sample
//     (DPP); lane d takes the N-way decision for A_i(d+1) and the N-1 entry decisions P(i:d+1, y:1); the entries go
//     to a skewed FIFO in LDS (column d of track i holds L-d+1 samples) from which the younger neuron's track reads
//     them L-d samples later, when the run ends.  The N-1 entries of one lane differ only in which candidate is
//     excluded, so the FIFO keeps (best, second best, owner of the best) instead of N-1 values: 120 KB of LDS at
//     N = 4, L = 59 where N-1 values per slot would need 170 KB;
//   * wavefront N is the junction: lane = target (Z, A_i(1), P(i:1,j:1)), a loop over the 1 + N + N(N-1)/2 sources.
// Everything around the sweep is the blocked engine's (generic_blocked.hip): warm-up from a flat column (every pair
// state present at the start: runs that "entered" before the start are priced from a table for the first L
// samples), full trellis columns at the block boundaries for the certificate (pair states materialised from their
// entries), back-pointer rows of the multi-source states in T2c, exact backtrace, ll.  The arithmetic is not the
// reference's operation order (gains, prefix sums, ~1e-12): every decision whose margin is below
// thr = 16 (L+2) ulp(|T1|max) + 1e-10 carries a flag in bit 15 of its back-pointer, flags ON THE DECODED PATH are
// counted in diag[7], and hmmsort_viterbi then decodes with the generic sweep (pair_sweep.hip has the same contract).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_map>

#include "generic_dev.h"
#include "hmmsort_internal.h"
#include "wave_common.h"   // DPP lane moves

namespace hmmsort {

// ---- device table (doubles) -------------------------------------------------------------------
// [0] c00 (k_pair_tail)  [12] m0 (k_pair_mag)
constexpr int kMT_A = 16;                       // N x 64      a_i(k) at [i*64 + k-1]
constexpr int kMT_CAA = kMT_A + 5 * 64;         // 8           A_i(k) -> A_i(k+1)
constexpr int kMT_CXA = kMT_CAA + 8;            // 32          [i*4+p] P(i:k, l:L) -> A_i(k+1), l = partner p of i
constexpr int kMT_CCT = kMT_CXA + 32;           // 20 x 64     [(i*4+p)*64 + lane] total correction of the run that exits into lane
constexpr int kMT_CT = kMT_CCT + 20 * 64;       // 16 x 16     junction constants [target*16 + source], -inf = no transition
constexpr int kMT_J = kMT_CT + 256;             // 10 x 2      per family: CC0tot, -
constexpr int kMT_H0 = kMT_J + 32;              // 10 x 64     H0_f(m) + cPP_f at [f*64 + m], m = 0..L (diagonal runs)
constexpr int kMT_V0 = kMT_H0 + 10 * 64;        // 10 x 64     H0_f(m) at [f*64+m] (virtual entries of the diagonal runs)
constexpr int kMT_CAP = kMT_V0 + 10 * 64;       // 32          [i*4+q] A_i(k) -> P(i:k+1, y:1), y = partner q of i (exact first step)
constexpr int kMT_CXP = kMT_CAP + 32;           // 128         [(i*4+p)*4+q] P(i:k, l:L) -> P(i:k+1, y:1), l = partner p, y = partner q
constexpr int kMT_TBL = kMT_CXP + 128;          // N*N x 4096  [(o*N+y)*4096 + d*64 + m] = H_oy(d,m) - delta_oy
constexpr int kMT_CPP = 8;                      // [8..11]: -, cPP (one value: verified equal for all families)

struct MultiArgs {
    const double *y;
    int64_t T;
    int S, B, H, L, nms;
    const double *tab;
    const double *mean;    // [S] per-state means (the first decisions of a recording are taken in the reference's own arithmetic)
    double c0, den;
    int16_t *T2c;
    double *endv, *warmv;
    double *qsum;
};

static inline int fam_index(int N, int i, int j) { return i * N - i * (i + 1) / 2 + (j - i - 1); }   // i < j

// Host: is this transition list the N-template overlap pattern, with values uniform where the sweep assumes so?
bool multi_analyze(const HostModel &m, std::vector<double> &tab)
{
    const int N = (int)m.N;
    if (N < 3 || N > 5 || m.K < 3) return false;
    const int L = (int)m.K - 1, NP = N * (N - 1) / 2;
    if (L > 63 || m.S != 1 + (int64_t)N * L + (int64_t)NP * L * L) return false;
    const int64_t S = m.S;
    auto A = [&](int i, int k) { return (int64_t)1 + (int64_t)i * L + (k - 1); };
    auto P = [&](int i, int k1, int j, int k2) {
        if (i > j) { std::swap(i, j); std::swap(k1, k2); }
        return (int64_t)1 + (int64_t)N * L + (int64_t)fam_index(N, i, j) * L * L + (int64_t)(k1 - 1) * L + (k2 - 1);
    };
    // state table of generate_states(N, K, true) (1-based rows of mu)
    auto st = [&](int neuron, int64_t j) { return (int)m.states[neuron + (int64_t)N * j] - 1; };
    for (int n = 0; n < N; n++) if (st(n, 0) != 0) return false;
    for (int i = 0; i < N; i++)
        for (int k = 1; k <= L; k++)
            for (int n = 0; n < N; n++) if (st(n, A(i, k)) != (n == i ? k : 0)) return false;
    for (int i = 0; i < N; i++)
        for (int j = i + 1; j < N; j++)
            for (int k1 = 1; k1 <= L; k1++)
                for (int k2 = 1; k2 <= L; k2++) {
                    const int64_t s = P(i, k1, j, k2);
                    for (int n = 0; n < N; n++)
                        if (st(n, s) != (n == i ? k1 : (n == j ? k2 : 0))) return false;
                }
    std::unordered_map<uint64_t, double> lp;
    lp.reserve((size_t)m.R * 2);
    for (const auto &t : m.tr) lp[(uint64_t)(t.src - 1) * (uint64_t)S + (uint64_t)(t.dst - 1)] = t.lp;
    if ((int64_t)lp.size() != m.R) return false;
    bool ok = true;
    int64_t count = 0;
    auto get = [&](int64_t s, int64_t d) {
        auto it = lp.find((uint64_t)s * (uint64_t)S + (uint64_t)d);
        if (it == lp.end() || !std::isfinite(it->second)) { ok = false; return 0.0; }
        count++;
        return it->second;
    };
    const double kUnset = -1.2345e300;
    auto uni = [&](double &slot, double v) { if (slot == kUnset) slot = v; else if (!(slot == v)) ok = false; };
    std::vector<double> cAA(N, kUnset), cAP(N * N, kUnset), cXA(N * N, kUnset), cXP(N * N * N, kUnset);
    double cPP = kUnset;
    // junction: targets/sources 0 = Z, 1+i = A(i,1) / A(i,L), 1+N+f = P(i:1,j:1) / P(i:L,j:L)
    std::vector<double> ct(256, -INFINITY);
    std::vector<std::pair<int, int>> fam;
    for (int i = 0; i < N; i++) for (int j = i + 1; j < N; j++) fam.push_back({i, j});
    auto tgt_state = [&](int q) { return q == 0 ? (int64_t)0 : (q <= N ? A(q - 1, 1) : P(fam[q - N - 1].first, 1, fam[q - N - 1].second, 1)); };
    auto src_state = [&](int s) { return s == 0 ? (int64_t)0 : (s <= N ? A(s - 1, L) : P(fam[s - N - 1].first, L, fam[s - N - 1].second, L)); };
    auto uses = [&](int q, int n) { return q == 0 ? false : (q <= N ? q - 1 == n : (fam[q - N - 1].first == n || fam[q - N - 1].second == n)); };
    const int NT = 1 + N + NP;
    for (int q = 0; q < NT && ok; q++)
        for (int s = 0; s < NT && ok; s++) {
            bool valid = true;
            for (int n = 0; n < N; n++) if (uses(q, n) && uses(s, n)) valid = false;   // a neuron that just ended cannot restart
            if (L == 1) valid = false;
            if (valid) ct[q * 16 + s] = get(src_state(s), tgt_state(q));
        }
    for (int i = 0; i < N && ok; i++)
        for (int k = 1; k < L && ok; k++) {
            uni(cAA[i], get(A(i, k), A(i, k + 1)));
            for (int y = 0; y < N; y++)
                if (y != i) uni(cAP[i * N + y], get(A(i, k), P(i, k + 1, y, 1)));
        }
    for (int f = 0; f < NP && ok; f++)
        for (int k1 = 1; k1 < L && ok; k1++)
            for (int k2 = 1; k2 < L; k2++)
                uni(cPP, get(P(fam[f].first, k1, fam[f].second, k2), P(fam[f].first, k1 + 1, fam[f].second, k2 + 1)));
    for (int o = 0; o < N && ok; o++)
        for (int y = 0; y < N && ok; y++) {
            if (y == o) continue;
            for (int mm = 1; mm < L && ok; mm++) {
                const int64_t s = P(o, L, y, mm);
                uni(cXA[y * N + o], get(s, A(y, mm + 1)));
                for (int l = 0; l < N; l++)
                    if (l != o && l != y) uni(cXP[(y * N + o) * N + l], get(s, P(y, mm + 1, l, 1)));
            }
        }
    if (!ok || count != m.R) return false;
    // the entry constants must be "continue" constant + a per-(track, starter) shift, whatever the source
    std::vector<double> delta(N * N, 0.0);
    for (int i = 0; i < N; i++)
        for (int y = 0; y < N; y++) {
            if (y == i) continue;
            delta[i * N + y] = cAP[i * N + y] - cAA[i];
            for (int o = 0; o < N; o++)
                if (o != i && o != y && std::fabs((cXP[(i * N + o) * N + y] - cXA[i * N + o]) - delta[i * N + y]) > 1e-12) return false;
        }
    // means: pair state = silent mean + both deviations (up to the rounding of the neuron-order sums)
    const double m0 = m.mean[0], den = 2.0 * (m.sigma * m.sigma);
    std::vector<double> a((size_t)N * 64, 0.0);
    double scale = std::fabs(m0) + 1e-300;
    for (int i = 0; i < N; i++)
        for (int k = 1; k <= L; k++) {
            a[i * 64 + k - 1] = m.mean[A(i, k)] - m0;
            scale = std::max(scale, std::fabs(a[i * 64 + k - 1]));
        }
    for (int f = 0; f < NP; f++)
        for (int k1 = 1; k1 <= L; k1++)
            for (int k2 = 1; k2 <= L; k2++)
                if (std::fabs(m.mean[P(fam[f].first, k1, fam[f].second, k2)] -
                              (m0 + a[fam[f].first * 64 + k1 - 1] + a[fam[f].second * 64 + k2 - 1])) > 1e-12 * scale)
                    return false;
    tab.assign((size_t)kMT_TBL + (size_t)N * N * 4096, 0.0);
    tab[0] = ct[0]; tab[12] = m0; tab[kMT_CPP + 1] = cPP;
    for (int i = 0; i < N; i++) {
        for (int k = 0; k < 64; k++) tab[kMT_A + i * 64 + k] = a[i * 64 + k];
        tab[kMT_CAA + i] = cAA[i];
    }
    for (int i = 0; i < 256; i++) tab[kMT_CT + i] = ct[i];
    auto cc = [&](int o, int ko, int y, int ky) { return 2.0 * a[o * 64 + ko - 1] * a[y * 64 + ky - 1] / den; };
    // ordered runs: o older by d phases; H_oy(d, m) = sum_{i=1..m} cc(o:d+i, y:i) - m cPP
    for (int o = 0; o < N; o++)
        for (int y = 0; y < N; y++) {
            if (o == y) continue;
            double *T = &tab[(size_t)kMT_TBL + (size_t)(o * N + y) * 4096];
            for (int d = 1; d <= L - 1; d++) {
                double h = 0.0;
                for (int mm = 0; mm <= L - d; mm++) {
                    if (mm >= 1) h += cc(o, d + mm, y, mm) - cPP;
                    T[d * 64 + mm] = h - delta[o * N + y];
                }
            }
        }
    for (int i = 0; i < N; i++)
        for (int p = 0; p < N - 1; p++) {
            const int l = p + (p >= i);     // older partner
            tab[kMT_CXA + i * 4 + p] = cXA[i * N + l];
            tab[kMT_CAP + i * 4 + p] = cAP[i * N + l];
            for (int q = 0; q < N - 1; q++) {
                const int yq = q + (q >= i);
                if (q != p) tab[kMT_CXP + (i * 4 + p) * 4 + q] = cXP[(i * N + l) * N + yq];
            }
            const double *T = &tab[(size_t)kMT_TBL + (size_t)(l * N + i) * 4096];
            for (int lane = 1; lane <= L - 1; lane++) {
                const int d = L - lane;
                tab[kMT_CCT + (i * 4 + p) * 64 + lane] = T[d * 64 + (L - d)] + cPP;
            }
        }
    for (int f = 0; f < NP; f++) {
        double h = 0.0;
        for (int mm = 0; mm <= L; mm++) {
            if (mm >= 1) h += cc(fam[f].first, mm, fam[f].second, mm) - cPP;
            tab[kMT_H0 + f * 64 + mm] = h + cPP;
            tab[kMT_V0 + f * 64 + mm] = h;
        }
        tab[kMT_J + 2 * f] = tab[kMT_H0 + f * 64 + L];
    }
    return true;
}

// expected back-pointer rows: the multi-source states in state order
bool multi_rows_ok(const HostModel &m, const std::vector<int32_t> &ms_states)
{
    const int N = (int)m.N, L = (int)m.K - 1, NP = N * (N - 1) / 2;
    std::vector<int32_t> want;
    want.push_back(0);
    for (int j = 1; j <= N * L; j++) want.push_back(j);
    for (int f = 0; f < NP; f++) {
        const int base = 1 + N * L + f * L * L;
        for (int k2 = 1; k2 <= L; k2++) want.push_back(base + (k2 - 1));
        for (int k1 = 2; k1 <= L; k1++) want.push_back(base + (k1 - 1) * L);
    }
    return want == ms_states;
}

size_t multi_lds_bytes(int N, int L)
{
    const int NP = N * (N - 1) / 2;
    const size_t slots = (size_t)L * (L + 1) / 2 - 1;
    size_t dbl = (size_t)N * slots * 2 + (size_t)NP * 64 + 64 + 256 + 24 + (size_t)(N + 1) * 64 + (size_t)N * 128;
    return dbl * 8 + 64 * 4 + (size_t)N * slots + 64;
}

template <int N>
__global__ __launch_bounds__(64 * (N + 1)) void multi_vit_block(MultiArgs a)
{
    constexpr int NP = N * (N - 1) / 2, NT = 1 + N + NP;
    extern __shared__ double lds[];
    const int L = a.L, S = a.S;
    const int slots = L * (L + 1) / 2 - 1;
    double *FV = lds;                               // N x slots x 2: (best, second) - prefix gain at entry
    double *E0 = FV + (size_t)N * slots * 2;        // NP x 64: diagonal runs P(i:1,j:1), FIFO of L+1 samples
    double *PUB = E0 + NP * 64;                     // 2 parities x 32: dL[i] at 0.., sL[i] at 8.., A1[i] at 16.., Z at 24
    double *CT = PUB + 64;                          // junction constants
    double *SRC = CT + 256;                         // junction source values (+ one slot that idle lanes write)
    double *YB = SRC + 24;                          // N + 1 x 64 samples
    double *DS = YB + (N + 1) * 64;                 // dump scratch: d, s of every track
    int *CB = reinterpret_cast<int *>(DS + N * 128);   // CB[d] = first slot of column d (d = 1..L-1)
    unsigned char *FO = reinterpret_cast<unsigned char *>(CB + 64);   // N x slots: owner of the best (0xFF: the single)
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, c = blockIdx.x;
    const int k = lane + 1;
    const bool on = lane < L;
    const bool hx = on && lane >= 1;                // lanes with exits / entries: new phase k = 2..L
    const int64_t s = (int64_t)c * a.B;
    const int64_t e = (s + a.B < a.T) ? s + a.B : a.T;
    const int64_t w = (s - a.H > 0) ? s - a.H : 0;
    const double *tab = a.tab;
    const double m0 = tab[12], c00 = tab[0], cPP = tab[kMT_CPP + 1];
    const double rden = 1.0 / a.den;
    const double mmax = fabs(a.c0) * (double)a.T + a.qsum[0] * rden + fabs(c00) * (double)a.T + 1.0;
    const double thr = ldexp(16.0 * (double)(L + 2), ilogb(mmax) - 52) + 1e-10;
    auto famidx = [&](int i, int j) { return i * N - i * (i + 1) / 2 + (j - i - 1); };   // i < j
    auto smod = [&](int64_t v, int n) { int r = (int)(v % n); return r < 0 ? r + n : r; };

    if (tid < 64) {
        int b = 0;
        for (int d = 1; d < lane; d++) b += L - d + 1;
        CB[lane] = b;                                // CB[0], CB[1] = 0
    }
    for (int i = tid; i < 256; i += 64 * (N + 1)) CT[i] = tab[kMT_CT + i];
    if (tid < 16) SRC[tid] = 0.0;                   // (slots beyond the last source are read with a -inf constant: keep them finite)
    // virtual entries of the diagonal runs: the pair (m, m) "entered" m-1 samples before w
    for (int i = tid; i < NP * 64; i += 64 * (N + 1)) {
        const int f = i >> 6, mm = (i & 63) + 1;
        if (mm <= L) E0[f * 64 + smod(w - mm + 1, L + 1)] = tab[kMT_V0 + f * 64 + mm - 1];
    }
    __syncthreads();

    // ---- per-wave state ------------------------------------------------------------------
    const bool track = wv < N;
    const int ti = track ? wv : 0;                  // this wave's neuron
    double d = 0.0, sp = 0.0;                       // A_i(k), prefix gain s_i(k) of the previous sample (tracks)
    double Zv = 0.0;                                // silent state (junction wave, wave-uniform)
    const double ak = (track && on) ? tab[kMT_A + ti * 64 + lane] : 0.0;
    const double akr = ak * rden;
    const double cAA = track ? tab[kMT_CAA + ti] : 0.0;
    double cXA[N - 1], CCt[N - 1];
    int idx[N - 1];                                 // back-pointer (1-based state id) of candidate "partner p just ended"
    int rowP[N - 1];                                // T2c row of the entry P(i:k, y:1), y = partner p
#pragma unroll
    for (int p = 0; p < N - 1; p++) {
        const int l = p + (p >= ti);
        cXA[p] = track ? tab[kMT_CXA + ti * 4 + p] : 0.0;
        CCt[p] = (track && hx) ? tab[kMT_CCT + (ti * 4 + p) * 64 + lane] : 0.0;
        const int f = ti < l ? famidx(ti, l) : famidx(l, ti);
        const int pbase = 1 + N * L + f * L * L;
        // source P(i at k-1, l at L), 1-based id
        idx[p] = 1 + (ti < l ? pbase + (k - 2) * L + (L - 1) : pbase + (L - 1) * L + (k - 2));
        const int rbase = 1 + N * L + f * (2 * L - 1);
        rowP[p] = ti < l ? rbase + L + (k - 2) : rbase + (k - 1);
    }
    const int ida = 1 + ti * L + (k - 1);           // 1-based id of A_i(k-1) = state index of A_i(k-1) + 1 = 1 + i L + (k-2) + 1
    const int rowA = 1 + ti * L + (k - 1);          // T2c row of A_i(k)
    const int dex = L - lane;                       // column this lane reads (the run that exits into it)
    const int nw = L - lane + 1, nr = lane + 1;     // FIFO lengths of the column written (d = lane) / read (d = L - lane)
    const int wbase = hx ? CB[lane] : 0, rbase_ = hx ? CB[dex] : 0;
    // junction wave: lane = target / source q
    const int q = lane;
    const bool jon = !track && q < NT;
    int jf_i = 0, jf_j = 1;                          // family of target / source q > N
    if (q > N && q < NT) {
        int f = q - N - 1, i = 0;
        while (f >= N - 1 - i) { f -= N - 1 - i; i++; }
        jf_i = i; jf_j = i + 1 + f;
    }
    const double jcc0 = (jon && q > N) ? tab[kMT_J + 2 * (q - N - 1)] : 0.0;
    const int jsid = q == 0 ? 1 : (q <= N ? 1 + (q - 1) * L + L : 1 + N * L + (q - N - 1) * L * L + L * L);   // 1-based id of source q
    // target role: lane = 4 * target + source group; group g folds the sources g, g + 4, g + 8, (g + 12), then the four
    // lanes of a target combine (two quad-permute steps); the lane with g = 0 writes the target
    const int tq = lane >> 2, tg = lane & 3;
    const bool ton = !track && tq < NT;
    const double ta1 = (ton && tq >= 1 && tq <= N) ? tab[kMT_A + (tq - 1) * 64] : 0.0, ta1r = ta1 * rden;   // deviation of A_(tq-1)(1)
    const int trow = tq == 0 ? 0 : (tq <= N ? 1 + (tq - 1) * L : 1 + N * L + (tq - N - 1) * (2 * L - 1));
    constexpr int NG = (NT + 3) / 4;                 // sources per group
    auto sidof = [&](int sx) { return sx == 0 ? 1 : (sx <= N ? 1 + (sx - 1) * L + L : 1 + N * L + (sx - N - 1) * L * L + L * L); };

    // ---- the first column (flat start, viterbi.jl:55-63 / warm-up) -------------------------
    {
        const double u = a.y[w] - m0;
        if (track) {
            const double g = on ? ((2.0 * u - ak) * akr) : 0.0;
            d = g; sp = g;
            if (lane == L - 1) { PUB[(w & 1) * 32 + ti] = g; PUB[(w & 1) * 32 + 8 + ti] = g; }
            if (lane == 0) PUB[(w & 1) * 32 + 16 + ti] = g;
        } else {
            const double q0 = a.c0 - (u * u) * rden;
            Zv = (w == 0) ? -q0 : 0.0;              // T1[1,1] = 0 (viterbi.jl:63) in the frame that drops c0 + q0 per sample
        }
    }
    __syncthreads();

    // full trellis column (frame of this block) at sample t
    auto dump = [&](double *col, int64_t t) {
        const int par = (int)(t & 1);
        if (track) {
            const double dfix = lane == 0 ? PUB[par * 32 + 16 + ti] : d;
            DS[ti * 128 + lane] = dfix;
            DS[ti * 128 + 64 + lane] = sp;
            if (on) col[1 + ti * L + lane] = dfix;
        } else if (lane == 0) col[0] = Zv;
        __syncthreads();
        for (int ix = tid; ix < NP * L * L; ix += 64 * (N + 1)) {
            const int f = ix / (L * L), r = ix - f * L * L, k1 = r / L + 1, k2 = r - (k1 - 1) * L + 1;
            int i = 0, ff = f;
            while (ff >= N - 1 - i) { ff -= N - 1 - i; i++; }
            const int j = i + 1 + ff;
            const double si = DS[i * 128 + 64 + k1 - 1], sj = DS[j * 128 + 64 + k2 - 1];
            const int dd = k1 - k2;
            double v;
            if (dd == 0) {
                v = ((E0[f * 64 + smod(t - k1 + 1, L + 1)] + si) + sj) - tab[kMT_H0 + f * 64 + k1];
            } else {
                const int o = dd > 0 ? i : j, yy = dd > 0 ? j : i, dc = dd > 0 ? dd : -dd, mm = dd > 0 ? k2 : k1;
                const int64_t tau = t - mm + 1;
                const double *TB = tab + kMT_TBL + (size_t)(o * N + yy) * 4096 + dc * 64;
                double ev;
                if (tau <= w) ev = TB[mm - 1 - (int)(t - w)];            // run older than the block: priced from the table
                else {
                    const int sl = o * slots + CB[dc] + smod(tau, L - dc + 1);
                    ev = FO[sl] == yy ? FV[2 * sl + 1] : FV[2 * sl];
                }
                v = ((ev + si) + sj) - (TB[mm] + cPP);
            }
            col[1 + N * L + ix] = v;
        }
        __syncthreads();
    };

    // junction constants of this lane's target: one register row instead of an LDS read per source and sample
    double ctr[NG];
#pragma unroll
    for (int j = 0; j < NG; j++) ctr[j] = (tq < NT && tg + 4 * j < NT) ? CT[tq * 16 + tg + 4 * j] : -INFINITY;
    // first maximum in list order (lowest source index among equal values), runner-up, arg: this lane's sources, then
    // the four lanes of the target
    auto jdecide = [&](double &best, double &sec, int &arg) {
        best = -INFINITY; sec = -INFINITY; arg = tg;
#pragma unroll
        for (int j = 0; j < NG; j++) {
            const double v = SRC[(tg + 4 * j) & 15] + ctr[j];
            const bool gt = v > best;
            sec = fmax(sec, fmin(best, v));
            best = fmax(best, v);
            arg = gt ? tg + 4 * j : arg;
        }
#define HS_JQ(CTRL)                                                                              \
        {                                                                                        \
            const double ob = dpp_mov<CTRL, 0xF>(best, best), os = dpp_mov<CTRL, 0xF>(sec, sec); \
            const int oi = __builtin_amdgcn_update_dpp(arg, arg, CTRL, 0xF, 0xF, false);         \
            const bool take = ob > best || (ob == best && oi < arg);                             \
            sec = fmax(fmax(os, sec), fmin(ob, best));                                           \
            best = fmax(ob, best);                                                               \
            arg = take ? oi : arg;                                                               \
        }
        HS_JQ(0xB1) HS_JQ(0x4E)                      // quad_perm [1,0,3,2], [2,3,0,1]
#undef HS_JQ
    };
    int e0w = smod(w + 1, L + 1);                   // slot of the diagonal-run FIFOs written at the current sample: t mod (L+1)
    double *myY = YB + wv * 64;
    auto ychunk = [&](int64_t tc) { const int64_t tt = tc + lane; return a.y[tt < e ? tt : e - 1]; };
    double ynxt = ychunk(w + 1);
    int16_t *psi = a.T2c + (int64_t)a.nms * (w + 1);    // back-pointer row of the current sample
    int wpos = hx ? smod(w + 1, nw) : 0, rpos = hx ? smod(w + 1 - lane, nr) : 0;
    for (int64_t tc = w + 1; tc < e; tc += 64) {
      myY[lane] = ynxt;
      ynxt = ychunk(tc + 64 < e ? tc + 64 : tc);
      const int nstep = (e - tc) < 64 ? (int)(e - tc) : 64;
      for (int si = 0; si < nstep; si++) {
        const int64_t t = tc + si;
        if (t == s && s > 0) dump(a.warmv + (int64_t)c * S, t - 1);   // (contains barriers; every wave gets here)
        const double u = myY[si] - m0;
        const bool own = t >= s;
        const int pp = (int)((t - 1) & 1), par = (int)(t & 1);
        const int jj = (int)(t - w);               // steps since the start: runs read in the first L steps may be older than the block
        if (track) {
            const double g = (2.0 * u - ak) * akr;
            const double A1 = PUB[pp * 32 + 16 + ti];
            d = lane == 0 ? A1 : d;
            const double pd = lane_prev(d, 0.0), ps = lane_prev(sp, 0.0);
            double bb[N];
            bb[0] = pd + cAA;
#pragma unroll
            for (int p = 0; p < N - 1; p++) {
                const int l = p + (p >= ti);
                const double sLl = PUB[pp * 32 + 8 + l];
                const int sl = l * slots + rbase_ + rpos;
                double ev = FO[sl] == ti ? FV[2 * sl + 1] : FV[2 * sl];
                if (jj <= L) {                       // wave-uniform: the only global loads of the sweep
                    const double tv = tab[kMT_TBL + (size_t)(l * N + ti) * 4096 + (hx ? dex : 1) * 64 + (lane >= jj ? lane - jj : 0)];
                    ev = lane >= jj ? tv : ev;
                }
                const double xv = ((ev + sLl) + ps) - CCt[p];
                bb[p + 1] = hx ? xv + cXA[p] : -INFINITY;
            }
            // A_i(k): first maximum in list order (single, then partners ascending), margin over the runner-up
            double best = bb[0], sec = -INFINITY;
            int o1 = 0;
#pragma unroll
            for (int cnd = 1; cnd < N; cnd++) {
                const bool gt = bb[cnd] > best;
                sec = fmax(sec, fmin(best, bb[cnd]));
                best = fmax(best, bb[cnd]);
                o1 = gt ? cnd : o1;
            }
            int argA = ida;
#pragma unroll
            for (int p = 0; p < N - 1; p++) argA = o1 == p + 1 ? idx[p] : argA;
            argA |= (best - sec) < thr ? 0x8000 : 0;
            // entries P(i:k, y:1): the same candidates without "y just ended"
            int argP[N - 1];
#pragma unroll
            for (int qy = 0; qy < N - 1; qy++) {
                double wn = bb[0], rn = -INFINITY;
                int ow = ida;
#pragma unroll
                for (int p = 0; p < N - 1; p++) {
                    if (p == qy) continue;
                    const bool gt = bb[p + 1] > wn;
                    rn = fmax(rn, fmin(wn, bb[p + 1]));
                    wn = fmax(wn, bb[p + 1]);
                    ow = gt ? idx[p] : ow;
                }
                argP[qy] = ow | ((wn - rn) < thr ? 0x8000 : 0);
            }
            if (w == 0 && t == 1) {
                // The first decisions of a recording compare emission-only values (viterbi.jl:55-63), and states whose
                // deviation is ~0 (the last phases of every template) tie there to the last bit: these decisions are
                // taken in the reference's own arithmetic, T1[k,1] + lp with strict '>' in list order (viterbi.jl:74-84).
                const double y0 = a.y[0];
                double ex[N];
                {
                    const double dd = y0 - a.mean[hx ? ida - 1 : 0];
                    ex[0] = a.c0 - (dd * dd) / a.den;
                }
#pragma unroll
                for (int p = 0; p < N - 1; p++) {
                    const double dd = y0 - a.mean[hx ? idx[p] - 1 : 0];
                    ex[p + 1] = a.c0 - (dd * dd) / a.den;
                }
                {
                    double bv = ex[0] + cAA;
                    argA = ida;
#pragma unroll
                    for (int p = 0; p < N - 1; p++) {
                        const double v = ex[p + 1] + cXA[p];
                        const bool gt = v > bv;
                        bv = gt ? v : bv;
                        argA = gt ? idx[p] : argA;
                    }
                }
#pragma unroll
                for (int qy = 0; qy < N - 1; qy++) {
                    double bv = ex[0] + tab[kMT_CAP + ti * 4 + qy];
                    int ow = ida;
#pragma unroll
                    for (int p = 0; p < N - 1; p++) {
                        if (p == qy) continue;
                        const double v = ex[p + 1] + tab[kMT_CXP + (ti * 4 + p) * 4 + qy];
                        const bool gt = v > bv;
                        bv = gt ? v : bv;
                        ow = gt ? idx[p] : ow;
                    }
                    argP[qy] = ow;
                }
            }
            if (own && hx) {
                psi[rowA] = (int16_t)argA;
#pragma unroll
                for (int qy = 0; qy < N - 1; qy++) psi[rowP[qy]] = (int16_t)argP[qy];
            }
            if (hx) {
                const int sl = ti * slots + wbase + wpos;
                FV[2 * sl] = best - ps;
                FV[2 * sl + 1] = sec - ps;
                FO[sl] = (unsigned char)(o1 == 0 ? 0xFF : (o1 - 1) + ((o1 - 1) >= ti));
            }
            const double nd = best + g;
            d = nd;
            sp = ps + g;
            if (lane == L - 1) { PUB[par * 32 + ti] = nd; PUB[par * 32 + 8 + ti] = sp; }
            wpos = wpos + 1 == nw ? 0 : wpos + 1;
            rpos = rpos + 1 == nr ? 0 : rpos + 1;
        } else {
            // junction: sources at t-1
            double sv = Zv;
            if (q >= 1 && q <= N) sv = PUB[pp * 32 + (q - 1)];
            if (q > N && q < NT)
                sv = ((E0[(q - N - 1) * 64 + (e0w == L ? 0 : e0w + 1)] + PUB[pp * 32 + 8 + jf_i]) + PUB[pp * 32 + 8 + jf_j]) - jcc0;   // entered at t - L
            // (stores without a lane predicate: under "if (q < NT)" hipcc sank the loads of the reduction below into the
            // predicated region, and the lanes beyond kept stale operands for the cross-lane steps)
            SRC[q < NT ? q : 16] = sv;
            double best, sec;
            int arg;
            jdecide(best, sec, arg);
            int flag = (best - sec) < thr ? 0x8000 : 0;
            const double res = best + (2.0 * u - ta1) * ta1r;      // (gain of A_i(1); zero deviation elsewhere)
            if (w == 0 && t == 1) {                                // exact first decisions (see the tracks)
                const double dd = a.y[0] - a.mean[jon ? jsid - 1 : 0];
                SRC[q < NT ? q : 16] = q == 0 ? 0.0 : a.c0 - (dd * dd) / a.den;    // T1[1,1] = 0 (viterbi.jl:63)
                double bx, sx;
                jdecide(bx, sx, arg);
                flag = 0;
            }
            if (ton && tg == 0) {
                if (own) psi[trow] = (int16_t)(sidof(arg) | flag);
                if (tq >= 1 && tq <= N) PUB[par * 32 + 16 + (tq - 1)] = res;
                if (tq > N) E0[(tq - N - 1) * 64 + e0w] = res;
            }
            Zv = wave_bcast(res, 0);
            e0w = e0w == L ? 0 : e0w + 1;
        }
        psi += a.nms;
        __syncthreads();
      }
    }
    dump(a.endv + (int64_t)c * S, e - 1);
}

int multi_sweep_launch(GenericDev *g, const double *d_y, hipStream_t st)
{
    MultiArgs a;
    a.y = d_y; a.T = g->T; a.S = (int)g->S; a.B = (int)g->B; a.H = (int)g->H; a.L = (int)g->K - 1; a.nms = g->nms;
    a.tab = g->d_pairtab;
    a.mean = g->d_mean;
    a.c0 = -kLog2Pi - g->lsig;
    a.den = 2.0 * (g->sigma * g->sigma);
    a.T2c = g->d_T2; a.endv = g->d_endv; a.warmv = g->d_warmv; a.qsum = g->d_qsum;
    int rc = pair_mag_launch(g, d_y, st);      // sum (y - m0)^2 -> qsum[0] (pair_sweep.hip)
    if (rc) return rc;
    const int N = (int)g->N, L = (int)g->K - 1;
    const size_t lds = multi_lds_bytes(N, L);
    auto go = [&](auto kern) -> int {
        HS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)g->nblk), dim3(64 * (N + 1)), lds, st, a);
        HS_HIP(hipGetLastError());
        return HMMSORT_OK;
    };
    if (N == 3) return go(multi_vit_block<3>);
    if (N == 4) return go(multi_vit_block<4>);
    if (N == 5) return go(multi_vit_block<5>);
    set_error("multi sweep: %d templates", N);
    return HMMSORT_EUNSUP;
}

}  // namespace hmmsort
