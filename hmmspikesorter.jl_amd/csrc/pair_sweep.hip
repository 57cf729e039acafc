// Blocked engine, structure-exploiting sweep for overlap models of TWO templates (reference
// types.jl:65-127 with allow_overlaps = true, N = 2: the model of the reference's own Viterbi test,
// test/runtests.jl:17-34, 3600 states at K = 60).
//
// isvalid_transition (types.jl:94-113) makes the overlap model a PRODUCT of per-neuron chains: silent
// (lpz per step), start (lp_i), a ring of L = K-1 deterministic steps, restricted to states that exist.  With
// A_k / B_k = "only neuron 1 / 2 active, phase k" and P(k1,k2) = both active:
//   * a pair state has exactly one predecessor: P(k1,k2) <- P(k1-1,k2-1), P(k+1,1) <- A_k, P(1,k+1) <- B_k,
//     P(1,1) <- silent, all inside the pair lattice with log-probability 0.  A pair run is therefore a pure
//     DELAY: what enters at P(k+1,1) leaves at P(L,L-k) after L-k-1 steps, having collected the emissions on
//     its diagonal -- and those separate: with means as deviations a, b from the silent mean,
//         (y-m0-a-b)^2 = (y-m0-a)^2 + (y-m0-b)^2 - (y-m0)^2 + 2ab,
//     so in gains g = q - q_silent a pair state's gain is gA + gB - 2ab/den, the last term a table;
//   * only the 2L+1 states silent, A_k, B_k take decisions: A_k <- {A_(k-1), P(k-1,L)} ("continue alone" or
//     "the partner just ended"), A_1 <- {silent, B_L}, silent <- {silent, A_L, B_L, P(L,L)}.
// A wavefront whose LANES ARE PHASES advances one sample per step: the single tracks dA, dB and the prefix
// gains sA[k](t) = sum_{j<=k} gA_j(t-k+j) shift by one lane (DPP), the pair entry that exits into lane k is read
// from a skewed L x L buffer in LDS (written by lane L-k+1 of the other track k-1 steps earlier), and a pair
// run's total gain is sA[L] - sA[k] + sB[L-k] - CC(k) with CC the table of summed 2ab/den along the diagonal.
// ~90 instructions per sample instead of 3600 x 13.
//
// Everything around the sweep is the blocked engine's (generic_blocked.hip): blocks with a warm-up from a flat
// column, full trellis columns at the block boundaries for the certificate (pair states are materialised from
// their entries there), back-pointers of the multi-source states in T2c, exact backtrace, ll.  The arithmetic
// differs from the reference's operation order (gains, prefix sums) by ~1e-12, so every decision whose margin is
// below thr = 16 (L+2) ulp(|T1|max) + 1e-10 -- the worst case of the reference's own rounding for paths apart for up
// to 8 (L+2) steps (wave_viterbi.hip) plus this sweep's noise -- carries a flag in bit 15 of its back-pointer, and
// the flagged decisions ON THE DECODED PATH are counted in diag[7]; hmmsort_viterbi then decodes with the generic
// blocked sweep (the reference's operation order) and, should that flag too, with the strict engine.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>

#include "generic_dev.h"
#include "hmmsort_internal.h"
#include "wave_common.h"   // DPP lane moves

namespace hmmsort {

constexpr int kPairTab = 16 + 4 * 64 + 2 * 64 * 64 + 64;   // doubles in the device table (layout below)
// table layout: [0..15] c00 c0A c0B c0P cAA cAP cA0 cAB cBB cBP cB0 cBA m0 CC0 - -
//   [16..79] a[k] (k = 1..L at index k-1), [80..143] b, [144..207] CCab[d], [208..271] CCba[d]  (d = 1..L-1)
//   [272 .. 272+4095] CCpartAB[d*64+n], then CCpartBA[d*64+n], then CCpart0[n]
constexpr int kPT_A = 16, kPT_B = 80, kPT_CCAB = 144, kPT_CCBA = 208, kPT_PAB = 272, kPT_PBA = 272 + 4096,
              kPT_P0 = 272 + 8192;

struct PairArgs {
    const double *y;
    int64_t T;
    int S, B, H, L, nms;
    const double *tab;
    const double *mean;    // [S] per-state means (first decisions of a recording: reference arithmetic)
    double c0, den;
    int16_t *T2c;
    double *endv, *warmv;
    double *qsum;          // [0]: sum over the recording of (y - m0)^2, for the near-tie threshold
};

// Host: is this transition list the two-template overlap pattern, with the values the sweep assumes?
bool pair_analyze(const HostModel &m, std::vector<double> &tab)
{
    if (m.N != 2 || m.K < 3) return false;
    const int L = (int)m.K - 1;
    if (L > 63 || m.S != 1 + 2 * (int64_t)L + (int64_t)L * L || m.R != (int64_t)(L + 2) * (L + 2)) return false;
    // state table of generate_states(2, K, true) (1-based rows of mu)
    auto st = [&](int neuron, int64_t j) { return (int)m.states[neuron + 2 * j] - 1; };
    if (st(0, 0) != 0 || st(1, 0) != 0) return false;
    for (int k = 1; k <= L; k++) {
        if (st(0, k) != k || st(1, k) != 0 || st(0, L + k) != 0 || st(1, L + k) != k) return false;
        for (int k2 = 1; k2 <= L; k2++) {
            const int64_t j = 2 * L + (int64_t)(k - 1) * L + k2;
            if (st(0, j) != k || st(1, j) != k2) return false;
        }
    }
    std::map<std::pair<int64_t, int64_t>, double> lp;
    for (const auto &t : m.tr) lp[{t.src - 1, t.dst - 1}] = t.lp;
    if ((int64_t)lp.size() != m.R) return false;
    auto A = [&](int k) { return (int64_t)k; };
    auto Bs = [&](int k) { return (int64_t)L + k; };
    auto P = [&](int k1, int k2) { return (int64_t)2 * L + (int64_t)(k1 - 1) * L + k2; };
    bool ok = true;
    auto get = [&](int64_t s, int64_t d) {
        auto it = lp.find({s, d});
        if (it == lp.end() || !std::isfinite(it->second)) { ok = false; return 0.0; }
        return it->second;
    };
    auto same = [&](double v, double ref) { if (!(v == ref)) ok = false; };
    const double c00 = get(0, 0), c0A = get(0, A(1)), c0B = get(0, Bs(1)), c0P = get(0, P(1, 1));
    const double cAA = L > 1 ? get(A(1), A(2)) : 0.0, cAP = L > 1 ? get(A(1), P(2, 1)) : 0.0;
    const double cBB = L > 1 ? get(Bs(1), Bs(2)) : 0.0, cBP = L > 1 ? get(Bs(1), P(1, 2)) : 0.0;
    const double cA0 = get(A(L), 0), cAB = get(A(L), Bs(1)), cB0 = get(Bs(L), 0), cBA = get(Bs(L), A(1));
    for (int k = 1; k < L && ok; k++) {
        same(get(A(k), A(k + 1)), cAA); same(get(A(k), P(k + 1, 1)), cAP);
        same(get(Bs(k), Bs(k + 1)), cBB); same(get(Bs(k), P(1, k + 1)), cBP);
        same(get(P(L, k), Bs(k + 1)), 0.0); same(get(P(k, L), A(k + 1)), 0.0);   // the partner ends: log-probability 0
        for (int k2 = 1; k2 < L && ok; k2++) same(get(P(k, k2), P(k + 1, k2 + 1)), 0.0);
    }
    if (ok) same(get(P(L, L), 0), 0.0);
    if (!ok) return false;
    // means: pair state = silent mean + both deviations (up to the rounding of the neuron-order sums)
    const double m0 = m.mean[0], den = 2.0 * (m.sigma * m.sigma);
    std::vector<double> a(L + 1, 0.0), b(L + 1, 0.0);
    double scale = std::fabs(m0) + 1e-300;
    for (int k = 1; k <= L; k++) {
        a[k] = m.mean[A(k)] - m0; b[k] = m.mean[Bs(k)] - m0;
        scale = std::max(scale, std::max(std::fabs(a[k]), std::fabs(b[k])));
    }
    for (int k1 = 1; k1 <= L; k1++)
        for (int k2 = 1; k2 <= L; k2++)
            if (std::fabs(m.mean[P(k1, k2)] - (m0 + a[k1] + b[k2])) > 1e-12 * scale) return false;
    tab.assign(kPairTab, 0.0);
    const double cs[14] = {c00, c0A, c0B, c0P, cAA, cAP, cA0, cAB, cBB, cBP, cB0, cBA, m0, 0.0};
    for (int i = 0; i < 14; i++) tab[i] = cs[i];
    auto cc = [&](int k1, int k2) { return 2.0 * a[k1] * b[k2] / den; };
    for (int k = 1; k <= L; k++) { tab[kPT_A + k - 1] = a[k]; tab[kPT_B + k - 1] = b[k]; }
    // A ahead by d (entered from A_d into P(d+1,1)): terms cc(d+1+i, 1+i), i = 0..L-d-1; CCpart(d,n) = first n terms
    for (int d = 0; d <= L - 1; d++) {
        double sab = 0.0, sba = 0.0;
        for (int n = 0; n <= L - d; n++) {
            if (d >= 1) { tab[kPT_PAB + d * 64 + n] = sab; tab[kPT_PBA + d * 64 + n] = sba; }
            else tab[kPT_P0 + n] = sab;
            if (n < L - d) { sab += cc(d + 1 + n, 1 + n); sba += cc(1 + n, d + 1 + n); }
        }
        if (d >= 1) { tab[kPT_CCAB + d] = sab; tab[kPT_CCBA + d] = sba; }
        else tab[13] = sab;
    }
    return true;
}

__device__ __forceinline__ double pair_bcast(double v, int lane) { return wave_bcast(v, lane); }

// Skewed entry buffer: an entry of column d (written by lane d of the track that is ahead by d phases) is read
// L - d steps later, so column d is a FIFO of pairfifo(L, d) = L - d + 1 slots (column 0, silent -> P(1,1): L + 1)
// addressed by running positions that wrap at the column's own length.  Columns are packed one after the other:
// 1 829 slots per direction at L = 59 instead of a 64 x 64 square (or 2 474 with power-of-two columns and masks) --
// 29 KB of LDS per wavefront, i.e. five wavefronts per CU.
__host__ __device__ inline int pairfifo(int L, int d) { return d == 0 ? L + 1 : L - d + 1; }
__host__ __device__ inline int pairbase(int L, int d)   // slots of the columns before d (column 0: silent -> P(1,1))
{
    int b = 0;
    for (int q = 0; q < d; q++) b += pairfifo(L, q);
    return b;
}

__global__ __launch_bounds__(64) void pair_vit_block(PairArgs a)
{
    extern __shared__ double eb[];            // EAB columns 0..L-1 | EBA columns 1..L-1 | 64 samples | column bases
    const int nAB = pairbase(a.L, a.L), nBA = nAB - pairfifo(a.L, 0), n0 = pairfifo(a.L, 0);
    auto smod = [](int64_t v, int n) { const int r = (int)(v % n); return r < 0 ? r + n : r; };
    double *EAB = eb, *EBA = eb + nAB - pairfifo(a.L, 0);          // EBA[pairbase(d)] is valid for d >= 1
    double *YB = eb + nAB + nBA;
    int *CB = reinterpret_cast<int *>(YB + 64);                    // CB[d] = base of column d, CB[64 + d] = its length
    const int lane = threadIdx.x, c = blockIdx.x, L = a.L, S = a.S;
    const int k = lane + 1;                   // this lane's phase
    const bool on = lane < L;
    const int64_t s = (int64_t)c * a.B;
    const int64_t e = (s + a.B < a.T) ? s + a.B : a.T;
    const int64_t w = (s - a.H > 0) ? s - a.H : 0;
    const double *tab = a.tab;
    const double c00 = tab[0], c0A = tab[1], c0B = tab[2], c0P = tab[3], cAA = tab[4], cAP = tab[5], cA0 = tab[6],
                 cAB = tab[7], cBB = tab[8], cBP = tab[9], cB0 = tab[10], cBA = tab[11], m0 = tab[12], CC0 = tab[13];
    const double rden = 1.0 / a.den;
    // near-tie threshold: magnitude the reference's trellis reaches on this recording (as wave_thr)
    const double mmax = fabs(a.c0) * (double)a.T + a.qsum[0] * rden + fabs(c00) * (double)a.T + 1.0;
    const double thr = ldexp(16.0 * (double)(L + 2), ilogb(mmax) - 52) + 1e-10;
    const double ak = on ? tab[kPT_A + lane] : 0.0, bk = on ? tab[kPT_B + lane] : 0.0;
    const double akr = ak * rden, bkr = bk * rden;
    // the pair run that exits into this lane entered with d = L - lane (lanes 1..L-1)
    const int dex = L - lane;
    const bool has_exit = on && lane >= 1;
    const double CCab_r = has_exit ? tab[kPT_CCAB + dex] : 0.0, CCba_r = has_exit ? tab[kPT_CCBA + dex] : 0.0;
    // state ids (1-based, as the back-pointers store them)
    const int idA = 1 + k, idB = 1 + L + k;
    const int idPA = 1 + 2 * L + (k - 2) * L + L;          // P(k-1, L) -> A_k
    const int idPB = 1 + 2 * L + (L - 1) * L + (k - 1);    // P(L, k-1) -> B_k
    const int idAL = 1 + L, idBL = 1 + 2 * L, idPLL = 1 + 2 * L + L * L;

    for (int i = lane; i < nAB + nBA; i += 64) eb[i] = 0.0;
    if (lane < L) { CB[lane] = pairbase(L, lane); CB[64 + lane] = pairfifo(L, lane); }
    __syncthreads();
    // this lane's columns: it WRITES column k (entries of the pair runs it starts), it READS column L - lane
    const int wcol = (on && k <= L - 1) ? CB[k] : 0, nwr = (on && k <= L - 1) ? CB[64 + k] : 1;
    const int rcol = has_exit ? CB[dex] : 0, nrd = has_exit ? CB[64 + dex] : 1;
    const int rposb = has_exit ? rcol : CB[1];      // EBA has no column 0
    // per-lane constants of the two-way decisions: first candidate's log-probability, ids of both candidates
    const double c1A_r = lane == 0 ? c0A : cAA, c1B_r = lane == 0 ? c0B : cBB;
    const int id1A = lane == 0 ? 1 : idA - 1, id2A = lane == 0 ? idBL : idPA;
    const int id1B = lane == 0 ? 1 : idB - 1, id2B = lane == 0 ? idAL : idPB;
    // virtual entries: every pair state is present in the first column with its emission (viterbi.jl:55-63; the
    // flat start of a warm-up alike).  The pair (d + n + 1, n + 1) "entered" n samples before w; its entry holds
    // the part of CC that lies before w, so that exits and materialised columns come out right.
    {
        const int d = lane;
        for (int n = 0; n < L; n++) {
            if (d == 0) EAB[smod(w - n, n0)] = tab[kPT_P0 + n];
            else if (d <= L - 1 - n) {
                const int pos = CB[d] + smod(w - n, CB[64 + d]);
                EAB[pos] = tab[kPT_PAB + d * 64 + n];
                EBA[pos] = tab[kPT_PBA + d * 64 + n];
            }
        }
    }
    __syncthreads();
    double D0, dA, dB, sA, sB;
    {
        const double u = a.y[w] - m0;
        const double gA = on ? ((2.0 * u - ak) * akr) : 0.0, gB = on ? ((2.0 * u - bk) * bkr) : 0.0;
        dA = gA; dB = gB; sA = gA; sB = gB;   // (idle lanes: zero deviations, zero gains)
        const double q0 = a.c0 - (u * u) * rden;
        D0 = (w == 0) ? -q0 : 0.0;            // T1[1,1] = 0 (viterbi.jl:63) in the frame that drops c0 + q0 per sample
    }
    // full trellis column (frame of this block) at the current sample: singles from the registers, pair states from
    // their entries
    auto dump = [&](double *col, int64_t t) {
        if (lane == 0) col[0] = D0;
        if (on) { col[k] = dA; col[L + k] = dB; }
        for (int k2 = 1; k2 <= L; k2++) {
            const double sBk2 = pair_bcast(sB, k2 - 1);
            if (on) {
                const int d = k - k2;
                double v;
                if (d > 0) v = EAB[CB[d] + smod(t - k2 + 1, CB[64 + d])] - tab[kPT_PAB + d * 64 + k2];
                else if (d == 0) v = EAB[smod(t - k2 + 1, n0)] - tab[kPT_P0 + k2];
                else v = EBA[CB[-d] + smod(t - k + 1, CB[64 - d])] - tab[kPT_PBA + (-d) * 64 + k];
                col[2 * L + (k - 1) * L + k2] = (v + sA) + sBk2;
            }
        }
    };
    // samples reach the wavefront 64 at a time through LDS (the next 64 are in flight meanwhile) and a step reads its
    // sample as an LDS broadcast: a register filled by a global load would make every step wait on vmcnt, i.e. for
    // the previous step's back-pointer stores to reach L2 (measured: 0.55 us per step instead of ~0.15)
    auto ychunk = [&](int64_t tc) { const int64_t ti = tc + lane; return a.y[ti < e ? ti : e - 1]; };
    double ynxt = ychunk(w + 1);
    // running slots of the sample about to be processed (t = w + 1): this lane's write / read column, column 0
    int wp = smod(w + 1, nwr), rp = smod(w + 1 - lane, nrd), p0w = smod(w + 1, n0);
    for (int64_t tc = w + 1; tc < e; tc += 64) {
      YB[lane] = ynxt;
      ynxt = ychunk(tc + 64 < e ? tc + 64 : tc);
      const int nstep = (e - tc) < 64 ? (int)(e - tc) : 64;
      for (int si = 0; si < nstep; si++) {
        const int64_t t = tc + si;
        const double yt = YB[si];
        const bool own = t >= s;
        if (t == s && s > 0) {                 // the column of sample s-1 is what the boundary certificate compares
            __syncthreads();
            dump(a.warmv + (int64_t)c * S, t - 1);
        }
        const double u = yt - m0;
        // (idle lanes beyond phase L compute along with zero deviations; nothing reads them: shifts move values to
        // HIGHER lanes only and the broadcasts read lane L-1)
        const double gA = (2.0 * u - ak) * akr, gB = (2.0 * u - bk) * bkr;
        // values of the previous sample: lane k-1 (lane 0: the silent state), lane L
        const double pdA = lane_prev(dA, D0), pdB = lane_prev(dB, D0);
        const double psA = lane_prev(sA, 0.0), psB = lane_prev(sB, 0.0);
        const double dAL = pair_bcast(dA, L - 1), dBL = pair_bcast(dB, L - 1);
        const double sAL = pair_bcast(sA, L - 1), sBL = pair_bcast(sB, L - 1);
        // pair runs that end now (entered lane steps ago by lane L-k+1 of the other track)
        const int rpos = rcol + rp;                                // (lanes without an exit read slot 0 of column 0: unused)
        const double xa = ((EBA[rposb + rp] + sBL) + psA) - CCba_r;      // P(k-1, L) at t-1
        const double xb = ((EAB[rpos] + sAL) + psB) - CCab_r;      // P(L, k-1) at t-1
        const double xp = ((EAB[p0w == L ? 0 : p0w + 1] + sAL) + sBL) - CC0;   // P(L, L) at t-1 (wave-uniform; entered at t - L)
        // entries of this sample (from the previous sample's singles)
        if (on && k <= L - 1) {
            const int wpos = wcol + wp;
            EAB[wpos] = (dA + cAP) - sA; EBA[wpos] = (dB + cBP) - sB;
        }
        if (lane == 0) EAB[p0w] = D0 + c0P;
        // decisions, list order = source index ascending, strict '>' (viterbi.jl:74-84): A_k <- {A_(k-1), P(k-1,L)},
        // A_1 <- {silent, B_L}, B alike (the first candidate's constant and both ids are per-lane registers)
        const double c1a = pdA + c1A_r, c1b = pdB + c1B_r;
        const double c2a = lane == 0 ? dBL + cBA : xa, c2b = lane == 0 ? dAL + cAB : xb;
        const bool wa = c2a > c1a, wb = c2b > c1b;
        const double nA = (wa ? c2a : c1a) + gA, nB = (wb ? c2b : c1b) + gB;
        const int argA = wa ? id2A : id1A, argB = wb ? id2B : id1B;
        double best = D0 + c00, sec = -INFINITY;   // margin of the winner over the runner-up (losers may tie freely)
        int arg0 = 1;
        {
            const double t1 = dAL + cA0, t2 = dBL + cB0;
            if (t1 > best) { sec = best; best = t1; arg0 = idAL; } else sec = fmax(sec, t1);
            if (t2 > best) { sec = best; best = t2; arg0 = idBL; } else sec = fmax(sec, t2);
            if (xp > best) { sec = best; best = xp; arg0 = idPLL; } else sec = fmax(sec, xp);
        }
        const double g0 = best - sec;
        int wpA = argA | (fabs(c2a - c1a) < thr ? 0x8000 : 0), wpB = argB | (fabs(c2b - c1b) < thr ? 0x8000 : 0);
        int wp0 = arg0 | (g0 < thr ? 0x8000 : 0);
        if (w == 0 && t == 1) {
            // The first decisions of a recording compare emission-only values (viterbi.jl:55-63), and the states whose
            // deviation is ~0 (the last phases of both templates, P(L,L)) tie there to the last bit: these decisions are
            // taken in the reference's own arithmetic, T1[k,1] + lp with strict '>' in list order (viterbi.jl:74-84).
            const double y0 = a.y[0];
            auto fex = [&](int id1) { const double dd = y0 - a.mean[id1 - 1]; return a.c0 - (dd * dd) / a.den; };
            const double e1a = (lane == 0 ? 0.0 : fex(on ? id1A : 1)) + c1A_r, e2a = fex(on ? id2A : 1) + (lane == 0 ? cBA : 0.0);
            const double e1b = (lane == 0 ? 0.0 : fex(on ? id1B : 1)) + c1B_r, e2b = fex(on ? id2B : 1) + (lane == 0 ? cAB : 0.0);
            wpA = e2a > e1a ? id2A : id1A;
            wpB = e2b > e1b ? id2B : id1B;
            double bv = 0.0 + c00;
            wp0 = 1;
            const double z1 = fex(idAL) + cA0, z2 = fex(idBL) + cB0, z3 = fex(idPLL) + 0.0;
            if (z1 > bv) { bv = z1; wp0 = idAL; }
            if (z2 > bv) { bv = z2; wp0 = idBL; }
            if (z3 > bv) { bv = z3; wp0 = idPLL; }
        }
        if (own) {   // (no vector-memory load inside the step loop, so these stores never make a step wait)
            int16_t *psi = a.T2c + (int64_t)a.nms * t;
            if (on) {
                psi[k] = (int16_t)wpA;
                psi[L + k] = (int16_t)wpB;
            }
            if (lane == 0) psi[0] = (int16_t)wp0;
        }
        D0 = best;
        dA = nA; dB = nB;
        sA = psA + gA; sB = psB + gB;
        wp = wp + 1 == nwr ? 0 : wp + 1;
        rp = rp + 1 == nrd ? 0 : rp + 1;
        p0w = p0w == L ? 0 : p0w + 1;
        // (LDS operations of one wavefront execute in order: the next sample's reads see this sample's entries)
      }
    }
    __syncthreads();
    dump(a.endv + (int64_t)c * S, e - 1);
}

// sum over the recording of (y - m0)^2 (one double; the magnitude of the reference's trellis)
__global__ __launch_bounds__(256) void k_pair_mag(const double *__restrict__ y, int64_t T, const double *__restrict__ tab,
                                                  double *__restrict__ out)
{
    __shared__ double red[4];
    const double m0 = tab[12];
    double acc = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
        const double u = y[t] - m0;
        acc = __builtin_fma(u, u, acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

// flagged decisions on the decoded path: the state at sample t is a deciding state (silent, A_k, B_k) whose
// back-pointer of that sample carries the flag.  (The final arg-max over the end states is the blocked engine's,
// on columns materialised to ~1e-12: its margin is checked by k_pair_tail.)
// (bt: the blocked engine's per-state code, <= 0 for a multi-source state = minus its row; null: row = state.)
__global__ __launch_bounds__(256) void k_pair_ties(const int16_t *__restrict__ x, const int16_t *__restrict__ T2c,
                                                   const int32_t *__restrict__ bt, int64_t T, int nms, unsigned long long *diag)
{
    unsigned long long n = 0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; t < T; t += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)x[t] - 1;
        const int row = bt ? (bt[j] <= 0 ? -bt[j] : -1) : (j < nms ? j : -1);
        if (row >= 0 && (T2c[t * nms + row] & (int16_t)0x8000)) n++;
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(&diag[7], n);
}

// margin of the final arg-max (viterbi.jl:90) over the last block's end column
__global__ __launch_bounds__(256) void k_pair_tail(const double *__restrict__ endv_last, int S, const double *__restrict__ qsum,
                                                   const double *__restrict__ tab, double c0, double den, int64_t T, int L,
                                                   unsigned long long *diag)
{
    __shared__ double b1[256], b2[256];
    double best = -INFINITY, sec = -INFINITY;
    for (int j = threadIdx.x; j < S; j += 256) {
        const double v = endv_last[j];
        if (v > best) { sec = best; best = v; } else sec = fmax(sec, v);
    }
    b1[threadIdx.x] = best; b2[threadIdx.x] = sec;
    __syncthreads();
    if (threadIdx.x == 0) {
        best = -INFINITY; sec = -INFINITY;
        for (int i = 0; i < 256; i++) {
            if (b1[i] > best) { sec = fmax(sec, best); best = b1[i]; } else sec = fmax(sec, b1[i]);
            sec = fmax(sec, b2[i]);
        }
        const double mmax = fabs(c0) * (double)T + qsum[0] / den + fabs(tab[0]) * (double)T + 1.0;
        const double thr = ldexp(16.0 * (double)(L + 2), ilogb(mmax) - 52) + 1e-10;
        if (best - sec < thr) atomicAdd(&diag[7], 1ull);
    }
}

int pair_mag_launch(GenericDev *g, const double *d_y, hipStream_t st)
{
    HS_HIP(hipMemsetAsync(g->d_qsum, 0, sizeof(double), st));
    hipLaunchKernelGGL(k_pair_mag, dim3((unsigned)std::min<int64_t>(1024, (g->T + 255) / 256)), dim3(256), 0, st, d_y, g->T,
                       g->d_pairtab, g->d_qsum);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int pair_sweep_launch(GenericDev *g, const double *d_y, hipStream_t st)
{
    PairArgs a;
    a.y = d_y; a.T = g->T; a.S = (int)g->S; a.B = (int)g->B; a.H = (int)g->H; a.L = (int)g->K - 1; a.nms = g->nms;
    a.tab = g->d_pairtab;
    a.mean = g->d_mean;
    a.c0 = -kLog2Pi - g->lsig;
    a.den = 2.0 * (g->sigma * g->sigma);
    a.T2c = g->d_T2; a.endv = g->d_endv; a.warmv = g->d_warmv; a.qsum = g->d_qsum;
    int rc = pair_mag_launch(g, d_y, st);
    if (rc) return rc;
    const int L_ = (int)g->K - 1;
    const size_t lds = (size_t)(2 * pairbase(L_, L_) - pairfifo(L_, 0) + 64) * sizeof(double) + 128 * sizeof(int);
    HS_HIP(hipFuncSetAttribute((const void *)pair_vit_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(pair_vit_block, dim3((unsigned)g->nblk), dim3(64), lds, st, a);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int pair_ties_launch(GenericDev *g, const int16_t *d_x, hipStream_t st)
{
    hipLaunchKernelGGL(k_pair_ties, dim3((unsigned)std::min<int64_t>(2048, (g->T + 255) / 256)), dim3(256), 0, st, d_x,
                       g->d_T2, g->multi_ok ? g->d_bt : nullptr, g->T, g->nms, g->d_bdiag);
    hipLaunchKernelGGL(k_pair_tail, dim3(1), dim3(256), 0, st, g->d_endv + (g->nblk - 1) * g->S, (int)g->S, g->d_qsum,
                       g->d_pairtab, -kLog2Pi - g->lsig, 2.0 * (g->sigma * g->sigma), g->T, (int)g->K - 1, g->d_bdiag);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

}  // namespace hmmsort
