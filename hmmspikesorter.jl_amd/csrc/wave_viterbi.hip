// Wave engine, part 2: time-parallel Viterbi (reference src/viterbi.jl:44-98), one wavefront per
// chain.  Specification: tests/wave_model.py (vit_chain / vit_decode).
//
// Per sample t the recursion keeps, per chain:
//   D0(t)     = delta_t(silent)
//   P_a(t')   = delta of ring a's LAST state at t'+L-1 for the onset at t' = U_a(t') + Rfull_a(t'),
//               U_a(t') = best predecessor score of state (a,1) at t'
// and decides the back-pointer psi in {0 = silent, b = ring b's last state} of the N+1 junction
// states; candidates are scanned in source-state order (silent first, rings ascending) with a strict
// '>' -- the reference's tie rule (viterbi.jl:74-84).  Frame: delta'_t = delta_t - A*(t+1).
//
// A wavefront advances W = min(L, 64) samples per super-step: lane j owns sample t0 + j, the ring
// exits X_a(t) = P_a(t-L) it needs were produced in earlier super-steps (LDS delay line), so
// D0(t) = max(D0(t-1) + c00 + q0(t), max_a(X_a(t) + cend_a) + q0(t)) is a first-order max-plus
// recurrence across the lanes: one 6-level scan.  Every other quantity of the W samples is lane-local.
//
// Exactness.  (1) Chain starts: a chain starts Hw samples early from "silent, rings empty"; after the
// sweep its state at tc-1 is compared with the state the previous chain really ended on, ALL 1 + N*L
// entries up to one common constant (tolerance 1e-9); a chain that misses the certificate is swept
// again from the previous chain's exact end state (kw_vit with redo = 1), so the stored
// back-pointers are those of one sequential sweep.  (2) Rounding: inside a chain the arithmetic
// differs from the reference's by rounding at the 1e-13 level (pre-summed ring scores, scan order,
// per-chain offsets instead of the reference's O(t) cumulative sum, whose own rounding unit is
// larger).  Every decision records whether its winner beat the runner-up by less than
// thr = 16 (L+2) ulp(|T1|max) + 4e-9 (wave_thr, wave_common.h: the most the reference's own rounding can
// move the difference between two paths that have been apart for up to 8 (L+2) steps).  The flagged
// decisions ON THE DECODED PATH are then re-decided with the reference's own serial arithmetic
// (wave_ties.hip: exact prefix T1 along the path, candidates replayed op for op, strict '>' in list
// order); diag[7] counts the decisions that procedure could not settle (0 in every test).
#include <algorithm>
#include <cmath>
#include <type_traits>

#include "wave_common.h"

namespace hmmsort {

constexpr double kVitTol = 1e-9;

template <int N>
struct WIn {
    double y;
    double R[N];
};

// Near-tie threshold (wave_thr, wave_common.h).  The reference compares candidates T1[k,t-1] + lp whose
// values have grown to |T1| ~ t |A - 1/2| (ulp 5e-10 after 1e7 samples, 4e-9 after 1e8).  Two candidate paths
// that separated d steps ago share the value at their common ancestor exactly and then collect 2d roundings of
// at most ulp/2 each, so the reference's computed margin differs from the exact one by at most (2d + 1) ulp.  A
// decision is flagged when its margin (in this engine's per-chain frame, rounding ~1e-13) is below
//     thr = 16 (L + 2) ulp(|T1|max) + 4e-9
// -- the worst case for paths apart for up to 8 (L + 2) steps, the 4e-9 covering the tolerance of the
// chain-boundary certificate.  Flagged decisions are not guessed: wave_ties.hip replays the reference's own
// serial arithmetic for every flagged decision on the decoded path and re-decides it exactly.
// UC ("uniform cx"): the log-probability of (b,L) -> (a,1) does not depend on b -- true for every
// list the reference builds (types.jl:94-113: lp[a] + (N-2) lpz, accumulated in an order in which the
// source ring only contributes an exact +0.0) and checked bitwise on the host (wave_set_model).  The
// best ring exit into ring a is then "largest X_b over b != a" + cxin_a: one top-3 pass over the N
// exits serves all N junctions (O(N) instead of O(N^2) per sample).  Adding the same constant is
// monotone, so the winner can differ from the reference's candidate-by-candidate scan only by a tie
// created in rounding; such a decision has a zero margin and is flagged like any other near-tie.
template <int N, bool UC>
__global__ __launch_bounds__(64, wave_occ<N>()) void kw_vit(WaveGeom g, const WaveConst *__restrict__ cst,
                                                            const double *__restrict__ y,
                                                            const double *__restrict__ Rf,
                                                            const double *__restrict__ virt,
                                                            const double *__restrict__ ysum,
                                                            uint32_t *__restrict__ psi, double *__restrict__ vpre,
                                                            double *__restrict__ vend,
                                                            const int32_t *__restrict__ vfail, uint32_t *__restrict__ trash,
                                                            int redo)
{
    constexpr int EB = wpsi_bits_c(N), EPW = wpsi_epw_c(N), PW = wpsi_words_c(N), D = wave_depth<N>();
    // Model constants live in LDS next to the delay line: they are wave-uniform and many; as
    // kernel-argument/global scalars the compiler hoists all their loads out of the sweep loop and spills
    // hundreds of SGPRs.  An LDS word that the loop's own stores may alias is re-read (broadcast) where
    // it is used.
    constexpr int KC0 = 3, KCEND = 3 + N, KCXT = 3 + 2 * N, KSIZE = 3 + 2 * N + N * N;
    extern __shared__ double lds_vit[];
    double *KC = lds_vit;           // c00 | mean0 | den | c0[N] | cend[N] | cxT[N*N] (UC: cxin[N] first)
    double *DL = lds_vit + KSIZE;   // [N][RB] delay line: P_a(t') at slot (t' - tinit + L) mod RB
    const int lane = threadIdx.x;
    const int cg = blockIdx.x, ch = cg / g.nch, c = cg % g.nch;
    if (redo && (c == 0 || vfail[cg] == 0 || vfail[cg - 1] != 0)) return;  // wave-uniform
    const int L = g.L, W = g.W, RB = g.RB, B = g.B;
    const int64_t T = g.T;
    const int64_t tc = (int64_t)c * B;
    const int nc = (int)((T - tc) < B ? (T - tc) : B);
    const int64_t tend = tc + nc;
    const double *yc = y + (int64_t)ch * T;
    const double *Rc = Rf + (int64_t)ch * N * T;
    uint32_t *psic = psi + (int64_t)ch * T;
    const int64_t SR = 1 + N * L;
    const int64_t planePsi = (int64_t)g.C * T;
    const double thr = wave_thr(g, cst[ch], ysum, ch);

    for (int i = lane; i < N * (RB + 1); i += 64) DL[i] = -INFINITY;
    {
        const WaveConst &Kg = cst[ch];
        if (lane == 0) { KC[0] = Kg.c00; KC[1] = Kg.mean0; KC[2] = Kg.den; }
        if (lane < N) { KC[KC0 + lane] = Kg.c0[lane]; KC[KCEND + lane] = Kg.cend[lane]; }
        if (UC) { if (lane < N) KC[KCXT + lane] = Kg.cxin[lane]; }
        else for (int i = lane; i < N * N; i += 64) KC[KCXT + i] = Kg.cxT[i];
    }
    __syncthreads();
    int64_t tinit;
    double D0;
    if (redo) {            // exact hand-off: the previous chain's end state
        tinit = tc - 1;
        D0 = vend[(cg - 1) * SR];
        for (int i = lane; i < N * L; i += 64) {
            const int a = i / L, j = i % L + 1;
            DL[a * (RB + 1) + (L + 1 - j)] = vend[(cg - 1) * SR + 1 + i];
        }
    } else if (c == 0) {   // the reference's first column (viterbi.jl:55-63): emission only, T1[1,1] = 0
        tinit = 0;
        D0 = -cst[ch].A;
        for (int i = lane; i < N * L; i += 64) {
            const int a = i / L, j = i % L + 1;  // virtual onset -j -> slot L - j
            DL[a * (RB + 1) + (L - j)] = virt[((int64_t)ch * N + a) * (L + 1) + j];
        }
        if (lane < N) DL[lane * (RB + 1) + L] = Rc[(int64_t)lane * T];
    } else {               // warm-up start: silent, rings empty
        tinit = tc - g.Hw;
        D0 = 0.0;
    }
    __syncthreads();

    const int n_total = (int)(tend - 1 - tinit);  // steps t = tinit+1 .. tend-1
    // inputs of the super-step that starts `off` steps into the sweep: uniform base + lane offset
    auto load = [&](WIn<N> &d, int off) {
        const int nact = n_total - off < W ? n_total - off : W;
        const int li = lane < nact ? lane : 0;
        int64_t tb = tinit + 1 + off;
        tb = tb < T ? tb : T - 1;
        d.y = (yc + tb)[li];
#pragma unroll
        for (int a = 0; a < N; a++) d.R[a] = (Rc + (int64_t)a * T + tb)[li];
    };
    int rs = (1 + lane) % RB, ws = (L + 1 + lane) % RB;
    // STORE = 0: warm-up super-step (no global stores at all); 1: owned super-step (psi stored by every
    // lane, idle lanes of the last partial step into a trash line).  Straight-line global accesses only:
    // with stores under divergent branches hipcc drains vmcnt to 0 every super-step, which serialises the
    // input pipeline on the HBM latency.
    auto run = [&](const WIn<N> &d, int off, auto store_tag) {
        constexpr int STORE = decltype(store_tag)::value;
        const int nact = n_total - off < W ? n_total - off : W;
        const bool live = lane < nact;
        const int64_t tb = tinit + 1 + off;
        double X[N];
#pragma unroll
        for (int a = 0; a < N; a++) {
            const double v = DL[a * (RB + 1) + rs];
            X[a] = live ? v : -INFINITY;
        }
        // ring exits into the silent state: best and runner-up among the rings (first maximum wins)
        double e1 = -INFINITY, e2 = -INFINITY;
        int earg = 0;
#pragma unroll
        for (int a = 0; a < N; a++) {
            const double v = X[a] + KC[KCEND + a];
            earg = v > e1 ? a + 1 : earg;
            e2 = fmax(e2, fmin(e1, v));
            e1 = fmax(e1, v);
        }
        const double c00 = KC[0];
        const double dd = d.y - KC[1];
        const double q0 = -((dd * dd) / KC[2]);
        double sa = live ? c00 + q0 : 0.0;
        double sb = live ? e1 + q0 : -INFINITY;
        scan_maxplus(sa, sb);
        const double Dn = fmax(D0 + sa, sb);
        const double Dprev = lane_prev(Dn, D0);
        uint32_t pw[PW];
#pragma unroll
        for (int w = 0; w < PW; w++) pw[w] = 0u;
        {
            const double v0 = Dprev + c00;
            const bool ring = e1 > v0;
            const double gap = ring ? e1 - fmax(e2, v0) : v0 - e1;
            const uint32_t fl = gap < thr ? 1u : 0u;
            pw[0] = (ring ? (uint32_t)earg : 0u) | (fl << (EB - 1));
        }
        // top three exits (first index wins ties) for the UC path
        double m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
        int i1 = 0, i2 = 0;
        if (UC) {
#pragma unroll
            for (int b = 0; b < N; b++) {
                const double v = X[b];
                const bool g1 = v > m1, g2 = v > m2;
                m3 = g2 ? m2 : fmax(m3, v);
                i2 = g1 ? i1 : (g2 ? b + 1 : i2);
                m2 = g1 ? m1 : (g2 ? v : m2);
                i1 = g1 ? b + 1 : i1;
                m1 = g1 ? v : m1;
            }
        }
        const int wsl = live ? ws : RB;   // idle lanes write a spare slot behind the ring (branch-free stores)
#pragma unroll
        for (int a = 0; a < N; a++) {
            double r1 = -INFINITY, r2 = -INFINITY;
            int rarg = 0;
            if (UC) {
                const double cin = KC[KCXT + a];
                const bool first = (a + 1 == i1), second = (a + 1 == i2);
                r1 = (first ? m2 : m1) + cin;
                rarg = first ? i2 : i1;
                r2 = ((first || second) ? m3 : m2) + cin;
                if (N == 1) { r1 = -INFINITY; r2 = -INFINITY; rarg = 0; }
            } else {
#pragma unroll
                for (int b = 0; b < N; b++) {
                    if (b == a) continue;
                    const double v = X[b] + KC[KCXT + a * N + b];   // (b,L) -> (a,1)
                    rarg = v > r1 ? b + 1 : rarg;
                    r2 = fmax(r2, fmin(r1, v));
                    r1 = fmax(r1, v);
                }
            }
            const double v0 = Dprev + KC[KC0 + a];
            const bool ring = r1 > v0;
            const double u = ring ? r1 : v0;
            const double gap = ring ? r1 - fmax(r2, v0) : v0 - r1;
            const uint32_t fl = gap < thr ? 1u : 0u;
            DL[a * (RB + 1) + wsl] = u + d.R[a];
            const uint32_t ent = (ring ? (uint32_t)rarg : 0u) | (fl << (EB - 1));
            pw[(a + 1) / EPW] |= ent << (((a + 1) % EPW) * EB);
            if (!UC) __builtin_amdgcn_sched_barrier(0);   // keep the junctions sequential: N^2 live candidates otherwise
        }
        if (STORE) {
#pragma unroll
            for (int w = 0; w < PW; w++) {
                uint32_t *dst = live ? psic + w * planePsi + tb + lane : trash + lane;
                *dst = pw[w];
            }
        }
        D0 = wave_bcast(Dn, nact - 1);  // the last live lane
        rs += W; rs = rs >= RB ? rs - RB : rs;
        ws += W; ws = ws >= RB ? ws - RB : ws;
    };
    auto dump = [&](double *rec, int64_t tref) {   // state "just before tref": D0 and P_a(tref - j), j = 1..L
        __syncthreads();
        if (lane == 0) rec[0] = D0;
        for (int i = lane; i < N * L; i += 64) {
            const int a = i / L, j = i % L + 1;
            const int slot = (int)((tref - j - tinit + L) % RB);
            rec[1 + i] = DL[a * (RB + 1) + slot];
        }
        __syncthreads();
    };
    // D super-steps of input in flight; the ring of buffers is indexed statically
    // The main loop is branch-free (whole groups of D super-steps, loads clamped past the end), so that
    // hipcc can count vmcnt across the back edge; the last < D super-steps run from the buffers the main
    // loop has already filled.
    auto sweep = [&](int from, int to, auto store_tag) {
        WIn<N> buf[D];
#pragma unroll
        for (int i = 0; i < D; i++) load(buf[i], from + i * W);
        int off = from;
        for (; off + D * W <= to; off += D * W) {
#pragma unroll
            for (int i = 0; i < D; i++) {
                run(buf[i], off + i * W, store_tag);
                load(buf[i], off + (i + D) * W);
            }
        }
#pragma unroll
        for (int i = 0; i < D; i++)
            if (off + i * W < to) run(buf[i], off + i * W, store_tag);
    };
    const int n_warm = (redo || c == 0) ? 0 : g.Hw - 1;
    if (n_warm > 0) {
        sweep(0, n_warm, std::integral_constant<int, 0>());
        dump(vpre + cg * SR, tc);
    }
    sweep(n_warm, n_total, std::integral_constant<int, 1>());
    dump(vend + cg * SR, tend);
    if (redo) {  // the hand-off was exact by construction
        for (int i = lane; i < SR; i += 64) vpre[cg * SR + i] = vend[(cg - 1) * SR + i];
    }
}

// Boundary certificate: one wavefront per chain boundary; all 1 + N*L entries of the warm-up state
// must equal the previous chain's end state up to one constant.
__global__ __launch_bounds__(64) void kw_vit_check(WaveGeom g, const double *__restrict__ vpre,
                                                   const double *__restrict__ vend,
                                                   int32_t *__restrict__ vfail, int64_t *__restrict__ diag,
                                                   int final_round, double *__restrict__ dbg)
{
    const int lane = threadIdx.x;
    const int cg = blockIdx.x, c = cg % g.nch;
    if (c == 0) { if (lane == 0) vfail[cg] = 0; return; }
    const int64_t SR = 1 + (int64_t)g.N * g.L;
    double lo = INFINITY, hi = -INFINITY;
    int64_t ilo = -1, ihi = -1, ibad = -1;
    bool bad = false;
    for (int64_t i = lane; i < SR; i += 64) {
        const double p = vpre[cg * SR + i], e = vend[(cg - 1) * SR + i];
        if (p == e) {   // -inf on both sides (an unreachable entry) says nothing about the frame constant
            if (fabs(p) < INFINITY) { lo = fmin(lo, 0.0); hi = fmax(hi, 0.0); }
            continue;
        }
        const double d = p - e;
        if (!(fabs(d) < INFINITY)) { bad = true; ibad = i; continue; }      // NaN or one-sided infinity
        if (d < lo) { lo = d; ilo = i; }
        if (d > hi) { hi = d; ihi = i; }
    }
    if (dbg && final_round) {   // debug record: the extreme entries of the first boundaries
        for (int o = 32; o > 0; o >>= 1) {
            const double l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
            const long long il2 = __shfl_xor((long long)ilo, o), ih2 = __shfl_xor((long long)ihi, o), ib2 = __shfl_xor((long long)ibad, o);
            if (l2 < lo) { lo = l2; ilo = il2; }
            if (h2 > hi) { hi = h2; ihi = ih2; }
            if (ib2 > ibad) ibad = ib2;
        }
        if (lane == 0 && c <= 3) {
            double *r = dbg + 16 + 8 * (c - 1);
            r[0] = (double)cg; r[1] = lo; r[2] = (double)ilo; r[3] = hi; r[4] = (double)ihi; r[5] = (double)ibad;
            r[6] = ilo >= 0 ? vpre[cg * SR + ilo] : 0.0; r[7] = ilo >= 0 ? vend[(cg - 1) * SR + ilo] : 0.0;
        }
    }
    lo = -wave_max(-lo); hi = wave_max(hi);
    const bool anybad = __any(bad);
    const double spread = hi - lo;
    const bool fail = anybad || !(spread <= kVitTol);
    if (lane == 0) {
        vfail[cg] = fail ? 1 : 0;
        if (final_round) {
            if (fail) atomicAdd((unsigned long long *)&diag[0], 1ull);
            if (!anybad && spread == spread && spread < INFINITY)
                atomicMax((unsigned long long *)&diag[2], (unsigned long long)__double_as_longlong(spread));
        } else if (fail) {
            atomicAdd((unsigned long long *)&diag[1], 1ull);  // chains swept again (all rounds)
        }
    }
}

// Final state = argmax over all S states at the last sample, first maximum in state order
// (viterbi.jl:90).  delta(a,k) at T-1 is P_a(T-k).  One wave per channel.
__global__ __launch_bounds__(64) void kw_vit_tail(WaveGeom g, const WaveConst *__restrict__ cst,
                                                  const double *__restrict__ ysum,
                                                  const double *__restrict__ vend,
                                                  int32_t *__restrict__ final_state, int64_t *__restrict__ tie_cnt)
{
    const int lane = threadIdx.x, ch = blockIdx.x;
    const int64_t SR = 1 + (int64_t)g.N * g.L;
    const double *rec = vend + ((int64_t)ch * g.nch + g.nch - 1) * SR;
    const int S = 1 + g.N * g.L;
    double best = -INFINITY, sec = -INFINITY;
    int bi = S;
    for (int j = lane; j < S; j += 64) {  // state j = 1 + a*L + (k-1) <-> record entry 1 + a*L + (k-1)
        const double v = rec[j];
        if (v > best) { sec = best; best = v; bi = j; }
        else sec = fmax(sec, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(best, o), os = __shfl_xor(sec, o);
        const int oi = __shfl_xor(bi, o);
        const bool take = ov > best || (ov == best && oi < bi);
        const double lose = take ? best : ov;
        sec = fmax(fmax(sec, os), lose);
        if (take) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        final_state[ch] = (bi >= S) ? 0 : bi;
        if ((best - sec) < wave_thr(g, cst[ch], ysum, ch)) {   // wave_ties.hip re-decides it exactly
            tie_cnt[ch * 8 + kTieTail] = 1;
            atomicAdd((unsigned long long *)&tie_cnt[ch * 8 + kTieTrig], 1ull);
        }
    }
}

// Backtrace (viterbi.jl:93-94).  The path is cut into segments of Bb samples, one LANE per segment;
// a lane starts its walk Hb samples after its segment's end from the silent state (from the true
// final state where that is the end of the data) and has merged with the true path by the time it
// enters its segment (checked: kw_stitch_*).  psi is in natural time order, so a tile of 64 segments
// x 64 samples is read with coalesced rows into LDS and walked column-wise; x leaves the same way.
// Flagged (near-tie) junction decisions met inside the owned segment are counted per channel (the trigger of
// the exact resolver, wave_ties.hip).
template <int N>
__global__ __launch_bounds__(64) void kw_backtrace(WaveGeom g, const uint32_t *__restrict__ psi,
                                                   const int32_t *__restrict__ final_state,
                                                   int16_t *__restrict__ x, int32_t *__restrict__ bstate,
                                                   int64_t *__restrict__ tie_cnt)
{
    constexpr int EB = wpsi_bits_c(N), EPW = wpsi_epw_c(N), PW = wpsi_words_c(N);
    constexpr int TS = PW <= 2 ? 64 : 32;   // samples per tile (static LDS stays below 64 KB)
    constexpr int RPI = 64 / TS;            // segment rows per load instruction
    constexpr int XW = TS / 2;              // decoded ids leave as pairs (two int16 per word)
    constexpr int XRI = 64 / XW;            // rows per store instruction
    __shared__ uint32_t tile[PW][64][TS + 1];
    __shared__ uint32_t xt[64][XW + 1];
    const int lane = threadIdx.x, ch = blockIdx.y;
    const int L = g.L, Bb = g.Bb, Hb = g.Hb;
    const int64_t T = g.T;
    const int64_t sg0 = (int64_t)blockIdx.x * 64, sg = sg0 + lane;
    const bool active = sg < g.nseg;
    const int64_t s_lo = sg * Bb;
    const int64_t s_hi = active ? ((s_lo + Bb) < T ? (s_lo + Bb) : T) : s_lo;
    const int64_t te = (s_hi + Hb) < T ? (s_hi + Hb) : T;  // walk starts at te-1
    const int64_t planePsi = (int64_t)g.C * T;
    const uint32_t *pc = psi + (int64_t)ch * T;
    int16_t *xc = x + (int64_t)ch * T;
    const bool xal = ((reinterpret_cast<uintptr_t>(xc)) & 3) == 0;   // odd T puts later channels on odd samples
    // times relative to the segment start: u = t - s_lo
    const int te_rel = active ? (int)(te - s_lo) : 0, hi_rel = (int)(s_hi - s_lo);
    const int t1 = s_lo >= 1 ? 0 : 1;                      // the step at t = 0 has no predecessor
    const int t2 = s_lo >= 2 ? 0 : (int)(2 - s_lo);        // psi(1) only decides x[0]: kw_first_state
    // walk state: id = state id at the current sample, rem = steps left inside the ring (0 at a junction),
    // (wi, sh) = word and bit offset of the psi entry the next junction decision reads (entry e = ring + 1)
    int id = 1, rem = 0, wi = 0, sh = 0;
    if (active && te == T) {
        const int fs = final_state[ch];
        if (fs > 0) {
            const int a0 = (fs - 1) / L, k0 = (fs - 1) % L + 1, e0 = a0 + 1;
            id = fs + 1; rem = k0 - 1; wi = e0 / EPW; sh = (e0 % EPW) * EB;
        }
    }
    int nflag = 0, bs = 0;
    const int lr = lane / TS, lc = lane % TS;
    constexpr bool ROWREG = PW <= 2;        // the lane's tile row in registers (wider rows stay in LDS)
    uint32_t rowv[ROWREG ? PW : 1][ROWREG ? TS : 1], xr[XW];
    // one tile of the walk, newest sample first.  FAST: every lane of the wave is inside its walk for the
    // whole tile and (OWN) inside / (not OWN) behind its own segment, so no per-lane time predicates.
    auto walk = [&](int q, auto fast_tag, auto own_tag) {
        constexpr bool FAST = decltype(fast_tag)::value, OWN = decltype(own_tag)::value;
#pragma unroll
        for (int i = TS - 1; i >= 0; i--) {
            const int u = TS * q + i;
            const bool interior = rem > 0;
            if (FAST ? OWN : true) {
                if (i & 1) xr[i / 2] = (uint32_t)id << 16;
                else xr[i / 2] |= (uint32_t)id & 0xffffu;
            }
            uint32_t wsel;
            if constexpr (ROWREG) {
                wsel = rowv[0][i];
#pragma unroll
                for (int w = 1; w < PW; w++) wsel = (wi == w) ? rowv[w][i] : wsel;
            } else {
                wsel = tile[wi][lane][i];
            }
            const uint32_t ent = wsel >> sh;
            const int pj = (int)(ent & ((1u << (EB - 1)) - 1u));
            const int flag = (int)((ent >> (EB - 1)) & 1u);
            // junction: predecessor p = 0 silent (id 1), else the last state of ring p-1 (id 1 + p L)
            const int jid = 1 + pj * L, jrem = pj ? L - 1 : 0;
            const int jwi = pj / EPW, jsh = (pj % EPW) * EB;
            if (FAST) {
                if (OWN) nflag += interior ? 0 : flag;
                id = interior ? id - 1 : jid;
                rem = interior ? rem - 1 : jrem;
                wi = interior ? wi : jwi;
                sh = interior ? sh : jsh;
            } else {
                const bool live = u < te_rel, step = live && u >= t1;
                bs = (live && u == hi_rel) ? id : bs;
                if (step && !interior && u < hi_rel && u >= t2) nflag += flag;
                id = step ? (interior ? id - 1 : jid) : id;
                rem = step ? (interior ? rem - 1 : jrem) : rem;
                wi = step ? (interior ? wi : jwi) : wi;
                sh = step ? (interior ? sh : jsh) : sh;
            }
        }
    };
    const int nq = (Bb + Hb) / TS;
    for (int q = nq - 1; q >= 0; q--) {
        // stage psi rows: row r = segment sg0 + r, samples s_lo(r) + TS q + lc.  Wave-uniform tile origin +
        // 32-bit lane offsets; rows that do not exist read the origin and are staged as zeros.
        {
            const int64_t tb = sg0 * Bb + (int64_t)TS * q;
            const uint32_t *base = pc + (tb < T ? tb : 0);
#pragma unroll 16
            for (int rr = 0; rr < 64; rr += RPI) {
                const int r = rr + lr;
                const bool ok = (sg0 + r) < g.nseg && tb + (int64_t)r * Bb + lc < T;
                const uint32_t off = ok ? (uint32_t)(r * Bb + lc) : 0u;
                uint32_t v[PW];
#pragma unroll
                for (int w = 0; w < PW; w++) v[w] = (base + w * planePsi)[off];
#pragma unroll
                for (int w = 0; w < PW; w++) tile[w][r][lc] = ok ? v[w] : 0u;
            }
        }
        __syncthreads();
        if constexpr (ROWREG) {
#pragma unroll
            for (int w = 0; w < PW; w++)
#pragma unroll
                for (int i = 0; i < TS; i++) rowv[w][i] = tile[w][lane][i];
        }
        const int u0 = TS * q, u1 = u0 + TS;
        const bool own_tile = u0 < Bb;
        // wave-uniform choice of the tile body
        const bool lane_fast = u0 >= t1 && u1 <= te_rel &&
                               (own_tile ? (u1 <= hi_rel && u0 >= t2) : (u0 > hi_rel));
        const bool fast = __all(lane_fast);
        if (fast && own_tile) walk(q, std::true_type(), std::true_type());
        else if (fast) walk(q, std::true_type(), std::false_type());
        else walk(q, std::false_type(), std::false_type());
        if (own_tile) {  // owned rows: write x out, coalesced
#pragma unroll
            for (int j = 0; j < XW; j++) xt[lane][j] = xr[j];
            __syncthreads();
            const int xrw = lane / XW, xcw = lane % XW;
#pragma unroll 8
            for (int rr = 0; rr < 64; rr += XRI) {
                const int r = rr + xrw;
                const int64_t t = (sg0 + r) * Bb + (int64_t)TS * q + 2 * xcw;
                if ((sg0 + r) < g.nseg && t < T) {
                    const uint32_t v = xt[r][xcw];
                    if (t + 1 < T && xal) *reinterpret_cast<uint32_t *>(xc + t) = v;
                    else {
                        xc[t] = (int16_t)(v & 0xffffu);
                        if (t + 1 < T) xc[t + 1] = (int16_t)(v >> 16);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (active && hi_rel < te_rel) bstate[(int64_t)ch * g.nseg + sg] = bs;
    for (int o = 32; o > 0; o >>= 1) nflag += __shfl_xor(nflag, o);
    if (lane == 0 && nflag) atomicAdd((unsigned long long *)&tie_cnt[ch * 8 + kTieTrig], (unsigned long long)nflag);
}

__device__ __forceinline__ void wwalk_step(const WaveGeom &g, const uint32_t *__restrict__ pc, int64_t planePsi,
                                           int64_t t, int &a, int &k, int64_t &nflag)
{
    if (a >= 0 && k > 1) { k--; return; }
    const int e = a + 1;
    const uint32_t w = pc[(int64_t)(e / g.epw) * planePsi + t];
    const uint32_t ent = (w >> ((e % g.epw) * g.EB)) & ((1u << g.EB) - 1u);
    const int p = (int)(ent & ((1u << (g.EB - 1)) - 1u));
    if (t >= 2) nflag += ent >> (g.EB - 1);  // psi(1) only decides x[0], which kw_first_state re-decides exactly
    if (p == 0) { a = -1; k = 0; }
    else { a = p - 1; k = g.L; }
}

// Stitch check: the state segment s's walk had on the first sample of segment s+1 must equal what
// segment s+1 emitted there; otherwise segment s is queued for a serial re-walk.
__global__ void kw_stitch_check(WaveGeom g, const int16_t *__restrict__ x, const int32_t *__restrict__ bstate,
                                int32_t *__restrict__ redo)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)g.C * g.nseg) return;
    const int ch = (int)(i / g.nseg);
    const int64_t sg = i % g.nseg;
    if (sg >= g.nseg - 1) return;
    if (bstate[i] != (int)x[(int64_t)ch * g.T + (sg + 1) * g.Bb]) {
        const int slot = atomicAdd(&redo[0], 1);
        redo[1 + slot] = (int32_t)i;
    }
}

// Parallel repair of the queued segments whose successor is not queued itself (the common case: isolated
// disagreements on busy signals with long rings).  One workgroup per segment: its psi goes global -> LDS in
// full rows, lane 0 walks it from the successor's first state, all lanes write x.  A repaired segment whose
// own first sample changed is caught by the second kw_stitch_check and the serial kw_stitch_fix below, which
// also takes the chains of consecutive failures this kernel leaves alone.
__global__ __launch_bounds__(64) void kw_stitch_fix_par(WaveGeom g, const uint32_t *__restrict__ psi,
                                                        int16_t *__restrict__ x, int32_t *__restrict__ bstate,
                                                        const int32_t *__restrict__ redo, int64_t *__restrict__ diag,
                                                        int64_t *__restrict__ tie_cnt)
{
    extern __shared__ uint32_t shp[];                 // [PW][Bb + 1] psi of the segment and the sample after it | Bb ids
    const int lane = threadIdx.x, n = redo[0];
    const int L = g.L, Bb = g.Bb, PW = g.PW;
    int16_t *xs = reinterpret_cast<int16_t *>(shp + (size_t)PW * (Bb + 1));
    const int64_t planePsi = (int64_t)g.C * g.T;
    for (int q = blockIdx.x; q < n; q += gridDim.x) {
        const int32_t id = redo[1 + q];
        const int ch = (int)(id / g.nseg);
        const int64_t sg = id % g.nseg;
        bool succ = false;
        for (int j = lane; j < n; j += 64) succ = succ || (redo[1 + j] == id + 1 && sg + 1 < g.nseg);
        if (__any(succ)) continue;
        const uint32_t *pc = psi + (int64_t)ch * g.T;
        int16_t *xc = x + (int64_t)ch * g.T;
        const int64_t lo = sg * Bb, tn = lo + Bb;     // tn = first sample of segment sg + 1 (< T: sg is not the last)
        const int want = xc[tn];
        if (bstate[(int64_t)ch * g.nseg + sg] == want) continue;
        for (int w = 0; w < PW; w++)
            for (int i = lane; i <= Bb; i += 64) shp[w * (Bb + 1) + i] = pc[w * planePsi + lo + i];
        __syncthreads();
        if (lane == 0) {
            int a = -1, k = 0;
            int64_t nflag = 0;
            if (want > 1) { a = (want - 2) / L; k = (want - 2) % L + 1; }
            auto step = [&](int i) {                  // the move from sample lo + i to lo + i - 1
                if (a >= 0 && k > 1) { k--; return; }
                const int e = a + 1;
                const uint32_t wv = shp[(e / g.epw) * (Bb + 1) + i];
                const uint32_t ent = (wv >> ((e % g.epw) * g.EB)) & ((1u << g.EB) - 1u);
                const int p = (int)(ent & ((1u << (g.EB - 1)) - 1u));
                if (lo + i >= 2) nflag += ent >> (g.EB - 1);
                if (p == 0) { a = -1; k = 0; }
                else { a = p - 1; k = L; }
            };
            step(Bb);
            for (int i = Bb - 1; i >= 0; i--) {
                xs[i] = (int16_t)((a < 0) ? 1 : 2 + a * L + (k - 1));
                if (lo + i == 0) break;
                if (i > 0) step(i);
            }
            bstate[(int64_t)ch * g.nseg + sg] = want;
            atomicAdd((unsigned long long *)&diag[1], 1ull);
            if (nflag) atomicAdd((unsigned long long *)&tie_cnt[ch * 8 + kTieTrig], (unsigned long long)nflag);
        }
        __syncthreads();
        for (int i = lane; i < Bb; i += 64) xc[lo + i] = xs[i];
        __syncthreads();
    }
}

__global__ void kw_stitch_fix(WaveGeom g, const uint32_t *__restrict__ psi, int16_t *__restrict__ x,
                              int32_t *__restrict__ bstate, const int32_t *__restrict__ redo,
                              int64_t *__restrict__ diag, int64_t *__restrict__ tie_cnt)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n = redo[0];
    const int L = g.L, Bb = g.Bb;
    const int64_t planePsi = (int64_t)g.C * g.T;
    int64_t fixes = 0;
    for (int q = 0; q < n; q++) {
        const int ch = (int)(redo[1 + q] / g.nseg);
        int64_t nflag = 0;
        int64_t sg = redo[1 + q] % g.nseg;
        const uint32_t *pc = psi + (int64_t)ch * g.T;
        int16_t *xc = x + (int64_t)ch * g.T;
        while (sg >= 0) {
            const int64_t tn = (sg + 1) * Bb;  // first sample of segment sg+1
            const int want = xc[tn];
            if (bstate[(int64_t)ch * g.nseg + sg] == want) break;
            bstate[(int64_t)ch * g.nseg + sg] = want;
            fixes++;
            int a = -1, k = 0;
            if (want > 1) { a = (want - 2) / L; k = (want - 2) % L + 1; }
            wwalk_step(g, pc, planePsi, tn, a, k, nflag);
            for (int64_t t = tn - 1; t >= sg * Bb; t--) {
                xc[t] = (int16_t)((a < 0) ? 1 : 2 + a * L + (k - 1));
                if (t == 0) break;
                if (t > sg * Bb) wwalk_step(g, pc, planePsi, t, a, k, nflag);
            }
            sg--;  // did the first sample of segment sg change?  then segment sg-1 must be re-checked
        }
        if (nflag) atomicAdd((unsigned long long *)&tie_cnt[ch * 8 + kTieTrig], (unsigned long long)nflag);
    }
    diag[1] += fixes;
}

// The first decoded state, exactly as the reference finds it: x[0] = psi_1(x[1]) and psi_1 only sees
// the first trellis column, which is plain emission (viterbi.jl:55-63).  Template tails are ~1e-16, so
// the "ring in its last phase at sample 0" candidates differ by a few ulps only; re-deciding this one
// sample with the reference's own operations (strict '>', list order) makes it exact.
__global__ void kw_first_state(WaveGeom g, const WaveConst *__restrict__ cst, const double *__restrict__ y,
                               const double *__restrict__ mean, const double *__restrict__ ctab_all,
                               int16_t *__restrict__ x)
{
    const int ch = blockIdx.x;
    if (threadIdx.x != 0 || g.T < 2) return;
    const int N = g.N, L = g.L, S = 1 + N * L;
    const double *ctab = ctab_all + (int64_t)ch * (1 + 2 * N + N * N + N * L);
    const double *c0 = ctab + 1, *cend = ctab + 1 + N, *cx = ctab + 1 + 2 * N, *cint = ctab + 1 + 2 * N + N * N;
    const double *mc = mean + (int64_t)ch * S;
    const double A = cst[ch].A, den = cst[ch].den;
    int16_t *xc = x + (int64_t)ch * g.T;
    const double y0 = y[(int64_t)ch * g.T];
    auto T1 = [&](int a, int k) {  // funcl, utils.jl:4
        const double dd = y0 - mc[1 + a * L + (k - 1)];
        return A - (dd * dd) / den;
    };
    double best = -INFINITY;
    int arg = 1;
    auto cand = [&](int state, double t1, double lp) {
        const double tt = t1 + lp;
        if (tt > best) { best = tt; arg = state; }
    };
    const int x1 = xc[1];
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
    xc[0] = (int16_t)arg;
}

// ll = sum_{t=1..T-1} T1[x_t, t]  (viterbi.jl:92-96) without the trellis:
//   T1[x_t,t] = T1[x_0,0] + sum_{u=1..t} inc_u  =>  ll = (T-1) T1[x_0,0] + sum_u (T-u) inc_u.
__global__ __launch_bounds__(256) void kw_ll_partial(WaveGeom g, const WaveConst *__restrict__ cst,
                                                     const double *__restrict__ y, const int16_t *__restrict__ x,
                                                     const double *__restrict__ mean,
                                                     const double *__restrict__ ctab_all, double *__restrict__ part)
{
    __shared__ double red[4];
    const int ch = blockIdx.y, N = g.N, L = g.L, S = 1 + N * L;
    const int64_t T = g.T;
    const double *yc = y + (int64_t)ch * T, *mc = mean + (int64_t)ch * S;
    const int16_t *xc = x + (int64_t)ch * T;
    const double *ctab = ctab_all + (int64_t)ch * (1 + 2 * N + N * N + N * L);
    const double A = cst[ch].A, den = cst[ch].den;
    double acc = 0.0;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; u < T;
         u += (int64_t)gridDim.x * blockDim.x) {
        const int xp = xc[u - 1], xn = xc[u];
        const double d = yc[u] - mc[xn - 1];
        const double inc = wpath_lp(N, L, ctab, xp, xn) + (A - (d * d) / den);
        acc += (double)(T - u) * inc;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int x0 = xc[0];
        if (x0 != 1) {  // T1[1,1] = 0 for the silent state (viterbi.jl:63)
            const double d = yc[0] - mc[x0 - 1];
            acc += (double)(T - 1) * (A - (d * d) / den);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(int64_t)ch * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void kw_sum_partials(const double *__restrict__ part, int n, double *__restrict__ out)
{
    __shared__ double red[4];
    const int ch = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += part[(int64_t)ch * n + i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[ch] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename Kern>
static int wave_lds_attr(Kern kern, size_t lds)
{
    if (lds > 64 * 1024)
        HS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return HMMSORT_OK;
}

constexpr int kVitRounds = 2;  // certificate + re-sweep rounds run unconditionally on device

int wave_viterbi_sweep(WaveDev *r, const double *d_y, hipStream_t st)
{
    const WaveGeom &g = r->g;
    const int nchT = g.C * g.nch;
    return dispatch_N(g.N, [&](auto n) {
        constexpr int N = decltype(n)::value;
        const size_t lds = ((size_t)N * (g.RB + 1) + 3 + 2 * N + N * N) * sizeof(double);
        auto kern = kw_vit<N, true>;
        if constexpr (N <= 8) { if (!r->uniform_cx) kern = kw_vit<N, false>; }   // per-source values: up to 8 rings (wave_supported)
        int rc = wave_lds_attr(kern, lds);
        if (rc) return rc;
        { WPROF(r, "kw_vit", st);
          hipLaunchKernelGGL(kern, dim3(nchT), dim3(64), lds, st, g, r->d_cst, d_y, r->Rf, r->virt, r->ysum,
                             r->psi, r->vpre, r->vend, r->vfail, (uint32_t *)r->trash, 0); }
        HS_HIP(hipGetLastError());
        return HMMSORT_OK;
    });
}

// boundary certificates with exact re-sweeps, final state, backtrace + stitch, x[0], ll
int wave_viterbi_post(WaveDev *r, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st)
{
    const WaveGeom &g = r->g;
    const int nchT = g.C * g.nch;
    int rc = dispatch_N(g.N, [&](auto n) {
        constexpr int N = decltype(n)::value;
        const size_t lds = ((size_t)N * (g.RB + 1) + 3 + 2 * N + N * N) * sizeof(double);
        auto kern = kw_vit<N, true>;
        if constexpr (N <= 8) { if (!r->uniform_cx) kern = kw_vit<N, false>; }   // per-source values: up to 8 rings (wave_supported)
        for (int round = 0; round < kVitRounds && g.nch > 1; round++) {
            { WPROF(r, "kw_vit_check", st);
              hipLaunchKernelGGL(kw_vit_check, dim3(nchT), dim3(64), 0, st, g, r->vpre, r->vend, r->vfail, r->diag, 0, nullptr); }
            { WPROF(r, "kw_vit_redo", st);
              hipLaunchKernelGGL(kern, dim3(nchT), dim3(64), lds, st, g, r->d_cst, d_y, r->Rf, r->virt,
                                 r->ysum, r->psi, r->vpre, r->vend, r->vfail, (uint32_t *)r->trash, 1); }
        }
        { WPROF(r, "kw_vit_check", st);
          hipLaunchKernelGGL(kw_vit_check, dim3(nchT), dim3(64), 0, st, g, r->vpre, r->vend, r->vfail, r->diag, 1, r->dbg); }
        HS_HIP(hipMemsetAsync(r->tie_cnt, 0, (size_t)g.C * 8 * sizeof(int64_t), st));
        { WPROF(r, "kw_vit_tail", st);
          hipLaunchKernelGGL(kw_vit_tail, dim3(g.C), dim3(64), 0, st, g, r->d_cst, r->ysum, r->vend, r->final_state,
                             r->tie_cnt); }
        HS_HIP(hipMemsetAsync(r->redo, 0, sizeof(int32_t), st));
        { WPROF(r, "kw_backtrace", st);
          hipLaunchKernelGGL((kw_backtrace<N>), dim3((unsigned)((g.nseg + 63) / 64), g.C), dim3(64), 0, st, g, r->psi,
                             r->final_state, d_x, r->bstate, r->tie_cnt); }
        HS_HIP(hipGetLastError());
        return HMMSORT_OK;
    });
    if (rc) return rc;
    const int64_t nsegT = (int64_t)g.C * g.nseg;
    { WPROF(r, "kw_stitch_check", st);
      hipLaunchKernelGGL(kw_stitch_check, dim3((unsigned)((nsegT + 255) / 256)), dim3(256), 0, st, g, d_x, r->bstate,
                         r->redo); }
    { WPROF(r, "kw_stitch_fix", st);
      // isolated failures in parallel, then a second check and the serial repair for what is left (chains of
      // consecutive failures, first samples that changed)
      const size_t ldsp = ((size_t)g.PW * (g.Bb + 1)) * sizeof(uint32_t) + (size_t)g.Bb * sizeof(int16_t) + 8;
      hipLaunchKernelGGL(kw_stitch_fix_par, dim3(256), dim3(64), ldsp, st, g, r->psi, d_x, r->bstate, r->redo, r->diag,
                         r->tie_cnt);
      HS_HIP(hipMemsetAsync(r->redo, 0, sizeof(int32_t), st));
      hipLaunchKernelGGL(kw_stitch_check, dim3((unsigned)((nsegT + 255) / 256)), dim3(256), 0, st, g, d_x, r->bstate,
                         r->redo);
      hipLaunchKernelGGL(kw_stitch_fix, dim3(1), dim3(64), 0, st, g, r->psi, d_x, r->bstate, r->redo, r->diag,
                         r->tie_cnt); }
    { WPROF(r, "kw_first_state", st);
      hipLaunchKernelGGL(kw_first_state, dim3(g.C), dim3(64), 0, st, g, r->d_cst, d_y, r->d_mean, r->d_ctab, d_x); }
    // flagged near-ties on the decoded path: re-decided with the reference's own arithmetic (no-ops otherwise)
    if ((rc = wave_tie_resolve(r, d_y, d_x, st))) return rc;
    { WPROF(r, "kw_ll_partial", st);
      hipLaunchKernelGGL(kw_ll_partial, dim3(r->nparts, g.C), dim3(256), 0, st, g, r->d_cst, d_y, d_x, r->d_mean,
                         r->d_ctab, r->part); }
    { WPROF(r, "kw_sum_partials", st);
      hipLaunchKernelGGL(kw_sum_partials, dim3(g.C), dim3(256), 0, st, r->part, r->nparts, d_ll); }
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int wave_viterbi(WaveDev *r, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st)
{
    return wave_graphed(r, 1, d_y, d_x, d_ll, nullptr, st, [&](hipStream_t s) -> int {
        int rc;
        HS_HIP(hipMemsetAsync(r->diag, 0, 8 * sizeof(int64_t), s));
        if ((rc = wave_prepare(r, d_y, s))) return rc;
        if ((rc = wave_viterbi_sweep(r, d_y, s))) return rc;
        return wave_viterbi_post(r, d_y, d_x, d_ll, s);
    });
}

}  // namespace hmmsort
