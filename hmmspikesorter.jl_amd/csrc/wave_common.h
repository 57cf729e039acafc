// Shared declarations of the WAVE engine (wave_engine.hip, wave_viterbi.hip, wave_estep.hip): the
// time-parallel engine for no-overlap ("ring") models, one WAVEFRONT per chain.
//
// Structure exploited (reference types.jl:94-113): N deterministic rings of
// L = K-1 states through one silent state; only the N+1 junction states (silent, each ring's first
// state) have more than one predecessor, and a ring is a delay line of L samples.  Hence, given the
// ring exits X_a(t) = P_a(t-L) of the last L samples, the recursion over the NEXT W <= L samples is
// first order in one scalar (delta / alpha / beta of the silent state):
//     Viterbi   D0(t) = max(D0(t-1) + a_t, b_t)            (max-plus)
//     forward   x_t   = al_t * x_{t-1} + be_t              (linear, per-lane scaled)
// Both are associative, so a wavefront advances W = min(L, 64) samples per step with a 6-level
// cross-lane scan; everything else of those W samples is lane-parallel.  The delay lines live in
// LDS (N x (L+W) doubles per wavefront), never in HBM; the recording is cut into a few thousand
// long chains (one per wavefront, ~4 per SIMD), so the warm-up in front of every chain is a few
// per cent of the sweep instead of half of it (lane-per-chain ring engine, ring_*.hip).
// All per-sample arrays are in natural time order.  tests/wave_model.py is the executable
// specification of the arithmetic (checked against the CPU oracle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "hmmsort_internal.h"
#include "ring_common.h"   // ProfEntry, dispatch_N, kRingMaxN

namespace hmmsort {

constexpr double kScFloor = -700.0;  // entry log-probabilities below this stay in the exponent

struct WaveGeom {
    int64_t T;        // samples per channel
    int C;            // channels (batched plans; 1 otherwise)
    int N, L;         // rings, ring length
    int W;            // samples per super-step = min(L, 64)
    int RB;           // delay-line length in LDS (>= L + W, multiple of 32)
    int B;            // chain length (owned samples), multiple of 64
    int Hw;           // forward-type warm-up: chain c > 0 starts at tc - Hw, Hw - 1 = m*W steps
    int He;           // backward warm-up in samples (multiple of W)
    int nch;          // chains per channel
    int EB, epw, PW;  // psi packing: bits per entry (incl. the near-tie flag), entries/word, words
    int Bb, Hb;       // backtrace segments: length and walk-in (multiples of 64)
    int64_t nseg;     // backtrace segments per channel
    int64_t own_lo, own_hi;  // time shard (hmmsort_plan_set_shard)
    int first, last;
    double thr_scale;        // test aid: multiplier of the near-tie threshold (option "tie_scale")
    int tie_debug;           // test aids (option "tie_debug"): 1 exact prefix folded to the end, 2 resolver off (flagged = unresolved)
};

// per-channel model constants (device table, wave-uniform scalar loads)
struct WaveConst {
    double c00, mean0, den, A, sc0, P00;
    double c0[kRingMaxN], cend[kRingMaxN], sc[kRingMaxN], CP0[kRingMaxN], PEND[kRingMaxN];
    double xishift[kRingMaxN];                 // c0_a - sc_a: log Xi_a = xishift + log Xi'_a
    double cx[kRingMaxN * kRingMaxN];          // [a*N+b]: (a,L) -> (b,1)
    double CPX[kRingMaxN * kRingMaxN];         // [a*N+b] = exp(cx[a,b] - sc_b), 0 on the diagonal
    double cxin[kRingMaxN];                    // cx[b,a] for any b != a when that is the same for all b (uniform_cx)
    double CPXin[kRingMaxN];                   // exp(cxin_a - sc_a)
    double cxT[kRingMaxN * kRingMaxN];         // [b*N+a] = cx[a,b]   (entries INTO ring b, by source ring a)
    double CPXT[kRingMaxN * kRingMaxN];        // [b*N+a] = CPX[a,b]
};

struct WaveDev {
    WaveGeom g{};
    bool prof_on = false;
    bool uniform_cx = true;           // every channel's cx[b,a] is independent of b (bitwise): O(N) junction updates
    std::vector<char> ucx;            // per channel
    const double *bound_y = nullptr;
    hipStream_t side = nullptr, side2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;
    std::vector<ProfEntry> prof;
    int64_t S = 0, K = 0;
    std::vector<RingModel> ring;      // per channel
    std::vector<std::vector<double>> mean;
    std::vector<double> sigma;
    // device model tables (per channel)
    WaveConst *d_cst = nullptr;       // C
    double *d_mean = nullptr;         // C x S
    double *d_meanT = nullptr;        // C x L x N: ring means lag-major (the pre-pass loads a lag's N means as one wide scalar load)
    double *d_msq = nullptr;          // C x N*(L+1)
    double *d_cint = nullptr;         // C x N*(L+1)
    double *d_ctab = nullptr;         // C x (1 + 2N + N*N + N*L)
    int16_t *d_states = nullptr;      // N x S
    // per-sample arrays, natural layout [C][...][T]
    double *Rf = nullptr;             // C x N x T ring scores
    double *W2 = nullptr;             // C x T: sum of y^2 over the L samples from t (0 past the end)
    double *virt = nullptr;           // C x N x (L+1) virtual onsets V[a][j]
    double *ysum = nullptr;           // C x 2: sum y, sum y^2
    uint32_t *psi = nullptr;          // PW x C x T
    double *vpre = nullptr, *vend = nullptr;   // C*nch x (1 + N*L) Viterbi boundary states
    int32_t *vfail = nullptr;         // C*nch
    int32_t *bstate = nullptr;        // C*nseg
    int32_t *redo = nullptr;          // stitch list
    int32_t *final_state = nullptr;   // C
    double *part = nullptr;           // reduction partials (ll)
    double *FA0 = nullptr;            // C x T   log alpha(silent)
    double *FV = nullptr;             // C x N x T scaled onset masses
    double *FREF = nullptr;           // C x T   their scale
    double *fpre = nullptr;           // C*nch x (1 + L*(N+1)): warm-up copy of la0(tc-1), fv/fref of tc-L..tc-1
    double *bpre = nullptr, *bown = nullptr;   // C*nch x (1 + L*(N+1)): backward boundary values (warm-up / own)
    double *rho = nullptr;            // C x N x T onset posteriors
    double *Zc = nullptr;             // C*nch
    double *partS = nullptr;          // C*nch x (3N+3)
    double *partG = nullptr;          // gsum partials
    double *yhead = nullptr;          // C x (N*L + 2): Yn_a(t), t < L (logs) | lb0(0) | z0
    double *extra = nullptr;          // C x 3*N*L
    double *pp = nullptr;             // C x S
    int64_t *diag = nullptr;          // 8
    double *trash = nullptr;          // 64 x 64 doubles: where idle lanes of a partial super-step store (branch-free stores)
    double *dbg = nullptr;            // 64 doubles: debug record of the first failing certificate
    // exact near-tie resolver (wave_ties.hip)
    int64_t *tie_cnt = nullptr;       // C x 8 counters (kTie* below)
    int64_t *tie_list = nullptr;      // C x kTieCap flagged decisions on the decoded path: t * 32 + entry, time order
    int64_t *tie_off = nullptr;       // C x (ntile + 1) list offsets of the tiles of 4096 samples
    int64_t tie_ntile = 0;
    int16_t *tie_walk = nullptr;      // C x kTieLanes x kTieWalk candidate paths (newest sample first)
    double *tie_guess = nullptr;      // C x nblk approximate trellis value at every block start
    double *tie_c = nullptr;          // C x nblk x 2 exact block increments for an even / odd start value
    int32_t *tie_ok = nullptr;        // C x nblk block increments usable (one binade, path unchanged)
    double *tie_v = nullptr;          // C x (nblk + 1) exact trellis values of the decoded path at block starts
    int64_t tie_nblk = 0;
    // captured launch sequences (wave_graphed below)
    struct GraphEntry { uint64_t key[12]; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    bool graphs_on = true;
    int64_t bytes = 0;
    int nparts = 0, gparts = 0;
};

struct WProfScope {
    WaveDev *r;
    hipStream_t st;
    ProfEntry e;
    WProfScope(WaveDev *r_, const char *name, hipStream_t st_) : r(r_), st(st_)
    {
        e.name = name; e.a = nullptr; e.b = nullptr;
        if (r->prof_on && hipEventCreate(&e.a) == hipSuccess && hipEventCreate(&e.b) == hipSuccess)
            (void)hipEventRecord(e.a, st);
    }
    ~WProfScope()
    {
        if (r->prof_on && e.a && e.b) {
            (void)hipEventRecord(e.b, st);
            r->prof.push_back(e);
        }
    }
};
#define WPROF(r, name, st) WProfScope wprof_scope_(r, name, st)

// psi packing with one near-tie flag bit per entry
constexpr int wpsi_bits_c(int N) { int b = 1; while ((1 << b) < N + 1) b++; return b + 1; }
constexpr int wpsi_epw_c(int N) { return 32 / wpsi_bits_c(N); }
constexpr int wpsi_words_c(int N) { return (N + 1 + wpsi_epw_c(N) - 1) / wpsi_epw_c(N); }

// ---- cross-lane helpers (wave64) -------------------------------------------------------------
__device__ __forceinline__ double wave_bcast(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// ---- DPP cross-lane moves (gfx9 DPP controls; 2-cycle VALU moves instead of LDS-crossbar permutes) ----
// row_shr:n = 0x110+n (within rows of 16 lanes), row_bcast:15 = 0x142 (lane 15 of every row to the
// next row), row_bcast:31 = 0x143 (lane 31 to rows 2 and 3), wave_shr:1 = 0x138.  Lanes without a
// source (and rows outside row_mask) keep `old`, which the scans set to the identity element.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov(double old, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROWMASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}

// value of lane-1 (lane 0 receives `carry`)
__device__ __forceinline__ double lane_prev(double v, double carry)
{
    return dpp_mov<0x138, 0xF>(carry, v);
}

// inclusive scan of f_j(x) = max(x + a_j, b_j) over the 64 lanes: on return f_j o ... o f_0 (x) =
// max(x + a, b).  Identity element (0, -inf): combining with it is exact, so no lane predicates.
__device__ __forceinline__ void scan_maxplus(double &a, double &b)
{
#define HS_MP_STEP(CTRL, RM)                                                      \
    {                                                                             \
        const double al = dpp_mov<CTRL, RM>(0.0, a), bl = dpp_mov<CTRL, RM>(-INFINITY, b); \
        b = fmax(bl + a, b);                                                      \
        a = al + a;                                                               \
    }
    HS_MP_STEP(0x111, 0xF) HS_MP_STEP(0x112, 0xF) HS_MP_STEP(0x114, 0xF) HS_MP_STEP(0x118, 0xF)
    HS_MP_STEP(0x142, 0xA) HS_MP_STEP(0x143, 0xC)
#undef HS_MP_STEP
}

// The same scan in single precision, for the forward/backward sweeps: there the max-plus envelope M_t is
// only the SCALE of the linear recursion (any value within a few hundred nats of the true log value
// works, tests/wave_model.py), and it is used consistently (the rounded float, converted back) in
// every exponent, so its precision does not enter the results.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_movf(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROWMASK, 0xF, false));
}
__device__ __forceinline__ void scan_maxplus_f32(float &a, float &b)
{
#define HS_MPF_STEP(CTRL, RM)                                                     \
    {                                                                             \
        const float al = dpp_movf<CTRL, RM>(0.0f, a), bl = dpp_movf<CTRL, RM>(-INFINITY, b); \
        b = fmaxf(bl + a, b);                                                     \
        a = al + a;                                                               \
    }
    HS_MPF_STEP(0x111, 0xF) HS_MPF_STEP(0x112, 0xF) HS_MPF_STEP(0x114, 0xF) HS_MPF_STEP(0x118, 0xF)
    HS_MPF_STEP(0x142, 0xA) HS_MPF_STEP(0x143, 0xC)
#undef HS_MPF_STEP
}
__device__ __forceinline__ float lane_prevf(float v, float carry)
{
    return dpp_movf<0x138, 0xF>(carry, v);
}

// inclusive scan of f_j(x) = a_j * x + b_j.  Identity: (1, 0).
__device__ __forceinline__ void scan_linear(double &a, double &b)
{
#define HS_LN_STEP(CTRL, RM)                                                      \
    {                                                                             \
        const double al = dpp_mov<CTRL, RM>(1.0, a), bl = dpp_mov<CTRL, RM>(0.0, b); \
        b = __builtin_fma(a, bl, b);                                              \
        a = al * a;                                                               \
    }
    HS_LN_STEP(0x111, 0xF) HS_LN_STEP(0x112, 0xF) HS_LN_STEP(0x114, 0xF) HS_LN_STEP(0x118, 0xF)
    HS_LN_STEP(0x142, 0xA) HS_LN_STEP(0x143, 0xC)
#undef HS_LN_STEP
}

// shuffle-based references of the three primitives (self-test only: hmmsort_selftest)
__device__ __forceinline__ double lane_prev_ref(double v, double carry, int lane)
{
    const double s = __shfl_up(v, 1);
    return lane == 0 ? carry : s;
}
__device__ __forceinline__ void scan_maxplus_ref(double &a, double &b, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double al = __shfl_up(a, d), bl = __shfl_up(b, d);
        if (lane >= d) { b = fmax(bl + a, b); a = al + a; }
    }
}
__device__ __forceinline__ void scan_linear_ref(double &a, double &b, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double al = __shfl_up(a, d), bl = __shfl_up(b, d);
        if (lane >= d) { b = __builtin_fma(a, bl, b); a = al * a; }
    }
}

// The model constants are wave-uniform and many (3 + 5N + 2N^2 doubles): left to itself the compiler
// hoists all their loads out of the sweep loop and spills hundreds of SGPRs into VGPRs.  Passing the
// table pointer through an empty asm once per super-step keeps them as scalar loads (scalar cache)
// next to their use.
__device__ __forceinline__ const WaveConst *reload_consts(const WaveConst *p)
{
    asm volatile("" : "+s"(p));
    return p;
}

// software-pipeline depth of the chain kernels (super-steps of input in flight per wavefront)
template <int N> constexpr int wave_depth() { return N <= 4 ? 4 : (N <= 8 ? 3 : 2); }
// register budget of the chain kernels: waves per SIMD the launch bounds ask for
template <int N> constexpr int wave_occ() { return N <= 4 ? 4 : (N <= 8 ? 2 : 1); }

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- exact near-tie resolver (wave_ties.hip) ------------------------------------------------------
constexpr int kTieBlk = 512;                 // samples per block of the exact prefix
constexpr int kTieCap = 65536;               // flagged decisions on the decoded path kept per channel
constexpr int kTieWalk = 8192;               // longest candidate walk before it must have met the decoded path
constexpr int kTieLanes = 64;                // candidates of one decision (junctions: silent + N ring exits; final arg-max: end states)
// tie_cnt[ch * 8 + i]
constexpr int kTieTrig = 0;                  // flagged decisions the backtrace met (trigger; may count one twice)
constexpr int kTieListed = 1;                // flagged decisions on the final path (t >= 2)
constexpr int kTieDone = 2;                  // decisions re-decided with the reference's arithmetic (incl. off-path ones)
constexpr int kTieFlips = 3;                 // ... of which the back-pointer changed
constexpr int kTieOpen = 4;                  // decisions left unresolved
constexpr int kTieTail = 5;                  // final arg-max flagged
constexpr int kTieLongest = 6;               // longest candidate walk
constexpr int kTieSerial = 7;                // prefix blocks run serially

// log-probability of the transition xp -> xc of a ring model (state ids 1-based), from the per-channel
// table ctab = c00 | c0[N] | cend[N] | cx[N*N] | cint[N*L]
__device__ __forceinline__ double wpath_lp(int N, int L, const double *__restrict__ ctab, int xp, int xc)
{
    if (xp == 1) return xc == 1 ? ctab[0] : ctab[1 + (xc - 2) / L];
    const int a = (xp - 2) / L, k = (xp - 2) % L + 1;
    if (k < L) return ctab[1 + 2 * N + N * N + a * L + k];
    if (xc == 1) return ctab[1 + N + a];
    return ctab[1 + 2 * N + a * N + (xc - 2) / L];
}

// Near-tie threshold of the Viterbi sweep: see wave_viterbi.hip.
__device__ __forceinline__ double wave_thr(const WaveGeom &g, const WaveConst &K, const double *__restrict__ ysum, int ch)
{
    // largest magnitude the reference's trellis reaches: |sum_t (A - d^2/den + lp)| <= ...
    const double T = (double)g.T;
    const double s1 = ysum[2 * ch], s2 = ysum[2 * ch + 1];
    const double sq = fmax((s2 - 2.0 * K.mean0 * s1) + T * K.mean0 * K.mean0, 0.0);
    const double mmax = fabs(K.A) * T + sq / K.den + fabs(K.c00) * T + 1.0;
    return (ldexp(16.0 * (double)(g.L + 2), ilogb(mmax) - 52) + 4.0e-9) * g.thr_scale;
}

// A call's launch sequence (two dozen kernels, memsets and cross-stream events on three streams) as ONE
// hipGraph launch: the sequence is captured the first time a call is made with a given set of buffer
// pointers and launch-relevant plan state, and replayed afterwards (model constants live in device tables,
// so hmmsort_plan_set_model does not invalidate it).  Not on the legacy null stream (capture is not allowed
// there), not while per-kernel profiling brackets the launches; any capture failure switches the plan back
// to plain launches for good.
template <typename F>
int wave_graphed(WaveDev *r, int kind, const void *p0, const void *p1, const void *p2, const void *p3,
                 hipStream_t st, F enqueue)
{
    if (!r->graphs_on || r->prof_on || st == nullptr) return enqueue(st);
    uint64_t key[12] = {(uint64_t)kind, (uint64_t)(uintptr_t)p0, (uint64_t)(uintptr_t)p1, (uint64_t)(uintptr_t)p2,
                        (uint64_t)(uintptr_t)p3, (uint64_t)(uintptr_t)st, (uint64_t)(r->bound_y == p0),
                        (uint64_t)r->uniform_cx, (uint64_t)r->g.own_lo, (uint64_t)r->g.own_hi,
                        (uint64_t)(r->g.first * 2 + r->g.last), (uint64_t)(uint32_t)r->g.tie_debug};
    {
        uint64_t tb;
        static_assert(sizeof(tb) == sizeof(r->g.thr_scale), "");
        memcpy(&tb, &r->g.thr_scale, sizeof(tb));
        key[11] ^= tb << 8;
    }
    for (auto &e : r->graphs)
        if (!memcmp(e.key, key, sizeof(key))) {
            HS_HIP(hipGraphLaunch(e.exec, st));
            return HMMSORT_OK;
        }
    if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        r->graphs_on = false;
        return enqueue(st);
    }
    const int rc = enqueue(st);
    hipGraph_t graph = nullptr;
    const hipError_t e1 = hipStreamEndCapture(st, &graph);
    hipGraphExec_t exec = nullptr;
    if (rc || e1 != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        r->graphs_on = false;
        return rc ? rc : enqueue(st);
    }
    (void)hipGraphDestroy(graph);
    if (r->graphs.size() >= 16) {   // callers that cycle through many buffers: forget the oldest
        (void)hipGraphExecDestroy(r->graphs.front().exec);
        r->graphs.erase(r->graphs.begin());
    }
    WaveDev::GraphEntry ge;
    memcpy(ge.key, key, sizeof(key));
    ge.exec = exec;
    r->graphs.push_back(ge);
    HS_HIP(hipGraphLaunch(exec, st));
    return HMMSORT_OK;
}

// wave_engine.hip
bool wave_supported(const HostModel &m, int64_t T, std::string *why);
int wave_create(WaveDev **out, const std::vector<HostModel> &models, int64_t T, int64_t block_req,
                int64_t halo_req);
int wave_set_model(WaveDev *r, int ch, const HostModel &m);
void wave_destroy(WaveDev *r);
int wave_prepare(WaveDev *r, const double *d_y, hipStream_t st);
int wave_bind(WaveDev *r, const double *d_y, hipStream_t st);
int64_t wave_stats_len(const WaveDev *r);
int wave_diagnostics(WaveDev *r, hipStream_t st, int64_t diag[8]);
int wave_profile_read(WaveDev *r, hipStream_t st, std::vector<std::string> &names,
                      std::vector<double> &ms, std::vector<int64_t> &calls);
// wave_viterbi.hip
int wave_viterbi_sweep(WaveDev *r, const double *d_y, hipStream_t st);
int wave_viterbi_post(WaveDev *r, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st);
int wave_viterbi(WaveDev *r, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st);
// wave_ties.hip
int wave_tie_resolve(WaveDev *r, const double *d_y, int16_t *d_x, hipStream_t st);
int wave_tie_stats(WaveDev *r, hipStream_t st, int64_t out[8]);
// wave_estep.hip
int wave_estep(WaveDev *r, const double *d_y, double *d_stats, hipStream_t st);
int wave_mstep(WaveDev *r, const double *d_stats, double *d_out, hipStream_t st);
int wave_decode_estep(WaveDev *r, const double *d_y, int16_t *d_x, double *d_ll, double *d_stats,
                      hipStream_t st);

}  // namespace hmmsort
