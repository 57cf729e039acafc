// Shared declarations of the time-parallel ring engine (ring_engine.hip, ring_viterbi.hip,
// ring_estep.hip).  See DESIGN.md "Ring engine" for the derivation; short version:
//
// Without overlaps the model (reference types.jl:94-113) is N deterministic rings of L = K-1
// states through one silent state.  Only N+1 states have more than one predecessor: the silent
// state and each ring's first state ("junctions").  A ring is a pure delay line: the value that
// enters ring a at time t' leaves it at t'+L-1 having collected the ring score
//     Rfull_a(t') = sum_{k=1..L} q(y[t'+k-1]; mean(a,k)) + sum_{k=1..L-1} lp((a,k)->(a,k+1)),
// which does not depend on the recursion and is computed for all t' in parallel ("pre-pass").
// The serial part of Viterbi / forward / backward then touches N+1 numbers per sample.
//
// Time parallelism: the recording is cut into chains of B samples; chain c is advanced by ONE
// LANE (64 chains per wavefront, no cross-lane traffic) from a warm-up start H samples early.
// All per-sample arrays are stored TRANSPOSED, element (row = t mod B, column = t div B), so that
// the 64 lanes of a wavefront (64 consecutive chains, same step) touch 512 contiguous bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <string>
#include <vector>

#include "hmmsort_internal.h"

namespace hmmsort {

constexpr int kRingMaxN = 16;
constexpr int kRingMinL = 16;

struct RingGeom {
    int64_t T;     // samples
    int N, L;      // rings, ring length (K-1)
    int B, H;      // chain length, warm-up length (multiples of 64, H <= B)
    int nch;       // chains = ceil(T/B); every chain has >= L samples
    int ncol;      // columns of the transposed arrays (nch rounded up to 64)
    int bits, epw, W;  // psi packing: bits per entry, entries per 32-bit word, words per sample
    // time shard of a longer recording (hmmsort_plan_set_shard): statistics are accumulated for
    // the owned samples/onsets [own_lo, own_hi) only; first/last: the shard starts/ends the recording
    int64_t own_lo, own_hi;
    int first, last;
};

// junction constants, passed to kernels by value (kernarg segment -> scalar loads)
template <int N>
struct JParams {
    double c00;         // silent -> silent
    double c0[N];       // silent -> (a,1)
    double cend[N];     // (a,L)  -> silent
    double cx[N * N];   // (a,L)  -> (b,1)   [a*N+b]
    double mean0;       // mean of the silent state
    double den;         // 2*sigma^2
    double A;           // -log2pi - log(sigma): the per-sample emission constant
};

// same constants in the linear domain for the log-sum-exp junction updates
template <int N>
struct EParams {
    double p00;
    double p0[N];
    double pend[N];
    double px[N * N];
    double mean0;
    double den;
};

// optional per-kernel timing with HIP events on the caller's stream (bench.py / profiling)
struct ProfEntry {
    const char *name;
    hipEvent_t a, b;
};

struct RingDev {
    RingGeom g{};
    bool prof_on = false;
    const double *bound_y = nullptr;  // signal whose yT/Rf are current (hmmsort_plan_bind)
    hipStream_t side = nullptr;        // internal stream: decode post-processing beside the E-step's
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_post = nullptr, ev_chk = nullptr,
               ev_edges = nullptr;
    std::vector<ProfEntry> prof;
    int64_t S = 0, K = 0;
    double sigma = 0, lsig = 0, A = 0, den = 0;
    RingModel ring;               // host copy of the transition constants
    std::vector<double> mean;     // host copy of per-state means
    // device model tables
    double *d_mean = nullptr;     // S
    double *d_msq = nullptr;      // N*(L+1): Msq[a][kk] = sum_{k=1}^{kk} mean(a,k)^2
    double *d_cint = nullptr;     // N*(L+1): Cint[a][kk] = sum_{k=1}^{kk-1} lp((a,k)->(a,k+1))
    double *d_ctab = nullptr;     // 1 + N + N + N*N + N*L: c00 | c0 | cend | cx | cint
    int16_t *d_states = nullptr;  // N x S (for the M-step pack)
    // device work arrays (allocated once per plan)
    double *yT = nullptr;         // B x ncol
    double *Rf = nullptr;         // N planes of B x ncol
    double *P = nullptr;          // N planes of (H+B) x ncol   (forward lp)
    double *Pv = nullptr;         // N planes of (H+B) x ncol   (Viterbi delay line)
    double *Q = nullptr;          // N planes of (L+B+H) x ncol (bwd ly)                 [E-step]
    double *A0 = nullptr;         // (1+B) x ncol fwd silent (row 0 = value before the chain)
    double *B0 = nullptr;         // B x ncol bwd silent
    uint32_t *psi = nullptr;      // W planes of B x ncol
    double *D0end = nullptr;      // ncol
    double *D0pre = nullptr;      // ncol: delta(silent) at tc-1 in chain c's warm-up frame
    int32_t *bstate = nullptr;    // ncol
    int32_t *redo = nullptr;      // list of chains to re-walk (capacity ncol)
    int16_t *xT = nullptr;        // B x ncol
    int32_t *final_state = nullptr;
    double *part = nullptr;       // reduction partials
    double *Zc = nullptr;         // ncol per-chain normalisers
    double *Zp = nullptr;         // 4 x 2 x ncol partial (max, sum) of the normaliser
    double *B0h = nullptr;        // ncol: bwd silent value one step past the chain (warm-up side)
    double *partA = nullptr;      // (ncol/64) x 2 x N*L per-wave spike-triggered sums (G1 | G2)
    double *partS = nullptr;      // (ncol/64) x (2N+3) per-wave scalar sums
    double *rhoT = nullptr;       // N planes of B x ncol: onset posteriors rho_a(t')
    double *extra = nullptr;      // 3*N*L edge corrections (virtual onsets, end of data)
    double *pp = nullptr;         // S: gamma[:,1] in the log domain
    int64_t *diag = nullptr;      // 8 device counters
    int64_t bytes = 0;
    int nparts = 0;
};

struct ProfScope {
    RingDev *r;
    hipStream_t st;
    ProfEntry e;
    ProfScope(RingDev *r_, const char *name, hipStream_t st_) : r(r_), st(st_)
    {
        e.name = name; e.a = nullptr; e.b = nullptr;
        if (r->prof_on && hipEventCreate(&e.a) == hipSuccess && hipEventCreate(&e.b) == hipSuccess)
            (void)hipEventRecord(e.a, st);
    }
    ~ProfScope()
    {
        if (r->prof_on && e.a && e.b) {
            (void)hipEventRecord(e.b, st);
            r->prof.push_back(e);
        }
    }
};
#define PROF(r, name, st) ProfScope prof_scope_(r, name, st)

template <typename F>
inline int dispatch_N(int N, F &&f)
{
    switch (N) {
    case 1: return f(std::integral_constant<int, 1>());
    case 2: return f(std::integral_constant<int, 2>());
    case 3: return f(std::integral_constant<int, 3>());
    case 4: return f(std::integral_constant<int, 4>());
    case 5: return f(std::integral_constant<int, 5>());
    case 6: return f(std::integral_constant<int, 6>());
    case 7: return f(std::integral_constant<int, 7>());
    case 8: return f(std::integral_constant<int, 8>());
    case 9: return f(std::integral_constant<int, 9>());
    case 10: return f(std::integral_constant<int, 10>());
    case 11: return f(std::integral_constant<int, 11>());
    case 12: return f(std::integral_constant<int, 12>());
    case 13: return f(std::integral_constant<int, 13>());
    case 14: return f(std::integral_constant<int, 14>());
    case 15: return f(std::integral_constant<int, 15>());
    case 16: return f(std::integral_constant<int, 16>());
    }
    set_error("ring engine: N = %d outside 1..%d", N, kRingMaxN);
    return HMMSORT_EUNSUP;
}

// psi packing as compile-time functions of N (must agree with RingGeom.bits/epw/W)
constexpr int psi_bits_c(int N) { int b = 1; while ((1 << b) < N + 1) b++; return b; }
constexpr int psi_epw_c(int N) { return 32 / psi_bits_c(N); }
constexpr int psi_words_c(int N) { return (N + 1 + psi_epw_c(N) - 1) / psi_epw_c(N); }

// steps per software-pipelined batch of the chain kernels
template <int N> constexpr int chain_unroll() { return N <= 4 ? 4 : (N <= 8 ? 2 : 1); }
// rows per thread of the pre-pass
template <int N> constexpr int prepass_rows() { return N <= 8 ? 8 : 4; }

template <int N>
inline JParams<N> make_jparams(const RingDev *r)
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

template <int N>
inline EParams<N> make_eparams(const RingDev *r)
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

// ring_viterbi.hip
int ring_viterbi_launch(RingDev *r, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st);
int ring_viterbi_post(RingDev *r, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st);
// ring_estep.hip
int ring_estep_launch(RingDev *r, const double *d_y, double *d_stats, hipStream_t st);
int ring_estep_post(RingDev *r, const double *d_y, double *d_stats, hipStream_t st);
// ring_fused.hip
int ring_decode_estep_launch(RingDev *r, const double *d_y, int16_t *d_x, double *d_ll, double *d_stats,
                             hipStream_t st);
int ring_mstep_launch(RingDev *r, const double *d_stats, double *d_out, hipStream_t st);
// shared launch helpers (ring_engine.hip)
int ring_launch_transpose_in(RingDev *r, const double *d_y, hipStream_t st);
int ring_launch_prepass(RingDev *r, hipStream_t st);
int ring_launch_virtual(RingDev *r, const double *d_y, double *dst_planes, int64_t plane_stride,
                        hipStream_t st, double *dst_planes2 = nullptr);
int ring_bind(RingDev *r, const double *d_y, hipStream_t st);
int ring_prepare(RingDev *r, const double *d_y, hipStream_t st);
int ring_profile_enable(RingDev *r, int on);
int ring_profile_read(RingDev *r, hipStream_t st, std::vector<std::string> &names,
                      std::vector<double> &ms, std::vector<int64_t> &calls);

}  // namespace hmmsort
