// Internal declarations shared by the C ABI (capi.cpp), the host-side state-space code
// (statespace.cpp) and the two device engines (generic_engine.hip, ring_engine.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>
#include "../../include/hmmsort.h"

namespace hmmsort {

void set_error(const char *fmt, ...);
const char *last_error();

#define HS_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            hmmsort::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),         \
                               __FILE__, __LINE__);                                           \
            return HMMSORT_EHIP;                                                              \
        }                                                                                     \
    } while (0)

#define HS_CHECK(cond, code, ...)                                                             \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            hmmsort::set_error(__VA_ARGS__);                                                  \
            return (code);                                                                    \
        }                                                                                     \
    } while (0)

// 0.5*log(2*pi)  (reference utils.jl:1); the double nearest to 0.918938533204672741780...
constexpr double kLog2Pi = 0.9189385332046727;

struct Options {
    int64_t engine = HMMSORT_ENGINE_AUTO;
    int64_t block = 0;
    int64_t halo = 0;
    int64_t escalate = 1;          // host-buffer entry points retry with a doubled warm-up
    int64_t plan_cache = 4;        // idle plans (+ device buffers) the host-buffer entry points keep
    int64_t strict_limit_mb = 0;   // largest back-pointer table the strict fallback may allocate (0 = what is free)
    int64_t tie_scale = 1;         // test aid: multiplies the wave engine's near-tie threshold (more decisions flagged)
    int64_t tie_debug = 0;         // test aids: 1 the resolver folds the exact prefix to the end, 2 resolver off
};
// process-wide options behind a mutex: entry points work on a snapshot taken when they start
Options options_get();
void options_modify(const std::function<void(Options &)> &f);
int64_t &last_escalations();       // per host thread: retries of its last host-buffer call

// ---- host-side model -----------------------------------------------------------------------
// Ring structure of a no-overlap model (reference types.jl:94-113 with allow_overlaps=false):
// N rings of L = K-1 states through one silent state.  All log-probabilities are taken from the
// caller's transition list, not recomputed.
struct RingModel {
    bool valid = false;
    int N = 0, L = 0;
    double c00 = 0;                 // silent -> silent
    std::vector<double> c0;         // [a]      silent -> (a,1)
    std::vector<double> cint;       // [a*L+k]  (a,k) -> (a,k+1), k = 1..L-1 (index k; [a*L+0] unused)
    std::vector<double> cend;       // [a]      (a,L) -> silent
    std::vector<double> cx;         // [a*N+b]  (a,L) -> (b,1), b != a
};

struct HostModel {
    int64_t N = 0, K = 0, S = 0, R = 0;
    std::vector<int16_t> states;    // N x S, 1-based
    std::vector<hmm_trans> tr;      // reference order
    std::vector<double> mu;         // K x N
    double sigma = 0;
    std::vector<double> mean;       // per-state mean, accumulated from 0.0 in neuron order
    // CSR by destination (incoming, list order kept) and by source (outgoing, list order)
    std::vector<int32_t> in_ptr, in_src;
    std::vector<double> in_lp;
    std::vector<int32_t> out_ptr, out_dst;
    std::vector<double> out_lp;
    RingModel ring;
};

int build_host_model(HostModel &m, const int16_t *states, int64_t N, int64_t K, int64_t S,
                     const hmm_trans *tr, int64_t R, const double *mu, double sigma);
int analyze_ring(const HostModel &m, RingModel &ring);

// ---- device engines ------------------------------------------------------------------------
struct GenericDev;  // generic_engine.hip
struct RingDev;     // ring_engine.hip

// generic (strict) engine: single sequential sweep in the reference's operation order
// blocked = time-parallel Viterbi over blocks with a certified warm-up (generic_blocked.hip)
int generic_create(GenericDev **g, const HostModel &m, int64_t T, bool blocked = false,
                   int64_t block_req = 0, int64_t halo_req = 0);
bool generic_is_blocked(const GenericDev *g);
// two-template overlap models: the blocked engine's structure-exploiting sweep (pair_sweep.hip) is in use /
// switch it off for this plan (host fallback to the generic blocked sweep when a near-tie is flagged on the path)
bool generic_pair_active(const GenericDev *g);
void generic_pair_disable(GenericDev *g);
int64_t generic_overlap_sweep(const GenericDev *g);   // 0 generic sweeps, 2 pair sweep, 3..5 multi sweep
void generic_geometry(const GenericDev *g, int64_t *block, int64_t *halo, int64_t *nblocks);
int generic_diagnostics(GenericDev *g, hipStream_t st, int64_t diag[8]);
int64_t blocked_min_samples();
int generic_set_model(GenericDev *g, const HostModel &m);
void generic_destroy(GenericDev *g);
int64_t generic_workspace_bytes(const GenericDev *g);
int generic_viterbi(GenericDev *g, const double *d_y, int16_t *d_x, double *d_ll,
                    hipStream_t st);
int generic_forward(GenericDev *g, const double *d_y, double *d_alpha, hipStream_t st);
int generic_backward(GenericDev *g, const double *d_y, double *d_beta, hipStream_t st);
// update() on device from materialised alpha/beta; d_out = [mu K*N | sigma | lp (nsrc1-1) | pp S]
int generic_update(GenericDev *g, const double *d_alpha, const double *d_beta, const double *d_y,
                   double *d_out, hipStream_t st);
int64_t generic_n_lp(const GenericDev *g);
// time-parallel E-step of the blocked generic engine (generic_estep.hip): sufficient statistics without
// S x T arrays.  stats = [G0 (S) | G1 (S) | X (n_lp + 1) | Gamma0 | sum y^2]
bool blocked_estep_supported(const GenericDev *g);
int64_t blocked_stats_len(const GenericDev *g);
int blocked_estep(GenericDev *g, const double *d_y, double *d_stats, hipStream_t st);
int blocked_mstep(GenericDev *g, const double *d_stats, double *d_out, hipStream_t st);

// ring (time-parallel) engine
int ring_create(RingDev **r, const HostModel &m, int64_t T, int64_t block_req, int64_t halo_req);
int ring_set_model(RingDev *r, const HostModel &m);
void ring_destroy(RingDev *r);
int64_t ring_workspace_bytes(const RingDev *r);
void ring_geometry(const RingDev *r, int64_t *block, int64_t *halo, int64_t *nchains);
int ring_viterbi(RingDev *r, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st);
int ring_estep(RingDev *r, const double *d_y, double *d_stats, hipStream_t st);
int ring_mstep(RingDev *r, const double *d_stats, double *d_out, hipStream_t st);
int64_t ring_stats_len(const RingDev *r);
int ring_diagnostics(RingDev *r, hipStream_t st, int64_t diag[8]);
bool ring_supported(const HostModel &m, int64_t T, std::string *why);

// misc device helpers (generic_engine.hip)
int dev_reconstruct(const int16_t *d_x, int64_t T, const int16_t *d_states, int64_t N, int64_t S,
                    const double *d_mu, int64_t K, double *d_out, hipStream_t st);
int dev_widen(const void *d_in, int dtype, int64_t T, int64_t stride, double *d_out, hipStream_t st);
constexpr int kSpikeChunkHost = 4096;
int dev_spike_compact(const int16_t *d_x, int64_t T, const uint32_t *d_match, int N, int S, int pass,
                      int64_t *d_cnt, const int64_t *d_offs, int64_t *d_times, int64_t cap,
                      hipStream_t st);
int dev_unroll(const int16_t *d_x, int64_t T, const int16_t *d_states, int64_t N, int64_t S,
               int16_t *d_out, hipStream_t st);

}  // namespace hmmsort
