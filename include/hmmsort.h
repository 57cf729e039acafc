/*
 * hmmsort.h -- C ABI of libhmmsort_hip.so: the MI355X (gfx950) implementation of the HMM hot
 * path of grero/HMMSpikeSorter.jl (forward/backward/update, viterbi, reconstruct_signal).
 *
 * The reference (Julia) has NO FFI on this path (no ccall anywhere in /root/reference); the
 * boundary is therefore placed at the narrowest existing seam: the Julia methods listed
 * below.  A drop-in keeps their signatures and replaces their bodies by one ccall each
 * (julia/HMMSpikeSorterHIP.jl).  Each entry point cites the reference method
 * it replaces as file:line relative to the reference repository root.
 *
 * Conventions (all inherited from the reference so Julia arrays can be passed as they are):
 *   - matrices are column-major; state ids and transition endpoints are 1-based;
 *   - `states` is the N x S Int16 matrix StateMatrix.states (types.jl:2): entry = row of mu,
 *     1 = silent;
 *   - `tr` is StateMatrix.transitions (types.jl:3): a Vector{Tuple{Int64,Int64,Float64}} is a
 *     contiguous array of 24-byte isbits tuples == struct hmm_trans, in the reference's order
 *     (source-major, destination ascending, types.jl:115-127).  The order is part of the
 *     contract: it fixes Viterbi tie-breaking and log-sum-exp fold order;
 *   - mu is K x N (baumwelch.jl:314); StateMatrix.pi is never read on this path and is not
 *     passed;
 *   - all buffers are owned by the caller; the library reads/writes them only during the call;
 *   - every function returns 0 on success or a negative HMMSORT_E* code and never throws;
 *     hmmsort_last_error() returns a thread-local message for the last failure;
 *   - calls are synchronous (they return after the GPU work and the copy-back finished).
 *     The library is safe to call from several host threads (per-call streams/workspaces).
 *
 * There is no CPU fallback: without a HIP device every compute entry point fails with
 * HMMSORT_EHIP.
 */
#ifndef HMMSORT_H
#define HMMSORT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMMSORT_OK 0
#define HMMSORT_EINVAL (-1)  /* bad argument / inconsistent model */
#define HMMSORT_ENOMEM (-2)  /* host or device allocation failed / problem too large */
#define HMMSORT_EHIP (-3)    /* HIP runtime error (includes: no device) */
#define HMMSORT_ENOCONV (-4) /* time-parallel engine could not certify its block boundaries */
#define HMMSORT_EUNSUP (-5)  /* model shape not supported by the requested engine */

/* Julia Tuple{Int64,Int64,Float64}  (types.jl:3) */
typedef struct hmm_trans {
    int64_t src;
    int64_t dst;
    double lp;
} hmm_trans;

/* engine selection for hmmsort_set_option("engine", ...) */
#define HMMSORT_ENGINE_AUTO 0   /* ring engine when the model is a no-overlap ring model and
                                   T is long enough, else the generic engine */
#define HMMSORT_ENGINE_STRICT 1 /* generic engine: one sequential sweep per signal in the
                                   reference's operation order (bit-exact Viterbi incl. ll) */
#define HMMSORT_ENGINE_RING 2   /* time-parallel ring engine or fail with HMMSORT_EUNSUP */
#define HMMSORT_ENGINE_BLOCKED 3 /* any transition list (overlap models): the strict recursion run
                                    time-parallel over blocks with a certified warm-up */
#define HMMSORT_ENGINE_WAVE 4   /* no-overlap ring models, one wavefront per chain of a few thousand
                                   samples (lane scans; delay lines in LDS); AUTO's choice for ring
                                   models.  HMMSORT_ENGINE_RING = the older lane-per-chain engine */

const char *hmmsort_last_error(void);
int hmmsort_version(void);
int hmmsort_device_count(int *count);
int hmmsort_set_device(int device);
/* keys: "engine" (above), "block" (chain length in samples, 0 = auto), "halo" (warm-up length, 0 = auto),
 *       "escalate" (host-buffer entry points retry with a wider warm-up / the strict engine when a
 *       boundary certificate or the near-tie guard fires; default 1), "plan_cache" (idle plans the
 *       host-buffer entry points keep between calls, default 4, 0 = none), "strict_limit_mb" (largest
 *       back-pointer table the strict fallback of hmmsort_viterbi may allocate, 0 = what is free); read-only
 *       "last_escalations" (retries of the calling thread's last host-buffer call; NEGATIVE = minus the
 *       number of near-tie decisions on a time-parallel path that was returned because the strict sweep's
 *       S x T back-pointers do not fit: the path can differ from the reference's at those decisions only,
 *       whose margins are inside the reference's own rounding noise; hmmsort_last_error has the text).
 * Process-wide defaults behind a mutex; an entry point works on the snapshot it takes when it starts,
 * so options may be changed while other host threads are inside the library.  hmmsort_last_error is
 * per thread. */
int hmmsort_set_option(const char *key, int64_t value);
int hmmsort_get_option(const char *key, int64_t *value);
/* hmmsort_viterbi / hmmsort_em_step leave their plan, workspace and signal buffers in a small cache
 * keyed by (device, T, state matrix, options) and re-arm it on the next call of the same shape (an EM
 * loop, the channels of a recording).  hmmsort_shutdown frees that cache; plans made with
 * hmmsort_plan_create stay the caller's. */
int hmmsort_shutdown(void);

/* ---- state space helpers (host side, no GPU needed) ------------------------------------ */

/* generate_states(N,K,allow_overlaps) .+ 1   (types.jl:65-92,150).  states_out is N x S Int16
 * (1-based rows of mu) or NULL to query S only.  Returns S (>0) or a negative error. */
int64_t hmmsort_generate_states(int64_t N, int64_t K, int allow_overlaps, int16_t *states_out);

/* get_valid_transitions(states, K, lp)  (types.jl:94-127) in closed form: O(R) instead of the
 * reference's O(S^2 N) all-pairs scan, emitting the SAME list in the SAME order with the same
 * floating-point values (log-probabilities are accumulated neuron by neuron exactly as
 * isvalid_transition does).  tr_out may be NULL to query the count.  Returns R or <0.
 * Replaces the rebuild `StateMatrix(states.-1, pp, K, xb[2:end])` of baumwelch.jl:265. */
int64_t hmmsort_build_transitions(int64_t N, int64_t K, const double *lp, int64_t nlp,
                                  int allow_overlaps, hmm_trans *tr_out, int64_t cap);

/* ---- host-buffer entry points: one per reference method ------------------------------- */

/* viterbi(y, lA::StateMatrix, mu, sigma) -> (x::Vector{Int16}, ll)      viterbi.jl:44-98 */
int hmmsort_viterbi(const double *y, int64_t T, const int16_t *states, int64_t N, int64_t K,
                    int64_t S, const hmm_trans *tr, int64_t R, const double *mu, double sigma,
                    int16_t *x_out, double *ll_out);

/* The same decode from the acquisition's raw samples: the reference's CLI converts the int16 channel to
 * Float64 on the host before fit (src/hmmsort.jl:79-88); here the 2-byte samples cross PCIe and are
 * widened in HBM (exact, so the decode is the one hmmsort_viterbi gives on the converted signal). */
int hmmsort_viterbi_i16(const int16_t *y, int64_t T, const int16_t *states, int64_t N, int64_t K,
                        int64_t S, const hmm_trans *tr, int64_t R, const double *mu, double sigma,
                        int16_t *x_out, double *ll_out);

/* forward(V, lA::StateMatrix, mu, sigma) -> alpha (S x T)            baumwelch.jl:25-51 */
int hmmsort_forward(const double *y, int64_t T, const int16_t *states, int64_t N, int64_t K,
                    int64_t S, const hmm_trans *tr, int64_t R, const double *mu, double sigma,
                    double *alpha_out);

/* backward(V, lA::StateMatrix, mu, sigma) -> beta (S x T)            baumwelch.jl:73-98 */
int hmmsort_backward(const double *y, int64_t T, const int16_t *states, int64_t N, int64_t K,
                     int64_t S, const hmm_trans *tr, int64_t R, const double *mu, double sigma,
                     double *beta_out);

/* update(alpha, beta, lA, mu, sigma, x) -> (StateMatrix, mu, sigma)  baumwelch.jl:205-309
 * mu_inout (K x N) is zeroed and rewritten in place as the reference does (:268).
 * lp_out receives xb[2:end] (:264-265): n_lp_out = (#transitions with src == 1) - 1 values
 * (== N without overlaps); pp_out receives gammaf[:,1] (S values).  The new StateMatrix is
 * rebuilt by the caller from lp_out (hmmsort_build_transitions or the reference constructor). */
int hmmsort_update(const double *alpha, const double *beta, const double *x, int64_t T,
                   const int16_t *states, int64_t N, int64_t K, int64_t S, const hmm_trans *tr,
                   int64_t R, double *mu_inout, double sigma, double *sigma_out, double *lp_out,
                   int64_t lp_cap, int64_t *n_lp_out, double *pp_out);

/* train_model(X, state_matrix, mu0, sigma0) = forward -> backward -> update
 *                                                                    baumwelch.jl:362-370
 * Same outputs as hmmsort_update; alpha/beta are never materialised by the ring engine. */
int hmmsort_em_step(const double *y, int64_t T, const int16_t *states, int64_t N, int64_t K,
                    int64_t S, const hmm_trans *tr, int64_t R, double *mu_inout, double sigma,
                    double *sigma_out, double *lp_out, int64_t lp_cap, int64_t *n_lp_out,
                    double *pp_out);

/* reconstruct_signal(x, lA, mu, sigma) -> Y2 (sigma is ignored)    reconstruction.jl:1-10 */
int hmmsort_reconstruct(const int16_t *x, int64_t T, const int16_t *states, int64_t N,
                        int64_t S, const double *mu, int64_t K, double *y_out);

/* unroll_mlseq(mlseq, state_matrix) -> N x T Int16                   extraction.jl:4-13 */
int hmmsort_unroll_mlseq(const int16_t *mlseq, int64_t T, const int16_t *states, int64_t N,
                         int64_t S, int16_t *out);

/* extract_spiketimes(model)                                              extraction.jl:15-24
 * For neuron i the spike time is every sample whose decoded state has neuron i at the row of its
 * template minimum (indmin(mu[:,i]), first minimum).  times_out is N rows of `cap` entries
 * (1-based sample indices, ascending, as Julia's findin returns them); counts_out[i] is the total
 * number found (entries beyond cap are dropped: call again with a larger cap). */
int hmmsort_extract_spiketimes(const int16_t *mlseq, int64_t T, const int16_t *states, int64_t N,
                               int64_t S, const double *mu, int64_t K, int64_t *times_out,
                               int64_t cap, int64_t *counts_out);

/* ---- device-resident plan API --------------------------------------------------------- */
/* Used by bench.py and by multi-GPU hosts: the signal stays in HBM, the caller owns device
 * buffers (plain device pointers) and the HIP stream (passed as void* == hipStream_t; NULL =
 * the null stream).  No call below synchronises the stream unless noted. */

typedef struct hmmsort_plan hmmsort_plan;

/* Analyse the model, pick the engine and allocate the device workspace for signals of length T
 * (one recording channel).  Synchronous. */
int hmmsort_plan_create(hmmsort_plan **plan_out, int64_t T, const int16_t *states, int64_t N,
                        int64_t K, int64_t S, const hmm_trans *tr, int64_t R, const double *mu,
                        double sigma);
/* Batched plan (SURVEY 8b "batched variants with a leading channel count C"): C recording channels of
 * the same length T and the same model shape, each with its own transition values, templates and sigma
 * (the reference sorts one channel per call with that channel's model, src/hmmsort.jl:79-83).
 *   tr: C lists of R records, mu: C matrices of K x N, sigma: C values.
 * Every plan call then takes channel-major device buffers: d_y [C][T], d_x [C][T], d_ll [C],
 * d_stats [C][hmmsort_plan_stats_len()], d_out [C][K*N + 1 + N + S]; one set of launches sweeps all
 * channels (chains = channels x chains per channel), so short channels still fill the GPU.  Summing the
 * per-channel statistics before hmmsort_plan_mstep gives pooled templates (an extension the reference
 * lacks).  Wave engine only (HMMSORT_EUNSUP otherwise); diagnostics are summed over the channels. */
int hmmsort_plan_create_batched(hmmsort_plan **plan_out, int64_t C, int64_t T, const int16_t *states,
                                int64_t N, int64_t K, int64_t S, const hmm_trans *tr, int64_t R,
                                const double *mu, const double *sigma);
int64_t hmmsort_plan_channels(const hmmsort_plan *plan);
/* hmmsort_plan_set_model for one channel of a batched plan */
int hmmsort_plan_set_model_channel(hmmsort_plan *plan, int64_t channel, const hmm_trans *tr, int64_t R,
                                   const double *mu, double sigma);
/* Replace transitions / mu / sigma (same N, K, S; R may shrink when a template has lost its entry
 * transitions) -- e.g. between EM iterations. */
int hmmsort_plan_set_model(hmmsort_plan *plan, const hmm_trans *tr, int64_t R, const double *mu,
                           double sigma);
int hmmsort_plan_destroy(hmmsort_plan *plan);
/* engine actually used (HMMSORT_ENGINE_STRICT or HMMSORT_ENGINE_RING), geometry, bytes */
int hmmsort_plan_info(const hmmsort_plan *plan, int64_t *engine, int64_t *block, int64_t *halo,
                      int64_t *nchains, int64_t *workspace_bytes);
/* Overlap models on the blocked engine: which sweep the plan's current model runs -- 0 the generic sweeps (any
 * transition list), 2 the two-template sweep (csrc/pair_sweep.hip), 3..5 the multi-template sweep (csrc/multi_sweep.hip);
 * it follows hmmsort_plan_set_model and drops to 0 when a host entry point fell back.  Other engines: 0. */
int64_t hmmsort_plan_overlap_sweep(const hmmsort_plan *plan);

/* Optional: compute the signal-dependent intermediates every ring-engine call needs (transposed
 * copy of y and the ring scores of the current model) ONCE and let the following
 * hmmsort_plan_viterbi / hmmsort_plan_estep calls on the same d_y reuse them.  The binding ends at
 * hmmsort_plan_set_model, hmmsort_plan_unbind or a call with a different pointer.  The caller
 * promises not to modify d_y while it is bound.  No-op for the generic engine. */
int hmmsort_plan_bind(hmmsort_plan *plan, const double *d_y, void *stream);
int hmmsort_plan_unbind(hmmsort_plan *plan);

/* Viterbi decode of d_y[0..T) into d_x[0..T) (device Int16).  d_ll: one device double. */
int hmmsort_plan_viterbi(hmmsort_plan *plan, const double *d_y, int16_t *d_x, double *d_ll,
                         void *stream);
/* E-step: forward-backward + sufficient statistics.  d_stats receives
 * hmmsort_plan_stats_len() doubles (layout: hmmsort_plan_stats_layout below); the vector is a
 * plain sum over time, so shards of one recording / pooled channels combine by a SUM
 * all-reduce (RCCL) before the M-step. */
int hmmsort_plan_estep(hmmsort_plan *plan, const double *d_y, double *d_stats, void *stream);
/* hmmsort_plan_viterbi + hmmsort_plan_estep of the same signal and model in one call: the three
 * serial sweeps (Viterbi, forward, backward) are independent and share one launch, so three times
 * as many wavefronts are resident (wave engine: the Viterbi sweep and its post-processing run on an
 * internal stream beside the forward/backward sweeps).  Same results as the two separate calls.
 * Wave and ring engines. */
int hmmsort_plan_decode_estep(hmmsort_plan *plan, const double *d_y, int16_t *d_x, double *d_ll,
                              double *d_stats, void *stream);
int64_t hmmsort_plan_stats_len(const hmmsort_plan *plan);
/* Time-sharding ONE recording over several plans/GPUs: the plan's signal is the slice
 * [own_lo - halo_before, own_hi + halo_after) of the recording (halos of >= a few ring lengths;
 * the slice's own ends act as warm-up), and the E-step accumulates statistics only for samples /
 * ring onsets in [own_lo, own_hi) (slice coordinates).  `first` / `last` say whether the slice
 * starts / ends the recording (the reference's first-column and terminal conditions apply there).
 * Summing the statistics of all shards (SUM all-reduce) and calling hmmsort_plan_mstep gives the
 * EM step of the whole recording.  pp (gamma[:,1]) is meaningful on the first shard only.
 * Default: the plan owns everything (one shard = the recording). */
int hmmsort_plan_set_shard(hmmsort_plan *plan, int64_t own_lo, int64_t own_hi, int first, int last);
/* M-step finish from (all-reduced) statistics, on device: d_out receives
 * [mu (K*N) | sigma (1) | lp_new | pp (S)]; lp_new = xb[2:end] (baumwelch.jl:264) has one entry per
 * transition leaving state 1 except the first: N entries for models without overlaps.  Blocked plans
 * (overlap models) take the same three calls estep / all-reduce / mstep; their statistics vector is
 * [G0 (S) | G1 (S) | X | Gamma0 | sum y^2]. */
int hmmsort_plan_mstep(hmmsort_plan *plan, const double *d_stats, double *d_out, void *stream);
/* doubles hmmsort_plan_mstep writes PER CHANNEL: K*N + 1 + n_lp + S, n_lp = N for wave/ring plans (also when
 * the list has lost a vanished template's entry transitions, types.jl:121: its slot stays, value -Inf or the
 * re-estimate), (#transitions leaving state 1) - 1 for the blocked engine (baumwelch.jl:226,264). */
int64_t hmmsort_plan_mstep_len(const hmmsort_plan *plan);
/* diagnostics of the last time-parallel call on this plan (synchronises the stream):
 * diag[0] = chain boundaries whose Viterbi warm-up missed the certificate (the warm-up's boundary
 *           scores must equal the previous chain's up to one constant; ring engine 1e-6 on the
 *           relevant entries, wave engine 1e-9 on all 1 + N L entries).  The wave engine re-sweeps a
 *           failing chain exactly from its predecessor's hand-off and counts only boundaries still
 *           open after two such rounds,
 * diag[1] = backtrace stitch repairs (wave engine: + chains re-swept exactly), diag[2] = largest spread seen by that certificate (bit
 *           pattern of a double), diag[3] / diag[5] = chain boundaries whose forward / backward warm-up
 *           missed the posterior-weighted tolerance 1e-9, diag[4] / diag[6] = the largest such
 *           error (IEEE-754 bit pattern of a double).  Viterbi calls fill [0..1], E-step calls
 *           [3..6] (blocked plans: [3..6] from the certificates on gamma of generic_estep.hip).
 *           Blocked engine: diag[0], diag[2] as above for its block boundaries, and
 *           diag[7] = blocks in which two candidates of a maximum came closer than the rounding
 *           granularity of the reference's trellis at that point (near-ties, e.g. duplicate
 *           templates): the blocks' additive frames may then break a tie the reference breaks by
 *           list order; hmmsort_viterbi re-decodes such signals with the strict engine.
 *           Wave engine: junction decisions whose margin is below 16 (L+2) ulp(|T1|max) + 4e-9 (the most the
 *           reference's own rounding can move a difference at the magnitude its trellis reaches on this
 *           signal) are flagged, and the flagged ones ON the decoded path are re-decided on device with the
 *           reference's serial arithmetic (hmmsort_plan_tie_stats); diag[7] = decisions that could NOT be
 *           settled that way (0 on every signal seen; same consequence as above otherwise). */
int hmmsort_plan_diagnostics(hmmsort_plan *plan, void *stream, int64_t diag[8]);

/* Wave engine: what the exact near-tie resolver did in the last decode (synchronises the stream), summed
 * over the channels: out[0] flagged junction decisions the backtrace met (trigger), out[1] flagged decisions
 * on the final path, out[2] decisions re-decided with the reference's serial arithmetic (viterbi.jl:74-84;
 * includes flagged decisions off the path that a candidate's own history ran through), out[3] decisions
 * whose back-pointer changed, out[4] decisions left unresolved (== diag[7]), out[5] channels whose final
 * arg-max (viterbi.jl:90) was flagged, out[6] longest candidate walk in samples (max), out[7] blocks of the
 * exact prefix that were folded serially.  Zeros for other engines. */
int hmmsort_plan_tie_stats(hmmsort_plan *plan, void *stream, int64_t out[8]);

/* reconstruct_signal (reconstruction.jl:1-10) and unroll_mlseq (extraction.jl:4-13) of a decoded
 * path in device memory into device buffers (T doubles / N x T Int16, column-major); the model
 * is the plan's current one.  Out-of-range state ids give NaN / 0 as in the host-buffer kernels'
 * device code; synchronise the stream. */
int hmmsort_plan_reconstruct(hmmsort_plan *plan, const int16_t *d_x, double *d_y_out, void *stream);
int hmmsort_plan_unroll_mlseq(hmmsort_plan *plan, const int16_t *d_x, int16_t *d_out, void *stream);

/* extract_spiketimes (extraction.jl:15-24) on a decoded path that is still in device memory
 * (the x written by hmmsort_plan_viterbi): only the spike times cross PCIe, not the 2 bytes per
 * sample of the path.  Model (states, mu) = the plan's current one.  Host outputs as in
 * hmmsort_extract_spiketimes; synchronises the stream. */
int hmmsort_plan_extract_spiketimes(hmmsort_plan *plan, const int16_t *d_x, int64_t *times_out,
                                    int64_t cap, int64_t *counts_out, void *stream);

/* Widening of raw samples already in device memory into the fp64 signal the plan calls take
 * (src/hmmsort.jl:79-88: `view(data, :, 1)` of the acquisition array, converted to Float64): element i
 * of the output is d_in[i * stride] (stride in elements: 1 for a channel stored contiguously, the
 * channel count for sample-major recordings).  Exact for every source type.  Asynchronous on `stream`. */
#define HMMSORT_SAMPLES_I16 0
#define HMMSORT_SAMPLES_I32 1
#define HMMSORT_SAMPLES_F32 2
#define HMMSORT_SAMPLES_F64 3
int hmmsort_samples_to_f64(const void *d_in, int sample_type, int64_t T, int64_t stride, double *d_out,
                           void *stream);

/* Per-kernel timing of the ring engine with HIP events recorded on the caller's stream (used by
 * bench.py for the roofline line).  hmmsort_plan_profile(plan, 1) switches bracketing on;
 * hmmsort_plan_profile_read synchronises the stream and returns, per kernel name, the total
 * milliseconds and the number of launches since the previous read.  `names` receives the kernel
 * names joined by '\n'. */
int hmmsort_plan_profile(hmmsort_plan *plan, int enable);
int hmmsort_plan_profile_read(hmmsort_plan *plan, void *stream, char *names, int64_t names_cap,
                              double *ms, int64_t *calls, int64_t cap, int64_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* HMMSORT_H */
