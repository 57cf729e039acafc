// Device-side state of the generic engine (shared by generic_engine.hip and generic_update.hip).
#pragma once
#include "hmmsort_internal.h"

namespace hmmsort {

struct GenericDev {
    int64_t N = 0, K = 0, S = 0, R = 0, T = 0;
    double sigma = 0, lsig = 0;
    int nsrc1 = 0;  // number of transitions leaving state 1 (update(): tidx)
    double *d_mean = nullptr, *d_in_lp = nullptr, *d_out_lp = nullptr, *d_mu = nullptr;
    int32_t *d_in_ptr = nullptr, *d_in_src = nullptr, *d_out_ptr = nullptr, *d_out_dst = nullptr;
    int16_t *d_states = nullptr;
    int16_t *d_T2 = nullptr;  // S x T back-pointers (viterbi.jl:53), allocated on first decode
    double *d_pv = nullptr;   // T path values
    double *d_last = nullptr; // S last trellis column
    double *d_upd = nullptr;  // update() scratch
    double *d_gbuf = nullptr; // 3*S state-vector scratch when LDS is too small
    bool use_global = false;
    int64_t upd_bytes = 0;
    int threads = 256;
    int64_t bytes = 0;
};

}  // namespace hmmsort
