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
    // back-pointers T2 (viterbi.jl:53) of the strict sweep, kept only where a choice exists: npsi x T entries for
    // the states with more than one incoming transition (N + 1 of a ring model's 1 + N L states; 236 of the 3600
    // of the reference's overlap test model); a single-source state's pointer is its source, a state nothing
    // leads to keeps the reference's initial 1.  d_psidx[j] >= 0: row of state j in T2; < 0: -(implied pointer).
    int16_t *d_T2 = nullptr;
    int32_t *d_psidx = nullptr;
    int npsi = 0;
    int64_t t2_rows = 0;      // rows d_T2 was allocated for
    double *d_pv = nullptr;   // T path values
    double *d_last = nullptr; // S last trellis column
    double *d_upd = nullptr;  // update() scratch
    double *d_gbuf = nullptr; // 3*S state-vector scratch when LDS is too small
    bool use_global = false;
    // blocked (time-parallel) Viterbi, generic_blocked.hip
    bool blocked = false, blk_cols_lds = true, blk_tail_lds = true, blk_onecol = false;
    int64_t B = 0, H = 0, nblk = 0;
    int ntail = -1;
    double *d_lp0 = nullptr, *d_tlp = nullptr, *d_endv = nullptr, *d_warmv = nullptr;
    double *d_llpart = nullptr, *d_blkbuf = nullptr;
    int32_t *d_src0 = nullptr, *d_tinfo = nullptr, *d_tsrc = nullptr, *d_bt = nullptr;
    void *d_ms = nullptr;  // MsRec[nms]
    double *d_lpdict = nullptr;  // <= 256 distinct first-transition log-probabilities
    uint8_t *d_lpidx = nullptr;  // [S] index into it
    int ndict = 0;               // 0: more than 256 distinct values, dictionary unusable
    int nms = 0;
    int16_t *d_fmap = nullptr, *d_endstate = nullptr, *d_fconst = nullptr;
    int16_t *d_segbuf = nullptr;   // guess | below of the segment backtrace
    int64_t *d_merged = nullptr;
    unsigned long long *d_bdiag = nullptr, *d_gapmin = nullptr;
    double *d_frame = nullptr;  // per block: frame constant relative to the previous block, |values|
    // structure-exploiting sweep for two-template overlap models (pair_sweep.hip)
    bool pair_ok = false, pair_off = false;   // pair_off: the host fell back to the generic sweep for this plan
    bool multi_ok = false;                    // pair_ok through the 3-5 template sweep (multi_sweep.hip)
    size_t pairtab_len = 0;
    double *d_pairtab = nullptr, *d_qsum = nullptr;
    // time-parallel E-step (generic_estep.hip), allocated on first use
    double *d_es_win = nullptr, *d_es_rec = nullptr, *d_es_partG = nullptr, *d_es_partX = nullptr;
    double *d_es_inw = nullptr, *d_es_outw = nullptr, *d_es_tmp = nullptr;
    unsigned long long *d_es_diag = nullptr;
    int es_grid = 0;
    int64_t upd_bytes = 0;
    int threads = 256;
    int64_t bytes = 0;
};

int blocked_create(GenericDev *g, const HostModel &m, int64_t block_req, int64_t halo_req);
// pair_sweep.hip
bool pair_analyze(const HostModel &m, std::vector<double> &tab);
int pair_sweep_launch(GenericDev *g, const double *d_y, hipStream_t st);
int pair_ties_launch(GenericDev *g, const int16_t *d_x, hipStream_t st);
int pair_mag_launch(GenericDev *g, const double *d_y, hipStream_t st);
// multi_sweep.hip (three to five templates)
bool multi_analyze(const HostModel &m, std::vector<double> &tab);
bool multi_rows_ok(const HostModel &m, const std::vector<int32_t> &ms_states);
size_t multi_lds_bytes(int N, int L);
int multi_sweep_launch(GenericDev *g, const double *d_y, hipStream_t st);
int blocked_set_model(GenericDev *g, const HostModel &m);
void blocked_destroy(GenericDev *g);
int blocked_viterbi(GenericDev *g, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st);
int blocked_diagnostics(GenericDev *g, hipStream_t st, int64_t diag[8]);
bool blocked_estep_supported(const GenericDev *g);
int64_t blocked_stats_len(const GenericDev *g);
int blocked_estep(GenericDev *g, const double *d_y, double *d_stats, hipStream_t st);
int blocked_mstep(GenericDev *g, const double *d_stats, double *d_out, hipStream_t st);
int blocked_estep_diagnostics(GenericDev *g, hipStream_t st, int64_t diag[8]);
void blocked_estep_destroy(GenericDev *g);
void blocked_geometry(int64_t T, int64_t L, int64_t block_req, int64_t halo_req, int64_t *B,
                      int64_t *H, int64_t *nblk);

}  // namespace hmmsort
