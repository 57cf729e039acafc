// placeholder, replaced below by the real ring engine
#include "hmmsort_internal.h"
namespace hmmsort {
struct RingDev { int dummy; };
bool ring_supported(const HostModel &, int64_t, std::string *why) { if (why) *why = "not built yet"; return false; }
int ring_create(RingDev **, const HostModel &, int64_t) { return HMMSORT_EUNSUP; }
int ring_set_model(RingDev *, const HostModel &) { return HMMSORT_EUNSUP; }
void ring_destroy(RingDev *) {}
int64_t ring_workspace_bytes(const RingDev *) { return 0; }
void ring_geometry(const RingDev *, int64_t *, int64_t *, int64_t *) {}
int ring_viterbi(RingDev *, const double *, int16_t *, double *, hipStream_t) { return HMMSORT_EUNSUP; }
int ring_estep(RingDev *, const double *, double *, hipStream_t) { return HMMSORT_EUNSUP; }
int ring_mstep(RingDev *, const double *, double *, hipStream_t) { return HMMSORT_EUNSUP; }
int64_t ring_stats_len(const RingDev *) { return 0; }
int ring_diagnostics(RingDev *, hipStream_t, int64_t *) { return HMMSORT_OK; }
}
