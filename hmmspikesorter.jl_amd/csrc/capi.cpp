// C ABI of libhmmsort_hip.so (see include/hmmsort.h).  Thin: argument checks, engine choice,
// device buffers for the host-pointer entry points.  No CPU compute path exists here.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>

#include "hmmsort_internal.h"
#include "ring_common.h"
#include "wave_common.h"

using namespace hmmsort;

struct hmmsort_plan {
    HostModel model;
    int64_t T = 0;
    int64_t engine = HMMSORT_ENGINE_STRICT;
    GenericDev *gen = nullptr;
    RingDev *ring = nullptr;
    WaveDev *wave = nullptr;
    int64_t C = 1;                      // channels (batched wave plans)
    std::vector<HostModel> models;      // per-channel models of a batched plan (models[0] == model)
};

namespace {

struct DevBuf {  // RAII device buffer for the host-pointer entry points
    void *p = nullptr;
    size_t cap = 0;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    int alloc(size_t bytes)
    {
        release();
        bytes = std::max<size_t>(bytes, 8);
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            set_error("hipMalloc of %zu bytes failed", bytes);
            p = nullptr;
            return HMMSORT_ENOMEM;
        }
        cap = bytes;
        return HMMSORT_OK;
    }
    // cached buffers of a host slot: keep when large enough, else replace (a re-armed or rebuilt plan may
    // need more: blocked statistics grow with the finite entry transitions, a wave plan needs 3NL+N+4)
    int ensure(size_t bytes) { return (p && cap >= std::max<size_t>(bytes, 8)) ? HMMSORT_OK : alloc(bytes); }
    template <typename Tv> Tv *as() { return static_cast<Tv *>(p); }
};

struct PlanGuard {
    hmmsort_plan *p = nullptr;
    ~PlanGuard() { if (p) hmmsort_plan_destroy(p); }
};

// ---- idle plans of the host-buffer entry points ---------------------------------------------
// hmmsort_viterbi / hmmsort_em_step are what a reference-side binding calls once per EM iteration or per
// channel: same T, same model shape, new numbers.  Creating the plan (workspace hipMalloc,
// geometry) and the signal/output buffers costs more than the sweeps, so an entry point leaves its plan and
// buffers here when it returns and the next call with the same key takes them and re-arms the plan with
// hmmsort_plan_set_model.  A slot is owned by exactly one call while in use (taken OUT of the list), so
// host threads never share a plan; the list itself is behind a mutex.  hmmsort_shutdown() empties it.
struct HostSlot {
    hmmsort_plan *plan = nullptr;
    DevBuf dy, dx, dll, dstats, dout;
    int64_t T = 0, engine_opt = 0, block = 0, halo = 0;
    int device = 0;
    // every slot works on a stream of its own and waits for that stream only: host threads that decode or
    // train at the same time overlap on the device instead of meeting in hipDeviceSynchronize
    hipStream_t st = nullptr;
    ~HostSlot()
    {
        if (plan) hmmsort_plan_destroy(plan);
        if (st) (void)hipStreamDestroy(st);
    }
    void drop_plan()
    {
        if (plan) hmmsort_plan_destroy(plan);
        plan = nullptr;
    }
};
std::mutex g_slots_mu;
std::vector<std::unique_ptr<HostSlot>> g_slots;  // idle, least recently used first

void trim_slots(size_t keep)
{
    std::vector<std::unique_ptr<HostSlot>> dead;
    {
        std::lock_guard<std::mutex> lk(g_slots_mu);
        while (g_slots.size() > keep) {
            dead.push_back(std::move(g_slots.front()));
            g_slots.erase(g_slots.begin());
        }
    }
    // hipFree outside the lock
}

std::unique_ptr<HostSlot> take_slot(int64_t T, const int16_t *states, int64_t N, int64_t K, int64_t S,
                                    const Options &opt)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_slots_mu);
    for (size_t i = g_slots.size(); i-- > 0;) {
        HostSlot &h = *g_slots[i];
        const HostModel &m = h.plan->model;
        if (h.T != T || h.device != dev || h.engine_opt != opt.engine || h.block != opt.block ||
            h.halo != opt.halo || m.N != N || m.K != K || m.S != S)
            continue;
        if (memcmp(m.states.data(), states, m.states.size() * sizeof(int16_t))) continue;
        std::unique_ptr<HostSlot> out = std::move(g_slots[i]);
        g_slots.erase(g_slots.begin() + i);
        return out;
    }
    return nullptr;
}

void give_slot(std::unique_ptr<HostSlot> slot, const Options &opt)
{
    if (!slot || !slot->plan || opt.plan_cache <= 0) return;
    {
        std::lock_guard<std::mutex> lk(g_slots_mu);
        g_slots.push_back(std::move(slot));
    }
    trim_slots((size_t)options_get().plan_cache);
}

std::unique_ptr<HostSlot> new_slot(int64_t T, const Options &opt)
{
    std::unique_ptr<HostSlot> h(new HostSlot());
    h->T = T;
    h->engine_opt = opt.engine;
    h->block = opt.block;
    h->halo = opt.halo;
    if (hipGetDevice(&h->device) != hipSuccess) (void)hipGetLastError();
    if (hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        h->st = nullptr;   // the null stream still works, it only serialises
    }
    return h;
}

// two rings with bit-identical templates and entry probabilities: every decision between them is a
// tie up to the rounding of the reference's own sums (near-ties)
bool ring_has_twins(const HostModel &m)
{
    if (!m.ring.valid) return false;
    const int N = m.ring.N, L = m.ring.L;
    for (int a = 0; a < N; a++)
        for (int b = a + 1; b < N; b++) {
            if (m.ring.c0[a] != m.ring.c0[b]) continue;
            bool same = true;
            for (int k = 0; k < L && same; k++)
                same = m.mean[1 + (size_t)a * L + k] == m.mean[1 + (size_t)b * L + k];
            if (same) return true;
        }
    return false;
}

int need_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1) {
        (void)hipGetLastError();
        set_error("no HIP device available (%s); libhmmsort_hip has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return HMMSORT_EHIP;
    }
    return HMMSORT_OK;
}

int plan_create_engine(hmmsort_plan **out, int64_t T, const int16_t *states, int64_t N, int64_t K,
                       int64_t S, const hmm_trans *tr, int64_t R, const double *mu, double sigma,
                       int64_t engine_req, int64_t halo_req = -1)
{
    HS_CHECK(out, HMMSORT_EINVAL, "plan_create: null output pointer");
    *out = nullptr;
    HS_CHECK(T >= 1, HMMSORT_EINVAL, "plan_create: T must be >= 1 (got %lld)", (long long)T);
    int rc = need_device();
    if (rc) return rc;
    const Options opt = options_get();
    hmmsort_plan *p = new hmmsort_plan();
    PlanGuard guard{p};
    p->T = T;
    rc = build_host_model(p->model, states, N, K, S, tr, R, mu, sigma);
    if (rc) return rc;
    std::string why;
    const bool wave_ok = wave_supported(p->model, T, &why);
    if (engine_req == HMMSORT_ENGINE_WAVE && !wave_ok) {
        set_error("wave engine unavailable for this model/signal: %s", why.c_str());
        return HMMSORT_EUNSUP;
    }
    const bool ring_ok = ring_supported(p->model, T, &why);
    if (engine_req == HMMSORT_ENGINE_RING && !ring_ok) {
        set_error("ring engine unavailable for this model/signal: %s", why.c_str());
        return HMMSORT_EUNSUP;
    }
    // the round-1 lane-per-chain engine is kept as a second implementation for cross-checks only: AUTO never
    // picks it (what the wave engine does not take goes to the blocked / strict engines, which take any list)
    const bool want_ring = engine_req == HMMSORT_ENGINE_RING;
    if ((engine_req == HMMSORT_ENGINE_AUTO || engine_req == HMMSORT_ENGINE_WAVE) && wave_ok) {
        p->engine = HMMSORT_ENGINE_WAVE;
        p->models.assign(1, p->model);
        rc = wave_create(&p->wave, p->models, T, opt.block, halo_req >= 0 ? halo_req : opt.halo);
    } else if (want_ring && ring_ok) {
        p->engine = HMMSORT_ENGINE_RING;
        rc = ring_create(&p->ring, p->model, T, opt.block, halo_req >= 0 ? halo_req : opt.halo);
    } else if (engine_req == HMMSORT_ENGINE_BLOCKED ||
               (engine_req == HMMSORT_ENGINE_AUTO && T >= blocked_min_samples())) {
        // overlap models and other lists the ring engine does not take: blocked sweep
        p->engine = HMMSORT_ENGINE_BLOCKED;
        rc = generic_create(&p->gen, p->model, T, true, opt.block, halo_req >= 0 ? halo_req : opt.halo);
        if (rc == HMMSORT_EUNSUP && engine_req == HMMSORT_ENGINE_AUTO) {
            // a list the blocked sweep does not take (in-degree > 256): op-for-op single sweep
            p->engine = HMMSORT_ENGINE_STRICT;
            p->gen = nullptr;
            rc = generic_create(&p->gen, p->model, T);
        }
    } else {
        p->engine = HMMSORT_ENGINE_STRICT;
        rc = generic_create(&p->gen, p->model, T);
    }
    if (rc) return rc;
    guard.p = nullptr;
    *out = p;
    return HMMSORT_OK;
}

}  // namespace

extern "C" {

const char *hmmsort_last_error(void) { return last_error(); }
int hmmsort_version(void) { return 100; /* 0.1.0 */ }

int hmmsort_device_count(int *count)
{
    HS_CHECK(count, HMMSORT_EINVAL, "device_count: null pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return HMMSORT_OK;
}

int hmmsort_set_device(int device)
{
    HS_HIP(hipSetDevice(device));
    return HMMSORT_OK;
}

int hmmsort_set_option(const char *key, int64_t value)
{
    HS_CHECK(key, HMMSORT_EINVAL, "set_option: null key");
    if (!strcmp(key, "engine")) {
        HS_CHECK(value >= 0 && value <= 4, HMMSORT_EINVAL, "set_option: engine must be 0..4");
        options_modify([&](Options &o) { o.engine = value; });
    } else if (!strcmp(key, "block")) {
        HS_CHECK(value >= 0, HMMSORT_EINVAL, "set_option: block must be >= 0");
        options_modify([&](Options &o) { o.block = value; });
    } else if (!strcmp(key, "halo")) {
        HS_CHECK(value >= 0, HMMSORT_EINVAL, "set_option: halo must be >= 0");
        options_modify([&](Options &o) { o.halo = value; });
    } else if (!strcmp(key, "escalate")) {
        options_modify([&](Options &o) { o.escalate = value != 0; });
    } else if (!strcmp(key, "strict_limit_mb")) {
        HS_CHECK(value >= 0, HMMSORT_EINVAL, "set_option: strict_limit_mb must be >= 0");
        options_modify([&](Options &o) { o.strict_limit_mb = value; });
    } else if (!strcmp(key, "tie_scale")) {
        HS_CHECK(value >= 1, HMMSORT_EINVAL, "set_option: tie_scale must be >= 1");
        options_modify([&](Options &o) { o.tie_scale = value; });
    } else if (!strcmp(key, "tie_debug")) {
        HS_CHECK(value >= 0 && value <= 3, HMMSORT_EINVAL, "set_option: tie_debug must be 0..3");
        options_modify([&](Options &o) { o.tie_debug = value; });
    } else if (!strcmp(key, "plan_cache")) {
        HS_CHECK(value >= 0 && value <= 64, HMMSORT_EINVAL, "set_option: plan_cache must be 0..64");
        options_modify([&](Options &o) { o.plan_cache = value; });
        trim_slots((size_t)value);
    } else {
        set_error("set_option: unknown key '%s'", key);
        return HMMSORT_EINVAL;
    }
    return HMMSORT_OK;
}

int hmmsort_get_option(const char *key, int64_t *value)
{
    HS_CHECK(key && value, HMMSORT_EINVAL, "get_option: null argument");
    const Options o = options_get();
    if (!strcmp(key, "engine")) *value = o.engine;
    else if (!strcmp(key, "block")) *value = o.block;
    else if (!strcmp(key, "halo")) *value = o.halo;
    else if (!strcmp(key, "escalate")) *value = o.escalate;
    else if (!strcmp(key, "plan_cache")) *value = o.plan_cache;
    else if (!strcmp(key, "strict_limit_mb")) *value = o.strict_limit_mb;
    else if (!strcmp(key, "tie_scale")) *value = o.tie_scale;
    else if (!strcmp(key, "tie_debug")) *value = o.tie_debug;
    else if (!strcmp(key, "last_escalations")) *value = last_escalations();
    else {
        set_error("get_option: unknown key '%s'", key);
        return HMMSORT_EINVAL;
    }
    return HMMSORT_OK;
}

// frees what the library keeps between calls: the idle plans and device buffers of the host-buffer
// entry points (plans the caller created stay the caller's to destroy)
int hmmsort_shutdown(void)
{
    trim_slots(0);
    return HMMSORT_OK;
}

// ---- plan API ------------------------------------------------------------------------------

int hmmsort_plan_create(hmmsort_plan **plan_out, int64_t T, const int16_t *states, int64_t N,
                        int64_t K, int64_t S, const hmm_trans *tr, int64_t R, const double *mu,
                        double sigma)
{
    return plan_create_engine(plan_out, T, states, N, K, S, tr, R, mu, sigma, options_get().engine);
}

// Batched plan: C recording channels of the same length and model SHAPE, each with its own transition
// values / templates / sigma (the reference sorts one channel per call with its own model,
// src/hmmsort.jl:79-83); one set of launches sweeps all channels (chains = channels x chains per channel).
int hmmsort_plan_create_batched(hmmsort_plan **plan_out, int64_t C, int64_t T, const int16_t *states,
                                int64_t N, int64_t K, int64_t S, const hmm_trans *tr, int64_t R,
                                const double *mu, const double *sigma)
{
    HS_CHECK(plan_out, HMMSORT_EINVAL, "plan_create_batched: null output pointer");
    *plan_out = nullptr;
    HS_CHECK(C >= 1 && C <= 4096 && T >= 1 && tr && mu && sigma, HMMSORT_EINVAL,
             "plan_create_batched: bad argument (C = %lld, T = %lld)", (long long)C, (long long)T);
    int rc = need_device();
    if (rc) return rc;
    hmmsort_plan *p = new hmmsort_plan();
    PlanGuard guard{p};
    p->T = T;
    p->C = C;
    p->models.resize(C);
    for (int64_t ch = 0; ch < C; ch++) {
        rc = build_host_model(p->models[ch], states, N, K, S, tr + ch * R, R, mu + ch * K * N, sigma[ch]);
        if (rc) return rc;
    }
    p->model = p->models[0];
    std::string why;
    if (!wave_supported(p->model, T, &why)) {
        set_error("plan_create_batched needs the wave engine: %s", why.c_str());
        return HMMSORT_EUNSUP;
    }
    p->engine = HMMSORT_ENGINE_WAVE;
    const Options opt = options_get();
    rc = wave_create(&p->wave, p->models, T, opt.block, opt.halo);
    if (rc) return rc;
    guard.p = nullptr;
    *plan_out = p;
    return HMMSORT_OK;
}

int64_t hmmsort_plan_channels(const hmmsort_plan *p) { return p ? p->C : 0; }

int hmmsort_plan_set_model_channel(hmmsort_plan *p, int64_t channel, const hmm_trans *tr, int64_t R,
                                   const double *mu, double sigma)
{
    HS_CHECK(p && tr && mu, HMMSORT_EINVAL, "plan_set_model_channel: null argument");
    HS_CHECK(p->wave, HMMSORT_EUNSUP, "plan_set_model_channel: needs a wave-engine plan");
    HS_CHECK(channel >= 0 && channel < p->C, HMMSORT_EINVAL, "plan_set_model_channel: channel %lld outside 0..%lld",
             (long long)channel, (long long)p->C - 1);
    HostModel m;
    std::vector<int16_t> st = p->model.states;
    int rc = build_host_model(m, st.data(), p->model.N, p->model.K, p->model.S, tr, R, mu, sigma);
    if (rc) return rc;
    HS_CHECK(m.ring.valid, HMMSORT_EUNSUP, "plan_set_model_channel: new model is not a ring model");
    if ((rc = wave_set_model(p->wave, (int)channel, m))) return rc;
    p->models[channel] = m;
    if (channel == 0) p->model = std::move(m);
    return HMMSORT_OK;
}

int hmmsort_plan_set_model(hmmsort_plan *p, const hmm_trans *tr, int64_t R, const double *mu,
                           double sigma)
{
    HS_CHECK(p && tr && mu, HMMSORT_EINVAL, "plan_set_model: null argument");
    // the ring engines take the list apart into junction constants, so a list that has lost the entry
    // transitions of a vanished template (types.jl:121 keeps finite entries only) fits the same plan
    HS_CHECK(R == p->model.R || p->wave || p->ring, HMMSORT_EINVAL,
             "plan_set_model: R changed (%lld -> %lld)", (long long)p->model.R, (long long)R);
    HostModel m;
    std::vector<int16_t> st = p->model.states;
    int rc = build_host_model(m, st.data(), p->model.N, p->model.K, p->model.S, tr, R, mu, sigma);
    if (rc) return rc;
    if (p->wave) {
        HS_CHECK(m.ring.valid, HMMSORT_EUNSUP, "plan_set_model: new model is not a ring model");
        rc = wave_set_model(p->wave, 0, m);
        if (!rc) p->models[0] = m;
    } else if (p->ring) {
        HS_CHECK(m.ring.valid, HMMSORT_EUNSUP, "plan_set_model: new model is not a ring model");
        rc = ring_set_model(p->ring, m);
    } else {
        rc = generic_set_model(p->gen, m);
    }
    if (rc) return rc;
    p->model = std::move(m);
    return HMMSORT_OK;
}

int hmmsort_plan_destroy(hmmsort_plan *p)
{
    if (!p) return HMMSORT_OK;
    if (p->gen) generic_destroy(p->gen);
    if (p->ring) ring_destroy(p->ring);
    if (p->wave) wave_destroy(p->wave);
    delete p;
    return HMMSORT_OK;
}

int hmmsort_plan_info(const hmmsort_plan *p, int64_t *engine, int64_t *block, int64_t *halo,
                      int64_t *nchains, int64_t *workspace_bytes)
{
    HS_CHECK(p, HMMSORT_EINVAL, "plan_info: null plan");
    int64_t b = 0, h = 0, n = 0, w = 0;
    if (p->wave) {
        b = p->wave->g.B; h = p->wave->g.Hw - 1; n = (int64_t)p->wave->g.nch * p->wave->g.C;
        w = p->wave->bytes;
    } else if (p->ring) {
        ring_geometry(p->ring, &b, &h, &n);
        w = ring_workspace_bytes(p->ring);
    } else if (p->gen) {
        generic_geometry(p->gen, &b, &h, &n);
        w = generic_workspace_bytes(p->gen);
    }
    if (engine) *engine = p->engine;
    if (block) *block = b;
    if (halo) *halo = h;
    if (nchains) *nchains = n;
    if (workspace_bytes) *workspace_bytes = w;
    return HMMSORT_OK;
}

int64_t hmmsort_plan_overlap_sweep(const hmmsort_plan *p)
{
    return (p && p->gen) ? generic_overlap_sweep(p->gen) : 0;
}

int hmmsort_plan_bind(hmmsort_plan *p, const double *d_y, void *stream)
{
    HS_CHECK(p && d_y, HMMSORT_EINVAL, "plan_bind: null argument");
    if (p->wave) return wave_bind(p->wave, d_y, (hipStream_t)stream);
    if (p->ring) return ring_bind(p->ring, d_y, (hipStream_t)stream);
    return HMMSORT_OK;
}

int hmmsort_plan_unbind(hmmsort_plan *p)
{
    HS_CHECK(p, HMMSORT_EINVAL, "plan_unbind: null plan");
    if (p->wave) p->wave->bound_y = nullptr;
    if (p->ring) p->ring->bound_y = nullptr;
    return HMMSORT_OK;
}

int hmmsort_plan_viterbi(hmmsort_plan *p, const double *d_y, int16_t *d_x, double *d_ll,
                         void *stream)
{
    HS_CHECK(p && d_y && d_x && d_ll, HMMSORT_EINVAL, "plan_viterbi: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (p->wave) return wave_viterbi(p->wave, d_y, d_x, d_ll, st);
    if (p->ring) return ring_viterbi(p->ring, d_y, d_x, d_ll, st);
    return generic_viterbi(p->gen, d_y, d_x, d_ll, st);
}

int hmmsort_plan_decode_estep(hmmsort_plan *p, const double *d_y, int16_t *d_x, double *d_ll,
                              double *d_stats, void *stream)
{
    HS_CHECK(p && d_y && d_x && d_ll && d_stats, HMMSORT_EINVAL, "plan_decode_estep: null argument");
    if (p->wave) return wave_decode_estep(p->wave, d_y, d_x, d_ll, d_stats, (hipStream_t)stream);
    HS_CHECK(p->ring, HMMSORT_EUNSUP, "plan_decode_estep: needs the wave or ring engine");
    return ring_decode_estep_launch(p->ring, d_y, d_x, d_ll, d_stats, (hipStream_t)stream);
}

int hmmsort_plan_set_shard(hmmsort_plan *p, int64_t own_lo, int64_t own_hi, int first, int last)
{
    HS_CHECK(p, HMMSORT_EINVAL, "plan_set_shard: null plan");
    HS_CHECK(p->ring || p->wave, HMMSORT_EUNSUP, "plan_set_shard: needs the wave or ring engine");
    HS_CHECK(own_lo >= 0 && own_lo <= own_hi && own_hi <= p->T, HMMSORT_EINVAL,
             "plan_set_shard: owned range [%lld, %lld) outside [0, %lld]", (long long)own_lo,
             (long long)own_hi, (long long)p->T);
    HS_CHECK((!first || own_lo == 0) && (!last || own_hi == p->T), HMMSORT_EINVAL,
             "plan_set_shard: a first/last shard must own its first/last sample");
    if (p->wave) {
        // The slice's own ends are arbitrary starts (emission-only first column, beta = 0 at the end): what
        // certifies that they have been forgotten where the owned range begins / ends is a certified chain
        // boundary INSIDE each halo -- kw_fb_check compares, at every chain boundary, a warm-up started from
        // "silent, rings empty" with the neighbouring chain's own sweep, and the two can only agree when
        // both have forgotten where they started.  So an interior shard edge must have a chain boundary
        // between the slice end and the owned range.
        const WaveGeom &g = p->wave->g;
        const int64_t B = g.B, last_boundary = (int64_t)(g.nch - 1) * B;
        HS_CHECK(first || (g.nch > 1 && own_lo >= B), HMMSORT_EINVAL,
                 "plan_set_shard: no chain boundary inside the leading halo (owned range starts at %lld, chains are "
                 "%lld samples): widen the halo or set a shorter chain length (option \"block\")",
                 (long long)own_lo, (long long)B);
        HS_CHECK(last || (g.nch > 1 && own_hi <= last_boundary), HMMSORT_EINVAL,
                 "plan_set_shard: no chain boundary inside the trailing halo (owned range ends at %lld, last chain "
                 "boundary at %lld): widen the halo or set a shorter chain length (option \"block\")",
                 (long long)own_hi, (long long)last_boundary);
        p->wave->g.own_lo = own_lo; p->wave->g.own_hi = own_hi;
        p->wave->g.first = first != 0; p->wave->g.last = last != 0;
        return HMMSORT_OK;
    }
    p->ring->g.own_lo = own_lo; p->ring->g.own_hi = own_hi;
    p->ring->g.first = first != 0; p->ring->g.last = last != 0;
    return HMMSORT_OK;
}

int64_t hmmsort_plan_stats_len(const hmmsort_plan *p)
{
    if (p && p->wave) return wave_stats_len(p->wave);
    if (p && p->gen && blocked_estep_supported(p->gen)) return blocked_stats_len(p->gen);
    if (!p || !p->ring) return 0;
    return ring_stats_len(p->ring);
}

// [mu K*N | sigma | lp_new | pp S] per channel: the wave and ring M-step kernels always write N entry
// log-probabilities (a template whose entry transitions were dropped from the list keeps its slot);
// the generic/blocked engines one per transition leaving state 1 except the first (baumwelch.jl:226,264)
int64_t hmmsort_plan_mstep_len(const hmmsort_plan *p)
{
    if (!p) return 0;
    const HostModel &m = p->model;
    const int64_t nlp = (p->wave || p->ring) ? m.N : (p->gen ? generic_n_lp(p->gen) : 0);
    return m.K * m.N + 1 + nlp + m.S;
}

int hmmsort_plan_estep(hmmsort_plan *p, const double *d_y, double *d_stats, void *stream)
{
    HS_CHECK(p && d_y && d_stats, HMMSORT_EINVAL, "plan_estep: null argument");
    if (p->wave) return wave_estep(p->wave, d_y, d_stats, (hipStream_t)stream);
    if (p->gen && blocked_estep_supported(p->gen)) return blocked_estep(p->gen, d_y, d_stats, (hipStream_t)stream);
    HS_CHECK(p->ring, HMMSORT_EUNSUP,
             "plan_estep: sufficient-statistics E-step needs the wave, ring or blocked engine (use hmmsort_em_step)");
    return ring_estep(p->ring, d_y, d_stats, (hipStream_t)stream);
}

int hmmsort_plan_mstep(hmmsort_plan *p, const double *d_stats, double *d_out, void *stream)
{
    HS_CHECK(p && d_stats && d_out, HMMSORT_EINVAL, "plan_mstep: null argument");
    if (p->wave) return wave_mstep(p->wave, d_stats, d_out, (hipStream_t)stream);
    if (p->gen && blocked_estep_supported(p->gen)) return blocked_mstep(p->gen, d_stats, d_out, (hipStream_t)stream);
    HS_CHECK(p->ring, HMMSORT_EUNSUP, "plan_mstep: needs the wave, ring or blocked engine");
    return ring_mstep(p->ring, d_stats, d_out, (hipStream_t)stream);
}

int hmmsort_plan_diagnostics(hmmsort_plan *p, void *stream, int64_t diag[8])
{
    HS_CHECK(p && diag, HMMSORT_EINVAL, "plan_diagnostics: null argument");
    for (int i = 0; i < 8; i++) diag[i] = 0;
    if (p->wave) return wave_diagnostics(p->wave, (hipStream_t)stream, diag);
    if (p->ring) return ring_diagnostics(p->ring, (hipStream_t)stream, diag);
    if (p->gen) return generic_diagnostics(p->gen, (hipStream_t)stream, diag);
    return HMMSORT_OK;
}

int hmmsort_plan_tie_stats(hmmsort_plan *p, void *stream, int64_t out[8])
{
    HS_CHECK(p && out, HMMSORT_EINVAL, "plan_tie_stats: null argument");
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (p->wave) return wave_tie_stats(p->wave, (hipStream_t)stream, out);
    return HMMSORT_OK;
}

// debugging aid (not part of the documented ABI): raw debug record of the wave engine
int hmmsort_plan_debug_record(hmmsort_plan *p, double *out64)
{
    HS_CHECK(p && out64 && p->wave, HMMSORT_EINVAL, "plan_debug_record: needs a wave plan");
    HS_HIP(hipDeviceSynchronize());
    HS_HIP(hipMemcpy(out64, p->wave->dbg, 64 * sizeof(double), hipMemcpyDeviceToHost));
    return HMMSORT_OK;
}

// debugging aid (not part of the documented ABI): copy an internal per-sample array of a wave plan
// which: 0 FA0 (log alpha silent), 1 FREF, 2 FV (N x T), 3 rho (N x T), 4 Rf (N x T), 5 vend, 6 vpre (chain records), 7 exact trellis values of the decoded path at block starts (wave_ties.hip)
int hmmsort_plan_debug_array(hmmsort_plan *p, int which, double *out, int64_t n)
{
    HS_CHECK(p && out && p->wave, HMMSORT_EINVAL, "plan_debug_array: needs a wave plan");
    HS_HIP(hipDeviceSynchronize());
    const WaveDev *w = p->wave;
    const int64_t CT = (int64_t)w->g.C * w->g.T, NCT = CT * w->g.N;
    const int64_t rec = (int64_t)w->g.C * w->g.nch * (1 + (int64_t)w->g.N * w->g.L);
    const double *srcs[8] = {w->FA0, w->FREF, w->FV, w->rho, w->Rf, w->vend, w->vpre, w->tie_v};
    const int64_t lens[8] = {CT, CT, NCT, NCT, NCT, rec, rec, (int64_t)w->g.C * (w->tie_nblk + 1)};
    HS_CHECK(which >= 0 && which < 8, HMMSORT_EINVAL, "plan_debug_array: unknown array %d", which);
    HS_CHECK(n >= 0 && n <= lens[which], HMMSORT_EINVAL, "plan_debug_array: %lld entries asked, array %d holds %lld",
             (long long)n, which, (long long)lens[which]);
    HS_HIP(hipMemcpy(out, srcs[which], n * sizeof(double), hipMemcpyDeviceToHost));
    return HMMSORT_OK;
}

int hmmsort_plan_profile(hmmsort_plan *p, int enable)
{
    HS_CHECK(p, HMMSORT_EINVAL, "plan_profile: null plan");
    if (p->wave) { p->wave->prof_on = enable != 0; return HMMSORT_OK; }
    if (p->ring) return ring_profile_enable(p->ring, enable);
    return HMMSORT_OK;
}

int hmmsort_plan_profile_read(hmmsort_plan *p, void *stream, char *names, int64_t names_cap,
                              double *ms, int64_t *calls, int64_t cap, int64_t *n_out)
{
    HS_CHECK(p && names && ms && calls && n_out, HMMSORT_EINVAL, "plan_profile_read: null argument");
    *n_out = 0;
    if (names_cap > 0) names[0] = 0;
    if (!p->ring && !p->wave) return HMMSORT_OK;
    std::vector<std::string> nm;
    std::vector<double> m;
    std::vector<int64_t> c;
    int rc = p->wave ? wave_profile_read(p->wave, (hipStream_t)stream, nm, m, c)
                     : ring_profile_read(p->ring, (hipStream_t)stream, nm, m, c);
    if (rc) return rc;
    std::string joined;
    int64_t n = std::min<int64_t>((int64_t)nm.size(), cap);
    for (int64_t i = 0; i < n; i++) {
        if (i) joined += "\n";
        joined += nm[i];
        ms[i] = m[i];
        calls[i] = c[i];
    }
    HS_CHECK((int64_t)joined.size() + 1 <= names_cap, HMMSORT_EINVAL, "plan_profile_read: names buffer too small");
    memcpy(names, joined.c_str(), joined.size() + 1);
    *n_out = n;
    return HMMSORT_OK;
}

// ---- host-buffer entry points --------------------------------------------------------------

// Ring-engine calls certify their own chain boundaries on device; when a check fails the
// host-buffer entry points retry with a doubled warm-up (up to the chain length of a single
// chain) and finally with the strict engine.
static int64_t next_halo(const hmmsort_plan *p)
{
    int64_t b = 0, h = 0, n = 0;
    if (p->wave) h = p->wave->g.Hw - 1;
    else if (p->ring) ring_geometry(p->ring, &b, &h, &n);
    else generic_geometry(p->gen, &b, &h, &n);
    return h * 2;
}

static int viterbi_host(const void *y, int sample_type, int64_t T, const int16_t *states, int64_t N,
                        int64_t K, int64_t S, const hmm_trans *tr, int64_t R, const double *mu,
                        double sigma, int16_t *x_out, double *ll_out)
{
    HS_CHECK(y && x_out && ll_out, HMMSORT_EINVAL, "viterbi: null argument");
    HS_CHECK(T >= 1, HMMSORT_EINVAL, "viterbi: empty signal (T = %lld)", (long long)T);
    int rc;
    if ((rc = need_device())) return rc;
    const Options opt = options_get();
    last_escalations() = 0;
    std::unique_ptr<HostSlot> slot = take_slot(T, states, N, K, S, opt);
    if (!slot) slot = new_slot(T, opt);
    HostSlot &h = *slot;
    if ((rc = h.dy.ensure(T * sizeof(double)))) return rc;
    if ((rc = h.dx.ensure(T * sizeof(int16_t)))) return rc;
    if ((rc = h.dll.ensure(sizeof(double)))) return rc;
    if (sample_type == HMMSORT_SAMPLES_F64) {
        HS_HIP(hipMemcpyAsync(h.dy.p, y, T * sizeof(double), hipMemcpyHostToDevice, h.st));
    } else {
        // raw samples: the decoded path's buffer has the size of an int16 signal and is free until the sweep
        HS_CHECK(sample_type == HMMSORT_SAMPLES_I16, HMMSORT_EINVAL, "viterbi: unsupported sample type");
        HS_HIP(hipMemcpyAsync(h.dx.p, y, T * sizeof(int16_t), hipMemcpyHostToDevice, h.st));
        if ((rc = dev_widen(h.dx.p, sample_type, T, 1, h.dy.as<double>(), h.st))) return rc;
    }
    // an idle plan of the same shape: new numbers in, workspace kept.  A list it cannot take (a ring
    // model that stopped being one) falls through to a fresh plan.
    if (h.plan && hmmsort_plan_set_model(h.plan, tr, R, mu, sigma)) h.drop_plan();
    bool keep = true;  // the plan is the one a first attempt with these options builds
    int64_t halo = -1, engine = opt.engine;
    for (int attempt = 0;; attempt++) {
        if (!h.plan) {
            rc = plan_create_engine(&h.plan, T, states, N, K, S, tr, R, mu, sigma, engine, halo);
            if (rc) return rc;
        }
        if (h.plan->ring && engine == HMMSORT_ENGINE_AUTO && ring_has_twins(h.plan->model)) {
            // duplicate templates: which twin the reference decodes hangs on the last bit of its
            // own sums; only the op-for-op sweep reproduces that
            engine = HMMSORT_ENGINE_STRICT;
            h.drop_plan();
            keep = false;
            continue;
        }
        rc = hmmsort_plan_viterbi(h.plan, h.dy.as<double>(), h.dx.as<int16_t>(), h.dll.as<double>(), h.st);
        if (rc) return rc;
        HS_HIP(hipStreamSynchronize(h.st));
        if (h.plan->engine == HMMSORT_ENGINE_STRICT) break;
        int64_t diag[8];
        if ((rc = hmmsort_plan_diagnostics(h.plan, h.st, diag))) return rc;
        const bool ties = (h.plan->engine == HMMSORT_ENGINE_BLOCKED || h.plan->engine == HMMSORT_ENGINE_WAVE) &&
                          diag[7] != 0;
        if ((diag[0] == 0 && !ties) || !opt.escalate) break;
        if (ties && diag[0] == 0 && opt.engine == HMMSORT_ENGINE_AUTO) {
            // Decisions the exact resolver could not settle (none on any signal seen): the op-for-op sweep decides.
            // It keeps back-pointers for the states with more than one incoming transition only (N + 1 of a
            // ring model: 3.4 GB at 4081 states x 10^8 samples instead of the reference's S x T table, 0.8 TB).
            // Should even that not fit, the time-parallel path stands: it differs from the reference's at most at
            // the open decisions, whose margins are inside the reference's own rounding noise.
            // last_escalations < 0 = minus the number of such decisions.
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
            int64_t nmulti = 0;
            for (int64_t j = 0; j < S; j++)
                nmulti += (h.plan->model.in_ptr[j + 1] - h.plan->model.in_ptr[j]) > 1;
            const double need = (double)std::max<int64_t>(nmulti, 1) * (double)T * 2.0 + 16.0 * (double)T;
            const double limit = opt.strict_limit_mb > 0 ? (double)opt.strict_limit_mb * 1048576.0 : 0.9 * (double)free_b;
            if (need > limit) {
                last_escalations() = -diag[7];
                set_error("viterbi: %lld near-tie decisions on the decoded path; the strict sweep needs %.1f GB of "
                          "back-pointers (limit %.1f GB): time-parallel path returned", (long long)diag[7], need / 1e9,
                          limit / 1e9);
                break;
            }
        }
        last_escalations() = attempt + 1;
        if (ties && diag[0] == 0 && h.plan->gen && generic_pair_active(h.plan->gen)) {
            // two-template overlap model: a decision on the path is inside the noise of the pair sweep's own
            // arithmetic -- decode again with the generic blocked sweep (the reference's operation order per block)
            generic_pair_disable(h.plan->gen);
            keep = false;
            continue;
        }
        halo = next_halo(h.plan);
        if (attempt >= 3 || halo > T || ties) {
            // near-ties depend on the frame, not on the warm-up: straight to the op-for-op sweep
            HS_CHECK(opt.engine == HMMSORT_ENGINE_AUTO, HMMSORT_ENOCONV,
                     "viterbi: %lld block boundaries fail the warm-up check, %lld blocks hold near-ties",
                     (long long)diag[0], (long long)diag[7]);
            engine = HMMSORT_ENGINE_STRICT;
        }
        h.drop_plan();
        keep = false;
    }
    HS_HIP(hipMemcpyAsync(x_out, h.dx.p, T * sizeof(int16_t), hipMemcpyDeviceToHost, h.st));
    HS_HIP(hipMemcpyAsync(ll_out, h.dll.p, sizeof(double), hipMemcpyDeviceToHost, h.st));
    HS_HIP(hipStreamSynchronize(h.st));
    if (keep) give_slot(std::move(slot), opt);
    return HMMSORT_OK;
}

int hmmsort_viterbi(const double *y, int64_t T, const int16_t *states, int64_t N, int64_t K,
                    int64_t S, const hmm_trans *tr, int64_t R, const double *mu, double sigma,
                    int16_t *x_out, double *ll_out)
{
    return viterbi_host(y, HMMSORT_SAMPLES_F64, T, states, N, K, S, tr, R, mu, sigma, x_out, ll_out);
}

int hmmsort_viterbi_i16(const int16_t *y, int64_t T, const int16_t *states, int64_t N, int64_t K,
                        int64_t S, const hmm_trans *tr, int64_t R, const double *mu, double sigma,
                        int16_t *x_out, double *ll_out)
{
    return viterbi_host(y, HMMSORT_SAMPLES_I16, T, states, N, K, S, tr, R, mu, sigma, x_out, ll_out);
}

int hmmsort_samples_to_f64(const void *d_in, int sample_type, int64_t T, int64_t stride, double *d_out,
                           void *stream)
{
    HS_CHECK(T >= 0 && stride >= 1 && (T == 0 || (d_in && d_out)), HMMSORT_EINVAL,
             "samples_to_f64: bad argument (T = %lld, stride = %lld)", (long long)T, (long long)stride);
    int rc = need_device();
    if (rc) return rc;
    return dev_widen(d_in, sample_type, T, stride, d_out, (hipStream_t)stream);
}

static int fwd_bwd_host(bool fwd, const double *y, int64_t T, const int16_t *states, int64_t N,
                        int64_t K, int64_t S, const hmm_trans *tr, int64_t R, const double *mu,
                        double sigma, double *out)
{
    HS_CHECK(y && out, HMMSORT_EINVAL, "forward/backward: null argument");
    HS_CHECK(T >= 1, HMMSORT_EINVAL, "forward/backward: empty signal");
    PlanGuard pg;
    // materialising S x T output is the generic engine's job whatever the model
    int rc = plan_create_engine(&pg.p, T, states, N, K, S, tr, R, mu, sigma, HMMSORT_ENGINE_STRICT);
    if (rc) return rc;
    DevBuf dy, da;
    if ((rc = dy.alloc(T * sizeof(double))) || (rc = da.alloc((size_t)S * T * sizeof(double))))
        return rc;
    HS_HIP(hipMemcpy(dy.p, y, T * sizeof(double), hipMemcpyHostToDevice));
    rc = fwd ? generic_forward(pg.p->gen, dy.as<double>(), da.as<double>(), nullptr)
             : generic_backward(pg.p->gen, dy.as<double>(), da.as<double>(), nullptr);
    if (rc) return rc;
    HS_HIP(hipDeviceSynchronize());
    HS_HIP(hipMemcpy(out, da.p, (size_t)S * T * sizeof(double), hipMemcpyDeviceToHost));
    return HMMSORT_OK;
}

int hmmsort_forward(const double *y, int64_t T, const int16_t *states, int64_t N, int64_t K,
                    int64_t S, const hmm_trans *tr, int64_t R, const double *mu, double sigma,
                    double *alpha_out)
{
    return fwd_bwd_host(true, y, T, states, N, K, S, tr, R, mu, sigma, alpha_out);
}

int hmmsort_backward(const double *y, int64_t T, const int16_t *states, int64_t N, int64_t K,
                     int64_t S, const hmm_trans *tr, int64_t R, const double *mu, double sigma,
                     double *beta_out)
{
    return fwd_bwd_host(false, y, T, states, N, K, S, tr, R, mu, sigma, beta_out);
}

// unpack [mu K*N | sigma | lp nlp | pp S] from the device into the caller's buffers
static int unpack_mstep(const double *d_out, int64_t K, int64_t N, int64_t S, int64_t nlp,
                        double *mu_inout, double *sigma_out, double *lp_out, int64_t lp_cap,
                        int64_t *n_lp_out, double *pp_out)
{
    std::vector<double> h(K * N + 1 + nlp + S);
    HS_HIP(hipMemcpy(h.data(), d_out, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    memcpy(mu_inout, h.data(), K * N * sizeof(double));
    *sigma_out = h[K * N];
    if (n_lp_out) *n_lp_out = nlp;
    HS_CHECK(lp_cap >= nlp, HMMSORT_EINVAL, "lp_out too small: need %lld entries, got %lld",
             (long long)nlp, (long long)lp_cap);
    memcpy(lp_out, h.data() + K * N + 1, nlp * sizeof(double));
    if (pp_out) memcpy(pp_out, h.data() + K * N + 1 + nlp, S * sizeof(double));
    return HMMSORT_OK;
}

int hmmsort_update(const double *alpha, const double *beta, const double *x, int64_t T,
                   const int16_t *states, int64_t N, int64_t K, int64_t S, const hmm_trans *tr,
                   int64_t R, double *mu_inout, double sigma, double *sigma_out, double *lp_out,
                   int64_t lp_cap, int64_t *n_lp_out, double *pp_out)
{
    HS_CHECK(alpha && beta && x && mu_inout && sigma_out && lp_out, HMMSORT_EINVAL,
             "update: null argument");
    HS_CHECK(T >= 2, HMMSORT_EINVAL, "update: need T >= 2");
    PlanGuard pg;
    int rc = plan_create_engine(&pg.p, T, states, N, K, S, tr, R, mu_inout, sigma,
                                HMMSORT_ENGINE_STRICT);
    if (rc) return rc;
    const int64_t nlp = generic_n_lp(pg.p->gen);
    DevBuf dy, da, db, dout;
    const size_t st = (size_t)S * T * sizeof(double);
    if ((rc = dy.alloc(T * sizeof(double))) || (rc = da.alloc(st)) || (rc = db.alloc(st)) ||
        (rc = dout.alloc((K * N + 1 + nlp + S) * sizeof(double))))
        return rc;
    HS_HIP(hipMemcpy(dy.p, x, T * sizeof(double), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(da.p, alpha, st, hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(db.p, beta, st, hipMemcpyHostToDevice));
    rc = generic_update(pg.p->gen, da.as<double>(), db.as<double>(), dy.as<double>(),
                        dout.as<double>(), nullptr);
    if (rc) return rc;
    HS_HIP(hipDeviceSynchronize());
    return unpack_mstep(dout.as<double>(), K, N, S, nlp, mu_inout, sigma_out, lp_out, lp_cap,
                        n_lp_out, pp_out);
}

int hmmsort_em_step(const double *y, int64_t T, const int16_t *states, int64_t N, int64_t K,
                    int64_t S, const hmm_trans *tr, int64_t R, double *mu_inout, double sigma,
                    double *sigma_out, double *lp_out, int64_t lp_cap, int64_t *n_lp_out,
                    double *pp_out)
{
    HS_CHECK(y && mu_inout && sigma_out && lp_out, HMMSORT_EINVAL, "em_step: null argument");
    HS_CHECK(T >= 2, HMMSORT_EINVAL, "em_step: need T >= 2");
    int rc;
    if ((rc = need_device())) return rc;
    const Options opt = options_get();
    last_escalations() = 0;
    std::unique_ptr<HostSlot> slot = take_slot(T, states, N, K, S, opt);
    if (!slot) slot = new_slot(T, opt);
    HostSlot &h = *slot;
    if ((rc = h.dy.ensure(T * sizeof(double)))) return rc;
    HS_HIP(hipMemcpyAsync(h.dy.p, y, T * sizeof(double), hipMemcpyHostToDevice, h.st));
    if (h.plan && hmmsort_plan_set_model(h.plan, tr, R, mu_inout, sigma)) h.drop_plan();
    bool keep = true;
    int64_t halo = -1, engine = opt.engine;
    for (int attempt = 0;; attempt++) {
        if (!h.plan) {
            rc = plan_create_engine(&h.plan, T, states, N, K, S, tr, R, mu_inout, sigma, engine, halo);
            if (rc) return rc;
        }
        const bool blocked_es = h.plan->gen && blocked_estep_supported(h.plan->gen);
        if (!h.plan->ring && !h.plan->wave && !blocked_es) {
            keep = false;
            if (h.plan->engine == HMMSORT_ENGINE_STRICT) break;
            engine = HMMSORT_ENGINE_STRICT;  // materialised alpha/beta are the strict engine's job
            h.drop_plan();
            continue;
        }
        const int64_t nlp = blocked_es ? generic_n_lp(h.plan->gen) : N;
        // sized for THIS plan: a cached slot's buffers may come from a plan of another engine or list
        // (the slot key holds neither R nor the engine)
        if ((rc = h.dstats.ensure(hmmsort_plan_stats_len(h.plan) * sizeof(double)))) return rc;
        if ((rc = h.dout.ensure(hmmsort_plan_mstep_len(h.plan) * sizeof(double)))) return rc;
        if ((rc = hmmsort_plan_estep(h.plan, h.dy.as<double>(), h.dstats.as<double>(), h.st))) return rc;
        if ((rc = hmmsort_plan_mstep(h.plan, h.dstats.as<double>(), h.dout.as<double>(), h.st))) return rc;
        HS_HIP(hipStreamSynchronize(h.st));
        int64_t diag[8];
        if ((rc = hmmsort_plan_diagnostics(h.plan, h.st, diag))) return rc;
        if ((diag[3] == 0 && diag[5] == 0) || !opt.escalate) {
            rc = unpack_mstep(h.dout.as<double>(), K, N, S, nlp, mu_inout, sigma_out, lp_out, lp_cap,
                              n_lp_out, pp_out);
            if (!rc && keep) give_slot(std::move(slot), opt);
            return rc;
        }
        last_escalations() = attempt + 1;
        halo = next_halo(h.plan);
        if (attempt >= 3 || halo > T) {
            HS_CHECK(opt.engine != HMMSORT_ENGINE_RING && opt.engine != HMMSORT_ENGINE_WAVE &&
                         opt.engine != HMMSORT_ENGINE_BLOCKED,
                     HMMSORT_ENOCONV, "em_step: %lld chain boundaries still fail the warm-up check",
                     (long long)(diag[3] + diag[5]));
            engine = HMMSORT_ENGINE_STRICT;
        }
        // a wider warm-up changes the geometry: statistics buffer and plan are rebuilt
        h.drop_plan();
        h.dstats.release();
        h.dout.release();
        keep = false;
    }
    // generic engine: forward -> backward -> update with materialised alpha/beta, all on device
    const int64_t nlp = generic_n_lp(h.plan->gen);
    DevBuf da, db, dout;
    const size_t st = (size_t)S * T * sizeof(double);
    if ((rc = da.alloc(st)) || (rc = db.alloc(st)) ||
        (rc = dout.alloc((K * N + 1 + nlp + S) * sizeof(double))))
        return rc;
    if ((rc = generic_forward(h.plan->gen, h.dy.as<double>(), da.as<double>(), h.st))) return rc;
    if ((rc = generic_backward(h.plan->gen, h.dy.as<double>(), db.as<double>(), h.st))) return rc;
    if ((rc = generic_update(h.plan->gen, da.as<double>(), db.as<double>(), h.dy.as<double>(),
                             dout.as<double>(), h.st)))
        return rc;
    HS_HIP(hipStreamSynchronize(h.st));
    return unpack_mstep(dout.as<double>(), K, N, S, nlp, mu_inout, sigma_out, lp_out, lp_cap,
                        n_lp_out, pp_out);
}

int hmmsort_reconstruct(const int16_t *x, int64_t T, const int16_t *states, int64_t N, int64_t S,
                        const double *mu, int64_t K, double *y_out)
{
    HS_CHECK(states && mu && (T == 0 || (x && y_out)), HMMSORT_EINVAL, "reconstruct: null argument");
    HS_CHECK(T >= 0 && N >= 1 && S >= 1 && K >= 1, HMMSORT_EINVAL, "reconstruct: bad sizes");
    if (T == 0) return HMMSORT_OK;  // reference returns an empty vector
    int rc = need_device();
    if (rc) return rc;
    for (int64_t i = 0; i < T; i++)
        HS_CHECK(x[i] >= 1 && x[i] <= S, HMMSORT_EINVAL,
                 "reconstruct: x[%lld] = %d outside 1..S (reference would throw BoundsError)",
                 (long long)i, (int)x[i]);
    DevBuf dx, dst, dmu, dout;
    if ((rc = dx.alloc(T * sizeof(int16_t))) || (rc = dst.alloc(N * S * sizeof(int16_t))) ||
        (rc = dmu.alloc(K * N * sizeof(double))) || (rc = dout.alloc(T * sizeof(double))))
        return rc;
    HS_HIP(hipMemcpy(dx.p, x, T * sizeof(int16_t), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(dst.p, states, N * S * sizeof(int16_t), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(dmu.p, mu, K * N * sizeof(double), hipMemcpyHostToDevice));
    rc = dev_reconstruct(dx.as<int16_t>(), T, dst.as<int16_t>(), N, S, dmu.as<double>(), K,
                         dout.as<double>(), nullptr);
    if (rc) return rc;
    HS_HIP(hipDeviceSynchronize());
    HS_HIP(hipMemcpy(y_out, dout.p, T * sizeof(double), hipMemcpyDeviceToHost));
    return HMMSORT_OK;
}

int hmmsort_unroll_mlseq(const int16_t *mlseq, int64_t T, const int16_t *states, int64_t N,
                         int64_t S, int16_t *out)
{
    HS_CHECK(states && (T == 0 || (mlseq && out)), HMMSORT_EINVAL, "unroll_mlseq: null argument");
    if (T == 0) return HMMSORT_OK;
    int rc = need_device();
    if (rc) return rc;
    for (int64_t i = 0; i < T; i++)
        HS_CHECK(mlseq[i] >= 1 && mlseq[i] <= S, HMMSORT_EINVAL,
                 "unroll_mlseq: mlseq[%lld] = %d outside 1..S", (long long)i, (int)mlseq[i]);
    DevBuf dx, dst, dout;
    if ((rc = dx.alloc(T * sizeof(int16_t))) || (rc = dst.alloc(N * S * sizeof(int16_t))) ||
        (rc = dout.alloc((size_t)N * T * sizeof(int16_t))))
        return rc;
    HS_HIP(hipMemcpy(dx.p, mlseq, T * sizeof(int16_t), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(dst.p, states, N * S * sizeof(int16_t), hipMemcpyHostToDevice));
    rc = dev_unroll(dx.as<int16_t>(), T, dst.as<int16_t>(), N, S, dout.as<int16_t>(), nullptr);
    if (rc) return rc;
    HS_HIP(hipDeviceSynchronize());
    HS_HIP(hipMemcpy(out, dout.p, (size_t)N * T * sizeof(int16_t), hipMemcpyDeviceToHost));
    return HMMSORT_OK;
}

// extract_spiketimes on a path that already lives in device memory
static int extract_from_device(const int16_t *d_x, int64_t T, const int16_t *states, int64_t N,
                               int64_t S, const double *mu, int64_t K, int64_t *times_out,
                               int64_t cap, int64_t *counts_out, hipStream_t st)
{
    // indmin(mu[:,i]): first minimum (extraction.jl:18); match table per state
    std::vector<uint32_t> match(S, 0u);
    for (int64_t i = 0; i < N; i++) {
        int64_t q = 0;
        for (int64_t k = 1; k < K; k++)
            if (mu[k + K * i] < mu[q + K * i]) q = k;
        for (int64_t j = 0; j < S; j++)
            if (states[i + N * j] == q + 1) match[j] |= (1u << i);
    }
    const int64_t nb = (T + kSpikeChunkHost - 1) / kSpikeChunkHost;
    DevBuf dm, dcnt, doff, dt;
    int rc;
    if ((rc = dm.alloc(S * sizeof(uint32_t))) || (rc = dcnt.alloc(nb * N * sizeof(int64_t))) ||
        (rc = doff.alloc(nb * N * sizeof(int64_t))) ||
        (rc = dt.alloc(std::max<int64_t>(1, N * cap) * sizeof(int64_t))))
        return rc;
    HS_HIP(hipMemcpyAsync(dm.p, match.data(), S * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if ((rc = dev_spike_compact(d_x, T, dm.as<uint32_t>(), (int)N, (int)S, 0, dcnt.as<int64_t>(), nullptr,
                                nullptr, cap, st)))
        return rc;
    std::vector<int64_t> cnt(nb * N), off(nb * N);
    HS_HIP(hipMemcpyAsync(cnt.data(), dcnt.p, cnt.size() * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HS_HIP(hipStreamSynchronize(st));
    for (int64_t i = 0; i < N; i++) {
        int64_t acc = 0;
        for (int64_t b = 0; b < nb; b++) { off[b * N + i] = acc; acc += cnt[b * N + i]; }
        counts_out[i] = acc;
    }
    if (cap == 0) return HMMSORT_OK;
    HS_HIP(hipMemcpyAsync(doff.p, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    if ((rc = dev_spike_compact(d_x, T, dm.as<uint32_t>(), (int)N, (int)S, 1, dcnt.as<int64_t>(),
                                doff.as<int64_t>(), dt.as<int64_t>(), cap, st)))
        return rc;
    for (int64_t i = 0; i < N; i++) {
        const int64_t n = std::min(counts_out[i], cap);
        HS_HIP(hipMemcpyAsync(times_out + i * cap, dt.as<int64_t>() + i * cap, n * sizeof(int64_t),
                              hipMemcpyDeviceToHost, st));
    }
    HS_HIP(hipStreamSynchronize(st));
    return HMMSORT_OK;
}

int hmmsort_extract_spiketimes(const int16_t *mlseq, int64_t T, const int16_t *states, int64_t N,
                               int64_t S, const double *mu, int64_t K, int64_t *times_out,
                               int64_t cap, int64_t *counts_out)
{
    HS_CHECK(states && mu && counts_out && (T == 0 || mlseq) && (cap == 0 || times_out), HMMSORT_EINVAL,
             "extract_spiketimes: null argument");
    HS_CHECK(N >= 1 && N <= 32 && S >= 1 && K >= 1 && T >= 0 && cap >= 0, HMMSORT_EINVAL,
             "extract_spiketimes: bad sizes");
    for (int64_t i = 0; i < N; i++) counts_out[i] = 0;
    if (T == 0) return HMMSORT_OK;
    int rc = need_device();
    if (rc) return rc;
    DevBuf dx;
    if ((rc = dx.alloc(T * sizeof(int16_t)))) return rc;
    HS_HIP(hipMemcpy(dx.p, mlseq, T * sizeof(int16_t), hipMemcpyHostToDevice));
    return extract_from_device(dx.as<int16_t>(), T, states, N, S, mu, K, times_out, cap, counts_out, nullptr);
}

// reconstruct_signal / unroll_mlseq on a path in device memory, results left in device memory
static int plan_path_op(hmmsort_plan *p, const int16_t *d_x, double *d_y_out, int16_t *d_unrolled,
                        hipStream_t st)
{
    const HostModel &m = p->model;
    DevBuf dst, dmu;
    int rc;
    if ((rc = dst.alloc(m.N * m.S * sizeof(int16_t))) || (rc = dmu.alloc(m.K * m.N * sizeof(double))))
        return rc;
    HS_HIP(hipMemcpyAsync(dst.p, m.states.data(), m.N * m.S * sizeof(int16_t), hipMemcpyHostToDevice, st));
    HS_HIP(hipMemcpyAsync(dmu.p, m.mu.data(), m.K * m.N * sizeof(double), hipMemcpyHostToDevice, st));
    if (d_y_out)
        rc = dev_reconstruct(d_x, p->T, dst.as<int16_t>(), m.N, m.S, dmu.as<double>(), m.K, d_y_out, st);
    else
        rc = dev_unroll(d_x, p->T, dst.as<int16_t>(), m.N, m.S, d_unrolled, st);
    if (rc) return rc;
    HS_HIP(hipStreamSynchronize(st));  // the temporaries die with this frame
    return HMMSORT_OK;
}

int hmmsort_plan_reconstruct(hmmsort_plan *p, const int16_t *d_x, double *d_y_out, void *stream)
{
    HS_CHECK(p && d_x && d_y_out, HMMSORT_EINVAL, "plan_reconstruct: null argument");
    return plan_path_op(p, d_x, d_y_out, nullptr, (hipStream_t)stream);
}

int hmmsort_plan_unroll_mlseq(hmmsort_plan *p, const int16_t *d_x, int16_t *d_out, void *stream)
{
    HS_CHECK(p && d_x && d_out, HMMSORT_EINVAL, "plan_unroll_mlseq: null argument");
    return plan_path_op(p, d_x, nullptr, d_out, (hipStream_t)stream);
}

int hmmsort_plan_extract_spiketimes(hmmsort_plan *p, const int16_t *d_x, int64_t *times_out, int64_t cap,
                                    int64_t *counts_out, void *stream)
{
    HS_CHECK(p && d_x && counts_out && (cap == 0 || times_out) && cap >= 0, HMMSORT_EINVAL,
             "plan_extract_spiketimes: bad argument");
    const HostModel &m = p->model;
    HS_CHECK(m.N <= 32, HMMSORT_EINVAL, "plan_extract_spiketimes: more than 32 neurons");
    for (int64_t i = 0; i < m.N; i++) counts_out[i] = 0;
    return extract_from_device(d_x, p->T, m.states.data(), m.N, m.S, m.mu.data(), m.K, times_out, cap,
                               counts_out, (hipStream_t)stream);
}

}  // extern "C"
