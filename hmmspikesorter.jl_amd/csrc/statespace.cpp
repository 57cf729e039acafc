// Host-side state space: enumeration (reference types.jl:65-92), closed-form transition list
// (types.jl:94-127 without the O(S^2 N) scan), model analysis for the device engines.
// Pure host code; no GPU needed.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <mutex>

#include "hmmsort_internal.h"

namespace hmmsort {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

static std::mutex g_opt_mu;
static Options g_opt;
Options options_get()
{
    std::lock_guard<std::mutex> lk(g_opt_mu);
    return g_opt;
}
void options_modify(const std::function<void(Options &)> &f)
{
    std::lock_guard<std::mutex> lk(g_opt_mu);
    f(g_opt);
}
int64_t &last_escalations()
{
    static thread_local int64_t n = 0;
    return n;
}

// number of states: types.jl:67-71
static int64_t nstates_of(int64_t N, int64_t K, bool ov)
{
    int64_t S = 1 + N * (K - 1);
    if (ov) S += (N * (N - 1) * (K - 1) * (K - 1)) / 2;
    return S;
}

// 0-based column of a phase vector with at most two active neurons (enumeration order of
// generate_states: silent; singles neuron-major; pairs (i<j) with k1 slow, k2 fast)
static int64_t state_index(int64_t N, int64_t K, int na, int a1, int k1, int a2, int k2)
{
    const int64_t L = K - 1;
    if (na == 0) return 0;
    if (na == 1) return 1 + (int64_t)a1 * L + (k1 - 1);
    // pair index of (a1 < a2) in the order i = 0..N-2, j = i+1..N-1
    int64_t pidx = (int64_t)a1 * N - ((int64_t)a1 * (a1 + 1)) / 2 + (a2 - a1 - 1);
    return 1 + N * L + pidx * L * L + (int64_t)(k1 - 1) * L + (k2 - 1);
}

}  // namespace hmmsort

using namespace hmmsort;

extern "C" int64_t hmmsort_generate_states(int64_t N, int64_t K, int allow_overlaps,
                                           int16_t *out)
{
    if (N < 1 || K < 2) {
        set_error("generate_states: need N >= 1 and K >= 2 (got N=%lld K=%lld)", (long long)N,
                  (long long)K);
        return HMMSORT_EINVAL;
    }
    const int64_t S = nstates_of(N, K, allow_overlaps != 0);
    if (S > 32767) {  // Int16 state ids (viterbi.jl:51,53)
        set_error("generate_states: %lld states exceed the Int16 id range", (long long)S);
        return HMMSORT_EINVAL;
    }
    if (!out) return S;
    const int64_t L = K - 1;
    for (int64_t i = 0; i < N * S; i++) out[i] = 1;  // phase 0 stored +1
    int64_t k = 1;
    for (int64_t i = 0; i < N; i++)
        for (int64_t k1 = 1; k1 <= L; k1++, k++) out[i + N * k] = (int16_t)(k1 + 1);
    if (allow_overlaps)
        for (int64_t i = 0; i < N - 1; i++)
            for (int64_t j = i + 1; j < N; j++)
                for (int64_t k1 = 1; k1 <= L; k1++)
                    for (int64_t k2 = 1; k2 <= L; k2++, k++) {
                        out[i + N * k] = (int16_t)(k1 + 1);
                        out[j + N * k] = (int16_t)(k2 + 1);
                    }
    return S;
}

// Closed-form transition list.  For every source state the reachable destinations are
// enumerated directly from the per-neuron rule of isvalid_transition (types.jl:94-113):
//   phase 0 -> 0 (adds lpz) or 0 -> 1 (adds lp[i]);  0 < s < K-1 -> s+1 (adds 0.0);
//   K-1 -> 0 (adds 0.0);  anything else is impossible.
// The log-probability is accumulated neuron by neuron in index order from 0.0 exactly as the
// reference loop does, so the doubles are identical given identical lp and lpz.
extern "C" int64_t hmmsort_build_transitions(int64_t N, int64_t K, const double *lp, int64_t nlp,
                                             int allow_overlaps, hmm_trans *tr_out, int64_t cap)
{
    if (N < 1 || K < 2 || !lp || nlp < N) {
        set_error("build_transitions: bad arguments (N=%lld K=%lld nlp=%lld)", (long long)N,
                  (long long)K, (long long)nlp);
        return HMMSORT_EINVAL;
    }
    const bool ov = allow_overlaps != 0;
    const int maxact = ov ? 2 : 1;
    const int64_t S = nstates_of(N, K, ov);
    const int64_t L = K - 1;
    // lpz = log1p(-exp(sum(lp)))  types.jl:96; Julia's sum is a left fold below 16 elements
    double slp = lp[0];
    for (int64_t i = 1; i < nlp; i++) slp = slp + lp[i];
    const double lpz = std::log1p(-std::exp(slp));

    std::vector<int> ph(N), nx(N);
    struct Cand { int64_t dst; double v; };
    std::vector<Cand> cands;
    int64_t r = 0;
    auto emit_row = [&](int64_t src) {
        // forced moves of the active neurons
        int nact_next = 0;
        std::vector<int> silent;
        for (int i = 0; i < N; i++) {
            if (ph[i] == 0) { silent.push_back(i); nx[i] = 0; }
            else if (ph[i] < L) { nx[i] = ph[i] + 1; nact_next++; }
            else { nx[i] = 0; }  // phase K-1 -> 0
        }
        cands.clear();
        const int ns = (int)silent.size();
        const int room = maxact - nact_next;  // how many silent neurons may start
        auto push = [&](int s1, int s2) {
            // s1, s2: indices into `silent` of the starters (-1 = none)
            for (int i = 0; i < N; i++) if (ph[i] == 0) nx[i] = 0;
            if (s1 >= 0) nx[silent[s1]] = 1;
            if (s2 >= 0) nx[silent[s2]] = 1;
            // destination index
            int na = 0, a1 = 0, k1 = 0, a2 = 0, k2 = 0;
            for (int i = 0; i < N; i++)
                if (nx[i] > 0) {
                    if (na == 0) { a1 = i; k1 = nx[i]; }
                    else { a2 = i; k2 = nx[i]; }
                    na++;
                }
            if (na > maxact) return;
            double lpt = 0.0;
            for (int i = 0; i < N; i++) {
                if (ph[i] == 0 && nx[i] == 0) lpt += lpz;
                else if (ph[i] == 0 && nx[i] == 1) lpt += lp[i];
                else lpt += 0.0;
            }
            if (!std::isfinite(lpt)) return;  // reference keeps only finite entries :121
            cands.push_back({state_index(N, K, na, a1, k1, a2, k2), lpt});
        };
        if (room >= 0) push(-1, -1);
        if (room >= 1)
            for (int s = 0; s < ns; s++) push(s, -1);
        if (room >= 2)
            for (int s = 0; s < ns; s++)
                for (int t = s + 1; t < ns; t++) push(s, t);
        std::sort(cands.begin(), cands.end(),
                  [](const Cand &a, const Cand &b) { return a.dst < b.dst; });
        for (auto &c : cands) {
            if (tr_out && r < cap) tr_out[r] = {src + 1, c.dst + 1, c.v};
            r++;
        }
    };
    // enumerate sources in state order
    std::fill(ph.begin(), ph.end(), 0);
    emit_row(0);
    int64_t src = 1;
    for (int i = 0; i < N; i++)
        for (int k1 = 1; k1 <= L; k1++, src++) {
            std::fill(ph.begin(), ph.end(), 0);
            ph[i] = k1;
            emit_row(src);
        }
    if (ov)
        for (int i = 0; i < N - 1; i++)
            for (int j = i + 1; j < N; j++)
                for (int k1 = 1; k1 <= L; k1++)
                    for (int k2 = 1; k2 <= L; k2++, src++) {
                        std::fill(ph.begin(), ph.end(), 0);
                        ph[i] = k1;
                        ph[j] = k2;
                        emit_row(src);
                    }
    if (src != S) {
        set_error("build_transitions: internal enumeration mismatch");
        return HMMSORT_EINVAL;
    }
    return r;
}

namespace hmmsort {

int build_host_model(HostModel &m, const int16_t *states, int64_t N, int64_t K, int64_t S,
                     const hmm_trans *tr, int64_t R, const double *mu, double sigma)
{
    HS_CHECK(states && tr && mu, HMMSORT_EINVAL, "model: null pointer argument");
    HS_CHECK(N >= 1 && K >= 1 && S >= 1 && R >= 1, HMMSORT_EINVAL,
             "model: need N,K,S,R >= 1 (got %lld %lld %lld %lld)", (long long)N, (long long)K,
             (long long)S, (long long)R);
    HS_CHECK(S <= 32767, HMMSORT_EINVAL, "model: %lld states exceed the Int16 id range",
             (long long)S);
    HS_CHECK(sigma > 0 && std::isfinite(sigma), HMMSORT_EINVAL, "model: sigma must be > 0");
    m.N = N; m.K = K; m.S = S; m.R = R; m.sigma = sigma;
    m.states.assign(states, states + N * S);
    m.tr.assign(tr, tr + R);
    m.mu.assign(mu, mu + K * N);
    for (int64_t i = 0; i < N * S; i++)
        HS_CHECK(states[i] >= 1 && states[i] <= K, HMMSORT_EINVAL,
                 "model: states[%lld] = %d outside 1..K", (long long)i, (int)states[i]);
    for (int64_t r = 0; r < R; r++)
        HS_CHECK(tr[r].src >= 1 && tr[r].src <= S && tr[r].dst >= 1 && tr[r].dst <= S,
                 HMMSORT_EINVAL, "model: transition %lld endpoints outside 1..S", (long long)r);
    // per-state mean, accumulated from 0.0 in neuron order (baumwelch.jl:29-35, viterbi.jl:68-71)
    m.mean.resize(S);
    for (int64_t j = 0; j < S; j++) {
        double a = 0.0;
        for (int64_t l = 0; l < N; l++) a += mu[(states[l + N * j] - 1) + K * l];
        m.mean[j] = a;
    }
    // CSR by destination keeping list order (stable counting sort)
    m.in_ptr.assign(S + 1, 0);
    m.out_ptr.assign(S + 1, 0);
    for (int64_t r = 0; r < R; r++) { m.in_ptr[tr[r].dst]++; m.out_ptr[tr[r].src]++; }
    for (int64_t j = 0; j < S; j++) { m.in_ptr[j + 1] += m.in_ptr[j]; m.out_ptr[j + 1] += m.out_ptr[j]; }
    m.in_src.resize(R); m.in_lp.resize(R); m.out_dst.resize(R); m.out_lp.resize(R);
    std::vector<int32_t> ci(m.in_ptr.begin(), m.in_ptr.end() - 1), co(m.out_ptr.begin(), m.out_ptr.end() - 1);
    for (int64_t r = 0; r < R; r++) {
        int32_t pi = ci[tr[r].dst - 1]++;
        m.in_src[pi] = (int32_t)(tr[r].src - 1);
        m.in_lp[pi] = tr[r].lp;
        int32_t po = co[tr[r].src - 1]++;
        m.out_dst[po] = (int32_t)(tr[r].dst - 1);
        m.out_lp[po] = tr[r].lp;
    }
    analyze_ring(m, m.ring);
    return HMMSORT_OK;
}

// Is the transition list the no-overlap ring pattern (in the reference's order)?  The reference keeps
// finite entries only (types.jl:121), so a template whose entry log-probability has become -Inf has lost
// its N entry transitions (silent -> (a,1) and every (b,L) -> (a,1)): those are accepted as missing and
// recorded as -Inf.  Everything else (silent -> silent, ring interiors, (a,L) -> silent) must be there.
int analyze_ring(const HostModel &m, RingModel &ring)
{
    ring = RingModel();
    const int64_t N = m.N, L = m.K - 1, S = m.S, R = m.R;
    if (L < 1 || S != 1 + N * L) return 0;
    if (R > N * (L - 1) + N * N + N + 1) return 0;
    // state table must be the single-active enumeration
    for (int64_t a = 0; a < N; a++)
        for (int64_t k = 1; k <= L; k++) {
            int64_t j = 1 + a * L + (k - 1);
            for (int64_t l = 0; l < N; l++) {
                int want = (l == a) ? (int)(k + 1) : 1;
                if (m.states[l + N * j] != want) return 0;
            }
        }
    for (int64_t l = 0; l < N; l++)
        if (m.states[l] != 1) return 0;
    ring.N = (int)N; ring.L = (int)L;
    ring.c0.assign(N, -INFINITY); ring.cint.assign(N * L, 0); ring.cend.assign(N, 0);
    ring.cx.assign(N * N, -INFINITY);
    int64_t r = 0;
    // a mandatory transition must be the next list entry; an entry transition may be absent
    auto expect = [&](int64_t s, int64_t d, double *out) {
        if (r >= R || m.tr[r].src != s + 1 || m.tr[r].dst != d + 1) return false;
        *out = m.tr[r].lp;
        r++;
        return true;
    };
    auto optional = [&](int64_t s, int64_t d, double *out) {
        if (r < R && m.tr[r].src == s + 1 && m.tr[r].dst == d + 1) { *out = m.tr[r].lp; r++; }
    };
    if (!expect(0, 0, &ring.c00)) return 0;
    for (int64_t a = 0; a < N; a++) optional(0, 1 + a * L, &ring.c0[a]);
    for (int64_t a = 0; a < N; a++) {
        for (int64_t k = 1; k < L; k++)
            if (!expect(1 + a * L + (k - 1), 1 + a * L + k, &ring.cint[a * L + k])) return 0;
        int64_t e = 1 + a * L + (L - 1);
        if (!expect(e, 0, &ring.cend[a])) return 0;
        for (int64_t b = 0; b < N; b++) {
            if (b == a) continue;
            optional(e, 1 + b * L, &ring.cx[a * N + b]);
        }
    }
    if (r != R) return 0;
    ring.valid = true;
    return 1;
}

}  // namespace hmmsort
