// Wave engine, part 1: geometry, workspace, model tables, the parallel pre-pass (ring scores).
// Design notes: wave_common.h / DESIGN.md section 3.3.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "wave_common.h"

namespace hmmsort {

static inline int64_t wround_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

constexpr int64_t kWaveLdsMax = 160 * 1024;

// exit -> entry log-probabilities (b,L) -> (a,1) independent of the source ring b, bit for bit: true for every
// list the reference builds (types.jl:94-113)
static bool ring_uniform_cx(const RingModel &R)
{
    const int N = R.N;
    for (int a = 0; a < N; a++) {
        double ref = 0.0;
        bool have = false;
        for (int b = 0; b < N; b++) {
            if (b == a) continue;
            const double v = R.cx[b * N + a];
            if (!have) { ref = v; have = true; }
            else if (!(v == ref)) return false;
        }
    }
    return true;
}

// per-source exit -> entry values take the O(N^2) junction code, which is built for up to kWaveMaxPerSource rings
// (with 12-13 rings of fewer than 32 states those kernels -- 200+ spilled scalar registers -- gave wrong forward
// values on gfx950, found by tests/test_gpu_wave_edges.py; such lists go to the blocked / strict engines)
constexpr int kWaveMaxPerSource = 8;

bool wave_supported(const HostModel &m, int64_t T, std::string *why)
{
    auto no = [&](const char *w) { if (why) *why = w; return false; };
    if (!m.ring.valid) return no("transition list is not the no-overlap ring pattern");
    if (m.ring.N > kRingMaxN) return no("more than 16 rings");
    if (m.ring.N > kWaveMaxPerSource && !ring_uniform_cx(m.ring))
        return no("exit->entry log-probabilities depend on the source ring and there are more than 8 rings");
    if (m.ring.L < 8) return no("rings shorter than 8 states");
    if (T < 4 * (int64_t)m.ring.L || T < 512) return no("signal shorter than 4 ring lengths / 512 samples");
    const int W = std::min(m.ring.L, 64);
    const int64_t RB = wround_up(m.ring.L + W, 32);
    // forward sweep: two delay lines per ring (one, in log form, above 8 rings); backward: N + 1 lines
    const int64_t lines = m.ring.N > 8 ? m.ring.N + 1 : 2 * (int64_t)m.ring.N;
    if (lines * (RB + 1) * 8 + 4096 > kWaveLdsMax) return no("delay lines exceed the LDS of one CU");
    return true;
}

static int make_geometry(WaveGeom &g, int64_t T, int C, int N, int L, int64_t block_req, int64_t halo_req)
{
    g.T = T; g.C = C; g.N = N; g.L = L;
    g.own_lo = 0; g.own_hi = T; g.first = 1; g.last = 1;
    { const Options o = options_get(); g.thr_scale = (double)o.tie_scale; g.tie_debug = (int)o.tie_debug; }
    g.W = std::min(L, 64);
    g.RB = (int)wround_up(L + g.W, 32);
    // warm-up: four ring lengths, at least 256 samples; with chains of thousands of samples this is
    // a few per cent of the sweep.  Every chain boundary is certified on device after the sweep
    // (Viterbi: exact hand-off and re-sweep of the chain when the certificate fails).
    int64_t H = halo_req > 0 ? halo_req : std::max<int64_t>(256, 4 * (int64_t)L);
    H = std::max<int64_t>(H, L + 8);
    const int64_t m = (H + g.W - 1) / g.W;
    g.Hw = (int)(1 + m * g.W);
    g.He = (int)(m * g.W);
    // chain length: 2 wavefronts per SIMD over all channels (1024 SIMDs; the backward sweep holds ~190
    // VGPRs, and the Viterbi sweep runs beside the forward/backward sweeps), at least 2 warm-ups
    int64_t B = block_req > 0 ? block_req : (T * C + 2047) / 2048;
    B = std::max<int64_t>(B, std::max<int64_t>(2 * g.Hw, 512));
    if (block_req > 0) B = std::max<int64_t>(block_req, std::max<int64_t>(g.Hw, L + 16));
    B = wround_up(B, 64);
    if (B > T) B = wround_up(T, 64);
    for (;;) {  // every chain owns >= L samples (the per-chain normaliser needs a full ring window)
        const int64_t nch = (T + B - 1) / B;
        const int64_t nlast = T - (nch - 1) * B;
        if (nch == 1 || nlast >= L) break;
        B += 64;
    }
    HS_CHECK(B < (1 << 30), HMMSORT_EINVAL, "wave engine: chain too long");
    g.B = (int)B;
    g.nch = (int)((T + B - 1) / B);
    int bits = 1;
    while ((1 << bits) < N + 1) bits++;
    g.EB = bits + 1; g.epw = 32 / g.EB; g.PW = (N + 1 + g.epw - 1) / g.epw;
    g.Bb = 512; g.Hb = 128;
    while (g.Hb < 2 * L + 64) g.Hb += 64;
    if (g.Bb < 2 * g.Hb) g.Bb = 2 * g.Hb;
    g.nseg = (T + g.Bb - 1) / g.Bb;
    return HMMSORT_OK;
}

template <typename Tv>
static int wmalloc(Tv **p, int64_t n, int64_t *bytes)
{
    if (hipMalloc((void **)p, (size_t)std::max<int64_t>(n, 1) * sizeof(Tv)) != hipSuccess) {
        (void)hipGetLastError();
        set_error("wave engine: hipMalloc of %.2f GB failed", n * sizeof(Tv) / 1e9);
        *p = nullptr;
        return HMMSORT_ENOMEM;
    }
    *bytes += n * (int64_t)sizeof(Tv);
    return HMMSORT_OK;
}

int wave_set_model(WaveDev *r, int ch, const HostModel &m)
{
    const WaveGeom &g = r->g;
    HS_CHECK(ch >= 0 && ch < g.C, HMMSORT_EINVAL, "wave set_model: channel %d outside 0..%d", ch, g.C - 1);
    HS_CHECK(m.ring.valid && m.ring.N == g.N && m.ring.L == g.L && m.S == r->S, HMMSORT_EINVAL,
             "wave set_model: model shape changed");
    HS_CHECK(g.N <= kWaveMaxPerSource || ring_uniform_cx(m.ring), HMMSORT_EUNSUP,
             "wave set_model: exit->entry log-probabilities depend on the source ring and there are more than 8 rings");
    r->bound_y = nullptr;
    r->ring[ch] = m.ring;
    r->mean[ch] = m.mean;
    r->sigma[ch] = m.sigma;
    const int N = g.N, L = g.L;
    const RingModel &R = m.ring;
    WaveConst k;
    memset(&k, 0, sizeof(k));
    k.c00 = R.c00;
    k.mean0 = m.mean[0];
    k.den = 2.0 * (m.sigma * m.sigma);
    k.A = -kLog2Pi - std::log(m.sigma);
    double sc0 = R.c00;
    for (int a = 0; a < N; a++) sc0 = std::max(sc0, R.cend[a]);
    if (!std::isfinite(sc0)) sc0 = 0.0;
    k.sc0 = sc0;
    k.P00 = std::exp(R.c00 - sc0);
    for (int a = 0; a < N; a++) {
        k.c0[a] = R.c0[a];
        k.cend[a] = R.cend[a];
        k.PEND[a] = std::exp(R.cend[a] - sc0);
        double sc = R.c0[a];                          // scale of the entries into ring a
        for (int b = 0; b < N; b++)
            if (b != a) sc = std::max(sc, R.cx[b * N + a]);
        if (!std::isfinite(sc)) sc = 0.0;
        sc = std::max(sc, kScFloor);
        k.sc[a] = sc;
        k.CP0[a] = std::exp(R.c0[a] - sc);
        k.xishift[a] = R.c0[a] - sc;
    }
    for (int a = 0; a < N; a++)
        for (int b = 0; b < N; b++) {
            k.cx[a * N + b] = (a == b) ? -INFINITY : R.cx[a * N + b];
            k.CPX[a * N + b] = (a == b) ? 0.0 : std::exp(R.cx[a * N + b] - k.sc[b]);
            k.cxT[b * N + a] = k.cx[a * N + b];
            k.CPXT[b * N + a] = k.CPX[a * N + b];
        }
    bool ucx = true;
    for (int a = 0; a < N; a++) {
        double ref = -INFINITY;
        bool have = false;
        for (int b = 0; b < N; b++) {
            if (b == a) continue;
            const double v = R.cx[b * N + a];
            if (!have) { ref = v; have = true; }
            else if (!(v == ref)) ucx = false;
        }
        k.cxin[a] = have ? ref : -INFINITY;
        k.CPXin[a] = have ? std::exp(ref - k.sc[a]) : 0.0;
    }
    r->ucx[ch] = ucx;
    r->uniform_cx = true;
    for (char u : r->ucx) r->uniform_cx = r->uniform_cx && u;
    HS_HIP(hipMemcpy(r->d_cst + ch, &k, sizeof(k), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(r->d_mean + (size_t)ch * m.S, m.mean.data(), m.S * sizeof(double), hipMemcpyHostToDevice));
    {
        const int Nn = m.ring.N, Ll = m.ring.L;
        std::vector<double> mt((size_t)Nn * Ll);
        for (int a = 0; a < Nn; a++)
            for (int k = 0; k < Ll; k++) mt[(size_t)k * Nn + a] = m.mean[1 + (size_t)a * Ll + k];
        HS_HIP(hipMemcpy(r->d_meanT + (size_t)ch * Nn * Ll, mt.data(), mt.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    std::vector<double> cint((size_t)N * (L + 1), 0.0), msq((size_t)N * (L + 1), 0.0);
    for (int a = 0; a < N; a++) {
        double acc = 0.0, acc2 = 0.0;
        for (int kk = 2; kk <= L; kk++) {
            acc += R.cint[(size_t)a * L + (kk - 1)];  // lp((a,kk-1)->(a,kk))
            cint[(size_t)a * (L + 1) + kk] = acc;
        }
        for (int kk = 1; kk <= L; kk++) {
            const double mv = m.mean[1 + (size_t)a * L + (kk - 1)];
            acc2 += mv * mv;
            msq[(size_t)a * (L + 1) + kk] = acc2;
        }
    }
    const size_t nt = (size_t)N * (L + 1);
    HS_HIP(hipMemcpy(r->d_cint + ch * nt, cint.data(), nt * sizeof(double), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(r->d_msq + ch * nt, msq.data(), nt * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> ctab;
    ctab.push_back(R.c00);
    ctab.insert(ctab.end(), R.c0.begin(), R.c0.end());
    ctab.insert(ctab.end(), R.cend.begin(), R.cend.end());
    ctab.insert(ctab.end(), R.cx.begin(), R.cx.end());
    ctab.insert(ctab.end(), R.cint.begin(), R.cint.end());
    HS_HIP(hipMemcpy(r->d_ctab + (size_t)ch * ctab.size(), ctab.data(), ctab.size() * sizeof(double),
                     hipMemcpyHostToDevice));
    if (ch == 0)
        HS_HIP(hipMemcpy(r->d_states, m.states.data(), m.states.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    return HMMSORT_OK;
}

int wave_create(WaveDev **out, const std::vector<HostModel> &models, int64_t T, int64_t block_req,
                int64_t halo_req)
{
    HS_CHECK(!models.empty(), HMMSORT_EINVAL, "wave engine: no model");
    const HostModel &m = models[0];
    std::string why;
    HS_CHECK(wave_supported(m, T, &why), HMMSORT_EUNSUP, "wave engine: %s", why.c_str());
    const int C = (int)models.size();
    WaveDev *r = new WaveDev();
    int rc = make_geometry(r->g, T, C, m.ring.N, m.ring.L, block_req, halo_req);
    if (rc) { delete r; return rc; }
    r->S = m.S; r->K = m.K;
    // captured launch sequences are opt-in: measured SLOWER than plain launches on ROCm 7.2 (headline step 1.24
    // against 1.11 ms: the replay does not overlap the decode branch with the E-step chain like three streams do)
    r->graphs_on = getenv("HMMSORT_GRAPHS") && atoi(getenv("HMMSORT_GRAPHS")) != 0;
    r->ring.resize(C); r->mean.resize(C); r->sigma.resize(C); r->ucx.assign(C, 1);
    const WaveGeom &g = r->g;
    const int64_t N = g.N, L = g.L, CT = (int64_t)C * T, nchT = (int64_t)C * g.nch;
    r->nparts = 1024;
    bool ok = true;
    auto A = [&](auto **p, int64_t n) { if (ok && wmalloc(p, n, &r->bytes)) ok = false; };
    A(&r->d_cst, C);
    A(&r->d_mean, C * m.S);
    A(&r->d_meanT, C * N * L);
    A(&r->d_cint, C * N * (L + 1));
    A(&r->d_msq, C * N * (L + 1));
    A(&r->d_ctab, C * (1 + 2 * N + N * N + N * L));
    A(&r->d_states, N * m.S);
    A(&r->Rf, N * CT);
    A(&r->W2, CT);
    A(&r->virt, C * N * (L + 1));
    A(&r->ysum, 2 * C);
    A(&r->psi, (int64_t)g.PW * CT);
    A(&r->vpre, nchT * (1 + N * L));
    A(&r->vend, nchT * (1 + N * L));
    A(&r->vfail, nchT + 8);
    A(&r->bstate, C * g.nseg);
    A(&r->redo, C * g.nseg + 8);
    A(&r->final_state, C + 8);
    A(&r->part, (int64_t)C * 4 * r->nparts);
    A(&r->FA0, CT);
    A(&r->FV, N * CT);
    A(&r->FREF, CT);
    A(&r->fpre, nchT * (1 + L * (N + 1)));
    A(&r->bpre, nchT * (1 + L * (N + 1)));
    A(&r->bown, nchT * (1 + L * (N + 1)));
    A(&r->rho, N * CT);
    A(&r->Zc, nchT);
    A(&r->partS, nchT * (3 * N + 3));
    r->gparts = (int)((T + 4095) / 4096);
    r->gparts = std::max<int>(r->gparts, g.nch);     // the fused backward sweep leaves one row per chain
    A(&r->partG, (int64_t)C * r->gparts * N * L);
    A(&r->yhead, C * (N * L + 2));
    A(&r->extra, C * 3 * N * L);
    A(&r->pp, C * m.S);
    A(&r->diag, 8);
    A(&r->dbg, 64);
    A(&r->trash, 64 * 64);
    r->tie_nblk = (T + kTieBlk - 1) / kTieBlk;
    A(&r->tie_cnt, (int64_t)C * 8);
    A(&r->tie_list, (int64_t)C * kTieCap);
    r->tie_ntile = (T + 4095) / 4096;
    A(&r->tie_off, (int64_t)C * (r->tie_ntile + 1));
    A(&r->tie_walk, (int64_t)C * kTieLanes * kTieWalk);
    A(&r->tie_guess, (int64_t)C * r->tie_nblk);
    A(&r->tie_c, (int64_t)C * r->tie_nblk * 2);
    A(&r->tie_ok, (int64_t)C * r->tie_nblk);
    A(&r->tie_v, (int64_t)C * (r->tie_nblk + 1));
    if (!ok) { wave_destroy(r); return HMMSORT_ENOMEM; }
    if (getenv("HMMSORT_POISON")) {  // test aid: NaN bit patterns in everything a kernel might read unwritten
        (void)hipMemset(r->Rf, 0xFF, N * CT * 8);
        (void)hipMemset(r->psi, 0xFF, (int64_t)g.PW * CT * 4);
        (void)hipMemset(r->vpre, 0xFF, nchT * (1 + N * L) * 8);
        (void)hipMemset(r->vend, 0xFF, nchT * (1 + N * L) * 8);
        (void)hipMemset(r->FA0, 0xFF, CT * 8);
        (void)hipMemset(r->FV, 0xFF, N * CT * 8);
        (void)hipMemset(r->FREF, 0xFF, CT * 8);
        (void)hipMemset(r->fpre, 0xFF, nchT * (1 + L * (N + 1)) * 8);
        (void)hipMemset(r->bpre, 0xFF, nchT * (1 + L * (N + 1)) * 8);
        (void)hipMemset(r->bown, 0xFF, nchT * (1 + L * (N + 1)) * 8);
        (void)hipMemset(r->rho, 0xFF, N * CT * 8);
        (void)hipMemset(r->Zc, 0xFF, nchT * 8);
        (void)hipMemset(r->partS, 0xFF, nchT * (3 * N + 3) * 8);
        (void)hipMemset(r->partG, 0xFF, (int64_t)C * r->gparts * N * L * 8);
        (void)hipMemset(r->yhead, 0xFF, C * (N * L + 2) * 8);
    }
    if (hipMemset(r->diag, 0, 8 * sizeof(int64_t)) != hipSuccess ||
        hipMemset(r->tie_cnt, 0, (size_t)C * 8 * sizeof(int64_t)) != hipSuccess ||
        hipMemset(r->vfail, 0, (nchT + 8) * sizeof(int32_t)) != hipSuccess) {
        wave_destroy(r);
        return HMMSORT_EHIP;
    }
    for (int ch = 0; ch < C; ch++) {
        const HostModel &mc = models[ch];
        if (mc.S != m.S || mc.K != m.K || mc.N != m.N) {
            set_error("wave engine: channel %d has a different model shape", ch);
            wave_destroy(r);
            return HMMSORT_EINVAL;
        }
        rc = wave_set_model(r, ch, mc);
        if (rc) { wave_destroy(r); return rc; }
    }
    // (a lowest-priority stream for the decode branch of hmmsort_plan_decode_estep changes nothing measurable:
    // workgroups already resident are not displaced)
    if (hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&r->side2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_b, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_c, hipEventDisableTiming) != hipSuccess) {
        set_error("wave engine: could not create the internal stream/events");
        wave_destroy(r);
        return HMMSORT_EHIP;
    }
    *out = r;
    return HMMSORT_OK;
}

void wave_destroy(WaveDev *r)
{
    if (!r) return;
    void *ptrs[] = {r->d_cst, r->d_mean, r->d_meanT, r->d_cint, r->d_msq, r->d_ctab, r->d_states, r->Rf, r->W2, r->virt, r->ysum,
                    r->psi, r->vpre, r->vend, r->vfail, r->bstate, r->redo, r->final_state, r->part, r->FA0,
                    r->FV, r->FREF, r->fpre, r->bpre, r->bown, r->rho, r->Zc, r->partS, r->partG, r->yhead,
                    r->extra, r->pp, r->diag, r->dbg, r->trash, r->tie_cnt, r->tie_list, r->tie_off, r->tie_walk, r->tie_guess,
                    r->tie_c, r->tie_ok, r->tie_v};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (r->ev_fork) (void)hipEventDestroy(r->ev_fork);
    if (r->ev_join) (void)hipEventDestroy(r->ev_join);
    if (r->ev_a) (void)hipEventDestroy(r->ev_a);
    if (r->ev_b) (void)hipEventDestroy(r->ev_b);
    if (r->ev_c) (void)hipEventDestroy(r->ev_c);
    if (r->side) (void)hipStreamDestroy(r->side);
    if (r->side2) (void)hipStreamDestroy(r->side2);
    for (auto &e : r->prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &ge : r->graphs) (void)hipGraphExecDestroy(ge.exec);
    delete r;
}

int wave_diagnostics(WaveDev *r, hipStream_t st, int64_t diag[8])
{
    HS_HIP(hipStreamSynchronize(st));
    HS_HIP(hipMemcpy(diag, r->diag, 8 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return HMMSORT_OK;
}

int64_t wave_stats_len(const WaveDev *r) { return 3 * (int64_t)r->g.N * r->g.L + r->g.N + 4; }

// ------------------------------------------------------------------------------------------
// pre-pass: ring scores for every onset time t' in [0, T) of every channel, natural layout
//   Rf[ch][a][t'] = Cint[a][kmax] - (sum_k y^2 - 2 sum_k y*mean(a,k) + Msq[a][kmax]) / den,
//   kmax = min(L, T - t')  (rings running off the end of the data are truncated: the reference's
//   terminal conditions, viterbi.jl:90 / baumwelch.jl:80); y beyond the end counts as 0.
// Block = 256 threads x R CONSECUTIVE onsets per thread: a thread's onsets share a sliding window of y in
// registers (one LDS read per lag instead of one per onset and lag), the means of a lag are wave-uniform
// scalar loads, and the results leave through LDS so that every global store is a full row segment.
// The y tile is padded by one slot per R (p(i) = i + i/R): lanes R apart then hit different banks.
// Accumulation order over the lags is the textbook one (k = 0..L-1, fused multiply-add).
// Also accumulates sum y and sum y^2 per channel (magnitude of the reference's trellis, for the
// near-tie threshold of the Viterbi sweep).
// ------------------------------------------------------------------------------------------
template <int N> constexpr int pre_rows() { return N <= 4 ? 8 : 4; }   // measured: N = 4: 4 rows 0.30 ms, 8 rows 0.19 ms, 16 rows 0.24 ms (10 M samples); N = 8: 8 rows 2.44 ms, 4 rows 1.92 ms (40 M)

template <int N>
__global__ __launch_bounds__(256) void kw_prepass(WaveGeom g, const WaveConst *__restrict__ cst,
                                                  const double *__restrict__ y,
                                                  const double *__restrict__ mean,
                                                  const double *__restrict__ cint,
                                                  const double *__restrict__ msq,
                                                  double *__restrict__ Rf, double *__restrict__ W2,
                                                  double *__restrict__ ysum)
{
    constexpr int R = pre_rows<N>(), TILE = 256 * R;
    extern __shared__ double ly[];  // y tile (TILE + L, padded); reused for the transposed results (TILE, padded)
    __shared__ double red[8];
    const int ch = blockIdx.y, L = g.L, tid = threadIdx.x;
    const int64_t T = g.T, t0 = (int64_t)blockIdx.x * TILE;
    const double *yc = y + (int64_t)ch * T;
    const double *mc = mean + (int64_t)ch * N * L;    // lag-major: mc[k * N + a]
    auto pad = [](int i) { return i + i / R; };
    double s1 = 0.0, s2 = 0.0;
    for (int i = tid; i < TILE + L; i += 256) {
        const int64_t t = t0 + i;
        const double v = yc[t < T ? t : T - 1];
        const double yv = t < T ? v : 0.0;
        ly[pad(i)] = yv;
        if (i < TILE) { s1 += yv; s2 = __builtin_fma(yv, yv, s2); }
    }
    __syncthreads();
    // window of R + 1 registers, slot (r + k) mod (R + 1) = y[t + k + r]; the free slot receives the sample
    // the NEXT lag needs (one lag ahead of its use, like the means), so the loop is unrolled by R + 1
    constexpr int U = R + 1;
    double dot[R][N], ysq[R], w[U];
#pragma unroll
    for (int r = 0; r < R; r++) {
        ysq[r] = 0.0;
        w[r] = ly[pad(R * tid + r)];
#pragma unroll
        for (int a = 0; a < N; a++) dot[r][a] = 0.0;
    }
    double mv[N];
#pragma unroll
    for (int a = 0; a < N; a++) mv[a] = mc[a];
    for (int k0 = 0; k0 < L; k0 += U) {
#pragma unroll
        for (int kk = 0; kk < U; kk++) {
            const int k = k0 + kk;
            if (k < L) {
                const int kn = k + 1 < L ? k + 1 : k;
                double mn[N];
#pragma unroll
                for (int a = 0; a < N; a++) mn[a] = mc[kn * N + a];   // wave-uniform (one wide scalar load)
                w[(kk + R) % U] = ly[pad(R * tid + k + R)];           // y[t + (k+1) + (R-1)]
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const double yv = w[(r + kk) % U];
                    ysq[r] = __builtin_fma(yv, yv, ysq[r]);
#pragma unroll
                    for (int a = 0; a < N; a++) dot[r][a] = __builtin_fma(yv, mv[a], dot[r][a]);
                }
#pragma unroll
                for (int a = 0; a < N; a++) mv[a] = mn[a];
            }
        }
    }
    // Rf = Cint - (ysq - 2 dot + Msq) / den with the reciprocal of den: one ulp from the division, far
    // inside the rounding noise the near-tie threshold of the Viterbi sweep already allows for this sum
    const double rden = 1.0 / cst[ch].den;
    const bool interior = t0 + TILE + L <= T;   // no onset of this tile runs off the end of the data
    auto value = [&](int a, int r, double msqL, double cintL, int64_t cb) {
        double mq = msqL, ci = cintL;
        if (!interior) {
            const int64_t rem = T - (t0 + R * tid + r);
            const int kmax = rem < L ? (rem > 0 ? (int)rem : 0) : L;
            mq = msq[cb + kmax]; ci = cint[cb + kmax];
        }
        return ci - ((ysq[r] - 2.0 * dot[r][a]) + mq) * rden;
    };
    {
        // ring by ring through LDS, thread-major in, time-major out (stores straight from the registers, lanes
        // 64 bytes apart per instruction, measured slower: 0.225 against 0.194 ms at 10 M samples)
#pragma unroll
        for (int a = 0; a <= N; a++) {
            __syncthreads();
            const int64_t cb = ((int64_t)ch * N + (a < N ? a : 0)) * (L + 1);
            const double msqL = msq[cb + L], cintL = cint[cb + L];
#pragma unroll
            for (int r = 0; r < R; r++)
                ly[pad(R * tid + r)] = a < N ? value(a < N ? a : 0, r, msqL, cintL, cb) : ysq[r];
            __syncthreads();
            double *dst = a < N ? Rf + ((int64_t)ch * N + a) * T : W2 + (int64_t)ch * T;
            for (int i = tid; i < TILE; i += 256)
                if (t0 + i < T) dst[t0 + i] = ly[pad(i)];
        }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = s1; red[4 + (tid >> 6)] = s2; }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&ysum[2 * ch], (red[0] + red[1]) + (red[2] + red[3]));
        atomicAdd(&ysum[2 * ch + 1], (red[4] + red[5]) + (red[6] + red[7]));
    }
}

// virtual onsets t' = -j, j = 1..L-1: rings already running at the first sample (the reference's
// first column is "emission only" for every state: viterbi.jl:55-62, baumwelch.jl:36).  Their score
// covers phases k = 1+j..L on samples 0..L-1-j.  V[ch][a][j]; V[.][.][L] = -inf ("no such onset").
__global__ void kw_virtual(WaveGeom g, const WaveConst *__restrict__ cst, const double *__restrict__ y,
                           const double *__restrict__ mean, const double *__restrict__ cint,
                           double *__restrict__ virt)
{
    const int L = g.L, N = g.N, ch = blockIdx.x, S = 1 + N * L;
    const double *yc = y + (int64_t)ch * g.T, *mc = mean + (int64_t)ch * S;
    const double *ci = cint + (int64_t)ch * N * (L + 1);
    const double den = cst[ch].den, A = cst[ch].A;
    for (int i = threadIdx.x; i < N * L; i += blockDim.x) {
        const int a = i / L, j = i % L + 1;  // j = 1..L
        double v;
        if (j == L) {
            v = -INFINITY;
        } else if (j == L - 1) {
            // A ring in its LAST phase at the first sample: one emission term.  Template tails are
            // ~1e-16 (sin(3*pi)), so these N candidates tie to the last bit in the reference, whose
            // first column is funcl = A - d*d/den (viterbi.jl:55-62); round exactly like it, then take
            // A out again, so that equal reference values stay equal here (lowest ring wins the tie).
            const double d = yc[0] - mc[1 + a * L + (L - 1)];
            v = (A - (d * d) / den) - A;
        } else {
            double acc = 0.0;
            for (int k = 1 + j; k <= L; k++) {
                const double d = yc[k - 1 - j] - mc[1 + a * L + (k - 1)];
                acc += d * d;
            }
            v = (ci[a * (L + 1) + L] - ci[a * (L + 1) + (1 + j)]) - acc / den;
        }
        virt[((int64_t)ch * N + a) * (L + 1) + j] = v;
    }
}

int wave_prepare(WaveDev *r, const double *d_y, hipStream_t st)
{
    if (r->bound_y == d_y) return HMMSORT_OK;
    r->bound_y = nullptr;
    const WaveGeom &g = r->g;
    HS_HIP(hipMemsetAsync(r->ysum, 0, 2 * g.C * sizeof(double), st));
    int rc = dispatch_N(g.N, [&](auto n) {
        constexpr int N = decltype(n)::value;
        WPROF(r, "kw_prepass", st);
        constexpr int kPreTile = 256 * pre_rows<N>();
        hipLaunchKernelGGL((kw_prepass<N>), dim3((unsigned)((g.T + kPreTile - 1) / kPreTile), g.C), dim3(256),
                           (size_t)((kPreTile + g.L) + (kPreTile + g.L) / pre_rows<N>() + 2) * sizeof(double), st, g, r->d_cst, d_y, r->d_meanT, r->d_cint,
                           r->d_msq, r->Rf, r->W2, r->ysum);
        HS_HIP(hipGetLastError());
        return HMMSORT_OK;
    });
    if (rc) return rc;
    { WPROF(r, "kw_virtual", st);
      hipLaunchKernelGGL(kw_virtual, dim3(g.C), dim3(256), 0, st, g, r->d_cst, d_y, r->d_mean, r->d_cint, r->virt); }
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int wave_bind(WaveDev *r, const double *d_y, hipStream_t st)
{
    r->bound_y = nullptr;
    int rc = wave_prepare(r, d_y, st);
    if (rc) return rc;
    r->bound_y = d_y;
    return HMMSORT_OK;
}

int wave_profile_read(WaveDev *r, hipStream_t st, std::vector<std::string> &names,
                      std::vector<double> &ms, std::vector<int64_t> &calls)
{
    HS_HIP(hipStreamSynchronize(st));
    if (r->side) HS_HIP(hipStreamSynchronize(r->side));
    if (r->side2) HS_HIP(hipStreamSynchronize(r->side2));
    for (auto &e : r->prof) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) { (void)hipGetLastError(); t = 0.f; }
        size_t i = 0;
        for (; i < names.size(); i++)
            if (names[i] == e.name) break;
        if (i == names.size()) { names.push_back(e.name); ms.push_back(0.0); calls.push_back(0); }
        ms[i] += t;
        calls[i] += 1;
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    r->prof.clear();
    return HMMSORT_OK;
}

}  // namespace hmmsort

// ---- self-test of the cross-lane primitives (DPP scans vs their shuffle references) ---------------
namespace hmmsort {
__global__ void kw_selftest_scans(const double *__restrict__ in, double *__restrict__ out)
{
    const int lane = threadIdx.x;
    double a = in[lane], b = in[64 + lane], c = in[128 + lane], d = in[192 + lane];
    double a2 = a, b2 = b, c2 = c, d2 = d;
    scan_maxplus(a, b);
    scan_maxplus_ref(a2, b2, lane);
    scan_linear(c, d);
    scan_linear_ref(c2, d2, lane);
    out[lane] = a; out[64 + lane] = b; out[128 + lane] = a2; out[192 + lane] = b2;
    out[256 + lane] = c; out[320 + lane] = d; out[384 + lane] = c2; out[448 + lane] = d2;
    out[512 + lane] = lane_prev(in[lane], 123.5);
    out[576 + lane] = lane_prev_ref(in[lane], 123.5, lane);
    out[640 + lane] = wave_bcast(in[lane], 63);
    float fa = (float)in[lane], fb = (float)in[64 + lane];
    scan_maxplus_f32(fa, fb);
    out[704 + lane] = fa; out[768 + lane] = fb;
    out[832 + lane] = lane_prevf((float)in[lane], 7.25f);
}
}  // namespace hmmsort

extern "C" int hmmsort_selftest_scans(const double *in256, double *out896)
{
    using namespace hmmsort;
    double *di = nullptr, *dout = nullptr;
    HS_HIP(hipMalloc((void **)&di, 256 * sizeof(double)));
    HS_HIP(hipMalloc((void **)&dout, 896 * sizeof(double)));
    HS_HIP(hipMemcpy(di, in256, 256 * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kw_selftest_scans, dim3(1), dim3(64), 0, nullptr, di, dout);
    HS_HIP(hipDeviceSynchronize());
    HS_HIP(hipMemcpy(out896, dout, 896 * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(di); (void)hipFree(dout);
    return HMMSORT_OK;
}
