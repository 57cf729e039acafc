// Ring engine, part 1: geometry, workspace, transposes, the parallel pre-pass (ring scores).
// Design notes: ring_common.h / DESIGN.md.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "ring_common.h"

namespace hmmsort {

// ------------------------------------------------------------------------------------------
// geometry
// ------------------------------------------------------------------------------------------
static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

bool ring_supported(const HostModel &m, int64_t T, std::string *why)
{
    auto no = [&](const char *w) { if (why) *why = w; return false; };
    if (!m.ring.valid) return no("transition list is not the no-overlap ring pattern");
    if (m.ring.N > kRingMaxN) return no("more than 16 rings");
    if (m.ring.L < kRingMinL) return no("rings shorter than 16 states");
    // k_halo_check keeps an L x 64 prefix table in LDS beside 8 KB of static arrays: 160 KB per workgroup
    if ((int64_t)m.ring.L * 64 * 8 + 8192 > 160 * 1024) return no("rings longer than 304 states");
    if (T < 4 * (int64_t)m.ring.L || T < 512) return no("signal shorter than 4 ring lengths / 512 samples");
    return true;
}

static int make_geometry(RingGeom &g, int64_t T, int N, int L, int64_t block_req, int64_t halo_req)
{
    g.T = T; g.N = N; g.L = L;
    g.own_lo = 0; g.own_hi = T; g.first = 1; g.last = 1;
    // warm-up: four ring lengths, at least 256 samples (scripts/sweep_halo.py: at 10 M samples the
    // certificates pass at 256 on sparse and dense signals, 128 is flagged 4..64 times).  Whether
    // a warm-up was long enough is CERTIFIED on device after every call (k_halo_check,
    // k_fb_check); the host-buffer entry points double it and retry when a certificate fails.
    int64_t H = halo_req > 0 ? halo_req : std::max<int64_t>(256, 4 * (int64_t)L);
    H = round_up(std::max<int64_t>(H, L + 40), 64);  // the boundary checks need L+8 settled steps
    // chain length: twice the warm-up, or longer when the signal alone fills 1024 SIMDs x 64 lanes
    int64_t B = block_req > 0 ? block_req : std::max<int64_t>(H, (T + 65535) / 65536);
    B = round_up(std::max<int64_t>(B, std::max<int64_t>(std::max<int64_t>(H, L + 16), 128)), 64);
    if (B > T) B = round_up(T, 64);
    if (H > B) H = B;
    // every chain must own >= L samples (the per-chain normaliser needs a full ring window)
    for (;;) {
        int64_t nch = (T + B - 1) / B;
        int64_t nlast = T - (nch - 1) * B;
        if (nch == 1 || nlast >= L) break;
        B += 64;
    }
    HS_CHECK(B + H + L < (1 << 30), HMMSORT_EINVAL, "ring engine: block too long");
    g.B = (int)B; g.H = (int)H;
    g.nch = (int)((T + B - 1) / B);
    g.ncol = (int)round_up(g.nch, 64);
    int bits = 1;
    while ((1 << bits) < N + 1) bits++;
    g.bits = bits; g.epw = 32 / bits; g.W = (N + 1 + g.epw - 1) / g.epw;
    return HMMSORT_OK;
}

template <typename Tv>
static int dmalloc(Tv **p, int64_t n, int64_t *bytes)
{
    if (hipMalloc((void **)p, (size_t)std::max<int64_t>(n, 1) * sizeof(Tv)) != hipSuccess) {
        (void)hipGetLastError();
        set_error("ring engine: hipMalloc of %.2f GB failed", n * sizeof(Tv) / 1e9);
        *p = nullptr;
        return HMMSORT_ENOMEM;
    }
    *bytes += n * (int64_t)sizeof(Tv);
    return HMMSORT_OK;
}

int ring_set_model(RingDev *r, const HostModel &m)
{
    HS_CHECK(m.ring.valid && m.ring.N == r->g.N && m.ring.L == r->g.L && m.S == r->S,
             HMMSORT_EINVAL, "ring set_model: model shape changed");
    r->bound_y = nullptr;  // ring scores depend on the model
    r->ring = m.ring;
    r->mean = m.mean;
    r->sigma = m.sigma;
    r->lsig = std::log(m.sigma);
    r->A = -kLog2Pi - r->lsig;
    r->den = 2.0 * (m.sigma * m.sigma);
    const int N = r->g.N, L = r->g.L;
    HS_HIP(hipMemcpy(r->d_mean, m.mean.data(), m.S * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> cint((size_t)N * (L + 1), 0.0);
    for (int a = 0; a < N; a++) {
        double acc = 0.0;
        cint[(size_t)a * (L + 1) + 0] = 0.0;
        cint[(size_t)a * (L + 1) + 1] = 0.0;
        for (int kk = 2; kk <= L; kk++) {
            acc += m.ring.cint[(size_t)a * L + (kk - 1)];  // lp((a,kk-1)->(a,kk))
            cint[(size_t)a * (L + 1) + kk] = acc;
        }
    }
    HS_HIP(hipMemcpy(r->d_cint, cint.data(), cint.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> msq((size_t)N * (L + 1), 0.0);  // Msq[a][kk] = sum_{k=1}^{kk} mean(a,k)^2
    for (int a = 0; a < N; a++) {
        double acc = 0.0;
        for (int kk = 1; kk <= L; kk++) {
            const double mv = m.mean[1 + (size_t)a * L + (kk - 1)];
            acc += mv * mv;
            msq[(size_t)a * (L + 1) + kk] = acc;
        }
    }
    HS_HIP(hipMemcpy(r->d_msq, msq.data(), msq.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> ctab;
    ctab.push_back(m.ring.c00);
    ctab.insert(ctab.end(), m.ring.c0.begin(), m.ring.c0.end());
    ctab.insert(ctab.end(), m.ring.cend.begin(), m.ring.cend.end());
    ctab.insert(ctab.end(), m.ring.cx.begin(), m.ring.cx.end());
    ctab.insert(ctab.end(), m.ring.cint.begin(), m.ring.cint.end());
    HS_HIP(hipMemcpy(r->d_ctab, ctab.data(), ctab.size() * sizeof(double), hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(r->d_states, m.states.data(), m.states.size() * sizeof(int16_t),
                     hipMemcpyHostToDevice));
    return HMMSORT_OK;
}

int ring_create(RingDev **out, const HostModel &m, int64_t T, int64_t block_req, int64_t halo_req)
{
    std::string why;
    HS_CHECK(ring_supported(m, T, &why), HMMSORT_EUNSUP, "ring engine: %s", why.c_str());
    RingDev *r = new RingDev();
    int rc = make_geometry(r->g, T, m.ring.N, m.ring.L, block_req, halo_req);
    if (rc) { delete r; return rc; }
    r->S = m.S; r->K = m.K;
    const RingGeom &g = r->g;
    const int64_t BC = (int64_t)g.B * g.ncol, N = g.N, L = g.L;
    r->nparts = 1024;
    bool ok = true;
    auto A = [&](auto **p, int64_t n) { if (ok && dmalloc(p, n, &r->bytes)) ok = false; };
    A(&r->d_mean, m.S);
    A(&r->d_cint, N * (L + 1));
    A(&r->d_msq, N * (L + 1));
    A(&r->d_ctab, 1 + 2 * N + N * N + N * L);
    A(&r->d_states, N * m.S);
    A(&r->yT, BC);
    A(&r->Rf, N * BC);
    A(&r->P, N * (int64_t)(g.H + g.B) * g.ncol);
    A(&r->Pv, N * (int64_t)(g.H + g.B) * g.ncol);
    A(&r->Q, N * (int64_t)(g.L + g.B + g.H) * g.ncol);
    A(&r->A0, (int64_t)(1 + g.B) * g.ncol);
    A(&r->B0, BC);
    A(&r->psi, (int64_t)g.W * BC);
    A(&r->D0end, g.ncol);
    A(&r->D0pre, g.ncol);
    A(&r->bstate, g.ncol);
    A(&r->redo, g.ncol + 8);
    A(&r->xT, BC);
    A(&r->final_state, 8);
    A(&r->part, 4 * r->nparts);
    A(&r->Zc, g.ncol);
    A(&r->Zp, 8 * g.ncol);
    A(&r->B0h, g.ncol);
    A(&r->partA, (int64_t)(g.ncol / 64) * 2 * N * L);
    A(&r->partS, (int64_t)(g.ncol / 64) * ((g.B + 31) / 32) * (2 * N + 3));
    A(&r->rhoT, N * BC);
    A(&r->extra, 3 * N * L);
    A(&r->pp, m.S);
    A(&r->diag, 8);
    if (!ok) { ring_destroy(r); return HMMSORT_ENOMEM; }
    if (getenv("HMMSORT_POISON")) {
        // test aid: fill the work arrays with NaN bit patterns so that a kernel reading an
        // element nobody wrote shows up as NaN instead of depending on stale memory
        (void)hipMemset(r->P, 0xFF, N * (int64_t)(g.H + g.B) * g.ncol * 8);
        (void)hipMemset(r->Pv, 0xFF, N * (int64_t)(g.H + g.B) * g.ncol * 8);
        (void)hipMemset(r->Q, 0xFF, N * (int64_t)(g.L + g.B + g.H) * g.ncol * 8);
        (void)hipMemset(r->A0, 0xFF, (int64_t)(1 + g.B) * g.ncol * 8);
        (void)hipMemset(r->B0, 0xFF, BC * 8);
        (void)hipMemset(r->Rf, 0xFF, N * BC * 8);
        (void)hipMemset(r->yT, 0xFF, BC * 8);
        (void)hipMemset(r->psi, 0xFF, (int64_t)g.W * BC * 4);
        (void)hipMemset(r->xT, 0xFF, BC * 2);
        (void)hipMemset(r->Zc, 0xFF, g.ncol * 8);
        (void)hipMemset(r->partA, 0xFF, (int64_t)(g.ncol / 64) * 2 * N * L * 8);
        (void)hipMemset(r->partS, 0xFF, (int64_t)(g.ncol / 64) * ((g.B + 31) / 32) * (2 * N + 3) * 8);
        (void)hipMemset(r->rhoT, 0xFF, N * BC * 8);
    }
    if (hipMemset(r->diag, 0, 8 * sizeof(int64_t)) != hipSuccess) { ring_destroy(r); return HMMSORT_EHIP; }
    rc = ring_set_model(r, m);
    if (rc) { ring_destroy(r); return rc; }
    if (hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_post, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_chk, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_edges, hipEventDisableTiming) != hipSuccess) {
        set_error("ring engine: could not create the internal stream/events");
        ring_destroy(r);
        return HMMSORT_EHIP;
    }
    *out = r;
    return HMMSORT_OK;
}

void ring_destroy(RingDev *r)
{
    if (!r) return;
    void *ptrs[] = {r->d_mean, r->d_cint, r->d_msq, r->d_ctab, r->d_states, r->yT, r->Rf, r->P, r->Pv, r->Q, r->A0,
                    r->B0, r->psi, r->D0end, r->D0pre, r->bstate, r->redo, r->xT, r->final_state,
                    r->part, r->Zc, r->Zp, r->B0h, r->partA, r->partS, r->rhoT, r->extra, r->pp, r->diag};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (r->ev_fork) (void)hipEventDestroy(r->ev_fork);
    if (r->ev_join) (void)hipEventDestroy(r->ev_join);
    if (r->ev_post) (void)hipEventDestroy(r->ev_post);
    if (r->ev_chk) (void)hipEventDestroy(r->ev_chk);
    if (r->ev_edges) (void)hipEventDestroy(r->ev_edges);
    if (r->side) (void)hipStreamDestroy(r->side);
    delete r;
}

int64_t ring_workspace_bytes(const RingDev *r) { return r->bytes; }
void ring_geometry(const RingDev *r, int64_t *block, int64_t *halo, int64_t *nchains)
{
    if (block) *block = r->g.B;
    if (halo) *halo = r->g.H;
    if (nchains) *nchains = r->g.nch;
}

int ring_diagnostics(RingDev *r, hipStream_t st, int64_t diag[8])
{
    HS_HIP(hipStreamSynchronize(st));
    HS_HIP(hipMemcpy(diag, r->diag, 8 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return HMMSORT_OK;
}

// ------------------------------------------------------------------------------------------
// transposes:  y[c*B + s]  <->  yT[s*ncol + c]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose_in(const double *__restrict__ y, int64_t T,
                                                      int B, int ncol, double *__restrict__ yT)
{
    __shared__ double tile[64][65];
    const int s0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int cl = ty + 4 * i;
        const int64_t t = (int64_t)(c0 + cl) * B + s0 + tx;
        tile[cl][tx] = (t < T) ? y[t] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int sl = ty + 4 * i;
        yT[(int64_t)(s0 + sl) * ncol + c0 + tx] = tile[tx][sl];
    }
}

int ring_launch_transpose_in(RingDev *r, const double *d_y, hipStream_t st)
{
    const RingGeom &g = r->g;
    { PROF(r, "k_transpose_in", st); hipLaunchKernelGGL(k_transpose_in, dim3(g.B / 64, g.ncol / 64), dim3(256), 0, st, d_y, g.T, g.B,
                       g.ncol, r->yT); }
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

// ------------------------------------------------------------------------------------------
// pre-pass: ring scores for every onset time t' in [0, T)
//   Rf[a][t'] = -(1/den) * sum_{k=1..kmax} (y[t'+k-1] - mean(a,k))^2 + Cint[a][kmax],
//   (the square is expanded: sum y^2 - 2 sum y*mean + sum mean^2; fp64, error ~1e-13 absolute)
//   kmax = min(L, T - t')   (rings that run off the end of the data are truncated: the
//   reference's terminal conditions, viterbi.jl:90 / baumwelch.jl:80).
// The per-sample constant A = -log2pi - log(sigma) is left out everywhere (it shifts every state
// of a time step equally and cancels from every decision and posterior); ll adds it back.
// One thread = one chain column x RS consecutive rows; a y sample is loaded once and feeds the
// RS overlapping windows from registers.  Means are wave-uniform (scalar loads).
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void k_prepass(RingGeom g, const double *__restrict__ yT,
                                                 const double *__restrict__ mean,
                                                 const double *__restrict__ cint,
                                                 const double *__restrict__ msq, double den,
                                                 double *__restrict__ Rf)
{
    constexpr int RS = prepass_rows<N>();  // windows (onset rows) per thread
    constexpr int WR = 2 * RS;             // y ring buffer: RS in use + RS fetched ahead
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int s0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * RS;
    const int L = g.L, B = g.B, ncol = g.ncol;
    const int64_t T = g.T;
    const int64_t tbase = (int64_t)c * B + s0;
    const bool cact = c < g.nch;
    auto Y = [&](int j) -> double {  // y[tbase + j]; 0 past the end of the data (truncated rings)
        const bool ok = cact && tbase + j < T;        // unconditional load from a clamped address
        const int row = ok ? s0 + j : 0;
        const int cc = cact ? c : 0;
        const int64_t o = (row < B) ? (int64_t)row * ncol + cc : (int64_t)(row - B) * ncol + cc + 1;
        const double v = yT[o];
        return ok ? v : 0.0;
    };
    // sum_k (y - m)^2 = sum y^2 - 2 sum y*m + sum m^2 : one fma per (window, ring, phase).
    // Phase-outer loop: the mean of phase k is wave-uniform (N scalar loads per phase) and the RS
    // windows slide over a register ring buffer of y (one coalesced load per phase, issued RS
    // phases before its first use).
    double dot[RS][N], ysq[RS], w[WR];
#pragma unroll
    for (int r = 0; r < RS; r++) {
        ysq[r] = 0.0;
#pragma unroll
        for (int a = 0; a < N; a++) dot[r][a] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < WR; i++) w[i] = Y(i);
    for (int kb = 0; kb < L; kb += WR) {
#pragma unroll
        for (int u = 0; u < WR; u++) {
            const int k = kb + u + 1;
            if (k <= L) {  // wave-uniform
                double mv[N];
#pragma unroll
                for (int a = 0; a < N; a++) mv[a] = mean[1 + a * L + (k - 1)];
#pragma unroll
                for (int r = 0; r < RS; r++) {
                    const double yv = w[(u + r) % WR];  // y[tbase + r + k - 1]
                    ysq[r] = __builtin_fma(yv, yv, ysq[r]);
#pragma unroll
                    for (int a = 0; a < N; a++) dot[r][a] = __builtin_fma(yv, mv[a], dot[r][a]);
                }
                w[u % WR] = Y(k - 1 + WR);  // slot of y[tbase + k - 1] is free now
            }
        }
    }
    if (!cact) return;
#pragma unroll
    for (int r = 0; r < RS; r++) {
        const int64_t t0 = tbase + r;
        if (t0 < T) {
            const int64_t rem = T - t0;
            const int kmax = rem < L ? (int)rem : L;
#pragma unroll
            for (int a = 0; a < N; a++) {
                const double ss = (ysq[r] - 2.0 * dot[r][a]) + msq[a * (L + 1) + kmax];
                Rf[(int64_t)a * B * ncol + (int64_t)(s0 + r) * ncol + c] = cint[a * (L + 1) + kmax] - ss / den;
            }
        }
    }
}

int ring_launch_prepass(RingDev *r, hipStream_t st)
{
    const RingGeom &g = r->g;
    return dispatch_N(g.N, [&](auto n) {
        constexpr int N = decltype(n)::value;
        constexpr int RS = prepass_rows<N>();
        { PROF(r, "k_prepass", st); hipLaunchKernelGGL((k_prepass<N>), dim3(g.ncol / 64, g.B / (4 * RS)), dim3(256), 0, st, g,
                           r->yT, r->d_mean, r->d_cint, r->d_msq, r->den, r->Rf); }
        HS_HIP(hipGetLastError());
        return HMMSORT_OK;
    });
}

// ------------------------------------------------------------------------------------------
// virtual onsets t' = -j, j = 1..L-1: rings already running at the first sample (the
// reference's first column is "emission only" for every state: viterbi.jl:55-62,
// baumwelch.jl:36).  Their score covers phases k = 1+j..L on samples 0..L-1-j.  Written into
// chain 0's (otherwise unused) warm-up rows of the delay-line array: row H-j, column 0; row H-L
// is the "no such onset" marker -Inf.
// ------------------------------------------------------------------------------------------
__global__ void k_virtual(RingGeom g, const double *__restrict__ y, const double *__restrict__ mean,
                          const double *__restrict__ cint, double den, double A,
                          double *__restrict__ dst, double *__restrict__ dst2, int64_t plane_stride)
{
    const int L = g.L, N = g.N;
    for (int i = threadIdx.x; i < N * L; i += blockDim.x) {
        const int a = i / L, j = i % L + 1;  // j = 1..L
        double v;
        if (j == L) {
            v = -INFINITY;
        } else if (j == L - 1) {
            // A ring in its LAST phase at the first sample: one emission term.  Template tails are
            // ~1e-16 (sin(3*pi)), so these N candidates tie to the last bit in the reference, whose
            // first column is funcl = A - d*d/den (viterbi.jl:55-62) with the small term absorbed
            // into A's rounding; with sigma < 0.4 (A > 0) one of them usually IS the decoded first
            // state.  Round exactly like the reference, then take A out again, so that equal
            // reference values stay equal here and the lowest ring wins the tie as it does there.
            const double d = y[0] - mean[1 + a * L + (L - 1)];
            v = (A - (d * d) / den) - A;
        } else {
            double acc = 0.0;
            for (int k = 1 + j; k <= L; k++) {
                const double d = y[k - 1 - j] - mean[1 + a * L + (k - 1)];
                acc += d * d;
            }
            v = (cint[a * (L + 1) + L] - cint[a * (L + 1) + (1 + j)]) - acc / den;
        }
        dst[(int64_t)a * plane_stride + (int64_t)(g.H - j) * g.ncol + 0] = v;
        if (dst2) dst2[(int64_t)a * plane_stride + (int64_t)(g.H - j) * g.ncol + 0] = v;
    }
}

int ring_launch_virtual(RingDev *r, const double *d_y, double *dst, int64_t plane_stride,
                        hipStream_t st, double *dst2)
{
    { PROF(r, "k_virtual", st); hipLaunchKernelGGL(k_virtual, dim3(1), dim3(256), 0, st, r->g, d_y, r->d_mean, r->d_cint, r->den,
                       r->A, dst, dst2, plane_stride); }
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

// transpose + pre-pass unless d_y is the bound signal
int ring_prepare(RingDev *r, const double *d_y, hipStream_t st)
{
    if (r->bound_y == d_y) return HMMSORT_OK;
    r->bound_y = nullptr;
    int rc;
    if ((rc = ring_launch_transpose_in(r, d_y, st))) return rc;
    return ring_launch_prepass(r, st);
}

int ring_bind(RingDev *r, const double *d_y, hipStream_t st)
{
    r->bound_y = nullptr;
    int rc = ring_prepare(r, d_y, st);
    if (rc) return rc;
    r->bound_y = d_y;
    return HMMSORT_OK;
}

int ring_profile_enable(RingDev *r, int on)
{
    r->prof_on = on != 0;
    return HMMSORT_OK;
}

// Synchronises the stream; returns per-kernel-name total milliseconds and call counts since the
// last read.
int ring_profile_read(RingDev *r, hipStream_t st, std::vector<std::string> &names,
                      std::vector<double> &ms, std::vector<int64_t> &calls)
{
    HS_HIP(hipStreamSynchronize(st));
    for (auto &e : r->prof) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) t = 0.f;
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

int ring_viterbi(RingDev *r, const double *d_y, int16_t *d_x, double *d_ll, hipStream_t st)
{
    return ring_viterbi_launch(r, d_y, d_x, d_ll, st);
}
int ring_estep(RingDev *r, const double *d_y, double *d_stats, hipStream_t st)
{
    return ring_estep_launch(r, d_y, d_stats, st);
}
int ring_mstep(RingDev *r, const double *d_stats, double *d_out, hipStream_t st)
{
    return ring_mstep_launch(r, d_stats, d_out, st);
}
int64_t ring_stats_len(const RingDev *r)
{
    // per ring state G0,G1,G2 | per ring: xi-sum | gamma0 sums (all t, t<T-1), gamma0*y^2, loglik
    return 3 * (int64_t)r->g.N * r->g.L + r->g.N + 4;
}

}  // namespace hmmsort
