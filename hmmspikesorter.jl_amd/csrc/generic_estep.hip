// Time-parallel E-step for an ARBITRARY transition list (overlap models, reference types.jl:78-90):
// forward (baumwelch.jl:25-51), backward (:73-98) and the sufficient statistics of update (:205-309)
// without materialising alpha/beta/gamma for the whole signal.
//
// The signal is cut into the blocked engine's blocks of B samples (generic_blocked.hip).  One
// workgroup takes one block at a time: a forward sweep over [lo-H, hi) from a flat column (emissions
// only), keeping the scaled alpha of its OWN samples in a per-workgroup HBM window of S x B doubles,
// then a backward sweep from hi+H down to lo which, on the owned samples, turns alpha*beta into gamma
// and adds it to the block's statistics in registers.  Workspace = (resident workgroups) x S x B
// doubles, whatever T is; the reference's S x T arrays (28.8 GB each at S = 3600, T = 10^6) never exist.
//
// Arithmetic: the recursions run in the linear domain with one scale per column (alpha_hat = alpha /
// sum, beta likewise) instead of the reference's log-sum-exp per transition; gamma is normalised per
// sample, so the scales (and the unknown constant of a warmed-up block) cancel exactly as the `- g`
// of baumwelch.jl:222 cancels them.  Emissions are shifted by the largest exponent of the column
// (exp(e_j - max_j e_j)), so a sample far from every state mean does not underflow the column.
// Differences to the log-domain sweep: 1e-13 relative (tests: 1e-8 against the strict engine).
//
// Whether H samples of warm-up were enough is CERTIFIED per boundary (bes_check): the posterior
// gamma at the boundary sample must not move (L1 distance <= tol) when the neighbour's exact column
// is replaced by the warmed-up one.  Failures are counted in diag[3] (forward) / diag[5] (backward);
// hmmsort_em_step then retries with a longer warm-up and finally with the strict engine.
//
// Statistics (per state j, summed over t):  G0_j = sum gamma_j(t),  G1_j = sum gamma_j(t) y_t;
// X_i = sum_{t<T-1} xi_i(t) for the transitions i leaving the silent state (:229-261, normalised by the
// all-transition total, which equals the gamma normaliser);  Gamma0 = sum_{t<T-1} gamma_1(t);  sum y^2.
// sigma follows from  sum_t sum_j (y_t - m_j)^2 gamma_j(t) = sum y^2 - 2 sum_j m_j G1_j + sum_j m_j^2 G0_j.
#include <cmath>

#include "fastmath.h"
#include "generic_dev.h"
#include "hmmsort_internal.h"

namespace hmmsort {

namespace {

struct BesArgs {
    const double *y;
    int64_t T;
    int S, B, H, nblk, nsrc1;
    const double *mean;
    const int32_t *in_ptr, *in_src;
    const double *in_w;
    const int32_t *out_ptr, *out_dst;
    const double *out_w;
    double rden;
    double *win;    // [gridDim.x][B][S] scaled alpha of the owned samples
    double *rec;    // [nblk][6][S]  0 alpha warm (lo-1)  1 alpha exact (hi-1)  2 gamma (hi-1)
                    //               3 beta warm (hi)     4 beta exact (lo)     5 gamma (lo)
    double *partG;  // [nblk][2 S]
    double *partX;  // [nblk][nsrc1 + 2]: X_i | Gamma0 | sum y^2
};

__device__ __forceinline__ double wsum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wmax(double v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// red[par][slot][wave]: slot 0 column sum, 1 emission exponent maximum of the NEXT column, 2 gamma normaliser
constexpr int kRedW = 16;

// NTH = threads per workgroup (the launch bound decides the register budget: 512 threads leave 256 VGPRs per
// lane, which the per-state constants of 8 states per thread need; under a 1024-thread bound they spill 776 B)
template <int SPT, int NTH>
__global__ __launch_bounds__(NTH) void bes_block(BesArgs a)
{
    extern __shared__ double sh[];
    const int S = a.S, B = a.B, H = a.H, tid = threadIdx.x, nt = blockDim.x;
    const int wv = tid >> 6, nw = nt >> 6;
    double *col[2] = {sh, sh + S};
    double *red = sh + 2 * S;                      // [2][3][kRedW]
    double *xterm = red + 2 * 3 * kRedW;           // [2][nsrc1]
    double *xw = xterm + 2 * a.nsrc1, *xd = xw + a.nsrc1;   // the silent state's outgoing transitions: weight, destination
    for (int i = tid; i < a.nsrc1; i += nt) { xw[i] = a.out_w[i]; xd[i] = (double)a.out_dst[i]; }
    const int64_t T = a.T;
    // per-thread constants of its SPT states.  93 % of the states of an overlap model have ONE incoming and one
    // outgoing transition (the interior of the pair lattice): the first edge of each list lives in registers, the
    // rest of a list is read from the (L2-resident) CSR arrays -- one dependent global load per edge and step was
    // what bounded the first version (98 -> see DESIGN 3.1c)
    double m[SPT], en[SPT], wi0[SPT], wo0[SPT];
    int p0[SPT], p1[SPT], q0[SPT], q1[SPT], si0[SPT], do0[SPT];
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        const int j = tid + k * nt;
        const bool ok = j < S;
        m[k] = ok ? a.mean[j] : 0.0;
        p0[k] = ok ? a.in_ptr[j] : 0;  p1[k] = ok ? a.in_ptr[j + 1] : 0;
        q0[k] = ok ? a.out_ptr[j] : 0; q1[k] = ok ? a.out_ptr[j + 1] : 0;
        const bool hi = p1[k] > p0[k], ho = q1[k] > q0[k];
        si0[k] = hi ? a.in_src[p0[k]] : 0;  wi0[k] = hi ? a.in_w[p0[k]] : 0.0;
        do0[k] = ho ? a.out_dst[q0[k]] : 0; wo0[k] = ho ? a.out_w[q0[k]] : 0.0;
        p0[k] += hi; q0[k] += ho;             // the lists now start at their second edge
    }
    double *win = a.win + (size_t)blockIdx.x * B * S;

    auto reduce3 = [&](int par, double s, double mx, double z) {
        s = wsum(s); mx = wmax(mx); z = wsum(z);
        if ((tid & 63) == 0) {
            red[(par * 3 + 0) * kRedW + wv] = s;
            red[(par * 3 + 1) * kRedW + wv] = mx;
            red[(par * 3 + 2) * kRedW + wv] = z;
        }
    };
    auto read3 = [&](int par, double &s, double &mx, double &z) {
        s = 0.0; mx = -INFINITY; z = 0.0;
        for (int w = 0; w < nw; w++) {
            s += red[(par * 3 + 0) * kRedW + w];
            mx = fmax(mx, red[(par * 3 + 1) * kRedW + w]);
            z += red[(par * 3 + 2) * kRedW + w];
        }
    };

    for (int blk = blockIdx.x; blk < a.nblk; blk += gridDim.x) {
        const int64_t lo = (int64_t)blk * B, hi = (lo + B) < T ? (lo + B) : T;
        double *rec = a.rec + (size_t)blk * 6 * S;
        // ------------------------------------------------------------------ forward
        {
            const int64_t t0 = lo > H ? lo - H : 0;   // a warm-up reaching the start of the data is the exact sweep
            // column t0 = shifted emissions (baumwelch.jl:36 at the start of the data; a flat start elsewhere)
            const double y0 = a.y[t0];
            double pm = -INFINITY;
#pragma unroll
            for (int k = 0; k < SPT; k++) {
                const double d = y0 - m[k];
                en[k] = -(d * d) * a.rden;
                if (tid + k * nt < S) pm = fmax(pm, en[k]);
            }
            __syncthreads();                       // previous block is done with red / col
            reduce3(0, 0.0, pm, 0.0);
            __syncthreads();
            double s, emax, z;
            read3(0, s, emax, z);
            __syncthreads();
            const double y1 = a.y[(t0 + 1) < T ? (t0 + 1) : t0];
            double ps = 0.0;
            pm = -INFINITY;
#pragma unroll
            for (int k = 0; k < SPT; k++) {
                const int j = tid + k * nt;
                if (j < S) {
                    const double v = fexp(en[k] - emax);
                    col[0][j] = v;
                    ps += v;
                    if (t0 >= lo) win[j] = v;
                    const double d = y1 - m[k];
                    en[k] = -(d * d) * a.rden;
                    pm = fmax(pm, en[k]);
                }
            }
            reduce3(0, ps, pm, 0.0);
            int par = 0;
            double ynext = a.y[(t0 + 2) < T ? (t0 + 2) : T - 1];   // global loads run one step ahead of their use
            for (int64_t t = t0 + 1; t < hi; t++) {
                const double yn = ynext;                            // y[t + 1]
                ynext = a.y[(t + 2) < T ? (t + 2) : T - 1];
                __syncthreads();
                read3(par, s, emax, z);
                const double inv = 1.0 / s;
                const double *prev = col[par];
                double *cur = col[par ^ 1];
                ps = 0.0; pm = -INFINITY;
#pragma unroll
                for (int k = 0; k < SPT; k++) {
                    const int j = tid + k * nt;
                    if (j < S) {
                        double acc = prev[si0[k]] * wi0[k];
                        for (int e = p0[k]; e < p1[k]; e++) acc += prev[a.in_src[e]] * a.in_w[e];   // :47
                        const double v = (acc * inv) * fexp(en[k] - emax);
                        cur[j] = v;
                        ps += v;
                        if (t >= lo) win[(size_t)(t - lo) * S + j] = v;
                        if (t == lo - 1) rec[j] = v;
                        if (t == hi - 1) rec[S + j] = v;
                        const double d = yn - m[k];
                        en[k] = -(d * d) * a.rden;
                        pm = fmax(pm, en[k]);
                    }
                }
                par ^= 1;
                reduce3(par, ps, pm, 0.0);
            }
        }
        // ------------------------------------------------------------------ backward + statistics
        {
            const int64_t te = (hi - 1 + H) < (T - 1) ? (hi - 1 + H) : (T - 1);
            double g[SPT], G0[SPT], G1[SPT];
#pragma unroll
            for (int k = 0; k < SPT; k++) { g[k] = 0.0; G0[k] = 0.0; G1[k] = 0.0; }
            double X = 0.0, Gam0 = 0.0;            // X: thread i < nsrc1 owns transition i; Gam0: thread 0
            // emission exponents of column te and their maximum
            const double ye = a.y[te];
            double pm = -INFINITY;
#pragma unroll
            for (int k = 0; k < SPT; k++) {
                const double d = ye - m[k];
                en[k] = -(d * d) * a.rden;
                if (tid + k * nt < S) pm = fmax(pm, en[k]);
            }
            __syncthreads();
            reduce3(0, 0.0, pm, 0.0);
            __syncthreads();
            double s, emax, z;
            read3(0, s, emax, z);
            __syncthreads();
            // column te: beta = 1 (:80 at the end of the data; a flat start elsewhere).  nxt = b(te) * beta(te)
            double ps = 0.0, pz = 0.0;
            pm = -INFINITY;
            {
                const double yp = a.y[te > 0 ? te - 1 : 0];
#pragma unroll
                for (int k = 0; k < SPT; k++) {
                    const int j = tid + k * nt;
                    if (j < S) {
                        const double cur = 1.0;
                        ps += cur;
                        if (te < hi) {             // the last sample of the data is an owned sample
                            const double al = win[(size_t)(te - lo) * S + j];
                            g[k] = al * cur;
                            pz += g[k];
                            if (te == lo) rec[4 * S + j] = cur;
                        }
                        if (te == hi) rec[3 * S + j] = cur;
                        col[0][j] = cur * fexp(en[k] - emax);
                        const double d = yp - m[k];
                        en[k] = -(d * d) * a.rden;
                        pm = fmax(pm, en[k]);
                    }
                }
            }
            reduce3(0, ps, pm, pz);
            int par = 0;
            int64_t tprev = te;                    // time whose g[] / xterm wait for their normaliser
            // global loads one step ahead: y[t-1] for the next emissions, alpha(t) of the owned samples
            double y_cur = a.y[te], y_m1 = a.y[te > 0 ? te - 1 : 0], y_m2 = a.y[te > 1 ? te - 2 : 0];
            double alc[SPT];
#pragma unroll
            for (int k = 0; k < SPT; k++) {
                const int j = tid + k * nt;
                alc[k] = (j < S && te - 1 < hi && te - 1 >= lo) ? win[(size_t)(te - 1 - lo) * S + j] : 0.0;
            }
            for (int64_t t = te - 1; t >= lo; t--) {
                // y_cur = y[t+1], y_m1 = y[t], y_m2 = y[t-1]; alc = alpha(t)
                const double yv = y_cur, yp = y_m2;
                y_cur = y_m1; y_m1 = y_m2;
                y_m2 = a.y[t > 1 ? t - 2 : 0];
                double aln[SPT];
                const bool own_next = t - 1 < hi && t - 1 >= lo;
#pragma unroll
                for (int k = 0; k < SPT; k++) {
                    const int j = tid + k * nt;
                    aln[k] = (j < S && own_next) ? win[(size_t)(t - 1 - lo) * S + j] : 0.0;
                }
                __syncthreads();
                read3(par, s, emax, z);
                const double inv = 1.0 / s;
                // lagged statistics of time tprev = t + 1 (its normaliser z has just arrived)
                if (tprev < hi) {
                    const double rz = 1.0 / z;
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        const int j = tid + k * nt;
                        if (j < S) {
                            const double gm = g[k] * rz;
                            G0[k] += gm;
                            G1[k] += gm * yv;
                            if (tprev == hi - 1) rec[2 * S + j] = gm;
                        }
                    }
                    if (tprev <= T - 2) {
                        if (tid < a.nsrc1) X += xterm[par * a.nsrc1 + tid] * rz;
                        if (tid == 0) Gam0 += g[0] * rz;
                    }
                }
                const double *nxt = col[par];
                double *out = col[par ^ 1];
                const bool own = t < hi;
                ps = 0.0; pz = 0.0; pm = -INFINITY;
#pragma unroll
                for (int k = 0; k < SPT; k++) {
                    const int j = tid + k * nt;
                    if (j < S) {
                        double cur = nxt[do0[k]] * wo0[k];
                        for (int e = q0[k]; e < q1[k]; e++) cur += nxt[a.out_dst[e]] * a.out_w[e];   // :94
                        ps += cur;
                        if (own) {
                            const double al = alc[k];
                            g[k] = al * cur;
                            pz += g[k];
                            if (t == lo) rec[4 * S + j] = cur;
                            if (j == 0 && t <= T - 2)
                                for (int i = 0; i < a.nsrc1; i++)   // :240  alpha_1(t) a_1j b_j(t+1) beta_j(t+1)
                                    xterm[(par ^ 1) * a.nsrc1 + i] = al * (nxt[(int)xd[i]] * xw[i]);
                        }
                        if (t == hi) rec[3 * S + j] = cur;
                        out[j] = (cur * inv) * fexp(en[k] - emax);
                        const double d = yp - m[k];
                        en[k] = -(d * d) * a.rden;
                        pm = fmax(pm, en[k]);
                    }
                }
                par ^= 1;
                reduce3(par, ps, pm, pz);
                tprev = t;
#pragma unroll
                for (int k = 0; k < SPT; k++) alc[k] = aln[k];
            }
            // flush the statistics of the block's first sample
            __syncthreads();
            read3(par, s, emax, z);
            if (tprev < hi) {
                const double rz = 1.0 / z, yv = y_cur;              // y[tprev]
#pragma unroll
                for (int k = 0; k < SPT; k++) {
                    const int j = tid + k * nt;
                    if (j < S) {
                        const double gm = g[k] * rz;
                        G0[k] += gm;
                        G1[k] += gm * yv;
                        if (tprev == hi - 1) rec[2 * S + j] = gm;
                        if (tprev == lo) rec[5 * S + j] = gm;
                    }
                }
                if (tprev <= T - 2) {
                    if (tid < a.nsrc1) X += xterm[par * a.nsrc1 + tid] * rz;
                    if (tid == 0) Gam0 += g[0] * rz;
                }
            }
            double *pG = a.partG + (size_t)blk * 2 * S;
#pragma unroll
            for (int k = 0; k < SPT; k++) {
                const int j = tid + k * nt;
                if (j < S) { pG[j] = G0[k]; pG[S + j] = G1[k]; }
            }
            double *pX = a.partX + (size_t)blk * (a.nsrc1 + 2);
            if (tid < a.nsrc1) pX[tid] = X;
            if (tid == 0) pX[a.nsrc1] = Gam0;
            double y2 = 0.0;
            for (int64_t t = lo + tid; t < hi; t += nt) { const double v = a.y[t]; y2 += v * v; }
            __syncthreads();
            reduce3(0, y2, 0.0, 0.0);
            __syncthreads();
            read3(0, s, emax, z);
            if (tid == 0) pX[a.nsrc1 + 1] = s;
        }
    }
}

// Boundary certificates.  Boundary c (between blocks c and c+1, samples hi-1 | hi):
//   forward : gamma(hi-1) of block c, with its exact alpha(hi-1) replaced by block c+1's warmed-up one;
//   backward: gamma(hi) of block c+1, with its exact beta(hi) replaced by block c's warmed-up one.
// err = sum_j | gamma_j r_j / sum(gamma r) - gamma_j |, r = warm / exact.  diag: [0] fwd failures,
// [1] bwd failures, [2], [3] largest errors (double bit patterns).
__global__ __launch_bounds__(256) void bes_check(int S, int nblk, double tol, const double *__restrict__ rec,
                                                 unsigned long long *__restrict__ diag)
{
    __shared__ double red[8];
    const int c = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const double *ra = rec + (size_t)c * 6 * S, *rb = rec + (size_t)(c + 1) * 6 * S;
    const double *gam = dir == 0 ? ra + 2 * S : rb + 5 * S;
    const double *exact = dir == 0 ? ra + 1 * S : rb + 4 * S;
    const double *warm = dir == 0 ? rb + 0 * S : ra + 3 * S;
    double sw = 0.0;
    for (int j = tid; j < S; j += 256) {
        const double gm = gam[j];
        if (gm > 0.0) sw += gm * (warm[j] / exact[j]);
    }
    sw = wsum(sw);
    if ((tid & 63) == 0) red[tid >> 6] = sw;
    __syncthreads();
    sw = (red[0] + red[1]) + (red[2] + red[3]);
    double err = 0.0;
    for (int j = tid; j < S; j += 256) {
        const double gm = gam[j];
        if (gm > 0.0) err += fabs(gm * (warm[j] / exact[j]) / sw - gm);
    }
    err = wsum(err);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = err;
    __syncthreads();
    if (tid == 0) {
        err = (red[4] + red[5]) + (red[6] + red[7]);
        if (!(err <= tol)) atomicAdd(&diag[dir], 1ull);
        if (!(err == err)) err = INFINITY;
        atomicMax(&diag[2 + dir], (unsigned long long)__double_as_longlong(err));
    }
}

// stats = [G0 (S) | G1 (S) | X (nsrc1) | Gamma0 | sum y^2], summed over the blocks in block order
__global__ __launch_bounds__(64) void bes_reduce(int S, int nblk, int nsrc1, const double *__restrict__ partG,
                                                 const double *__restrict__ partX, double *__restrict__ stats)
{
    const int i = blockIdx.x, lane = threadIdx.x, n1 = 2 * S;
    double acc = 0.0;
    if (i < n1) for (int b = lane; b < nblk; b += 64) acc += partG[(size_t)b * n1 + i];
    else for (int b = lane; b < nblk; b += 64) acc += partX[(size_t)b * (nsrc1 + 2) + (i - n1)];
    acc = wsum(acc);
    if (lane == 0) stats[i] = acc;
}

// M-step finish from the statistics (baumwelch.jl:262-307): out = [mu (K x N) | sigma | xb[2:end] | pp]
__global__ __launch_bounds__(256) void bes_mstep(const int16_t *__restrict__ states, int N, int K, int S, int nsrc1,
                                                 const double *__restrict__ stats, const double *__restrict__ pp,
                                                 double *__restrict__ gg, double *__restrict__ mean_new,
                                                 double *__restrict__ out)
{
    __shared__ double red[12];
    const int tid = threadIdx.x, KN = K * N;
    const double *G0 = stats, *G1 = stats + S, *X = stats + 2 * S;
    double *mu = out;
    for (int i = tid; i < KN; i += 256) { mu[i] = 0.0; gg[i] = 0.0; }
    __syncthreads();
    if (tid == 0) {                                   // :266-287, states with exactly one active neuron
        for (int j = 0; j < S; j++) {
            int nact = 0;
            for (int l = 0; l < N; l++) nact += (states[l + N * j] >= 2);
            if (nact != 1) continue;
            for (int l = 0; l < N; l++) {
                const int ss = states[l + N * j];
                if (ss > 1) { mu[(ss - 1) + K * l] += G1[j]; gg[(ss - 1) + K * l] += G0[j]; }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < KN; i += 256)
        if (i % K != 0) mu[i] /= gg[i];
    __syncthreads();
    double x2 = 0.0, qq = 0.0;
    for (int j = tid; j < S; j += 256) {              // :288-305 with the NEW means
        double mj = 0.0;
        for (int l = 0; l < N; l++) mj += mu[(states[l + N * j] - 1) + K * l];
        mean_new[j] = mj;
        x2 += (mj * mj) * G0[j] - (2.0 * mj) * G1[j];
        qq += G0[j];
    }
    x2 = wsum(x2); qq = wsum(qq);
    if ((tid & 63) == 0) { red[tid >> 6] = x2; red[4 + (tid >> 6)] = qq; }
    __syncthreads();
    if (tid == 0) {
        const double X2 = ((red[0] + red[1]) + (red[2] + red[3])) + X[nsrc1 + 1];
        const double QQ = (red[4] + red[5]) + (red[6] + red[7]);
        out[KN] = sqrt(X2 / QQ);                      // :306-307
    }
    for (int i = 1 + tid; i < nsrc1; i += 256) out[KN + i] = log(X[i]) - log(X[nsrc1]);   // :264 xb[2:end]
    for (int j = tid; j < S; j += 256) out[KN + nsrc1 + j] = log(pp[j]);                  // :263
}

__global__ void bes_weights(const double *__restrict__ lp, int n, double *__restrict__ w)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = exp(lp[i]);
}

template <typename Tv>
int balloc(Tv **p, size_t n, int64_t *bytes)
{
    if (*p) return HMMSORT_OK;
    if (hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(Tv)) != hipSuccess) {
        (void)hipGetLastError();
        set_error("blocked E-step: hipMalloc of %.2f GB failed", (double)n * sizeof(Tv) / 1e9);
        return HMMSORT_ENOMEM;
    }
    *bytes += (int64_t)(n * sizeof(Tv));
    return HMMSORT_OK;
}

}  // namespace

bool blocked_estep_supported(const GenericDev *g)
{
    // two columns of S doubles + reduction scratch in LDS; 16 states per thread at 1024 threads
    return g->blocked && g->nsrc1 >= 1 && g->nsrc1 <= 256 && g->T >= 2 &&
           (2 * (size_t)g->S + 2 * 3 * kRedW + 4 * (size_t)g->nsrc1) * 8 <= 156 * 1024 && g->S <= 16 * 1024;
}

int64_t blocked_stats_len(const GenericDev *g) { return 2 * g->S + g->nsrc1 + 2; }

int blocked_estep(GenericDev *g, const double *d_y, double *d_stats, hipStream_t st)
{
    HS_CHECK(blocked_estep_supported(g), HMMSORT_EUNSUP, "blocked E-step: model too large for the LDS columns");
    const size_t S = (size_t)g->S, nb = (size_t)g->nblk;
    const int nt = g->S <= 256 ? 256 : (g->S <= 4096 ? 512 : 1024);
    const int spt = (int)((S + nt - 1) / nt);
    int dev = 0, ncu = 256;
    HS_HIP(hipGetDevice(&dev));
    HS_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const size_t lds = (2 * S + 2 * 3 * kRedW + 4 * (size_t)g->nsrc1) * sizeof(double);
    const int per_cu = lds <= 76 * 1024 && nt <= 512 ? 2 : 1;
    const int grid = (int)std::min<size_t>(nb, (size_t)ncu * per_cu);
    int rc;
    if ((rc = balloc(&g->d_es_win, (size_t)grid * g->B * S, &g->bytes)) ||
        (rc = balloc(&g->d_es_rec, nb * 6 * S, &g->bytes)) ||
        (rc = balloc(&g->d_es_partG, nb * 2 * S, &g->bytes)) ||
        (rc = balloc(&g->d_es_partX, nb * (g->nsrc1 + 2), &g->bytes)) ||
        (rc = balloc(&g->d_es_inw, (size_t)g->R, &g->bytes)) || (rc = balloc(&g->d_es_outw, (size_t)g->R, &g->bytes)) ||
        (rc = balloc(&g->d_es_diag, 4, &g->bytes)) || (rc = balloc(&g->d_es_tmp, 2 * S + g->K * g->N, &g->bytes)))
        return rc;
    g->es_grid = grid;
    hipLaunchKernelGGL(bes_weights, dim3((unsigned)((g->R + 255) / 256)), dim3(256), 0, st, g->d_in_lp, (int)g->R, g->d_es_inw);
    hipLaunchKernelGGL(bes_weights, dim3((unsigned)((g->R + 255) / 256)), dim3(256), 0, st, g->d_out_lp, (int)g->R, g->d_es_outw);
    HS_HIP(hipMemsetAsync(g->d_es_diag, 0, 4 * sizeof(unsigned long long), st));
    HS_HIP(hipMemsetAsync(g->d_es_rec, 0, nb * 6 * S * sizeof(double), st));
    BesArgs a;
    a.y = d_y; a.T = g->T; a.S = (int)g->S; a.B = (int)g->B; a.H = (int)g->H; a.nblk = (int)g->nblk;
    a.nsrc1 = g->nsrc1; a.mean = g->d_mean;
    a.in_ptr = g->d_in_ptr; a.in_src = g->d_in_src; a.in_w = g->d_es_inw;
    a.out_ptr = g->d_out_ptr; a.out_dst = g->d_out_dst; a.out_w = g->d_es_outw;
    a.rden = 1.0 / (2.0 * (g->sigma * g->sigma));
    a.win = g->d_es_win; a.rec = g->d_es_rec; a.partG = g->d_es_partG; a.partX = g->d_es_partX;
    auto launch = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            HS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, st, a);
        return HMMSORT_OK;
    };
    // register budget by launch bound: 1-2 states per thread fit 128 VGPRs (bound 1024: more waves per SIMD,
    // S = 900: 15.5 against 22 ms per 10^6 samples); 4-8 states per thread need the 256 of a 512-thread bound
    // (S = 3600: 54 against 96 ms)
    if (nt <= 512) rc = spt <= 1 ? launch(bes_block<1, 1024>) : spt <= 2 ? launch(bes_block<2, 1024>)
                        : spt <= 4 ? launch(bes_block<4, 512>) : launch(bes_block<8, 512>);
    else rc = spt <= 8 ? launch(bes_block<8, 1024>) : launch(bes_block<16, 1024>);
    if (rc) return rc;
    HS_HIP(hipGetLastError());
    if (nb > 1)
        hipLaunchKernelGGL(bes_check, dim3((unsigned)(nb - 1), 2), dim3(256), 0, st, (int)S, (int)nb, 1e-9, g->d_es_rec,
                           g->d_es_diag);
    hipLaunchKernelGGL(bes_reduce, dim3((unsigned)blocked_stats_len(g)), dim3(64), 0, st, (int)S, (int)nb, g->nsrc1,
                       g->d_es_partG, g->d_es_partX, d_stats);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int blocked_mstep(GenericDev *g, const double *d_stats, double *d_out, hipStream_t st)
{
    HS_CHECK(g->d_es_rec, HMMSORT_EINVAL, "blocked M-step: no E-step has run on this plan");
    // pp = gamma[:,1] (:263): the first block's posterior on the first sample
    hipLaunchKernelGGL(bes_mstep, dim3(1), dim3(256), 0, st, g->d_states, (int)g->N, (int)g->K, (int)g->S, g->nsrc1,
                       d_stats, g->d_es_rec + 5 * (size_t)g->S, g->d_es_tmp + 2 * g->S, g->d_es_tmp, d_out);
    HS_HIP(hipGetLastError());
    return HMMSORT_OK;
}

int blocked_estep_diagnostics(GenericDev *g, hipStream_t st, int64_t diag[8])
{
    if (!g->d_es_diag) return HMMSORT_OK;
    unsigned long long h[4];
    HS_HIP(hipMemcpyAsync(h, g->d_es_diag, sizeof(h), hipMemcpyDeviceToHost, st));
    HS_HIP(hipStreamSynchronize(st));
    diag[3] = (int64_t)h[0];
    diag[5] = (int64_t)h[1];
    diag[4] = (int64_t)h[2];
    diag[6] = (int64_t)h[3];
    return HMMSORT_OK;
}

void blocked_estep_destroy(GenericDev *g)
{
    void *ptrs[] = {g->d_es_win, g->d_es_rec, g->d_es_partG, g->d_es_partX, g->d_es_inw, g->d_es_outw,
                    g->d_es_diag, g->d_es_tmp};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
}

}  // namespace hmmsort
