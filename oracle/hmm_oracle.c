/*
 * hmm_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the HMM hot path of
 * grero/HMMSpikeSorter.jl (reference @ /root/reference, v0.2.0).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (hmmspikesorter.jl_amd/, libhmmsort_hip.so) never does.
 *
 * PARITY PIN STATUS: the reference is Julia and no Julia runtime exists in the build
 * container, so the reference could not be executed.  What pins this restatement:
 *   - exact known answers held by the reference's own tests: state enumeration
 *     ("Unroll" test/runtests.jl:36-42) and the template numerics constant
 *     (test/runtests.jl:55);
 *   - the reference's statistical windows (test/runtests.jl:31-34, :85-94).
 * The reference holds NO golden alpha/beta/delta/path vectors, so for those quantities
 * this file is a line-by-line restatement and nothing more: "parity unpinned" for
 * alpha/beta/gamma/mu/sigma/path numerics.
 *
 * Every function cites the reference file:line it follows.  Arithmetic is IEEE fp64,
 * compiled with -O2 -ffp-contract=off (Julia does not contract a*b+c into fma).
 * Indices in this file are 0-based; state ids stored in arrays are 1-based exactly as the
 * reference stores them (states = generate_states() .+ 1, types.jl:150).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* utils.jl:1   const log2pi = 0.5*log(2*pi) */
static double log2pi_(void) { return 0.5 * log(2.0 * 3.141592653589793); }

/* utils.jl:3   funcl(x, mu, sigma) = (s2 = s*s; dd=x-mu; -log2pi-log(s) - dd*dd/(2*s2)) */
double hmm_oracle_funcl3(double x, double mu, double sigma)
{
    double s2 = sigma * sigma;
    double dd = x - mu;
    return (-log2pi_() - log(sigma)) - (dd * dd) / (2.0 * s2);
}

/* utils.jl:4   4-argument form with precomputed log(sigma) */
double hmm_oracle_funcl4(double x, double mu, double sigma, double lsigma)
{
    double s2 = sigma * sigma;
    double dd = x - mu;
    return (-log2pi_() - lsigma) - (dd * dd) / (2.0 * s2);
}

/* utils.jl:24-32 */
double hmm_oracle_logsumexpl(double xp, double yp)
{
    double z;
    if (xp > yp)
        z = xp + log1p(exp(yp - xp));
    else
        z = yp + log1p(exp(xp - yp));
    return z;
}

/* types.jl:65-92  generate_states(N,K,allow_overlaps); values are 0-based neuron phases.
 * states is N x S column-major.  Returns S.  Pass states==NULL to query S only. */
int64_t hmm_oracle_generate_states(int64_t N, int64_t K, int allow_overlaps, int16_t *states)
{
    int64_t S = 1 + N * (K - 1);
    if (allow_overlaps)
        S += (N * (N - 1) * (K - 1) * (K - 1)) / 2;
    if (!states)
        return S;
    memset(states, 0, sizeof(int16_t) * (size_t)(N * S));
    int64_t k = 1; /* 0-based column; reference starts at k = 2 (1-based) */
    for (int64_t i = 0; i < N; i++)
        for (int64_t k1 = 1; k1 <= K - 1; k1++) {
            states[i + N * k] = (int16_t)k1;
            k++;
        }
    if (allow_overlaps)
        for (int64_t i = 0; i < N - 1; i++)
            for (int64_t j = i + 1; j < N; j++)
                for (int64_t k1 = 1; k1 <= K - 1; k1++)
                    for (int64_t k2 = 1; k2 <= K - 1; k2++) {
                        states[i + N * k] = (int16_t)k1;
                        states[j + N * k] = (int16_t)k2;
                        k++;
                    }
    return S;
}

/* types.jl:94-113  isvalid_transition(states,K,lp,j1,j2); states hold 0-based phases.
 * sum(lp) is Julia's sum(): sequential left fold for length < 16 (Base reduce.jl). */
double hmm_oracle_isvalid_transition(const int16_t *states, int64_t N, int64_t K,
                                     const double *lp, int64_t nlp, int64_t j1, int64_t j2)
{
    double lpt = 0.0;
    double slp = 0.0;
    if (nlp > 0) {
        slp = lp[0];
        for (int64_t i = 1; i < nlp; i++)
            slp = slp + lp[i];
    }
    double lpz = log1p(-exp(slp));
    for (int64_t i = 0; i < N; i++) {
        int s1 = states[i + N * j1];
        int s2 = states[i + N * j2];
        double lpi = lp[i];
        if (s1 == 0 && s2 == 0)
            lpt += lpz;
        else if (s1 == 0 && s2 == 1)
            lpt += lpi;
        else if ((s2 - s1 == 1) || (s1 == K - 1 && s2 == 0))
            lpt += 0.0;
        else {
            lpt = -INFINITY;
            break;
        }
    }
    return lpt;
}

/* types.jl:115-127  get_valid_transitions: all-pairs scan, source-major, dest ascending.
 * src/dst are written 1-based like the reference tuples.  Returns R; pass NULLs to count. */
int64_t hmm_oracle_get_valid_transitions(const int16_t *states, int64_t N, int64_t S, int64_t K,
                                         const double *lp, int64_t nlp,
                                         int64_t *src, int64_t *dst, double *val, int64_t cap)
{
    int64_t r = 0;
    for (int64_t i = 0; i < S; i++)
        for (int64_t j = 0; j < S; j++) {
            double aa = hmm_oracle_isvalid_transition(states, N, K, lp, nlp, i, j);
            if (isfinite(aa)) {
                if (src && r < cap) {
                    src[r] = i + 1;
                    dst[r] = j + 1;
                    val[r] = aa;
                }
                r++;
            }
        }
    return r;
}

/* per-state mean  _mu[j] = sum_l mu[states[l,j], l]  accumulated from 0.0 in neuron order
 * (baumwelch.jl:29-35,81-86,210-215; viterbi.jl:57-60,68-71).  states are 1-based rows. */
static void state_means(const int16_t *states1, int64_t N, int64_t S, const double *mu, int64_t K,
                        double *m)
{
    for (int64_t j = 0; j < S; j++) {
        double a = 0.0;
        for (int64_t l = 0; l < N; l++)
            a += mu[(states1[l + N * j] - 1) + K * l];
        m[j] = a;
    }
}

/* baumwelch.jl:25-51  forward(V, lA, mu, sigma) -> a (S x T col-major, log domain, unscaled) */
int hmm_oracle_forward(const double *V, int64_t T, const int16_t *states1, int64_t N, int64_t K,
                       int64_t S, const int64_t *src, const int64_t *dst, const double *val,
                       int64_t R, const double *mu, double sigma, double *a)
{
    double *m = (double *)malloc(sizeof(double) * (size_t)S);
    if (!m) return -1;
    state_means(states1, N, S, mu, K, m);
    for (int64_t i = 0; i < S * T; i++) a[i] = -INFINITY;
    for (int64_t i = 0; i < S; i++)
        a[i] = hmm_oracle_funcl3(V[0], m[i], sigma); /* :36 (pi written :31 then overwritten) */
    for (int64_t t = 1; t < T; t++) {
        double v = V[t];
        double *at = a + S * t;
        const double *ap = a + S * (t - 1);
        for (int64_t q = 0; q < R; q++) {
            int64_t k = src[q] - 1, j = dst[q] - 1;
            double lp = val[q];
            double b = hmm_oracle_funcl3(v, m[j], sigma);
            at[j] = hmm_oracle_logsumexpl(at[j], (ap[k] + lp) + b); /* :47 */
        }
    }
    free(m);
    return 0;
}

/* baumwelch.jl:73-98  backward(V, lA, mu, sigma) */
int hmm_oracle_backward(const double *V, int64_t T, const int16_t *states1, int64_t N, int64_t K,
                        int64_t S, const int64_t *src, const int64_t *dst, const double *val,
                        int64_t R, const double *mu, double sigma, double *a)
{
    double *m = (double *)malloc(sizeof(double) * (size_t)S);
    if (!m) return -1;
    state_means(states1, N, S, mu, K, m);
    for (int64_t i = 0; i < S * T; i++) a[i] = -INFINITY;
    for (int64_t i = 0; i < S; i++) a[i + S * (T - 1)] = 0.0; /* :80 */
    for (int64_t t = T - 2; t >= 0; t--) {
        double v = V[t + 1];
        double *at = a + S * t;
        const double *an = a + S * (t + 1);
        for (int64_t q = 0; q < R; q++) {
            int64_t j = src[q] - 1, k = dst[q] - 1;
            double lp = val[q];
            double b = hmm_oracle_funcl3(v, m[k], sigma);
            at[j] = hmm_oracle_logsumexpl(at[j], (an[k] + lp) + b); /* :94 */
        }
    }
    free(m);
    return 0;
}

/* baumwelch.jl:205-309  update(alpha, beta, lA, mu, sigma, x)
 * Outputs: mu (K x N, zeroed and rewritten IN PLACE as the reference does :268),
 *          *sigma_out, xb_tail = xb[2:end] (length ntidx-1; the new lp),
 *          pp = gammaf[:,1] (length S), *ntidx_out = number of transitions with src==1.
 * The StateMatrix rebuild (:265) is the caller's job (types.jl:148). */
int hmm_oracle_update(const double *alpha, const double *beta, int64_t T, const int16_t *states1,
                      int64_t N, int64_t K, int64_t S, const int64_t *src, const int64_t *dst,
                      const double *val, int64_t R, double *mu, double sigma, const double *x,
                      double *sigma_out, double *xb_tail, int64_t xb_cap, double *pp,
                      int64_t *ntidx_out)
{
    int rc = 0;
    double *gam = (double *)malloc(sizeof(double) * (size_t)(S * T));
    double *m = (double *)calloc((size_t)S, sizeof(double));
    int64_t *tidx = (int64_t *)malloc(sizeof(int64_t) * (size_t)R);
    double *xi = NULL, *xx = NULL, *gg = NULL;
    if (!gam || !m || !tidx) { rc = -1; goto done; }
    state_means(states1, N, S, mu, K, m); /* :210-215 */
    for (int64_t t = 0; t < T; t++) {     /* :216-224 */
        double g = -INFINITY;
        for (int64_t j = 0; j < S; j++)
            g = hmm_oracle_logsumexpl(g, alpha[j + S * t] + beta[j + S * t]);
        for (int64_t j = 0; j < S; j++)
            gam[j + S * t] = (alpha[j + S * t] + beta[j + S * t]) - g;
    }
    int64_t nt = 0; /* :226 */
    for (int64_t q = 0; q < R; q++)
        if (src[q] == 1) tidx[nt++] = q;
    *ntidx_out = nt;
    xi = (double *)calloc((size_t)(nt * (T > 1 ? T - 1 : 1)), sizeof(double));
    xx = (double *)malloc(sizeof(double) * (size_t)(nt > 0 ? nt : 1));
    gg = (double *)calloc((size_t)(K * N), sizeof(double));
    if (!xi || !xx || !gg) { rc = -1; goto done; }
    for (int64_t t = 0; t < T - 1; t++) { /* :229-253 */
        double _x = x[t + 1];
        for (int64_t i = 0; i < nt; i++) {
            int64_t j = dst[tidx[i]] - 1;
            double lp = val[tidx[i]];
            double bb = hmm_oracle_funcl3(_x, m[j], sigma);
            xi[i + nt * t] = ((alpha[0 + S * t] + lp) + beta[j + S * (t + 1)]) + bb; /* :240 */
        }
        double q = -INFINITY;
        for (int64_t r = 0; r < R; r++) {
            int64_t i = src[r] - 1, j = dst[r] - 1;
            double lp = val[r];
            double bb = hmm_oracle_funcl3(_x, m[j], sigma);
            q = hmm_oracle_logsumexpl(q, ((alpha[i + S * t] + lp) + beta[j + S * (t + 1)]) + bb);
        }
        for (int64_t i = 0; i < nt; i++) xi[i + nt * t] -= q;
    }
    double bbs = -INFINITY; /* :254-261 */
    for (int64_t i = 0; i < nt; i++) xx[i] = -INFINITY;
    for (int64_t t = 0; t < T - 1; t++) {
        bbs = hmm_oracle_logsumexpl(bbs, gam[0 + S * t]);
        for (int64_t j = 0; j < nt; j++)
            xx[j] = hmm_oracle_logsumexpl(xx[j], xi[j + nt * t]);
    }
    for (int64_t j = 0; j < S; j++) pp[j] = gam[j]; /* :263 */
    for (int64_t j = 1; j < nt; j++)                /* :264-265 xb[2:end] */
        if (j - 1 < xb_cap) xb_tail[j - 1] = xx[j] - bbs;
    /* :266-287 mean update; mu zeroed in place */
    for (int64_t i = 0; i < K * N; i++) mu[i] = 0.0;
    /* :269 sidx = states with exactly one active neuron (tidx[] is free for reuse: S <= R) */
    for (int64_t j = 0; j < S && j < R; j++) {
        int nact = 0;
        for (int64_t l = 0; l < N; l++) nact += (states1[l + N * j] >= 2);
        tidx[j] = (nact == 1);
    }
    for (int64_t t = 0; t < T; t++) {
        double _x = x[t];
        for (int64_t j = 0; j < S; j++) {
            if (!tidx[j]) continue;
            double eg = exp(gam[j + S * t]);
            for (int64_t l = 0; l < N; l++) {
                int ss = states1[l + N * j];
                if (ss > 1) {
                    mu[(ss - 1) + K * l] += _x * eg;
                    gg[(ss - 1) + K * l] += eg;
                }
            }
        }
    }
    for (int64_t l = 0; l < N; l++)
        for (int64_t j = 1; j < K; j++) mu[j + K * l] /= gg[j + K * l];
    state_means(states1, N, S, mu, K, m); /* :288-293 with the NEW mu */
    double x2 = 0.0, qq = 0.0;            /* :295-305 */
    for (int64_t t = 0; t < T; t++)
        for (int64_t j = 0; j < S; j++) {
            double _x = x[t];
            double eg = exp(gam[j + S * t]);
            double d = _x - m[j];
            x2 += (d * d) * eg;
            qq += eg;
        }
    *sigma_out = sqrt(x2 / qq); /* :306-307 */
done:
    free(gam); free(m); free(tidx); free(xi); free(xx); free(gg);
    return rc;
}

/* viterbi.jl:44-98  viterbi(y, lA, mu, sigma) -> (x::Vector{Int16} 1-based, ll)
 * lean != 0: T1 is kept as two columns only and ll is re-accumulated along the decoded path
 * with the identical op order ((T1[k]+lp)+q) -- same values bit for bit, 8*S*T bytes less.
 * T1_out (S x T) may be NULL; only filled when lean == 0. */
int hmm_oracle_viterbi(const double *y, int64_t T, const int16_t *states1, int64_t N, int64_t K,
                       int64_t S, const int64_t *src, const int64_t *dst, const double *val,
                       int64_t R, const double *mu, double sigma, int16_t *x, double *ll_out,
                       int lean, double *T1_out)
{
    double lsig = log(sigma); /* :47 */
    double *m = (double *)malloc(sizeof(double) * (size_t)S);
    double *q = (double *)malloc(sizeof(double) * (size_t)S);
    int16_t *T2 = (int16_t *)malloc(sizeof(int16_t) * (size_t)(S * T));
    double *T1 = NULL;
    int rc = 0;
    if (!m || !q || !T2) { rc = -1; goto done; }
    T1 = (double *)malloc(sizeof(double) * (size_t)(lean ? 2 * S : S * T));
    if (!T1) { rc = -1; goto done; }
    state_means(states1, N, S, mu, K, m); /* same value as the per-sample recomputation :68-71 */
    for (int64_t i = 0; i < S * T; i++) T2[i] = 1;       /* :53 ones(Int16,...) */
    for (int64_t i = 0; i < S; i++)                      /* :55-62 */
        T1[i] = hmm_oracle_funcl4(y[0], m[i], sigma, lsig);
    T1[0] = 0.0;                                         /* :63 */
    for (int64_t t = 1; t < T; t++) {                    /* :65-88 */
        double yi = y[t];
        double *cur = lean ? T1 + S * (t & 1) : T1 + S * t;
        const double *prv = lean ? T1 + S * ((t - 1) & 1) : T1 + S * (t - 1);
        int16_t *psi = T2 + S * t;
        for (int64_t j = 0; j < S; j++) {
            q[j] = hmm_oracle_funcl4(yi, m[j], sigma, lsig);
            cur[j] = -INFINITY; /* :52 fill(-Inf) */
        }
        for (int64_t r = 0; r < R; r++) {
            int64_t k = src[r] - 1, j = dst[r] - 1;
            double tt = prv[k] + val[r];
            if (tt > cur[j]) {
                cur[j] = tt;
                psi[j] = (int16_t)(k + 1);
            }
        }
        for (int64_t j = 0; j < S; j++) cur[j] += q[j];
    }
    { /* :90 argmax = first maximal index */
        const double *last = lean ? T1 + S * ((T - 1) & 1) : T1 + S * (T - 1);
        int64_t best = 0;
        for (int64_t j = 1; j < S; j++)
            if (last[j] > last[best]) best = j;
        x[T - 1] = (int16_t)(best + 1);
    }
    for (int64_t i = T - 1; i >= 1; i--) /* :93-94 */
        x[i - 1] = T2[(x[i] - 1) + S * i];
    if (!lean) {
        double ll = 0.0; /* :92-96 summed from i = nobs down to 2 */
        for (int64_t i = T - 1; i >= 1; i--) ll += T1[(x[i] - 1) + S * i];
        *ll_out = ll;
        if (T1_out) memcpy(T1_out, T1, sizeof(double) * (size_t)(S * T));
    } else {
        /* values along the path, forward, with the reference's op order; then the same
         * descending-order sum */
        double *pv = (double *)malloc(sizeof(double) * (size_t)T);
        if (!pv) { rc = -1; goto done; }
        pv[0] = (x[0] == 1) ? 0.0 : hmm_oracle_funcl4(y[0], m[x[0] - 1], sigma, lsig);
        for (int64_t t = 1; t < T; t++) {
            double lp = -INFINITY;
            for (int64_t r = 0; r < R; r++) /* small R; only used by the lean path */
                if (src[r] == x[t - 1] && dst[r] == x[t]) { lp = val[r]; break; }
            pv[t] = (pv[t - 1] + lp) + hmm_oracle_funcl4(y[t], m[x[t] - 1], sigma, lsig);
        }
        double ll = 0.0;
        for (int64_t i = T - 1; i >= 1; i--) ll += pv[i];
        *ll_out = ll;
        free(pv);
    }
done:
    free(m); free(q); free(T2); free(T1);
    return rc;
}

/* reconstruction.jl:1-10 */
void hmm_oracle_reconstruct(const int16_t *x, int64_t T, const int16_t *states1, int64_t N,
                            int64_t S, const double *mu, int64_t K, double *Y2)
{
    (void)S;
    for (int64_t i = 0; i < T; i++) {
        double a = 0.0;
        for (int64_t j = 0; j < N; j++)
            a += mu[(states1[j + N * (x[i] - 1)] - 1) + K * j];
        Y2[i] = a;
    }
}

/* extraction.jl:4-13  unroll_mlseq -> N x T col-major */
void hmm_oracle_unroll_mlseq(const int16_t *mlseq, int64_t T, const int16_t *states1, int64_t N,
                             int16_t *out)
{
    for (int64_t i = 0; i < T; i++)
        for (int64_t j = 0; j < N; j++)
            out[j + N * i] = states1[j + N * (mlseq[i] - 1)];
}

/* fit.jl:11-42  chunked decode + stitch (with the undefined gc() call :19 removed).
 * ml_seq is initialised to ones (:15).  Returns total ll (:37). */
int hmm_oracle_fit_chunked(const double *X, int64_t n, int64_t chunksize, const int16_t *states1,
                           int64_t N, int64_t K, int64_t S, const int64_t *src,
                           const int64_t *dst, const double *val, int64_t R, const double *mu,
                           double sigma, int16_t *ml_seq, double *ll_out)
{
    int64_t i = 1, j = 1; /* 1-based like the reference */
    double ll = 0.0;
    for (int64_t t = 0; t < n; t++) ml_seq[t] = 1;
    int16_t *x = (int16_t *)malloc(sizeof(int16_t) * (size_t)(chunksize > 0 ? chunksize : 1));
    if (!x) return -1;
    while (j < n) {
        j = (i + chunksize - 1 < n) ? i + chunksize - 1 : n;
        int64_t k = j - i + 1;
        int64_t l = 1;
        double _ll = 0.0;
        int rc = hmm_oracle_viterbi(X + (i - 1), k, states1, N, K, S, src, dst, val, R, mu, sigma,
                                    x, &_ll, 1, NULL);
        if (rc) { free(x); return rc; }
        if (i > 1) {
            while (l <= k && x[l - 1] > 1) l++;
            if (l > k) { free(x); *ll_out = ll; return -3; } /* reference: BoundsError at x[l] */
        }
        if (j < n)
            while (k >= 1 && x[k - 1] > 1) { j--; k--; }
        for (int64_t u = l; u <= k; u++) ml_seq[(i + u - 1) - 1] = x[u - 1];
        ll += _ll;
        if (j <= i) { /* the reference would loop forever here (no silent sample in the chunk) */
            free(x);
            *ll_out = ll;
            return -2;
        }
        i = j;
    }
    free(x);
    *ll_out = ll;
    return 0;
}

/* extraction.jl:15-24  extract_spiketimes: for neuron i the spike time is every sample whose
 * decoded state has neuron i at the row of its template minimum (indmin = first minimum).
 * times is N x cap (row per neuron), 1-based sample indices ascending; counts[i] = total found. */
void hmm_oracle_extract_spiketimes(const int16_t *ml_seq, int64_t T, const int16_t *states1,
                                   int64_t N, int64_t S, const double *mu, int64_t K,
                                   int64_t *times, int64_t cap, int64_t *counts)
{
    (void)S;
    for (int64_t i = 0; i < N; i++) {
        int64_t q = 0;
        for (int64_t k = 1; k < K; k++)
            if (mu[k + K * i] < mu[q + K * i]) q = k;
        int64_t n = 0;
        for (int64_t t = 0; t < T; t++)
            if (states1[i + N * (ml_seq[t] - 1)] == q + 1) {
                if (n < cap) times[i * cap + n] = t + 1;
                n++;
            }
        counts[i] = n;
    }
}
