// fp64 exp / log for the junction log-sum-exp updates.  ROCm's device-library log() is a
// ~1-ulp double-double routine (about 250 VALU instructions; measured in the forward chain's ISA);
// these are plain polynomial versions (exp: 13-term Taylor after ln2 range reduction, < 1.5 ulp;
// log: atanh series after frexp, < 2 ulp), i.e. relative error ~3e-16 -- eight orders of
// magnitude inside the 1e-6 bar on mu/sigma.  The Viterbi path never uses them (no
// transcendentals there).  fma() is used explicitly: the translation unit is compiled with
// -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace hmmsort {

// exp(x) for any x <= ~700 (callers pass x <= 0 after max-subtraction); exp(-inf) = 0
__device__ __forceinline__ double fexp(double x)
{
    x = fmax(x, -746.0);
    const double n = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                     // 1/13!
    p = __builtin_fma(p, r, 2.0876756987868099e-09);       // 1/12!
    p = __builtin_fma(p, r, 2.5052108385441719e-08);       // 1/11!
    p = __builtin_fma(p, r, 2.7557319223985891e-07);       // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06);       // 1/9!
    p = __builtin_fma(p, r, 2.4801587301587302e-05);       // 1/8!
    p = __builtin_fma(p, r, 1.9841269841269841e-04);       // 1/7!
    p = __builtin_fma(p, r, 1.3888888888888889e-03);       // 1/6!
    p = __builtin_fma(p, r, 8.3333333333333332e-03);       // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);       // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01);       // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// log(x) for finite x >= 0 (log(0) = -inf)
__device__ __forceinline__ double flog(double x)
{
    int e;
    double m = frexp(x, &e);                               // m in [0.5, 1)
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m * 2.0 : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;                              // [-0.2929, 0.4142]
    const double s = f / (2.0 + f);                        // |s| <= 0.1716
    const double z = s * s;
    double p = 4.3478260869565216e-02;                     // 1/23
    p = __builtin_fma(p, z, 4.7619047619047616e-02);       // 1/21
    p = __builtin_fma(p, z, 5.2631578947368418e-02);       // 1/19
    p = __builtin_fma(p, z, 5.8823529411764705e-02);       // 1/17
    p = __builtin_fma(p, z, 6.6666666666666666e-02);       // 1/15
    p = __builtin_fma(p, z, 7.6923076923076927e-02);       // 1/13
    p = __builtin_fma(p, z, 9.0909090909090912e-02);       // 1/11
    p = __builtin_fma(p, z, 1.1111111111111110e-01);       // 1/9
    p = __builtin_fma(p, z, 1.4285714285714285e-01);       // 1/7
    p = __builtin_fma(p, z, 2.0000000000000001e-01);       // 1/5
    p = __builtin_fma(p, z, 3.3333333333333331e-01);       // 1/3
    const double ed = (double)e;
    const double s2 = s + s;
    // log(x) = e*ln2 + 2s + 2s*z*p
    double r = __builtin_fma(s2 * z, p, ed * 1.90821492927058770002e-10);
    r = r + s2;
    r = __builtin_fma(ed, 6.93147180369123816490e-01, r);
    return x > 0.0 ? r : -INFINITY;
}

// M independent exponentials / logarithms at once, coefficient-outer: consecutive instructions
// belong to different values, so the dependent Horner chains overlap instead of each waiting
// ~10 cycles for its predecessor (a single wave per SIMD has nothing else to issue).
template <int M>
__device__ __forceinline__ void fexp_n(double (&x)[M])
{
    double n[M], r[M], p[M];
#pragma unroll
    for (int i = 0; i < M; i++) {
        x[i] = fmax(x[i], -746.0);
        n[i] = __builtin_rint(x[i] * 1.4426950408889634074);
    }
#pragma unroll
    for (int i = 0; i < M; i++) r[i] = __builtin_fma(n[i], -6.93147180369123816490e-01, x[i]);
#pragma unroll
    for (int i = 0; i < M; i++) r[i] = __builtin_fma(n[i], -1.90821492927058770002e-10, r[i]);
    constexpr double c[13] = {2.0876756987868099e-09, 2.5052108385441719e-08, 2.7557319223985891e-07,
                              2.7557319223985893e-06, 2.4801587301587302e-05, 1.9841269841269841e-04,
                              1.3888888888888889e-03, 8.3333333333333332e-03, 4.1666666666666664e-02,
                              1.6666666666666666e-01, 0.5, 1.0, 1.0};
#pragma unroll
    for (int i = 0; i < M; i++) p[i] = 1.6059043836821613e-10;
#pragma unroll
    for (int k = 0; k < 13; k++)
#pragma unroll
        for (int i = 0; i < M; i++) p[i] = __builtin_fma(p[i], r[i], c[k]);
#pragma unroll
    for (int i = 0; i < M; i++) x[i] = ldexp(p[i], (int)n[i]);
}

template <int M>
__device__ __forceinline__ void flog_n(double (&x)[M])
{
    double s[M], z[M], p[M], ed[M];
#pragma unroll
    for (int i = 0; i < M; i++) {
        int e;
        double m = frexp(x[i], &e);
        const bool lo = m < 0.70710678118654752440;
        m = lo ? m * 2.0 : m;
        e = lo ? e - 1 : e;
        ed[i] = (double)e;
        const double f = m - 1.0;
        s[i] = f / (2.0 + f);
    }
#pragma unroll
    for (int i = 0; i < M; i++) z[i] = s[i] * s[i];
    constexpr double c[10] = {4.7619047619047616e-02, 5.2631578947368418e-02, 5.8823529411764705e-02,
                              6.6666666666666666e-02, 7.6923076923076927e-02, 9.0909090909090912e-02,
                              1.1111111111111110e-01, 1.4285714285714285e-01, 2.0000000000000001e-01,
                              3.3333333333333331e-01};
#pragma unroll
    for (int i = 0; i < M; i++) p[i] = 4.3478260869565216e-02;
#pragma unroll
    for (int k = 0; k < 10; k++)
#pragma unroll
        for (int i = 0; i < M; i++) p[i] = __builtin_fma(p[i], z[i], c[k]);
#pragma unroll
    for (int i = 0; i < M; i++) {
        const double s2 = s[i] + s[i];
        double r = __builtin_fma(s2 * z[i], p[i], ed[i] * 1.90821492927058770002e-10);
        r = r + s2;
        r = __builtin_fma(ed[i], 6.93147180369123816490e-01, r);
        x[i] = x[i] > 0.0 ? r : -INFINITY;
    }
}

}  // namespace hmmsort
