// kmpc_math.h -- scalar device math shared by the compile-time-horizon kernels (kmpc_fast.hip: one wave per problem;
// kmpc_wide.hip: one 4-wave workgroup per problem).  gfx950 only.
#pragma once
#include "kmpc_common.h"

DEV void pin(double &x) { asm volatile("" : "+v"(x)); }
DEV void pin(float &x) { asm volatile("" : "+v"(x)); }

DEV double rsqrt_(double d) {
    double y = __builtin_amdgcn_rsq(d);          // ~2^-26 relative
    double e = fma(-d * y, y, 1.0);
    y = fma(y * e, fma(e, 0.375, 0.5), y);      // cubic step: ~2^-78
    return y;
}
// reciprocal without the IEEE division's scaling / fix-up sequence (operands here are well inside the normal range): ~1 ulp
// lowest mantissa bit as a one-bit mark on a positive value whose last ulp carries no information (kmpc_ipm.h: degenerate-pair mark on 1/slack)
DEV bool lsb_get(double x) { return (__double2loint(x) & 1) != 0; }
DEV bool lsb_get(float x) { return (__float_as_int(x) & 1) != 0; }
DEV double lsb_set(double x, bool f) { return __hiloint2double(__double2hiint(x), (__double2loint(x) & ~1) | (f ? 1 : 0)); }
DEV float lsb_set(float x, bool f) { return __int_as_float((__float_as_int(x) & ~1) | (f ? 1 : 0)); }
// 1 or KMPC_DEGEN_THETA by the mark in the lowest mantissa bit, recomputed at every call (the empty asm keeps the result from being shared between calls)
template <typename T> DEV T dg_theta(T r)
{
    int b = lsb_get(r) ? 1 : 0;
    asm volatile("" : "+v"(b));
    return (T)1 - (T)(1.0 - KMPC_DEGEN_THETA) * (T)b;
}
DEV int lsb2_get(double x) { return __double2loint(x) & 3; }
DEV int lsb2_get(float x) { return __float_as_int(x) & 3; }
DEV double lsb2_set(double x, int b) { return __hiloint2double(__double2hiint(x), (__double2loint(x) & ~3) | b); }
DEV float lsb2_set(float x, int b) { return __int_as_float((__float_as_int(x) & ~3) | b); }
DEV double rcp_(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    return fma(y, e, y);
}
DEV float rcp_(float d) {
    const float y = __builtin_amdgcn_rcpf(d);
    return fmaf(y, fmaf(-d, y, 1.0f), y);
}
DEV float rsqrt_(float d) {
    float y = __builtin_amdgcn_rsqf(d);
    const float e = fmaf(-d * y, y, 1.0f);
    return fmaf(y * e, fmaf(e, 0.375f, 0.5f), y);
}

// Polynomial coefficients.  An fp64 literal cannot be an instruction operand: LLVM materialises each one into a VGPR pair, hoists
// that out of the iteration loop and keeps it there for the whole solve (~50 VGPRs at N = 20, with the excess spilled to scratch).
// The fp64 kernels therefore read them from a small LDS table (uniform address: one broadcast pass, no VALU work); fp32 literals
// are instruction operands and stay literals.
enum { KC_S = 0, KC_C = 8, KC_2OPI = 16, KC_PIO2H, KC_PIO2L, KC_LG1, KC_LG2, KC_LG3, KC_LG4, KC_LG5, KC_LG6, KC_LG7, KC_LN2H, KC_LN2L,
       KC_SQRTH, KC_COUNT = 32 };
static __constant__ const double kmpc_coef[KC_COUNT] = {
    // sin: x + x^3 (S0 + z (S1 + ...)), listed from the highest power down (Horner order)
    1.0 / 355687428096000.0, -1.0 / 1307674368000.0, 1.0 / 6227020800.0, -1.0 / 39916800.0, 1.0 / 362880.0, -1.0 / 5040.0, 1.0 / 120.0, -1.0 / 6.0,
    // cos: 1 + z (-1/2 + z (C7 + ...)), Horner order; the -1/2 is an inline constant
    -1.0 / 6402373705728000.0, 1.0 / 20922789888000.0, -1.0 / 87178291200.0, 1.0 / 479001600.0, -1.0 / 3628800.0, 1.0 / 40320.0, -1.0 / 720.0, 1.0 / 24.0,
    0.63661977236758134308, 1.57079632679489655800e+00, 6.12323399573676603587e-17,
    // log (fdlibm e_log.c): Lg1..Lg7, ln2_hi, ln2_lo, sqrt(1/2)
    6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01, 1.818357216161805012e-01,
    1.531383769920937332e-01, 1.479819860511658591e-01, 6.93147180369123816490e-01, 1.90821492927058770002e-10, 0.70710678118654752440,
    0.0, 0.0, 0.0 };
template <typename T> struct Coef {  // fp32: literals
    const T *tab;
    DEV T operator[](int i) const { return (T)kmpc_coef[i]; }
};
template <> struct Coef<double> {    // fp64: LDS table
    const double *tab;
    DEV double operator[](int i) const { return tab[i]; }
};

// sin / cos for |x| <= 0.8 (tyre angles are bounded by steer_max <= 0.5 rad): Taylor to x^17 / x^18,
// truncation < 2e-19 relative; larger arguments (non-default steer_max) take the libm path
template <typename T> DEV void sincos_small(T x, T *s, T *c, Coef<T> kc) {
    if (fabs(x) > (T)0.8) { Real<T>::sincos_(x, s, c); return; }
    const T z = x * x;
    T ps = kc[KC_S];
#pragma unroll
    for (int i = 1; i < 8; ++i) ps = fma(ps, z, kc[KC_S + i]);
    *s = fma(x * z, ps, x);
    T pc = kc[KC_C];
#pragma unroll
    for (int i = 1; i < 8; ++i) pc = fma(pc, z, kc[KC_C + i]);
    pc = fma(pc, z, (T)(-0.5));
    *c = fma(z, pc, (T)1);
}

// sin / cos for |x| up to ~1e3 rad (headings are unwrapped relative to the reference, a few rad):
// Cody-Waite reduction by pi/2 in two pieces + the same polynomials on |r| <= pi/4
template <typename T> DEV void sincos_mid(T x, T *s, T *c, Coef<T> kc) {
    if (!(fabs(x) < (T)1000)) { Real<T>::sincos_(x, s, c); return; }
    const T k = rint(x * kc[KC_2OPI]);
    T r = fma(-k, kc[KC_PIO2H], x);
    r = fma(-k, kc[KC_PIO2L], r);
    T sr, cr;
    sincos_small(r, &sr, &cr, kc);
    const int q = (int)k & 3;
    const T ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

// natural logarithm of a positive normal number (the slack products of the barrier function): fdlibm's e_log.c on the table, < 1 ulp
DEV double log_pos(double x, Coef<double> kc) {
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);         // x = m 2^e, m in [1/2, 1)
    const bool lo = m < kc[KC_SQRTH];
    m = lo ? m + m : m; e = lo ? e - 1 : e;               // m in [sqrt(1/2), sqrt(2))
    const double f = m - 1.0, sq = f * rcp_(2.0 + f), z = sq * sq, w = z * z;
    const double t1 = w * fma(w, fma(w, kc[KC_LG6], kc[KC_LG4]), kc[KC_LG2]);
    const double t2 = z * fma(w, fma(w, fma(w, kc[KC_LG7], kc[KC_LG5]), kc[KC_LG3]), kc[KC_LG1]);
    const double R = t1 + t2, hfsq = 0.5 * f * f, dk = (double)e;
    return fma(dk, kc[KC_LN2H], -((hfsq - fma(sq, hfsq + R, dk * kc[KC_LN2L])) - f));
}
DEV float log_pos(float x, Coef<float>) { return logf(x); }

