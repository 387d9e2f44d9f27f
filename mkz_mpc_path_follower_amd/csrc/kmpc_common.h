// kmpc_common.h -- device primitives shared by the generic (kmpc_kernels.hip) and the
// compile-time-horizon (kmpc_fast.hip) solver kernels.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "kmpc_device.h"

#define DEV __device__ __forceinline__

// Tuned constants of the non-convex phase (kmpc_ipm.h and the generic kernel; the CPU checker carries the same values).  Chosen on the
// pooled worst-of-4096 statistics of 48 seeded 4096-problem batches: shift growth after a failed first trial
// 8 -> 3 (the first trial is last/3, so x3 returns to the shift that worked last iteration instead of overshooting it 2.7-fold),
// barrier floor in shifted iterations rd/1000 -> rd/100 (-> rd/40 later in round 3).
#ifndef KMPC_IKRD
#define KMPC_IKRD 1e-3
#endif
#ifndef KMPC_IKRD_NC
#define KMPC_IKRD_NC 2.5e-2   // (rd/100 until the hybrid strategy switched at the first failure; re-measured on 12 pooled batches: rd/40)
#endif
#ifndef KMPC_DW_GROW
#define KMPC_DW_GROW 3
#endif
// A trial point is also accepted when the PREDICTED decrease of the merit function is below the noise of its evaluation, KMPC_NOISE_ACCEPT * eps * |phi|
// (the cost is resolved to ~50-70 eps: e = x - x_ref carries eps |x| with |x| >> |e|; Ipopt's 10 eps slack is not enough): near a low-cost optimum the
// Armijo test otherwise fails on rounding alone and the search backtracks max_ls times (~100 wasted roll-outs on ~0.25 % of the problems)
#ifndef KMPC_NOISE_ACCEPT
#define KMPC_NOISE_ACCEPT 100
#endif
// Degenerate complementarity pairs (slack and multiplier vanish together -- structurally the last acceleration input a_{N-1}, which only the rate cost
// ties to a_{N-2}: whenever the acceleration saturates to the end of the horizon its bound is active with a zero multiplier).  Newton's method treats such
// a pair as a double root: s and lambda halve per iteration (x0.375 with Mehrotra's corrector), five to eight iterations of end game.  A pair seen
// shrinking that way enters the KKT matrix with theta * lambda/s (theta < 1 lengthens its step: s+ = 0.13 s at theta = 0.6; below 0.45 the corrected
// step overshoots the bound and the fraction-to-the-boundary rule cuts EVERY component).  K stays positive definite and the right-hand side is the
// barrier gradient, so the direction is a descent direction of phi_mu as before.  CPU port, 12 x 4096 seeded problems at N = 20: mean iterations 7.45 ->
// 7.10, E[worst of 4096] 21.7 -> 19.6; same minima (costs agree to 2e-8).  0.75: 20.1; 0.5: 19.4; 0.4: 25.6.
// barrier floor without its mu_cur cap after an accepted step shorter than this (kmpc_ipm.h, the Mehrotra barrier update): un-sticks warm starts from a wrong point
#ifndef KMPC_UNSTICK
#define KMPC_UNSTICK 1e-4
#endif
#ifndef KMPC_DEGEN_THETA
#define KMPC_DEGEN_THETA 0.6
#endif
// Run-time guard of the slack iterates (kmpc_ipm.h, ipm::solve): an iterate is reported Optimal only if every live slack agrees with the freshly
// evaluated b -/+ a_f^T U to this tolerance, relative to max(1, |bound|, |a_f^T U|).  Rounding alone: <= 2.7e-15 (fp64) / <= 1.6e-6 (fp32) measured over
// seeded draws of up to 262 144 problems at N = 8 ... 50, every kernel family (tools/drift_probe.py -> profiles/r4_slack_drift.txt) -- the tolerances sit
// five / two decades above that, and two / one decade below
// the drift that moved round 3's non-KKT "Optimal" (complementarity 9e-3).
#ifndef KMPC_DRIFT_TOL_F64
#define KMPC_DRIFT_TOL_F64 1e-9
#endif
#ifndef KMPC_DRIFT_TOL_F32
#define KMPC_DRIFT_TOL_F32 1e-4
#endif

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <typename T> struct Real;
template <> struct Real<double> {
    typedef double4_t acc_t;
    static DEV acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // C/D of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
    static DEV int row_of(int lane, int reg) { return (lane >> 4) + 4 * reg; }
    static DEV int q_of_row(int rr) { return rr & 3; }
    static DEV int reg_of_row(int rr) { return rr >> 2; }
    static DEV void sincos_(double x, double *s, double *c) { sincos(x, s, c); }
    static DEV double eps() { return 2.220446049250313e-16; }
    static DEV double tiny() { return 1e-300; }
};
template <> struct Real<float> {
    typedef float4_t acc_t;
    static DEV acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // C/D of v_mfma_f32_16x16x4_f32: col = lane&15, row = 4*(lane>>4) + reg
    static DEV int row_of(int lane, int reg) { return 4 * (lane >> 4) + reg; }
    static DEV int q_of_row(int rr) { return rr >> 2; }
    static DEV int reg_of_row(int rr) { return rr & 3; }
    static DEV void sincos_(float x, float *s, float *c) { sincosf(x, s, c); }
    static DEV float eps() { return 1.1920929e-07f; }
    static DEV float tiny() { return 1e-30f; }
};

// ---- wave primitives (64 lanes, one wave per workgroup) -------------------------------------
template <typename T> DEV T wave_sum(T x) { for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o); return x; }
template <typename T> DEV T wave_max(T x) { for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o)); return x; }
template <typename T> DEV T wave_min(T x) { for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o)); return x; }
template <typename T> DEV T scan_prefix(T x, int lane) {  // inclusive, lane 0 -> 63
    for (int d = 1; d < 64; d <<= 1) { T t = __shfl_up(x, d); if (lane >= d) x += t; }
    return x;
}
template <typename T> DEV T scan_suffix(T x, int lane) {  // inclusive, lane 63 -> 0
    for (int d = 1; d < 64; d <<= 1) { T t = __shfl_down(x, d); if (lane + d < 64) x += t; }
    return x;
}
// Every solver kernel runs ONE wave per workgroup: cross-lane exchange through LDS needs program order only (a wave's LDS
// operations execute in order), so the "workgroup barrier" is a wave-level fence for the compiler -- __syncthreads() would add
// s_waitcnt lgkmcnt(0) + s_barrier, a full drain of the LDS queue, at every exchange.
#define WFENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define WSYNC() WFENCE()

// phase stamps: diagnostic builds only (-DKMPC_STAMPS); the shipped library has none.  -DKMPC_TRACE (diagnostic too) turns the
// stamp buffer into a per-iteration record of one problem: row `it` = 8 doubles (tools/trace_problem.py)
// (X_AT(obj, ..) are the forms the shared code of kmpc_ipm.h uses on the solver object; inside the solver structs STAMP(i) = STAMP_AT(*this, i))
#ifdef KMPC_TRACE
#define STAMP_MEMBERS
#define STAMP_DECL_AT(o)
#define STAMP_AT(o, i)
#define STAMP_OUT_AT(o, ptr, b)
#define TRACE8(ptr, it, a0, a1, a2, a3, a4, a5, a6, a7) do { if ((ptr) && threadIdx.x == 0 && (it) < 256) { double *q_ = (double *)(ptr) + 8 * (it); \
    q_[0] = (double)(a0); q_[1] = (double)(a1); q_[2] = (double)(a2); q_[3] = (double)(a3); q_[4] = (double)(a4); q_[5] = (double)(a5); q_[6] = (double)(a6); q_[7] = (double)(a7); } } while (0)
#elif defined(KMPC_STAMPS)
#define STAMP_MEMBERS unsigned long long st_t0_, st_acc_[16];
#define STAMP_DECL_AT(o) do { (o).st_t0_ = __builtin_readcyclecounter(); for (int i_ = 0; i_ < 16; ++i_) (o).st_acc_[i_] = 0; } while (0);
#define STAMP_AT(o, i) do { unsigned long long t_ = __builtin_readcyclecounter(); (o).st_acc_[i] += t_ - (o).st_t0_; (o).st_t0_ = t_; } while (0)
#define STAMP_OUT_AT(o, ptr, b) do { if ((ptr) && threadIdx.x == 0) for (int i_ = 0; i_ < 16; ++i_) (ptr)[(size_t)(b) * 16 + i_] = (o).st_acc_[i_]; } while (0)
#else
#define STAMP_MEMBERS
#define STAMP_DECL_AT(o)
#define STAMP_AT(o, i)
#define STAMP_OUT_AT(o, ptr, b)
#endif
#define STAMP_DECL STAMP_DECL_AT(*this)
#define STAMP(i) STAMP_AT(*this, i)
#define STAMP_OUT(ptr, b) STAMP_OUT_AT(*this, ptr, b)
#ifndef TRACE8
#define TRACE8(ptr, it, a0, a1, a2, a3, a4, a5, a6, a7)
#endif


// ---- DPP / readlane primitives (no LDS traffic) ----------------------------------------------
// dpp_ctrl encodings (GFX9 / CDNA): row_shl:n 0x100+n, row_shr:n 0x110+n, row_bcast15 0x142, row_bcast31 0x143
template <int CTRL, int RM> DEV double dpp_mov0(double x) {  // value of the source lane, 0 where invalid / masked
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, RM, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, RM, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int RM> DEV float dpp_mov0(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, RM, 0xf, true));
}
template <int CTRL> DEV double dpp_mov_keep(double x, double ident) {  // source lane value, `ident` where invalid
    int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(x), CTRL, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL> DEV float dpp_mov_keep(float x, float ident) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(x), CTRL, 0xf, 0xf, false));
}
DEV double readlane_(double x, int l) {  // l wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
DEV float readlane_(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
// a value that is identical in every lane, handed to the compiler AS wave-uniform (scalar registers): branches on it become scalar branches instead of
// exec-masked regions -- which is also where this toolchain's allocator has put spill code ahead of the mask restore
DEV double uniform_(double x) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x))); }
DEV float uniform_(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

// inclusive prefix sum over lanes 0..63 (ROWS = number of 16-lane rows that carry data)
template <int ROWS, typename T> DEV T dpp_scan_prefix(T x) {
    T v = x + dpp_mov0<0x111, 0xf>(x);
    v += dpp_mov0<0x112, 0xf>(x);
    v += dpp_mov0<0x113, 0xf>(x);
    v += dpp_mov0<0x114, 0xf>(v);
    v += dpp_mov0<0x118, 0xf>(v);
    if (ROWS > 1) v += dpp_mov0<0x142, 0xa>(v);
    if (ROWS > 2) v += dpp_mov0<0x143, 0xc>(v);
    return v;
}
// inclusive suffix sum (lane k gets sum over lanes >= k)
template <int ROWS, typename T> DEV T dpp_scan_suffix(T x, int lane) {
    T v = x + dpp_mov0<0x101, 0xf>(x);
    v += dpp_mov0<0x102, 0xf>(x);
    v += dpp_mov0<0x103, 0xf>(x);
    v += dpp_mov0<0x104, 0xf>(v);
    v += dpp_mov0<0x108, 0xf>(v);
    if (ROWS > 1) {
        const T t1 = readlane_(v, 16);
        const T t2 = ROWS > 2 ? readlane_(v, 32) : (T)0, t3 = ROWS > 3 ? readlane_(v, 48) : (T)0;
        const int row = lane >> 4;
        v += row == 0 ? t1 + t2 + t3 : (row == 1 ? t2 + t3 : (row == 2 ? t3 : (T)0));
    }
    return v;
}
template <typename T> DEV T dpp_sum(T x) {  // wave-wide sum, returned uniform (a reduction tree: one DPP step fewer than the scan)
    T v = x + dpp_mov0<0x111, 0xf>(x);
    v += dpp_mov0<0x112, 0xf>(v);
    v += dpp_mov0<0x114, 0xf>(v);
    v += dpp_mov0<0x118, 0xf>(v);   // lane 15 of each row: the row's sum
    v += dpp_mov0<0x142, 0xa>(v);
    v += dpp_mov0<0x143, 0xc>(v);
    return readlane_(v, 63);
}
// max without the canonicalising v_max(x, x) the compiler puts in front of fmax() for operands that are raw bit moves (DPP / readlane)
DEV double max_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int CTRL, int RM> DEV double dpp_mov_keep_rows(double x, double ident) {  // as dpp_mov_keep, rows outside RM keep `ident`
    int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(x), CTRL, RM, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(x), CTRL, RM, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int RM> DEV float dpp_mov_keep_rows(float x, float ident) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(x), CTRL, RM, 0xf, false));
}
template <typename T> DEV T dpp_max(T x) {  // wave-wide maximum, returned uniform (NaN operands are dropped, as by fmax)
    const T ninf = -INFINITY;
    x = max_raw(x, dpp_mov_keep<0x111>(x, ninf));
    x = max_raw(x, dpp_mov_keep<0x112>(x, ninf));
    x = max_raw(x, dpp_mov_keep<0x114>(x, ninf));
    x = max_raw(x, dpp_mov_keep<0x118>(x, ninf));            // lane 15 of each row: the row's maximum
    x = max_raw(x, dpp_mov_keep_rows<0x142, 0xa>(x, ninf));  // row_bcast15 into rows 1 and 3
    x = max_raw(x, dpp_mov_keep_rows<0x143, 0xc>(x, ninf));  // row_bcast31 into rows 2 and 3: lane 63 holds the maximum
    return readlane_(x, 63);
}
template <typename T> DEV T dpp_max_nn(T x) {  // wave-wide maximum of NON-NEGATIVE values: zero fill instead of a -inf identity (no moves to set it up)
    x = max_raw(x, dpp_mov0<0x111, 0xf>(x));
    x = max_raw(x, dpp_mov0<0x112, 0xf>(x));
    x = max_raw(x, dpp_mov0<0x114, 0xf>(x));
    x = max_raw(x, dpp_mov0<0x118, 0xf>(x));
    x = max_raw(x, dpp_mov0<0x142, 0xa>(x));
    x = max_raw(x, dpp_mov0<0x143, 0xc>(x));
    return readlane_(x, 63);
}
template <typename T> DEV T dpp_min(T x) { return -dpp_max(-x); }
