// kmpc_sim.hip -- batched vehicle simulator for closed-loop runs, gfx950 only.
// Restates scripts/vehicle_simulator.py:58-113 for B vehicles at once: one thread per vehicle (the model is a
// 6-state ODE + 2 actuator lags -- there is nothing to share between lanes), fp64 like the reference, state kept in
// registers across all sub-steps of a call, so a 0.1 s control period (10 model updates = 100 Euler sub-steps) costs one
// read and one write of 64 B per vehicle.  FP contraction is off so that every product/sum rounds as in the reference; sin / cos / atan2 are the device
// math library's or short polynomials on their small-argument ranges (both <= 1-2 ulp from numpy's), so parity with the numpy checker is to rounding,
// not bit-exact (tests/test_closed_loop.py states the tolerance).
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

// The 100 serial sub-steps of a control period are what this kernel's time is (one thread per vehicle): per sub-step two atan2, a sin / cos
// pair of the heading, cos of the tyre angle and an fmod.  In the states a path follower visits their arguments are small, so the common case
// takes a short polynomial (<= 1 ulp-level error, like libm's) and anything else the library call:
//   atan2(y, x), x > 0, |y / x| <= 1/8 (slip angles): odd Taylor series to t^19 (next term 8^-21 / 21 < 1e-20 relative);
//   cos(d), |d| <= 0.6 (tyre angle, actuator-lagged command within +-0.5): even series to d^18;
//   sin / cos(psi), psi wrapped to [-pi, pi): quadrant reduction + the Taylor polynomials on |r| <= pi/4;
//   the heading wrap only when psi + pi has left [0, 2 pi).
// Round 3: the sub-step is ONE straight-line block.  With one wave per SIMD (4096 vehicles = 64 waves) nothing hides the latency of the dependent fp64
// chains, so the five polynomial evaluations of a sub-step must overlap each other -- as separate `if (small argument) poly else library` branches each sat
// in its own basic block and the scheduler could not interleave them (79 us per control period).  Now every polynomial is evaluated unconditionally (with
// explicit fma: they are approximations, not reference operations -- the MODEL arithmetic below keeps contraction off and rounds like numpy), the two
// slip-angle quotients share one reciprocal, and a single wave-uniform test sends the wave through the library calls when ANY lane's argument is outside a
// polynomial's range (never, for states a path follower visits).
__device__ __forceinline__ double sim_rcp(double x)   // 1 / x to ~1 ulp (operands are speeds of 1e-6 ... 1e2 m/s)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ double sim_atan_poly(double t)   // |t| <= 1/8: odd Taylor series to t^19
{
    const double z = t * t;
    double p = -1.0 / 19.0;
    p = fma(p, z, 1.0 / 17.0); p = fma(p, z, -1.0 / 15.0); p = fma(p, z, 1.0 / 13.0); p = fma(p, z, -1.0 / 11.0); p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, -1.0 / 7.0); p = fma(p, z, 1.0 / 5.0); p = fma(p, z, -1.0 / 3.0);
    return fma(t, z * p, t);
}
__device__ __forceinline__ double sim_cos_poly(double z)   // z = d^2, |d| <= pi/4 (0.6 for the tyre angle): even series to d^18
{
    double p = -1.0 / 6402373705728000.0;
    p = fma(p, z, 1.0 / 20922789888000.0); p = fma(p, z, -1.0 / 87178291200.0); p = fma(p, z, 1.0 / 479001600.0); p = fma(p, z, -1.0 / 3628800.0);
    p = fma(p, z, 1.0 / 40320.0); p = fma(p, z, -1.0 / 720.0); p = fma(p, z, 1.0 / 24.0); p = fma(p, z, -0.5);
    return fma(z, p, 1.0);
}
__device__ __forceinline__ double sim_sin_poly(double r, double z)   // |r| <= pi/4: odd series to r^17
{
    double p = 1.0 / 355687428096000.0;
    p = fma(p, z, -1.0 / 1307674368000.0); p = fma(p, z, 1.0 / 6227020800.0); p = fma(p, z, -1.0 / 39916800.0); p = fma(p, z, 1.0 / 362880.0);
    p = fma(p, z, -1.0 / 5040.0); p = fma(p, z, 1.0 / 120.0); p = fma(p, z, -1.0 / 6.0);
    return fma(r * z, p, r);
}
// sin / cos of the heading (kept wrapped to [-pi, pi) by the model): Cody-Waite reduction by pi/2 in two pieces + the polynomials on |r| <= pi/4
__device__ __forceinline__ void sim_sincos_poly(double x, double *s, double *c)
{
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632679489655800e+00, x);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    const double sr = sim_sin_poly(r, z), cr = sim_cos_poly(z);
    const int q = (int)k & 3;
    const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

__global__ __launch_bounds__(256) void kmpc_sim_kernel(int B, double *__restrict__ state, const double *__restrict__ cmd, int n_updates)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const double lf = 1.152, lr = 1.693, m = 1840.0, Iz = 3477.0;   // vehicle_simulator.py:61-65
    const double C_alpha_f = 4.0703e4, C_alpha_r = 6.4495e4;        // :66-67
    const double deltaT = 0.01 / 10.0;                              // dt_model / disc_steps, :24,:69
    const double pi = 3.141592653589793;
    double *s = state + 8 * (size_t)i;
    double X = s[0], Y = s[1], psi = s[2], vx = s[3], vy = s[4], wz = s[5], acc = s[6], df = s[7];
    const double acc_des = cmd[2 * (size_t)i], df_des = cmd[2 * (size_t)i + 1];
    for (int it = 0; it < n_updates * 10; ++it) {
        const bool moving = fabs(vx) > 1e-6;                        // :75
        const double yf = vy + lf * wz, yr = vy - lf * wz;          // :76, :77 (lf where lr is expected -- as in the reference)
        const double rvx = sim_rcp(vx);
        const double tf = yf * rvx, tr = yr * rvx;
        // polynomial ranges: x > 0 and |y / x| <= 1/8 for the slip angles (a caller-written state may carry vx < 0: atan2's own quadrant logic),
        // |d_f| <= 0.6, |psi| <= 4
        const bool in_range = (!moving || (vx > 0.0 && fabs(tf) <= 0.125 && fabs(tr) <= 0.125)) && fabs(df) <= 0.6 && fabs(psi) <= 4.0;
        double af = sim_atan_poly(tf), ar = sim_atan_poly(tr), cd = sim_cos_poly(df * df), sp, cp;
        sim_sincos_poly(psi, &sp, &cp);
        if (__any(!in_range)) {   // wave-uniform: some lane is outside a polynomial's range -> the library calls (per lane, where needed)
            if (moving && !(vx > 0.0 && fabs(tf) <= 0.125 && fabs(tr) <= 0.125)) { af = atan2(yf, vx); ar = atan2(yr, vx); }
            if (!(fabs(df) <= 0.6)) cd = cos(df);
            if (!(fabs(psi) <= 4.0)) sincos(psi, &sp, &cp);
        }
        const double alpha_f = moving ? df - af : 0.0;              // :76
        const double alpha_r = moving ? -ar : 0.0;                  // :77
        const double Fyf = C_alpha_f * alpha_f, Fyr = C_alpha_r * alpha_r;  // :80-81
        // :84 reads `acc - 1/m*Fyf*np.sin(self.df) + self.wz*self.vy` with m = 1840 an int: the reference is Python 2 (print statements,
        // no `from __future__ import division`), so 1/m is INTEGER division = 0 and the lateral-force drag term vanishes (:88 uses 1.0/m)
        const double vx_n = fmax(0.0, vx + deltaT * (acc + wz * vy));
        const bool fwd = vx_n > 1e-6;                               // :87
        const double vy_c = vy + deltaT * (1.0 / m * (Fyf * cd + Fyr) - wz * vx);               // :88
        const double wz_c = wz + deltaT * (1.0 / Iz * (lf * Fyf * cd - lr * Fyr));              // :89
        const double vy_n = fwd ? vy_c : 0.0, wz_n = fwd ? wz_c : 0.0;                          // :91-92
        const double psi_n = psi + deltaT * wz;                     // :94
        const double X_n = X + deltaT * (vx * cp - vy * sp);        // :95
        const double Y_n = Y + deltaT * (vx * sp + vy * cp);        // :96
        X = X_n; Y = Y_n;
        const double a = psi_n + pi, p2 = 2.0 * pi;                 // :101  python's float % : result has the divisor's sign
        double md = a;
        const bool wrap = !(a >= 0.0 && a < p2);                    // (a % p2 == a exactly while a is inside [0, p2))
        if (__any(wrap)) {
            if (wrap) {
                md = fmod(a, p2);
                if (md < 0.0) md += p2;
            }
        }
        psi = md - pi;
        vx = vx_n; vy = vy_n; wz = wz_n;
        acc = 5.0 * (acc_des - acc) * deltaT + acc;                 // :112
        df = 5.0 * (df_des - df) * deltaT + df;                     // :113
    }
    s[0] = X; s[1] = Y; s[2] = psi; s[3] = vx; s[4] = vy; s[5] = wz; s[6] = acc; s[7] = df;
}

hipError_t kmpc_launch_sim(int B, double *state, const double *cmd, int n_updates, hipStream_t st)
{
    hipLaunchKernelGGL(kmpc_sim_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, state, cmd, n_updates);
    return hipGetLastError();
}

// The command stage of the node's loop for B vehicles (mpc_cmd_pub.jl): the waypoint helper's stop flag latches (:100-103); a latched vehicle
// is commanded accel -1.0 / steer 0.0 (:148-153) and keeps its rate-limit anchor, any other publishes the solver's first input -- whatever the
// solver status (:120-132, Q7) -- and remembers it as the anchor of the next solve (:140).
__global__ __launch_bounds__(256) void kmpc_command_kernel(int B, const double *__restrict__ u0, const int32_t *__restrict__ stop, uint8_t *__restrict__ latch,
                                                           double *__restrict__ u_prev, double *__restrict__ cmd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const bool st = latch[i] != 0 || stop[i] != 0;
    latch[i] = st ? 1 : 0;
    const double a = u0[2 * (size_t)i], d = u0[2 * (size_t)i + 1];
    cmd[2 * (size_t)i] = st ? -1.0 : a;
    cmd[2 * (size_t)i + 1] = st ? 0.0 : d;
    if (!st) { u_prev[2 * (size_t)i] = a; u_prev[2 * (size_t)i + 1] = d; }
}

hipError_t kmpc_launch_command(int B, const double *u0, const int32_t *stop, uint8_t *latch, double *u_prev, double *cmd, hipStream_t st)
{
    hipLaunchKernelGGL(kmpc_command_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, u0, stop, latch, u_prev, cmd);
    return hipGetLastError();
}
