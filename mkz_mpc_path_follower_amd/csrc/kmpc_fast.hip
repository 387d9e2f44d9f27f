// kmpc_fast.hip -- compile-time-horizon solver kernel for gfx950 (2N + 1 <= 64 lanes, N % 4 == 0).
//
// Same algorithm and same results as the generic kernel in kmpc_kernels.hip (one wavefront per
// problem; see the header there).  What this kernel is organised around, in order of importance
// (measured on MI355X, profiles/):
//   * INSTRUCTION FOOTPRINT.  A fully unrolled version of this kernel (109 KB of code) ran at
//     ~1 byte of instructions per cycle per wave with no benefit from 4 waves per SIMD: it was
//     instruction-fetch bound (the I-cache is 64 KB per CU pair).  So every loop over stages /
//     columns is ROLLED and the roll-out, the linearisation and the factorisation each have ONE
//     call site (the iteration is a small state machine: a line-search trial and the next
//     iterate's evaluation are the same code).
//   * INSTRUCTION COUNT.  With the chip full the kernel is VALU-issue-bound (SQ counters in
//     profiles/): ~4.6 k VALU instructions per iteration at 4 issue cycles each.  Things that cost
//     instructions without doing arithmetic, found in the opcode histogram of the loop: SGPR-spill
//     reloads of the kernel-argument tuples (-> scalars read from an LDS table), fp64 literals parked
//     in VGPRs (-> LDS coefficient table), canonicalising v_max in front of fmax on DPP results
//     (-> raw v_max), selects (-> 0/1 masks, exec-masked regions), IEEE divisions (-> rcp + Newton);
//   * no ds_bpermute / LDS round trips for scans and reductions: DPP row_shr / row_shl / row_bcast;
//   * condensing (Cartesian model) in O(N^2): lane j carries column j of the Hessian through an
//     adjoint recursion over the stages (condense_adjoint; ~30 instructions per stage, no matrix
//     product); the Frenet functor, whose stage Jacobians are dense, contracts G^T W G on the matrix
//     cores (condense_frenet).  The KKT matrix is completed in the MFMA C-layout tile registers
//     (barrier terms, shift, rhs row);
//   * Cholesky: 4-column panels on the matrix cores (the 4x4 diagonal block redundantly in every
//     lane, panel rows in MFMA fragment layout, one MFMA per trailing tile); what is stored (packed
//     column-major LDS image, n(n+3)/2 words, 6.9 KB at N = 20) is the block-LDL^T form L~ = L D^-1,
//     so the substitutions are n/4 dependent block steps and the rhs row comes out solved through L~ and D;
//   * wave-uniform scalars that are read once or twice per iteration, the best iterate, the
//     gradient and the corrector terms live in LDS, not in VGPRs (2 waves per SIMD = 256 VGPRs,
//     20 KB of LDS per wave = exactly 8 waves per CU at N = 20).
#include "kmpc_math.h"


template <typename T> struct StageF {  // lane k: state k / input k at the evaluated point
    T a, d, v, x, y, psi, c, s, sinb, cosb, b1, b2, ex, ey, ep, ev;
    T K, Kp, iden, dsdt;  // Frenet functor only: curvature, dK/ds, 1 / (1 - e_y K), ds/dt at the stage
};
// linearisation scalars of stage k, parked in LDS (16 words per stage):
// A02 A03 A12 A13 A23 Bdx Bdy Bdp mpp mpv mpd mvd mdd 0
constexpr int LIN_STRIDE = 16;

#ifndef KMPC_IKRD
#define KMPC_IKRD 1e-3
#endif
// MODEL 0: Cartesian kinematic bicycle (MKZMPCPathFollower.jl); MODEL 1: Frenet-frame functor (MKZMPCPathFollowerFrenet.jl:112-123): states
// (s, e_y, e_psi, v) in the (x, y, psi, v) slots, zero cost references, curvature polynomial K(s); only roll-out, costates and the
// sensitivity recursion differ (the dynamics couple s, e_y, e_psi, so they are serial recursions on wave-uniform values and the stage
// Jacobians are dense, and the condensing is a matrix-core contraction instead of the adjoint recursion) -- forms, barrier method, KKT
// assembly, factorisation and substitutions are shared.
template <typename T, int N, int MODEL = 0> struct FastSolver {
    static constexpr int LSTR = MODEL == 1 ? KMPC_STG : LIN_STRIDE;  // stage record stride (Frenet: 13 Jacobian + 3 roll-out + 4 costate + 15 Hessian)
    static constexpr int n = 2 * N, R = 2 * (N - 1), nf = 5 * N - 2;
    static constexpr int NF = (nf + 63) / 64;
    static constexpr int NT = (n + 15) / 16, NTT = NT * (NT + 1) / 2;
    static constexpr int SROWS = (N + 1 + 15) / 16;  // 16-lane rows that carry stage data
    static constexpr int NROWS = (n + 1 + 15) / 16;  // rows that carry n-vector data
    static constexpr int LC = n * (n + 3) / 2;       // packed lower triangle + rhs row, column-major
    static_assert(n + 1 <= 64 && n % 8 == 0, "fast kernel needs 2N + 1 <= 64 and N % 4 == 0");
    typedef typename Real<T>::acc_t acc_t;
    static constexpr int lds_elems() { return ((LC + 1) & ~1) + 64 + 64 * NF + 64 + LSTR * (N + 1) + 8 * 64 + 16 + 64 + 64 + 2 * 64 * NF + 16 * (n / 4) + (sizeof(T) == 8 ? KC_COUNT : 0); }
    // start of column j minus j, so that element (row i, col j) lives at offc(j) + i
    static constexpr int offc(int j) { return j * (n + 1) - j * (j - 1) / 2 - j; }
    static DEV int offc_rt(int j) { return j * (n + 1) - ((j * (j - 1)) >> 1) - j; }

    STAMP_MEMBERS
    const KP &P;
    int lane;  // re-materialised (opaque) at the top of every iteration: stops LICM from hoisting the
               // lane-derived index / mask arithmetic of every phase out of the loop into long-lived VGPRs
    T *Lc, *xb, *wb, *cb, *lin, *opb, *gnb, *cs, *ubest, *gb, *cub, *clb, *sinvb;
    Coef<T> kc;  // polynomial coefficients (LDS table in fp64)
    T x0, y0, psi0, v0, vt, up0, up1, rx, ry, rp, xoff, yoff;
    T kp0, kp1, kp2, kp3;  // Frenet: K(s) = kp0 s^3 + kp1 s^2 + kp2 s + kp3
    // Kernel-argument scalars (model constants, limits, tolerances, the doubled weights 2*C_i, products like dt^2 that the host
    // computed because there is no scalar fp64 ALU) are copied once into an LDS table, pt = cb + 32 (cb holds N <= 28 suffix sums), and
    // read from there: the kernarg segment arrives as 16-SGPR tuples which the allocator spills and then reloads WHOLE at every use
    // of any member (~600 v_readlane per iteration before this table).
    enum { PT_DT = 0, PT_DTC, PT_RR, PT_DT2, PT_DTL, PT_LB, PT_TOL, PT_GAP_TOL, PT_TOL_X100, PT_TOL_X1000, PT_TOL_D100, PT_TOL_D10,
           PT_STEER_MAX, PT_A_MAX, PT_STEER_DMAX, PT_A_DMAX, PT_W = 16, PT_V_MIN = 24, PT_V_MAX, PT_RELAX, PT_WARM_PUSH, PT_WARM_MU, PT_MU_INIT,
           PT_INV2NF /* 1 / (2 nf): an fp64 literal in the loop would be hoisted into (and spilled from) a VGPR pair */ };
    const T *pt, *cwt;

    DEV FastSolver(const KP &p, unsigned char *smem) : P(p), lane(threadIdx.x)
    {
        Lc = reinterpret_cast<T *>(smem);
        xb = Lc + ((LC + 1) & ~1);
        wb = xb + 64;
        cb = wb + 64 * NF;
        lin = cb + 64;
        opb = lin + LSTR * (N + 1);
        gnb = opb + 4 * 64;     // terminal sensitivities G_N [3][64] (Cartesian model): the Cholesky panel scratch uses opb[0 .. 4 * 64) only
        cs = opb + 8 * 64;      // wave-uniform scalars that are read once or twice per iteration live here, not in VGPRs
        ubest = cs + 16;        // last iterate that passed Ipopt's test
        gb = ubest + 64;        // gradient of the current linearisation (lane j: g_j)
        cub = gb + 64; clb = cub + 64 * NF;  // corrector terms
        sinvb = clb + 64 * NF;  // D_j^-1 of the factor's 4x4 diagonal blocks (row-major, 16 per 4-column panel)
        for (int e = lane; e < 16 * (n / 4); e += 64) sinvb[e] = (T)0;
        kc.tab = sinvb + 16 * (n / 4);
        if (sizeof(T) == 8 && lane < KC_COUNT) const_cast<T *>(kc.tab)[lane] = (T)kmpc_coef[lane];
        static_assert(N <= 32, "the scalar table shares the 64-entry cb buffer with the N suffix sums");
        pt = cb + 32; cwt = pt + PT_W;
        if (lane == 0) {
            T *q = cb + 32;
            q[PT_DT] = (T)p.dt; q[PT_DTC] = (T)p.dtc; q[PT_RR] = (T)p.r; q[PT_DT2] = (T)p.dt2; q[PT_DTL] = (T)p.dt_over_Lb; q[PT_LB] = (T)p.L_b;
            q[PT_TOL] = (T)p.tol; q[PT_GAP_TOL] = (T)p.gap_tol; q[PT_TOL_X100] = (T)p.tol_x100; q[PT_TOL_X1000] = (T)p.tol_x1000;
            q[PT_TOL_D100] = (T)p.tol_d100; q[PT_TOL_D10] = (T)p.tol_d10;
            q[PT_STEER_MAX] = (T)p.steer_max; q[PT_A_MAX] = (T)p.a_max; q[PT_STEER_DMAX] = (T)p.steer_dmax; q[PT_A_DMAX] = (T)p.a_dmax;
            q[PT_W + 0] = (T)p.C2[0]; q[PT_W + 1] = (T)p.C2[1]; q[PT_W + 2] = (T)p.C2[2]; q[PT_W + 3] = (T)p.C2[3];
            q[PT_W + 4] = (T)p.C2[4]; q[PT_W + 5] = (T)p.C2[5]; q[PT_W + 6] = (T)p.C2[6]; q[PT_W + 7] = (T)p.C2[7];
            q[PT_V_MIN] = (T)p.v_min; q[PT_V_MAX] = (T)p.v_max; q[PT_RELAX] = (T)p.relax; q[PT_WARM_PUSH] = (T)p.warm_push;
            q[PT_WARM_MU] = (T)p.warm_mu; q[PT_MU_INIT] = (T)p.mu_init; q[PT_INV2NF] = (T)(1.0 / (2 * nf));
        }
        // the tables are written by a few lanes and read by all: order the reads behind the writes (found with poisoned LDS in the
        // Frenet instantiation, whose first reads of the table were scheduled ahead of lane 0's stores -- invisible whenever the
        // previous occupant of the LDS was the same kernel, because the constants it left behind are the same)
        WSYNC();
    }

    DEV void load_problem(const T *z0, const T *ref, const T *vtp, const T *upp, int b)
    {
        if (MODEL == 1) {  // `ref` carries k_poly [B,4]; (s, e_y) are not translation-invariant (K depends on s); zero cost references
            xoff = yoff = (T)0; x0 = z0[4 * (size_t)b]; y0 = z0[4 * (size_t)b + 1];
            psi0 = z0[4 * (size_t)b + 2]; v0 = z0[4 * (size_t)b + 3];
            vt = vtp[b];
            up0 = upp[2 * (size_t)b]; up1 = upp[2 * (size_t)b + 1];
            rx = ry = rp = (T)0;
            const T *kp = ref + 4 * (size_t)b;
            kp0 = kp[0]; kp1 = kp[1]; kp2 = kp[2]; kp3 = kp[3];
            return;
        }
        // the NLP is invariant under a translation of (x, y): solve it in vehicle-centred coordinates (recorded paths live hundreds
        // of metres from their origin; positions would carry ~1e-13 m of rounding = ~1e-12 in the cost, above the Armijo
        // decrease of the last iterations); predictions are shifted back on output
        kp0 = kp1 = kp2 = kp3 = (T)0;
        xoff = z0[4 * (size_t)b]; yoff = z0[4 * (size_t)b + 1]; x0 = (T)0; y0 = (T)0;
        psi0 = z0[4 * (size_t)b + 2]; v0 = z0[4 * (size_t)b + 3];
        vt = vtp[b];
        up0 = upp[2 * (size_t)b]; up1 = upp[2 * (size_t)b + 1];
        rx = ry = rp = (T)0;
        if (lane <= N) {
            const T *r = ref + ((size_t)b * (N + 1) + lane) * 3;
            rx = r[0] - xoff; ry = r[1] - yoff; rp = r[2];
        }
    }

    DEV void form_bounds(int f, T &bu, T &bl) const
    {
        const T relax = pt[PT_RELAX];
        if (f < n) {
            const T ub = pt[(f & 1) ? PT_STEER_MAX : PT_A_MAX];
            bu = bl = ub + relax * fmax((T)1, ub);
        } else if (f < n + R) {
            const int r = f - n, jj = r & 1, kk = r >> 1;
            const T d = pt[jj ? PT_STEER_DMAX : PT_A_DMAX] * pt[kk == 0 ? PT_DTC : PT_DT];
            const T u = kk == 0 ? (jj ? up1 : up0) : (T)0;
            bu = d + relax * fmax((T)1, d) + u; bl = d + relax * fmax((T)1, d) - u;
        } else if (f < nf) {
            const T vmax = pt[PT_V_MAX], vmin = pt[PT_V_MIN];
            bu = vmax + relax * fmax((T)1, fabs(vmax)) - v0;
            bl = -vmin + relax * fmax((T)1, fabs(vmin)) + v0;
        } else { bu = bl = (T)1; }
    }
    DEV T form_relax(int f, bool upper) const
    {
        const T relax = pt[PT_RELAX];
        if (f < n) return relax * fmax((T)1, pt[(f & 1) ? PT_STEER_MAX : PT_A_MAX]);
        if (f < n + R) { const int r = f - n; return relax * fmax((T)1, pt[(r & 1) ? PT_STEER_DMAX : PT_A_DMAX] * pt[(r >> 1) == 0 ? PT_DTC : PT_DT]); }
        return relax * fmax((T)1, fabs(pt[upper ? PT_V_MAX : PT_V_MIN]));
    }

    // y_f = a_f^T x   (x: lane j holds x_j)
    DEV void forms_apply(T x, T (&y)[NF])
    {
        if (lane < n) xb[lane] = x;
        WSYNC();
        T a = lane < N ? xb[2 * lane] : (T)0;
        a = dpp_scan_prefix<SROWS>(a);
        if (lane < N) cb[lane] = a;
        WSYNC();
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = lane + 64 * i;
            T v = (T)0;
            if (f < n) v = xb[f];
            else if (f < n + R) { const int r = f - n; v = r < 2 ? xb[r] : xb[r + 2] - xb[r]; }
            else if (f < nf) v = pt[PT_DT] * cb[f - n - R];
            y[i] = v;
        }
        WSYNC();
    }
    DEV void stage_form_weights(const T (&w)[NF])
    {
#pragma unroll
        for (int i = 0; i < NF; ++i) { const int f = lane + 64 * i; if (f < nf) wb[f] = w[i]; }
        WSYNC();
        T s = lane < N ? wb[n + R + lane] : (T)0;
        s = dpp_scan_suffix<SROWS>(s, lane);
        if (lane < N) cb[lane] = s;
        WSYNC();
    }
    DEV T forms_applyT(const T (&w)[NF])  // returns (A^T w)_lane
    {
        stage_form_weights(w);
        T o = (T)0;
        const int j = lane;
        if (j < n) {
            o = wb[j];
            if (j < 2) o += wb[n + j];
            if (j >= 4) o += wb[n + j - 2];
            if (j >= 2 && j < R) o -= wb[n + j];
            if (!(j & 1)) o += pt[PT_DT] * cb[j >> 1];
        }
        WSYNC();
        return o;
    }
    // roll-out (MKZMPCPathFollower.jl:115-122 as prefix scans) + objective (:97-103) at U (lane j: U_j)
    DEV T eval(T U, StageF<T> &S)
    {
        if constexpr (MODEL == 1) return eval_frenet(U, S);
        const T dt = pt[PT_DT], rr_ = pt[PT_RR], dtL = pt[PT_DTL];
        if (lane < n) xb[lane] = U;
        WSYNC();
        const int k = lane;
        const bool st = k < N;
        const T a = st ? xb[2 * k] : (T)0, d = st ? xb[2 * k + 1] : (T)0;
        const T an = (k + 1 < N) ? xb[2 * k + 2] : a, dn = (k + 1 < N) ? xb[2 * k + 3] : d;
        WSYNC();
        S.a = a; S.d = d;
        const T ia = dpp_scan_prefix<SROWS>(a);
        const T v = v0 + dt * (ia - a);
        T sd, cd;
        sincos_small(d, &sd, &cd, kc);
        const T Dn = cd * cd + rr_ * rr_ * sd * sd;
        const T rs = rsqrt_(Dn);
        S.sinb = rr_ * sd * rs;
        S.cosb = cd * rs;
        const T iD = rs * rs;  // 1 / Dn
        S.b1 = rr_ * iD;
        S.b2 = rr_ * ((T)1 - rr_ * rr_) * ((T)2 * sd * cd) * (iD * iD);
        const T wp = st ? v * S.sinb : (T)0;
        const T ip = dpp_scan_prefix<SROWS>(wp);
        const T psi = psi0 + dtL * (ip - wp);
        T sp, cp;
        sincos_mid(psi, &sp, &cp, kc);
        S.c = cp * S.cosb - sp * S.sinb;
        S.s = sp * S.cosb + cp * S.sinb;
        const T wx = st ? v * S.c : (T)0, wy = st ? v * S.s : (T)0;
        const T ix = dpp_scan_prefix<SROWS>(wx), iy = dpp_scan_prefix<SROWS>(wy);
        S.x = x0 + dt * (ix - wx);
        S.y = y0 + dt * (iy - wy);
        S.v = v; S.psi = psi;
        const bool cs = (k >= 1 && k <= N);
        S.ex = cs ? S.x - rx : (T)0;
        S.ey = cs ? S.y - ry : (T)0;
        S.ep = cs ? psi - rp : (T)0;
        S.ev = (k >= 1 && k <= N - 1) ? v - vt : (T)0;
        // (the weights are held doubled -- the form every derivative needs; halving the sum is exact)
        const T Cx2 = cwt[0], Cy2 = cwt[1], Cp2 = cwt[2], Cv2 = cwt[3], Cda2 = cwt[4], Cdd2 = cwt[5], Ca2 = cwt[6], Cd2 = cwt[7];
        T Jl = Cx2 * S.ex * S.ex + Cy2 * S.ey * S.ey + Cp2 * S.ep * S.ep + Cv2 * S.ev * S.ev;
        if (st) Jl += Ca2 * a * a + Cd2 * d * d;
        if (k < N - 1) Jl += Cda2 * (an - a) * (an - a) + Cdd2 * (dn - d) * (dn - d);
        Jl *= (T)0.5;
        return dpp_sum(Jl);
    }

    // costates by suffix scans -> gradient (returned, lane j: g_j); per-stage scalars go to LDS (Lc alias)
    DEV T linearize(const StageF<T> &S, bool exact)
    {
        if constexpr (MODEL == 1) return linearize_frenet(S, exact);
        const T dt = pt[PT_DT], dtL = pt[PT_DTL];
        const int k = lane;
        const bool st = k < N;
        const T Cx2 = cwt[0], Cy2 = cwt[1], Cp2 = cwt[2], Cv2 = cwt[3], Cda2 = cwt[4], Cdd2 = cwt[5], Ca2 = cwt[6], Cd2 = cwt[7];
        const T lx = Cx2 * S.ex, ly = Cy2 * S.ey, lp = Cp2 * S.ep, lv = Cv2 * S.ev;
        const T px = dpp_scan_suffix<SROWS>(lx, lane), py = dpp_scan_suffix<SROWS>(ly, lane);
        const T px1 = dpp_mov0<0x130, 0xf>(px), py1 = dpp_mov0<0x130, 0xf>(py);  // wave_shl:1 -> value of lane+1
        const T A02 = st ? -dt * S.v * S.s : (T)0, A12 = st ? dt * S.v * S.c : (T)0;
        const T A03 = st ? dt * S.c : (T)0, A13 = st ? dt * S.s : (T)0, A23 = st ? dtL * S.sinb : (T)0;
        const T tp = lp + (st ? A02 * px1 + A12 * py1 : (T)0);
        const T pp = dpp_scan_suffix<SROWS>(tp, lane);
        const T pp1 = dpp_mov0<0x130, 0xf>(pp);
        const T tv = lv + (st ? A03 * px1 + A13 * py1 + A23 * pp1 : (T)0);
        const T pv = dpp_scan_suffix<SROWS>(tv, lane);
        const T pv1 = dpp_mov0<0x130, 0xf>(pv);
        const T Bdx = st ? -dt * S.v * S.s * S.b1 : (T)0, Bdy = st ? dt * S.v * S.c * S.b1 : (T)0;
        const T Bdp = st ? dtL * S.v * S.cosb * S.b1 : (T)0;
        const T aprev = dpp_mov0<0x138, 0xf>(S.a), dprev = dpp_mov0<0x138, 0xf>(S.d);  // wave_shr:1 -> lane-1
        const T anext = dpp_mov0<0x130, 0xf>(S.a), dnext = dpp_mov0<0x130, 0xf>(S.d);
        T ga = dt * pv1 + Ca2 * S.a, gd = Bdx * px1 + Bdy * py1 + Bdp * pp1 + Cd2 * S.d;
        if (k >= 1) { ga += Cda2 * (S.a - aprev); gd += Cdd2 * (S.d - dprev); }
        if (k < N - 1) { ga -= Cda2 * (anext - S.a); gd -= Cdd2 * (dnext - S.d); }
        if (st) { xb[2 * k] = ga; xb[2 * k + 1] = gd; }
        if (ADJ) {
            // Terminal sensitivities for condense_adjoint: column j of G_N = Phi(N, k+1) B_k e_j, k = j / 2.  The stage Jacobians are unit
            // upper triangular (x, y <- psi, v; psi <- v), so the transition matrix is made of suffix sums over the later stages:
            //   d psi_N / d v = P3(k) = sum_{s>k} A23_s,   d x_N / d psi = X2(k) = sum_{s>k} A02_s,
            //   d x_N / d v = sum_{s>k} [A03_s + A02_s (P3(k) - R_s)],  R_s = sum_{t>=s} A23_t          (y alike)
            const T R = dpp_scan_suffix<SROWS>(A23, lane);
            const T ux = A03 - A02 * R, uy = A13 - A12 * R;
            const T X2 = dpp_scan_suffix<SROWS>(A02, lane) - A02, Y2 = dpp_scan_suffix<SROWS>(A12, lane) - A12;
            const T zx = dpp_scan_suffix<SROWS>(ux, lane) - ux, zy = dpp_scan_suffix<SROWS>(uy, lane) - uy;
            const T P3 = R - A23;
            if (st) {
                T *q = gnb + 2 * k;
                q[0] = dt * fma(P3, X2, zx); q[64] = dt * fma(P3, Y2, zy); q[128] = dt * P3;   // acceleration column: B = (0, 0, 0, dt)
                q[1] = fma(X2, Bdp, Bdx); q[65] = fma(Y2, Bdp, Bdy); q[129] = Bdp;               // steering column
            }
        }
        T mpp = 0, mpv = 0, mpd = 0, mvd = 0, mdd = 0;
        if (exact && st) {
            const T v = S.v, c = S.c, s = S.s, b1 = S.b1, b2 = S.b2;
            mpp = px1 * (-dt * v * c) + py1 * (-dt * v * s);
            mpv = px1 * (-dt * s) + py1 * (dt * c);
            mpd = px1 * (-dt * v * c * b1) + py1 * (-dt * v * s * b1);
            mvd = px1 * (-dt * s * b1) + py1 * (dt * c * b1) + pp1 * (dtL * S.cosb * b1);
            mdd = px1 * (-dt * v * (c * b1 * b1 + s * b2)) + py1 * (dt * v * (-s * b1 * b1 + c * b2)) +
                  pp1 * (dtL * v * (-S.sinb * b1 * b1 + S.cosb * b2));
        }
        {
            T *q = lin + LSTR * (k <= N ? k : N);
            T *dmy = xb + 62;  // lanes > N write past the n <= 56 gradient entries held in xb
            T *w0 = k <= N ? q : dmy;
            w0[0] = A02; w0[1] = A03;
            if (k <= N) { q[2] = A12; q[3] = A13; q[4] = A23; q[5] = Bdx; q[6] = Bdy; q[7] = Bdp;
                          q[8] = mpp; q[9] = mpv; q[10] = mpd; q[11] = mvd; q[12] = mdd; q[13] = (T)0; }
        }
        WSYNC();
        const T g = lane < n ? xb[lane] : (T)0;
        WSYNC();
        return g;
    }


    // ================= Frenet functor (MODEL 1), MKZMPCPathFollowerFrenet.jl:112-123 ======================================================
    // Roll-out: a serial recursion over the stages on wave-uniform values (every lane runs it; lane k keeps stage k).
    DEV T eval_frenet(T U, StageF<T> &S)
    {
        const T dt = pt[PT_DT], rr_ = pt[PT_RR], dtL = pt[PT_DTL];
        if (lane < n) xb[lane] = U;
        WSYNC();
        const int k = lane;
        const bool st = k < N;
        const T a = st ? xb[2 * k] : (T)0, d = st ? xb[2 * k + 1] : (T)0;
        const T an = (k + 1 < N) ? xb[2 * k + 2] : a, dn = (k + 1 < N) ? xb[2 * k + 3] : d;
        S.a = a; S.d = d;
        T sd, cd;
        sincos_small(d, &sd, &cd, kc);
        const T Dn = cd * cd + rr_ * rr_ * sd * sd;
        const T rs = rsqrt_(Dn);
        S.sinb = rr_ * sd * rs;  // sin(atan(r tan d))   (:113)
        S.cosb = cd * rs;
        const T iD = rs * rs;
        S.b1 = rr_ * iD;                                                    // d beta / d d_f
        S.b2 = rr_ * ((T)1 - rr_ * rr_) * ((T)2 * sd * cd) * (iD * iD);     // d2 beta / d d_f2
        if (st) { T *q = lin + LSTR * k; q[13] = a; q[14] = S.sinb; q[15] = S.cosb; }  // slots 0..12 hold the stage Jacobians
        WSYNC();
        T s_ = x0, ey_ = y0, ep_ = psi0, v_ = v0;
        S.x = S.y = S.psi = S.v = S.c = S.s = S.K = S.Kp = S.iden = S.dsdt = (T)0;
#pragma nounroll
        for (int kk = 0; kk <= N; ++kk) {
            const T K = ((kp0 * s_ + kp1) * s_ + kp2) * s_ + kp3;               // :112
            const T Kp = ((T)3 * kp0 * s_ + (T)2 * kp1) * s_ + kp2;
            if (lane == kk) { S.x = s_; S.y = ey_; S.psi = ep_; S.v = v_; S.K = K; S.Kp = Kp; }
            if (kk == N) break;
            const T *q = lin + LSTR * kk;
            const T ak = q[13], sb = q[14], cbt = q[15];
            T sp, cp;
            sincos_mid(ep_, &sp, &cp, kc);
            const T c = cp * cbt - sp * sb, sn = sp * cbt + cp * sb;            // cos / sin(e_psi + beta)
            const T iden = rcp_((T)1 - ey_ * K), dsdt = v_ * c * iden;          // :114
            if (lane == kk) { S.c = c; S.s = sn; S.iden = iden; S.dsdt = dsdt; }
            s_ += dt * dsdt;                                                     // :118
            ey_ += dt * (v_ * sn);                                               // :119
            ep_ += dtL * v_ * sb - dt * dsdt * K;                                // :120
            v_ += dt * ak;                                                       // :121
        }
        WSYNC();
        const bool cs = (k >= 1 && k <= N);
        S.ex = cs ? S.x : (T)0;      // zero references (the s weight is 0: Frenet.jl:97-98 has no s term)
        S.ey = cs ? S.y : (T)0;
        S.ep = cs ? S.psi : (T)0;
        S.ev = (k >= 1 && k <= N - 1) ? S.v - vt : (T)0;
        const T Cx2 = cwt[0], Cy2 = cwt[1], Cp2 = cwt[2], Cv2 = cwt[3], Cda2 = cwt[4], Cdd2 = cwt[5], Ca2 = cwt[6], Cd2 = cwt[7];
        T Jl = Cx2 * S.ex * S.ex + Cy2 * S.ey * S.ey + Cp2 * S.ep * S.ep + Cv2 * S.ev * S.ev;
        if (st) Jl += Ca2 * a * a + Cd2 * d * d;
        if (k < N - 1) Jl += Cda2 * (an - a) * (an - a) + Cdd2 * (dn - d) * (dn - d);
        Jl *= (T)0.5;
        return dpp_sum(Jl);
    }

    // stage record of the Frenet functor: A00 A01 A02 A03 A12 A13 A20 A21 A22 A23 Bs Bey Bep  (A11 = A33 = 1, B_v,acc = dt), roll-out
    // scratch 13..15, costate of the state 16..19, second-order block 20..34 (upper triangle over (s, e_y, e_psi, v, d_f), row-major)
    DEV T linearize_frenet(const StageF<T> &S, bool exact)
    {
        const T dt = pt[PT_DT], Lb = pt[PT_LB], iLb = pt[PT_DTL] * rcp_(dt);
        const T Cx2 = cwt[0], Cy2 = cwt[1], Cp2 = cwt[2], Cv2 = cwt[3], Cda2 = cwt[4], Cdd2 = cwt[5], Ca2 = cwt[6], Cd2 = cwt[7];
        (void)Lb;
        const int k = lane;
        const bool st = k < N;
        if (k <= N) {
            T *q = lin + LSTR * k;
            const T v = S.v, c = S.c, sn = S.s, K = S.K, Kp = S.Kp, iden = S.iden, dsdt = S.dsdt, ey = S.y, b1 = S.b1;
            const T ds_s = v * c * ey * Kp * iden * iden, ds_ey = v * c * K * iden * iden, ds_ep = -v * sn * iden, ds_v = c * iden,
                    ds_d = -v * sn * iden * b1;
            q[0] = st ? (T)1 + dt * ds_s : (T)0; q[1] = st ? dt * ds_ey : (T)0; q[2] = st ? dt * ds_ep : (T)0; q[3] = st ? dt * ds_v : (T)0;
            q[4] = st ? dt * v * c : (T)0; q[5] = st ? dt * sn : (T)0;
            q[6] = st ? dt * (-ds_s * K - dsdt * Kp) : (T)0; q[7] = st ? -dt * ds_ey * K : (T)0;
            q[8] = st ? (T)1 - dt * ds_ep * K : (T)0; q[9] = st ? dt * (S.sinb * iLb - ds_v * K) : (T)0;
            q[10] = st ? dt * ds_d : (T)0; q[11] = st ? dt * v * c * b1 : (T)0; q[12] = st ? dt * (v * iLb * S.cosb * b1 - ds_d * K) : (T)0;
            T *l = wb + 4 * k;  // stage cost gradient (wb is scratch here; stage_form_weights rewrites it later)
            l[0] = Cx2 * S.ex; l[1] = Cy2 * S.ey; l[2] = Cp2 * S.ep; l[3] = Cv2 * S.ev;
        }
        WSYNC();
        T l0 = wb[4 * N], l1 = wb[4 * N + 1], l2 = wb[4 * N + 2], l3 = wb[4 * N + 3];  // costate of state N
        if (lane == 0) { T *ql = lin + LSTR * N + 16; ql[0] = l0; ql[1] = l1; ql[2] = l2; ql[3] = l3; }
#pragma nounroll
        for (int kk = N - 1; kk >= 0; --kk) {  // costates: a serial recursion on wave-uniform values
            const T *q = lin + LSTR * kk;
            if (lane == 0) { xb[2 * kk] = dt * l3; xb[2 * kk + 1] = q[10] * l0 + q[11] * l1 + q[12] * l2; }  // B_k^T lambda_{k+1}
            const T t0 = q[0] * l0 + q[6] * l2;
            const T t1 = q[1] * l0 + l1 + q[7] * l2;
            const T t2 = q[2] * l0 + q[4] * l1 + q[8] * l2;
            const T t3 = q[3] * l0 + q[5] * l1 + q[9] * l2 + l3;
            const T *l = wb + 4 * kk;
            l0 = t0 + l[0]; l1 = t1 + l[1]; l2 = t2 + l[2]; l3 = t3 + l[3];
            if (lane == 0) { T *ql = lin + LSTR * kk + 16; ql[0] = l0; ql[1] = l1; ql[2] = l2; ql[3] = l3; }
        }
        WSYNC();
        if (k <= N) {
            T *q = lin + LSTR * k + 20;  // upper triangle, row-major: ss se sp sv sd | ee ep ev ed | pp pv pd | vv vd | dd
            T m[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) m[i] = (T)0;
            if (exact && st) {
                // second derivatives of the Euler step wrt (s, e_y, e_psi, v, d_f), contracted with the costate of state k+1
                const T *ln = lin + LSTR * (k + 1) + 16;
                const T m0 = ln[0], m1 = ln[1], m2 = ln[2];
                const T s_ = S.x, ey = S.y, v = S.v, C = S.c, Sn = S.s, K = S.K, K1 = S.Kp, K2 = (T)6 * kp0 * s_ + (T)2 * kp1;
                const T D = S.iden, b1 = S.b1, b2 = S.b2, gq = S.dsdt;
                const T Ds = ey * K1 * D * D, De = K * D * D;
                const T Dss = ey * K2 * D * D + (T)2 * ey * K1 * D * Ds, Dse = K1 * D * D + (T)2 * ey * K1 * D * De, Dee = (T)2 * K * D * De;
                const T g_s = v * C * Ds, g_e = v * C * De, g_p = -v * Sn * D, g_v = C * D, g_d = -v * Sn * b1 * D;
                const T w = m0 - m2 * K, a2 = m2 * K1;
                m[0] = dt * (w * (v * C * Dss) - (T)2 * a2 * g_s - m2 * gq * K2);
                m[1] = dt * (w * (v * C * Dse) - a2 * g_e);
                m[2] = dt * (w * (-v * Sn * Ds) - a2 * g_p);
                m[3] = dt * (w * (C * Ds) - a2 * g_v);
                m[4] = dt * (w * (-v * Sn * b1 * Ds) - a2 * g_d);
                m[5] = dt * (w * (v * C * Dee));
                m[6] = dt * (w * (-v * Sn * De));
                m[7] = dt * (w * (C * De));
                m[8] = dt * (w * (-v * Sn * b1 * De));
                m[9] = dt * (w * (-v * C * D) + m1 * (-v * Sn));
                m[10] = dt * (w * (-Sn * D) + m1 * C);
                m[11] = dt * (w * (-v * C * b1 * D) + m1 * (-v * Sn * b1));
                m[13] = dt * (w * (-Sn * b1 * D) + m1 * (C * b1) + m2 * (S.cosb * b1 * iLb));
                m[14] = dt * (w * (v * D * (-C * b1 * b1 - Sn * b2)) + m1 * (v * (-Sn * b1 * b1 + C * b2)) + m2 * (v * (-S.sinb * b1 * b1 + S.cosb * b2) * iLb));
            }
#pragma unroll
            for (int i = 0; i < 15; ++i) q[i] = m[i];   // zero when the Gauss-Newton matrix is wanted (and in record N)
        }
        // input-cost terms, Frenet.jl:99-102 (same as the Cartesian model)
        const T aprev = dpp_mov0<0x138, 0xf>(S.a), dprev = dpp_mov0<0x138, 0xf>(S.d);  // wave_shr:1 -> lane-1
        const T anext = dpp_mov0<0x130, 0xf>(S.a), dnext = dpp_mov0<0x130, 0xf>(S.d);
        if (st) {
            T ga = xb[2 * k] + Ca2 * S.a, gd = xb[2 * k + 1] + Cd2 * S.d;
            if (k >= 1) { ga += Cda2 * (S.a - aprev); gd += Cdd2 * (S.d - dprev); }
            if (k < N - 1) { ga -= Cda2 * (anext - S.a); gd -= Cdd2 * (dnext - S.d); }
            xb[2 * k] = ga; xb[2 * k + 1] = gd;
        }
        WSYNC();
        const T g = lane < n ? xb[lane] : (T)0;
        WSYNC();
        return g;
    }

    // Condensing with dense stage Jacobians, on the matrix cores: H = sum_s G_s^T (2 Q_s + M_s^{zz}) G_s accumulates in `acc` (lower 16x16
    // tiles, MFMA C layout, unscaled).  Lane j keeps all four components of column j of G (16 FMAs per stage); each stage's MFMA fragments
    // -- A = G_s, B = (2 Q_s + M_s) G_s with the full symmetric 4x4 weight -- go through a small component-major LDS buffer (opb[8][64]:
    // conflict-free writes by column, conflict-free reads in fragment layout), software-pipelined: while the fragments of stage s are in
    // flight / on the matrix cores the VALU advances the recursion.  Trips with the same number of live tile rows share one basic block.
    // The d_f rows of the second-order term (every ODD row of the image) are written, scaled, straight into the packed K image by the
    // lane of each column; build_tiles reads them back into the C-layout tiles.
    template <int ROWS> DEV void condense_trips_frenet(int s0, int s1, T (&g)[4], T (&fa)[NT], T (&fb)[NT], acc_t (&acc)[NTT], T sc)
    {
        const int kk = lane >> 4, c = lane & 15;
        const T dtv = pt[PT_DT];
        T *colK = Lc + offc_rt(lane < n ? lane : 0);
        const T Cx2 = cwt[0], Cy2 = cwt[1], Cp2 = cwt[2], Cv2 = cwt[3];
#pragma nounroll
        for (int s = s0; s < s1; ++s) {
            const T *q = lin + LSTR * s;           // Jacobians and d_f row of the second-order block of state s
            const T *mn = lin + LSTR * (s + 1) + 20;  // state part of the second-order block of state s+1 (record N: zero)
            const T A00 = q[0], A01 = q[1], A02 = q[2], A03 = q[3], A12 = q[4], A13 = q[5], A20 = q[6], A21 = q[7], A22 = q[8], A23 = q[9];
            const T Bs = q[10], Bey = q[11], Bep = q[12];
            const T msd = q[24], med = q[28], mpd = q[31], mvd = q[33], mdd = q[34];
            const int rho = 2 * s + 1;
            const T val = sc * (msd * g[0] + med * g[1] + mpd * g[2] + mvd * g[3]) + (lane == rho ? sc * mdd : (T)0);
            T *dst = (lane <= rho && lane < n) ? colK + rho : xb + lane;
            *dst = val;
            // advance the recursion to state s+1: G_{s+1} = [A_s G_s | B_s]
            const T n0 = A00 * g[0] + A01 * g[1] + A02 * g[2] + A03 * g[3];
            const T n1 = g[1] + A12 * g[2] + A13 * g[3];
            const T n2 = A20 * g[0] + A21 * g[1] + A22 * g[2] + A23 * g[3];
            g[0] = n0; g[1] = n1; g[2] = n2;
            if (lane == 2 * s) { g[0] = (T)0; g[1] = (T)0; g[2] = (T)0; g[3] = dtv; }
            if (lane == 2 * s + 1) { g[0] = Bs; g[1] = Bey; g[2] = Bep; g[3] = (T)0; }
            // products of state s (fragments requested at the end of the previous trip)
#pragma unroll
            for (int ti = 0; ti < ROWS; ++ti)
#pragma unroll
                for (int tj = 0; tj <= ti; ++tj)
                    acc[ti * (ti + 1) / 2 + tj] = Real<T>::mfma(fa[ti], fb[tj], acc[ti * (ti + 1) / 2 + tj]);
            // weights of state s+1
            const T Cv1 = s + 1 <= N - 1 ? Cv2 : (T)0;
            const T W00 = Cx2 + mn[0], W01 = mn[1], W02 = mn[2], W03 = mn[3], W11 = Cy2 + mn[5], W12 = mn[6], W13 = mn[7],
                    W22 = Cp2 + mn[9], W23 = mn[10], W33 = Cv1 + mn[12];
            opb[0 * 64 + lane] = g[0]; opb[1 * 64 + lane] = g[1]; opb[2 * 64 + lane] = g[2]; opb[3 * 64 + lane] = g[3];
            opb[4 * 64 + lane] = W00 * g[0] + W01 * g[1] + W02 * g[2] + W03 * g[3];
            opb[5 * 64 + lane] = W01 * g[0] + W11 * g[1] + W12 * g[2] + W13 * g[3];
            opb[6 * 64 + lane] = W02 * g[0] + W12 * g[1] + W22 * g[2] + W23 * g[3];
            opb[7 * 64 + lane] = W03 * g[0] + W13 * g[1] + W23 * g[2] + W33 * g[3];
            WFENCE();
#pragma unroll
            for (int t = 0; t < (ROWS + 1 < NT ? ROWS + 1 : NT); ++t) {
                fa[t] = opb[kk * 64 + 16 * t + c]; fb[t] = opb[(4 + kk) * 64 + 16 * t + c];
            }
        }
    }
    DEV void condense_frenet(T sc, acc_t (&acc)[NTT])
    {
#pragma unroll
        for (int t = 0; t < NTT; ++t) acc[t] = acc_t{0, 0, 0, 0};
        T g[4] = {0, 0, 0, 0}, fa[NT], fb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) fa[t] = fb[t] = (T)0;
        condense_trips_frenet<0>(0, 1, g, fa, fb, acc, sc);
        condense_trips_frenet<1>(1, N < 9 ? N : 9, g, fa, fb, acc, sc);
        if (NT >= 2 && N > 9) condense_trips_frenet<(NT >= 2 ? 2 : 1)>(9, N < 17 ? N : 17, g, fa, fb, acc, sc);
        if (NT >= 3 && N > 17) condense_trips_frenet<(NT >= 3 ? 3 : 1)>(17, N < 25 ? N : 25, g, fa, fb, acc, sc);
        if (NT >= 4 && N > 25) condense_trips_frenet<(NT >= 4 ? 4 : 1)>(25, N, g, fa, fb, acc, sc);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj)
                acc[ti * (ti + 1) / 2 + tj] = Real<T>::mfma(fa[ti], fb[tj], acc[ti * (ti + 1) / 2 + tj]);
    }

    // stage record as the condensing recursions read it (uniform address: one LDS broadcast per field)
    struct Rec { T a02, a03, a12, a13, a23, bx, by, bp, mpp, mpv, mpd, mvd, mdd; };
    DEV void load_rec(Rec &r, int s) const
    {
        const T *q = lin + LSTR * s;  // record N is all zero (linearize)
        r.a02 = q[0]; r.a03 = q[1]; r.a12 = q[2]; r.a13 = q[3]; r.a23 = q[4]; r.bx = q[5]; r.by = q[6]; r.bp = q[7];
        r.mpp = q[8]; r.mpv = q[9]; r.mpd = q[10]; r.mvd = q[11]; r.mdd = q[12];  // zero when the Gauss-Newton matrix is wanted
    }
    // The second-order entries of the stage records decide between the exact and the Gauss-Newton matrix: linearize writes them only
    // when the exact Hessian is wanted, and a fallback inside an iteration clears them (drop_second_order).
    DEV void drop_second_order()
    {
        T z = (T)0;
        pin(z);  // materialised here: hoisted out of the iteration loop this zero would occupy (and spill) a VGPR pair for the whole solve
        if (MODEL == 1) {
            if (lane <= N) { T *q = lin + LSTR * lane + 20; for (int i = 0; i < 15; ++i) q[i] = z; }
        } else if (lane <= N) { T *q = lin + LSTR * lane; q[8] = z; q[9] = z; q[10] = z; q[11] = z; q[12] = z; }
        WSYNC();
    }
    // Condensing in O(N^2) (Cartesian model): column j of sc * H, H = sum_s G_s^T W_s G_s + the second-order d_f rows, by an ADJOINT
    // recursion with lane j = column j -- no matrix product at all:
    //   start   : column j of G_N, the sensitivity of the terminal state (closed form in suffix sums of the stage Jacobians: linearize);
    //   backward: p(s) = sum_{k >= s} Phi(k,s)^T W_k G_k[:,j] = W_s G_s[:,j] + A_s^T p(s+1), and with it the two rows of stage s,
    //             H[2s][j] = dt p_v(s+1),  H[2s+1][j] = B_s^T p(s+1) + (mpd, mvd) . G_s[(psi, v), j]   (the m_dd diagonal: build_tiles);
    //             G_s[:,j] comes from G_{s+1}[:,j] through the exact inverse of the unit upper-triangular A_s (5 FMAs, nothing stored).
    // Each lane writes its column (rows >= j) of the packed K image; build_tiles reads the tiles from there.  A lane whose column is
    // born at stage j/2 carries meaningless (finite) values below that stage; they are never stored.
    DEV void condense_adjoint(T sc)
    {
        const T dtv = pt[PT_DT];
        const T Cx2 = cwt[0], Cy2 = cwt[1], Cp2 = cwt[2], Cv2 = cwt[3];
        // column j of G_N: linearize left it in the table (it depends on the linearisation only, not on the barrier weights or the shift);
        // everything downstream is linear in G, so the scaling of the objective goes in here, once
        const int jr = lane < n ? lane : 0;
        T gx = sc * gnb[jr], gy = sc * gnb[64 + jr], gp = sc * gnb[128 + jr], gv = (lane & 1) ? (T)0 : sc * dtv;
        // p(N) = W_N G_N: no second-order part and no speed cost on the terminal state
        T px = Cx2 * gx, py = Cy2 * gy, pp = Cp2 * gp, pv = (T)0;
        // lane j is alive while 2s+1 >= j.  Lane 2s+1 has no row 2s: its store lands on (row n, column 2s) of the image -- the rhs row, which
        // nobody reads before the factorisation writes it -- so one address serves both rows
        T *colK = Lc + offc_rt(lane < n ? lane : 0);
        Rec cur;
        load_rec(cur, N - 1);
#pragma unroll 2
        for (int s = N - 1; s >= 0; --s) {
            Rec nxt;
            load_rec(nxt, s > 0 ? s - 1 : 0);
            const T ra = dtv * pv;
            T rd = fma(cur.bp, pp, fma(cur.by, py, cur.bx * px));
            // G_s from G_{s+1}
            gp = fma(-cur.a23, gv, gp);
            gx = fma(-cur.a03, gv, fma(-cur.a02, gp, gx));
            gy = fma(-cur.a13, gv, fma(-cur.a12, gp, gy));
            const T cross = fma(cur.mvd, gv, cur.mpd * gp);
            rd += lane < 2 * s ? cross : (T)0;   // G_s is exactly zero in columns 2s, 2s+1 (what the lanes hold there is not)
            if (lane <= 2 * s + 1 && lane < n) { colK[2 * s] = ra; colK[2 * s + 1] = rd; }
            // p(s) = A_s^T p(s+1) + W_s G_s  (states 1 .. N-1 carry the speed weight; p(0) is never used)
            pv = fma(cur.a23, pp, fma(cur.a13, py, fma(cur.a03, px, pv)));
            pp = fma(cur.a12, py, fma(cur.a02, px, pp));
            px = fma(Cx2, gx, px);
            py = fma(Cy2, gy, py);
            pp = fma(cur.mpv, gv, fma(Cp2 + cur.mpp, gp, pp));
            pv = fma(cur.mpv, gp, fma(Cv2, gv, pv));
            cur = nxt;
        }
        WSYNC();
    }
    DEV void condense(T sc, acc_t (&acc)[NTT])
    {
        if constexpr (MODEL == 1) condense_frenet(sc, acc);   // dense stage Jacobians: MFMA contraction into `acc`
        else condense_adjoint(sc);                            // Cartesian model: into the packed image, `acc` untouched
    }

    // KKT tiles for the factorisation, built in registers: K = sc*(H + input Hessian) + A^T W A + reg*I in MFMA C-layout tiles (lower
    // 16x16 tiles; the strict upper part of the diagonal tiles is never read), with the rhs -sc*g as row n.
    // Needs stage_form_weights(w) done (wb = form weights, cb = suffix sums of the speed weights).
    // The dense part is the accumulators; A^T W A is structured: the speed rows add dt^2 * S[row/2] to every (even,even) entry --
    // (even,even) is a per-lane property in the C layout -- and the box / rate rows touch only the diagonal and the (j+2, j)
    // entries, which lane j computes together with the input-cost Hessian (MKZMPCPathFollower.jl:99-102) and hands over through
    // a 2 x n staging buffer.  The second-order ODD rows come from the packed image, where condense put them.
    static constexpr int NTF = (n + 1 + 15) / 16, NTTF = NTF * (NTF + 1) / 2;
    static constexpr bool ADJ = MODEL == 0;  // the whole lower triangle of sc*H is in the packed image (condense_adjoint), nothing in kt
    // In place: on entry the first NTT tiles of kt are the accumulators of condense (same packed lower-triangular tile order).
    DEV void build_tiles(T sc, T reg, acc_t (&kt)[NTTF])
    {
        const int c = lane & 15;
        const T dt2 = pt[PT_DT2];
        T *dgs = cub, *sbs = clb;  // the corrector buffers are dead between the accepted step and the end of the factorisation
        if (lane < n) {
            const int j = lane, jj = j & 1, k = j >> 1;
            const T Cu2 = cwt[jj ? 7 : 6], Cdl2 = cwt[jj ? 5 : 4];
            T dg = wb[j] + sc * (Cu2 + Cdl2 * (T)((k > 0) + (k < N - 1))) + reg;
            if (ADJ && jj) dg += sc * lin[LSTR * k + 12];  // m_dd of stage k: the second-order (d_f, d_f) entry
            if (j < 2) dg += wb[n + j];
            if (j >= 4) dg += wb[n + j - 2];
            const bool rate = j >= 2 && j < R;
            const T wr = rate ? wb[n + j] : (T)0;
            dgs[j] = dg + wr;
            sbs[j] = -wr - sc * Cdl2;
        }
        WSYNC();
#pragma unroll
        for (int tj = 0; tj < NTF; ++tj) {
            const int col = 16 * tj + c;
            const bool colok = col < n;
            const int cs_ = colok ? col : 0;
            const T dgv = dgs[cs_], sbv = sbs[cs_], rhv = -sc * gb[cs_];
            const T *colK = Lc + offc_rt(cs_);
#pragma unroll
            for (int ti = tj; ti < NTF; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + Real<T>::row_of(lane, r);
                    const bool oddrow = row & 1;  // fp64 C layout: a property of the lane; fp32: of the register index
                    // boundary conditions are spelled per tile position so that interior tiles carry no selects for them
                    T v = (T)0;
                    if (ti < NT) {
                        const T evm = (!oddrow && !(c & 1)) ? dt2 : (T)0;
                        v = evm * cb[(ti == NTF - 1 ? (row < n ? row : 0) : row) >> 1];
                        if (!ADJ) v = fma(sc, kt[ti * (ti + 1) / 2 + tj][r], v);
                        bool rd = ADJ || oddrow;
                        if (ti == NTF - 1) rd = rd && row < n;
                        if (ti == tj) rd = rd && col <= row;
                        if (tj == NTF - 1) rd = rd && colok;
                        const T *src = rd ? colK + row : pt;  // pt[0] is finite: the product with the zero mask below is exact
                        const T odm = rd ? (T)1 : (T)0;
                        v = fma(odm, *src, v);
                        if (ti == tj) v += row == col ? dgv : (T)0;
                        if (ti <= tj + 1) v += row == col + 2 ? sbv : (T)0;
                    }
                    if (ti == NTF - 1) { v = row == n ? rhv : v; v = row <= n ? v : (T)0; }
                    if (tj == NTF - 1) v = colok ? v : (T)0;
                    kt[ti * (ti + 1) / 2 + tj][r] = v;
                }
            // one tile column at a time: keeps the scheduler from stretching every tile's live range over the whole build
#pragma unroll
            for (int ti = tj; ti < NTF; ++ti) asm volatile("" : "+v"(kt[ti * (ti + 1) / 2 + tj]));
        }
        WFENCE();
    }

    // ---- blocked Cholesky on the matrix cores -----------------------------------------------------------------
    // K (with the rhs as row n) is held in MFMA C-layout tiles kt[] (lower 16x16 tiles, diagonal tiles symmetric).
    // One block-step takes a 4-column panel j0..j0+3:
    //   1. the lanes holding those columns put them into the LDS scratch pan[row][4];
    //   2. every lane reads the 4x4 diagonal block (uniform addresses) and factors it redundantly (4 rsqrt chains);
    //   3. lane (kk = lane>>4, c = lane&15) solves the panel rows 16t + c against it -- x L_dd^T = a, 10 FMAs -- keeps
    //      component kk: exactly the A/B fragment layout of the 16x16x4 MFMA -- and stores it as column j0+kk of L;
    //   4. trailing tiles -= P P^T, one MFMA per live tile.
    // 2N/4 block-steps of ~170 instructions replace 2N rank-1 column steps; dead rows are zeroed in the fragments,
    // so finished entries are never touched again.
    template <int TJ> DEV bool chol_blocks(acc_t (&kt)[NTTF], int jb0, int jb1)
    {
        const int c = lane & 15, kk = lane >> 4;
        T *pan = opb;
#pragma nounroll
        for (int jb = jb0; jb < jb1; ++jb) {
            const int j0 = 4 * jb, kp = c - (j0 & 15);
            const bool holder = kp >= 0 && kp < 4;
            if (holder) {  // one exec-masked region instead of an address select per store
                T *dst = pan + 4 * Real<T>::row_of(lane, 0) + kp;
#pragma unroll
                for (int ti = TJ; ti < NTF; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dst[4 * (16 * ti + Real<T>::row_of(0, r))] = kt[ti * (ti + 1) / 2 + TJ][r];
            }
            WFENCE();
            const T *pd = pan + 4 * j0;
            const T d00 = pd[0], d10 = pd[4], d11 = pd[5], d20 = pd[8], d21 = pd[9], d22 = pd[10];
            const T d30 = pd[12], d31 = pd[13], d32 = pd[14], d33 = pd[15];
            // panel rows are independent of the diagonal factor until the solve: issue their loads now
            T a[NTF][4];
#pragma unroll
            for (int t = TJ; t < NTF; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) a[t][k] = pan[4 * (16 * t + c) + k];
            const T r0 = rsqrt_(d00);
            const T l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
            const T e11 = fma(-l10, l10, d11), r1 = rsqrt_(e11);
            const T l21 = fma(-l20, l10, d21) * r1, l31 = fma(-l30, l10, d31) * r1;
            const T e22 = fma(-l21, l21, fma(-l20, l20, d22)), r2 = rsqrt_(e22);
            const T l32 = fma(-l31, l21, fma(-l30, l20, d32)) * r2;
            const T e33 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33))), r3 = rsqrt_(e33);
            const T dmin = fmin(fmin(d00, e11), fmin(e22, e33)), dmax = fmax(fmax(d00, e11), fmax(e22, e33));
            if (!(dmin > Real<T>::tiny() && dmax < (T)1e300)) return false;  // not positive definite (wave-uniform)
            // block-LDL^T view of the same factor: K = L~ S L~^T with L~ = L D^-1 (D = blockdiag of the 4x4 diagonal factors, so L~ has
            // unit diagonal blocks) and S^-1 = D^-T D^-1.  The substitutions then have no dependency inside a block (4 readlanes + 4
            // FMAs per block step) and the rhs row n comes out as S^-1 L~^-1 rhs.  inv = D_j^-1 (lower triangular):
            const T i10 = -l10 * r0 * r1;
            const T i21 = -l21 * r1 * r2, i20 = -fma(l21, i10, l20 * r0) * r2;
            const T i32 = -l32 * r2 * r3, i31 = -fma(l32, i21, l31 * r1) * r3, i30 = -fma(l32, i20, fma(l31, i10, l30 * r0)) * r3;
            if (lane == 0) {  // D_j^-1, row-major 4x4 (the strict upper part stays zero from construction)
                T *sq = sinvb + 16 * jb;
                sq[0] = r0; sq[4] = i10; sq[5] = r1; sq[8] = i20; sq[9] = i21; sq[10] = r2; sq[12] = i30; sq[13] = i31; sq[14] = i32; sq[15] = r3;
            }
            // column kk of inv, for this lane's component of L~: read back from the block just stored (4 LDS reads instead of 9 selects;
            // they feed only the L~ stores, which are off the panel -> MFMA critical path)
            WFENCE();
            const T *sqc = sinvb + 16 * jb + kk;
            const T c0 = sqc[0], c1 = sqc[4], c2 = sqc[8], c3 = sqc[12];
            T pf[NTF];
            const int jc = j0 + kk;
            T *colL = Lc + offc_rt(jc < n ? jc : 0);
#pragma unroll
            for (int t = TJ; t < NTF; ++t) {
                const int row = 16 * t + c;
                const T x0 = a[t][0] * r0;
                const T x1 = fma(-x0, l10, a[t][1]) * r1;
                const T x2 = fma(-x1, l21, fma(-x0, l20, a[t][2])) * r2;
                const T x3 = fma(-x2, l32, fma(-x1, l31, fma(-x0, l30, a[t][3]))) * r3;
                const T xs = kk == 0 ? x0 : (kk == 1 ? x1 : (kk == 2 ? x2 : x3));
                const bool live = row >= jc && row <= n;
                pf[t] = live ? xs : (T)0;                       // component kk of L (MFMA fragment of the trailing update)
                if (live) colL[row] = fma(x3, c3, fma(x2, c2, fma(x1, c1, x0 * c0)));  // component kk of L~ = L D^-1
            }
            if (j0 + 4 < n) {
                const int tmin = (j0 + 4) >> 4;  // first tile column that still has live entries
#pragma unroll
                for (int tjj = TJ; tjj < NTF; ++tjj)
                    if (tjj >= tmin) {
                        const T nb = -pf[tjj];
#pragma unroll
                        for (int ti = tjj; ti < NTF; ++ti)
                            kt[ti * (ti + 1) / 2 + tjj] = Real<T>::mfma(pf[ti], nb, kt[ti * (ti + 1) / 2 + tjj]);
                    }
            }
        }
        return true;
    }

    DEV bool factor(acc_t (&kt)[NTTF])
    {
        static_assert(n % 16 == 0 || n % 16 == 8, "block-steps are split at 16-column tile boundaries");
        bool ok = chol_blocks<0>(kt, 0, (n < 16 ? n : 16) / 4);
        STAMP(12);
        if (NTF > 1 && n > 16) { if (ok) ok = chol_blocks<(NTF > 1 ? 1 : 0)>(kt, 4, (n < 32 ? n : 32) / 4); }
        STAMP(13);
        if (NTF > 2 && n > 32) { if (ok) ok = chol_blocks<(NTF > 2 ? 2 : 0)>(kt, 8, (n < 48 ? n : 48) / 4); }
        if (NTF > 3 && n > 48) { if (ok) ok = chol_blocks<(NTF > 3 ? 3 : 0)>(kt, 12, n / 4); }
        STAMP(14);
        if (!ok) return false;
        WFENCE();
        return true;
    }

    // Substitutions on the block-LDL^T factor (lane j <-> component j): K^-1 b = L~^-T S^-1 L~^-1 b.  L~ has unit 4x4 diagonal
    // blocks, so a block step is 4 v_readlane + 4 FMAs on the lanes below (above) the block -- n/4 dependent steps instead of n.
    DEV T fwd_subst(T b)  // L~ y = b
    {
        T wv = lane < n ? b : (T)0;
        T lr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) lr[k] = Lc[offc_rt(k) + lane];  // L~[lane][j0+k]
#pragma unroll 2
        for (int jb = 0; jb < n / 4 - 1; ++jb) {
            const int j0 = 4 * jb;
            T ln[4];  // next block's coefficients are in flight while this block's chain runs
#pragma unroll
            for (int k = 0; k < 4; ++k) ln[k] = Lc[offc_rt(j0 + 4 + k) + lane];
            const T t0 = readlane_(wv, j0), t1 = readlane_(wv, j0 + 1), t2 = readlane_(wv, j0 + 2), t3 = readlane_(wv, j0 + 3);
            const T upd = fma(lr[3], t3, lr[2] * t2) + fma(lr[1], t1, lr[0] * t0);
            wv = (lane >= j0 + 4 && lane < n) ? wv - upd : wv;
#pragma unroll
            for (int k = 0; k < 4; ++k) lr[k] = ln[k];
        }
        return wv;
    }
    DEV T back_subst(T z)  // L~^T x = z
    {
        const T *pc = Lc + offc_rt(lane < n ? lane : 0);  // column `lane` of L~: L~[j0+k][lane] = pc[j0+k]
        T wv = lane < n ? z : (T)0;
        T lr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) lr[k] = pc[n - 4 + k];
#pragma unroll 2
        for (int jb = n / 4 - 1; jb >= 1; --jb) {
            const int j0 = 4 * jb;
            T ln[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ln[k] = pc[j0 - 4 + k];
            const T t0 = readlane_(wv, j0), t1 = readlane_(wv, j0 + 1), t2 = readlane_(wv, j0 + 2), t3 = readlane_(wv, j0 + 3);
            const T upd = fma(lr[3], t3, lr[2] * t2) + fma(lr[1], t1, lr[0] * t0);
            wv = lane < j0 ? wv - upd : wv;
#pragma unroll
            for (int k = 0; k < 4; ++k) lr[k] = ln[k];
        }
        return wv;
    }
    DEV T diag_solve(T y)  // S^-1 y = D^-T (D^-1 y): every 4x4 block at once, operands of the quad through DPP quad_perm
    {
        const int a = lane & 3;
        const T *blk = sinvb + 16 * ((lane < n ? lane : 0) >> 2);
        const T *dr = blk + 4 * a;   // row a of D^-1 (zero above the diagonal)
        const T *dc = blk + a;       // column a of D^-1 (zero above the diagonal): dc[4 m]
        const T y0 = dpp_mov0<0x00, 0xf>(y), y1 = dpp_mov0<0x55, 0xf>(y), y2 = dpp_mov0<0xaa, 0xf>(y), y3 = dpp_mov0<0xff, 0xf>(y);
        const T u = fma(dr[3], y3, dr[2] * y2) + fma(dr[1], y1, dr[0] * y0);
        const T u0 = dpp_mov0<0x00, 0xf>(u), u1 = dpp_mov0<0x55, 0xf>(u), u2 = dpp_mov0<0xaa, 0xf>(u), u3 = dpp_mov0<0xff, 0xf>(u);
        const T zz = fma(dc[12], u3, dc[8] * u2) + fma(dc[4], u1, dc[0] * u0);
        return lane < n ? zz : (T)0;
    }

    DEV bool interior_point(T &Uf)
    {
        const T relax = pt[PT_RELAX], dt = pt[PT_DT], dtc = pt[PT_DTC];
        const T steer_max = pt[PT_STEER_MAX], a_max = pt[PT_A_MAX], steer_dmax = pt[PT_STEER_DMAX], a_dmax = pt[PT_A_DMAX];
        const T v_min = pt[PT_V_MIN], v_max = pt[PT_V_MAX];
        // first guess of the solution inside the bounds (same rule as the CPU checker): accelerations approach
        // the reference speed (time constant 1 s), steering the kinematic feed-forward of the reference's mean curvature;
        // reference points 1..N only -- point 0 is a dead input (Q3)
        const T frac = (T)0.6, rr = pt[PT_RR];
        T len, kap;
        {
            const T rxn = __shfl_down(rx, 1), ryn = __shfl_down(ry, 1);
            const T seg = (lane >= 1 && lane < N) ? sqrt((rxn - rx) * (rxn - rx) + (ryn - ry) * (ryn - ry)) : (T)0;
            len = dpp_sum(seg);
            kap = (readlane_(rp, N) - readlane_(rp, 1)) / fmax(len, (T)1e-6);
        }
        if (MODEL == 1) kap = ((kp0 * x0 + kp1) * x0 + kp2) * x0 + kp3;  // Frenet: curvature of the polynomial at s0
        const T vref = MODEL == 1 ? vt : len / ((T)(N - 1) * dt);
        const T sb = fmin(fmax(pt[PT_LB] * kap, (T)-0.9), (T)0.9);
        const T dff = fmin(fmax(atan(sb * rsqrt_((T)1 - sb * sb) / rr)  /* tan(asin(sb)) = sb / sqrt(1 - sb^2), |sb| <= 0.9 */, -frac * steer_max), frac * steer_max);
        const T aff = fmin(fmax(vref - v0, -frac * a_max), frac * a_max);
        T u0[2];
        // Q5: v[1] = v0 is itself bounded in the reference model -> any v0 outside the (relaxed) speed bounds is infeasible
        bool ok = v0 >= v_min - relax * fmax((T)1, fabs(v_min)) && v0 <= v_max + relax * fmax((T)1, fabs(v_max));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const T ub = j ? steer_max : a_max;
            const T d0 = (j ? steer_dmax : a_dmax) * dtc;
            const T up = j ? up1 : up0;
            T lo = fmax(-ub - relax * fmax((T)1, ub), up - d0 - relax * fmax((T)1, d0));
            T hi = fmin(ub + relax * fmax((T)1, ub), up + d0 + relax * fmax((T)1, d0));
            if (j == 0) {
                lo = fmax(lo, (v_min - relax * fmax((T)1, fabs(v_min)) - v0) / dt);
                hi = fmin(hi, (v_max + relax * fmax((T)1, fabs(v_max)) - v0) / dt);
            }
            if (!(lo < hi)) ok = false;
            const T push = (T)0.25 * (hi - lo);
            u0[j] = fmin(fmax(j ? dff : aff, lo + push), hi - push);
        }
        const T vm = fmin((T)1, (T)0.25 * (v_max - v_min)), acap = (T)0.5 * a_max;
        const T astep = frac * a_dmax * dt, dstep = frac * steer_dmax * dt;
        T v = v0 + dt * u0[0], ap = u0[0], dp = u0[1];
        Uf = lane == 0 ? u0[0] : (lane == 1 ? u0[1] : (T)0);
#pragma nounroll
        for (int k = 1; k < N; ++k) {  // uniform scalar recurrence
            T a = fmin(fmax(vref - v, -frac * a_max), frac * a_max);
            a = fmin(fmax(a, ap - astep), ap + astep);
            if (v + dt * a < v_min + vm) a = fmin(v_min + vm - v, acap);
            else if (v + dt * a > v_max - vm) a = fmax(v_max - vm - v, -acap);
            const T d = fmin(fmax(dff, dp - dstep), dp + dstep);
            if (lane == 2 * k) Uf = a;
            if (lane == 2 * k + 1) Uf = d;
            v += dt * a; ap = a; dp = d;
        }
        return ok;
    }

    // The whole solve as one small state machine (one call site per phase -- see the file header):
    //   TRIAL : Ut was just evaluated; Armijo-test it (the very first point and refactor passes skip the test)
    //   after acceptance: duals, linearise, optimality test, mu, condense, factor, direction, first trial
    //   FINAL : last evaluation, for the predicted states, then exit
    DEV void solve(const KIO<T> &io, int b)
    {
        const T kappa_eps = 10, kappa_mu = (T)0.2, tau_min = (T)0.99, kappa_sigma = (T)1e10, eta_phi = (T)1e-8, s_max = 100;
        // the integer options are copied out of the kernel arguments once (the argument tuple is not touched inside the loop)
        const int max_ls = P.max_ls, max_iter = P.max_iter, indef_cfg = P.indef_strategy;
        const bool warm = P.warm != 0;
        const bool exact = P.hessian == 1;
        T U, Ut, du = 0;
        // slacks are iterates, advanced by s -/+ alpha * a_f^T du (as in Ipopt): recomputing b - a_f^T U would lose 7 digits to
        // cancellation once an active slack is ~1e-9; corrector terms are parked in LDS (cub/clb)
        T sup[NF], slo[NF], isu[NF], isl[NF], lu[NF], ll[NF], aut[NF], w[NF];  // isu/isl = 1/slack, refreshed when the slacks move
        bool fv[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) { const int f = lane + 64 * i; fv[i] = f < nf; lu[i] = ll[i] = sup[i] = slo[i] = isu[i] = isl[i] = aut[i] = (T)0; }
        int status = 1, iters = 0, ls = 0, attempt = 0, n_polish = 0, n_accept = 0, gn_hold = 0;
        enum { C_ERR = 0, C_RDS, C_DWL, C_DWS, C_HMAX, C_MUF, C_PHI0, C_DPHI, C_AD, C_J, C_LGS, C_JP };
        cs[C_ERR] = (T)1e30; cs[C_RDS] = 0; cs[C_DWL] = 0; cs[C_DWS] = 0; cs[C_HMAX] = 0; cs[C_AD] = 0; cs[C_J] = 0; cs[C_JP] = (T)1e30;
        int indef = indef_cfg == 2 ? 0 : indef_cfg, n_fail = 0;  // 2 = hybrid: GN fallback, delta_w shift from the 2nd failure on
        bool have_best = false;
        T mu = pt[warm ? PT_WARM_MU : PT_MU_INIT], sc = 1, Jt = 0, alpha = 0, reg = 0;
        bool use_exact = exact;
        enum { FIRST = 0, TRIAL = 1, REFACTOR = 2, FINAL = 3, RESTEP = 4 };
        const bool pc = P.mu_strategy == 1;
        bool corr_active = false, first_attempt = true, tiny_stop = false;
        int n_tiny = 0, n_flat = 0;
        bool final_reuse = false;  // FINAL reached with St / Jt already holding the evaluation of the returned iterate
#pragma unroll
        for (int i = 0; i < NF; ++i) cub[lane + 64 * i] = clb[lane + 64 * i] = (T)0;
        int mode = FIRST;
        StageF<T> St;
        STAMP_DECL

        {
            T Uf;
            const bool feas = interior_point(Uf);
            if (!feas) {
                status = 2;
                const T ub = pt[(lane & 1) ? PT_STEER_MAX : PT_A_MAX];
                U = lane < n ? fmin(fmax((lane & 1) ? up1 : up0, -ub), ub) : (T)0;
                mode = FINAL;
            } else if (warm && io.warmU) {
                const T dw = lane < n ? io.warmU[(size_t)b * n + lane] - Uf : (T)0;
                forms_apply(Uf, w);
                forms_apply(dw, aut);
                T th = 1;
#pragma unroll
                for (int i = 0; i < NF; ++i)
                    if (fv[i]) {
                        T bu_, bl_;
                        form_bounds(lane + 64 * i, bu_, bl_);
                        if (aut[i] > 0) th = fmin(th, (bu_ - w[i]) / aut[i]);
                        if (aut[i] < 0) th = fmin(th, (bl_ + w[i]) / -aut[i]);
                    }
                th = dpp_min(th) * ((T)1 - pt[PT_WARM_PUSH]);
                U = Uf + th * dw;
            } else U = Uf;
        }
        Ut = U;
        STAMP(0);
#pragma nounroll
        for (;;) {
            asm volatile("" : "+v"(lane));
            if (mode == FINAL && have_best && !tiny_stop && !(status == 0 && cs[C_ERR] <= pt[PT_TOL])) {
                // any later trouble (polishing noise, line-search failure, iteration cap) returns the iterate that passed
                Ut = ubest[lane]; U = Ut; status = 0; final_reuse = false;
            }
            // refactor / restep passes re-use the linearisation of U; a stop decided on the iterate that was just evaluated re-uses that too
            if (mode != REFACTOR && mode != RESTEP && !final_reuse) Jt = eval(Ut, St);
            STAMP(9);
            if (mode == FINAL) break;
            if (mode == TRIAL) {
                // sum of log(slack) over the forms of this lane: one log of the product (fp64 range is ample; fp32 takes one per register)
                T lgt = 0, lpr = 1;
                bool okp = true;
#pragma unroll
                // (no validity guard: the unused forms of the last register carry s = 1, ds = 0, lambda = 0, 1/s := 0 throughout, so they
                // contribute exact zeros / ones to every sum, product and maximum below -- see the start-point block)
                for (int i = 0; i < NF; ++i) {
                    const T a_ = sup[i] - alpha * aut[i], b_ = slo[i] + alpha * aut[i];
                    const bool pos = a_ > 0 && b_ > 0;
                    okp = okp && pos;
                    if (sizeof(T) == 8) lpr *= pos ? a_ * b_ : (T)1;
                    else lgt += log_pos(pos ? a_ * b_ : (T)1, kc);
                }
                if (sizeof(T) == 8) lgt = log_pos(lpr, kc);
                okp = __all(okp);
                const T slg = dpp_sum(lgt);
                const T phi = sc * Jt - mu * slg;
                const T phi0 = cs[C_PHI0];
                if (!(okp && phi - phi0 - (T)10 * Real<T>::eps() * fabs(phi0) <= eta_phi * alpha * cs[C_DPHI])) {
                    // safeguard: the corrected direction is tried at the full step only; redo the step without the corrector term
                    if (corr_active) { mode = RESTEP; Ut = U; continue; }
                    if (++ls >= max_ls) {
                        status = cs[C_ERR] <= pt[PT_TOL_X100] ? 0 : 3; mode = FINAL; Ut = U; continue;  // acceptable level
                    }
                    alpha *= (T)0.5;
                    Ut = U + alpha * du;
                    continue;
                }
                // Ipopt's tiny-step rule: two accepted steps in a row below 10 eps relative to the iterate -> the arithmetic cannot improve
                // it; Optimal if the error is within 1e3 tol (where the rounding floor of the fp32 dual residual sits), else Error
                {
                    const T stepn = dpp_max_nn(fabs(alpha * du)), umax = fmax((T)1, dpp_max_nn(fabs(U)));
                    n_tiny = stepn <= (T)10 * Real<T>::eps() * umax ? n_tiny + 1 : 0;
                    if (n_tiny >= 2) { U = Ut; status = cs[C_ERR] <= pt[PT_TOL_X1000] ? 0 : 3; tiny_stop = true; mode = FINAL; final_reuse = true; continue; }
                }
                // accepted: dual step from the pre-step slacks, then the slacks advance with the step
                cs[C_LGS] = slg;  // = sum log(slack) of the new iterate: the next barrier value re-uses it
                const T ad = cs[C_AD];
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const T su = sup[i], sl = slo[i];
                    lu[i] += ad * ((mu - cub[lane + 64 * i] - lu[i] * su) * isu[i] + lu[i] * isu[i] * aut[i]);
                    ll[i] += ad * ((mu - clb[lane + 64 * i] - ll[i] * sl) * isl[i] - ll[i] * isl[i] * aut[i]);
                    sup[i] = su - alpha * aut[i];
                    slo[i] = sl + alpha * aut[i];
                    isu[i] = fv[i] ? rcp_(sup[i]) : (T)0; isl[i] = fv[i] ? rcp_(slo[i]) : (T)0;
                }
            }
            const bool restep = mode == RESTEP;
            if (!restep) {
            if (mode != REFACTOR) {
            U = Ut; cs[C_J] = Jt;
            const T g = linearize(St, exact && gn_hold == 0);  // = use_exact of this iteration (set below, before gn_hold counts down)
            gb[lane] = g;
            STAMP(1);
                if (mode == FIRST) {
                    forms_apply(U, w);
#pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        T bu_, bl_;
                        form_bounds(lane + 64 * i, bu_, bl_);
                        sup[i] = bu_ - w[i]; slo[i] = bl_ + w[i];
                        isu[i] = fv[i] ? (T)1 / sup[i] : (T)0; isl[i] = fv[i] ? (T)1 / slo[i] : (T)0;
                    }
                    {
                        T lg0 = 0;
#pragma unroll
                        for (int i = 0; i < NF; ++i) if (fv[i]) lg0 += log_pos(sup[i] * slo[i], kc);
                        cs[C_LGS] = dpp_sum(lg0);
                    }
                    const T gm = dpp_max_nn(fabs(g));
                    sc = gm > (T)100 ? (T)100 / gm : (T)1;  // Ipopt nlp_scaling_max_gradient
#pragma unroll
                    for (int i = 0; i < NF; ++i) { lu[i] = mu * isu[i]; ll[i] = mu * isl[i]; }
                } else {
#pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        lu[i] = fmax(fmin(lu[i], kappa_sigma * mu * isu[i]), mu * isu[i] * ((T)1 / kappa_sigma));
                        ll[i] = fmax(fmin(ll[i], kappa_sigma * mu * isl[i]), mu * isl[i] * ((T)1 / kappa_sigma));
                    }
                }
                if (iters >= max_iter) { mode = FINAL; Ut = U; final_reuse = true; continue; }  // status stays ITERATION_LIMIT
                ++iters;
                // optimality error (Ipopt's scaled test + unscaled duality-gap bound)
#pragma unroll
                for (int i = 0; i < NF; ++i) w[i] = lu[i] - ll[i];
                const T rd = sc * g + forms_applyT(w);
                T lsum = 0, cm0 = 0, gap = 0;
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const T cu = sup[i] * lu[i], cl = slo[i] * ll[i];
                    lsum += lu[i] + ll[i]; gap += cu + cl; cm0 = fmax(cm0, fmax(cu, cl));
                }
                const T rdm = dpp_max_nn(fabs(rd));
                lsum = dpp_sum(lsum); cm0 = dpp_max_nn(cm0); gap = dpp_sum(gap);
                const T inv2nf = pt[PT_INV2NF];
                const T isd = s_max * rcp_(fmax(s_max, lsum * inv2nf));  // 1 / s_d
                const T err0 = fmax(rdm, cm0) * isd;
                const T tol = pt[PT_TOL];
                const T gap_lim = pt[PT_GAP_TOL] * fmax((T)1, fabs(Jt));
                cs[C_ERR] = err0; cs[C_RDS] = rdm * isd;
                TRACE8(io.stamps, iters, err0, rdm * isd, cm0 * isd, mu, Jt, alpha, ls, (use_exact ? 1 : 0) + 2 * indef + 4 * (int)corr_active + 8 * n_tiny);
                // Ipopt's test (+ gap bound, pursued for at most 1 more iteration once Ipopt's test is met), or
                // Ipopt's "acceptable level" (error <= 100*tol for 15 iterations in a row)
                bool done = false;
                if (err0 <= tol) { ubest[lane] = U; have_best = true; }  // last iterate passing Ipopt's test
                if (err0 <= tol) {
                    if (gap <= gap_lim * sc || n_polish >= 1) done = true; else ++n_polish;
                } else if (n_polish > 0 && ++n_polish > 1) done = true;
                n_accept = err0 <= pt[PT_TOL_X100] ? n_accept + 1 : 0;
                // rounding floor: the objective has not moved by more than 20 eps |J| for 12 iterations in a row -> the arithmetic cannot
                // improve the iterate (fp32, large costs: the dual residual never settles below 100 tol); Optimal within 1e3 tol
                n_flat = fabs(Jt - cs[C_JP]) <= (T)20 * Real<T>::eps() * fmax((T)1, fabs(Jt)) ? n_flat + 1 : 0;
                cs[C_JP] = Jt;
                if (n_flat >= 12 && err0 <= pt[PT_TOL_X1000]) done = true;
                if (done || n_accept >= 15) { status = 0; mode = FINAL; Ut = U; final_reuse = true; continue; }
                const T mu_min = fmax(pt[PT_TOL_D100], fmin(pt[PT_TOL_D10], (T)0.1 * gap_lim * sc * inv2nf));
                cs[C_MUF] = mu_min;
#pragma nounroll
                for (; !pc;) {  // monotone barrier update (mu_strategy 0)
                    T cmu = 0;
#pragma unroll
                    for (int i = 0; i < NF; ++i)
                        if (fv[i]) cmu = fmax(cmu, fmax(fabs(sup[i] * lu[i] - mu), fabs(slo[i] * ll[i] - mu)));
                    cmu = dpp_max_nn(cmu);
                    if (fmax(rdm, cmu) * isd <= kappa_eps * mu && mu > mu_min) mu = fmax(mu_min, fmin(kappa_mu * mu, mu * sqrt(mu)));
                    else break;
                }
                use_exact = exact && gn_hold == 0; reg = 0; attempt = 0;  // GN is held for 2 iterations after an indefinite exact Hessian
                if (gn_hold > 0) --gn_hold;
                // in shift mode the previous iteration's delta_w / 3 is the first trial (dropped below 1e-9 * max|H_jj|)
                if (use_exact && indef == 1 && cs[C_DWS] > (T)0) { reg = cs[C_DWS] / (T)3; if (reg < (T)1e-9 * cs[C_HMAX]) reg = 0; }
                first_attempt = true;
                STAMP(2);
            }
            // K = sc*H + A^T Sigma A with the affine right-hand side -sc*g riding along as row n
#pragma unroll
            for (int i = 0; i < NF; ++i) w[i] = lu[i] * isu[i] + ll[i] * isl[i];
            stage_form_weights(w);
            STAMP(6);
            bool factored;
            {
                acc_t kt[NTTF];  // condense accumulates into the first NTT tiles; build_tiles turns them into K in place
                acc_t (&acc)[NTT] = reinterpret_cast<acc_t (&)[NTT]>(kt);
                condense(sc, acc);
                if (ADJ) {
                    if (use_exact && indef == 1 && first_attempt)  // max |sc * H_jj|: scale of the delta_w shift
                        cs[C_HMAX] = dpp_max_nn(lane < n ? fabs(Lc[offc_rt(lane) + lane]) : (T)0);
                } else if (use_exact && indef == 1 && first_attempt) {  // max |sc * H_jj| over the diagonal of the tiles
                    T hm = 0;
#pragma unroll
                    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (Real<T>::row_of(lane, r) == (lane & 15)) hm = fmax(hm, fabs(sc * acc[ti * (ti + 1) / 2 + ti][r]));
                    cs[C_HMAX] = dpp_max_nn(hm);
                }
                first_attempt = false;
                STAMP(3);
                build_tiles(sc, reg, kt);
                STAMP(4);
                factored = factor(kt);
            }
            STAMP(5);
            if (!factored) {
                // Indefinite exact Hessian: strategy 0 -> Gauss-Newton for this and the next 2 iterations; 1 -> Ipopt's inertia
                // correction K + delta_w*I, delta_w = 1e-2*max|sc*H_jj| (x10) the first time, last/3 (x8) afterwards;
                // 2 -> 0 until the second failure, 1 from then on (Gauss-Newton leaves a saddle only slowly)
                if (++attempt >= 40) { status = 3; mode = FINAL; Ut = U; final_reuse = true; continue; }
                if (use_exact && indef == 1) {
                    const T hmax = cs[C_HMAX], dw_last = cs[C_DWL];
                    if (reg == (T)0) reg = dw_last > (T)0 ? fmax((T)1e-10 * hmax, dw_last / (T)3) : (T)1e-2 * hmax;
                    else reg *= dw_last > (T)0 ? (T)8 : (T)10;
                    if (reg > (T)1e2 * hmax) { use_exact = false; reg = 0; drop_second_order(); }
                } else if (use_exact) {
                    use_exact = false; gn_hold = 2; drop_second_order();
                    if (indef_cfg == 2 && ++n_fail >= 2) { indef = 1; gn_hold = 0; }
                } else reg = reg == (T)0 ? (T)1e-8 : reg * (T)100;  // last resort: shift the Gauss-Newton matrix
                mode = REFACTOR; Ut = U;
                continue;
            }
            if (use_exact && reg > (T)0) cs[C_DWL] = reg;
            if (use_exact) cs[C_DWS] = reg;
            // S^-1 L~^-1 (-sc*g) sits in row n of the factor image; re-read where needed rather than held in registers
#pragma unroll
            for (int i = 0; i < NF; ++i) cub[lane + 64 * i] = clb[lane + 64 * i] = (T)0;
            corr_active = false;
            if (pc) {
                // Mehrotra predictor: affine-scaling step on the same factor -> this iteration's barrier target
                const T dua = back_subst(lane < n ? Lc[offc_rt(lane) + n] : (T)0);
                forms_apply(dua, aut);
                // step lengths to the boundary as reciprocals: 1/alpha = max(1, max_f(-ds/s)); for the affine step -dlam/lam = 1 + ds/s
                T rpa = 1, rda = 1, mucur = 0, muaff = 0;
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const T qu = aut[i] * isu[i], ql = aut[i] * isl[i];  // -ds_u/s_u, ds_l/s_l
                    rpa = fmax(rpa, fmax(qu, -ql));
                    rda = fmax(rda, fmax((T)1 - qu, (T)1 + ql));
                    mucur += sup[i] * lu[i] + slo[i] * ll[i];
                }
                const T apa = rcp_(dpp_max_nn(rpa)), ada = rcp_(dpp_max_nn(rda));
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const T su = sup[i], sl = slo[i], dsu = -aut[i], dsl = aut[i];
                    const T dlu = -lu[i] - lu[i] * isu[i] * dsu, dll = -ll[i] - ll[i] * isl[i] * dsl;
                    muaff += (su + apa * dsu) * (lu[i] + ada * dlu) + (sl + apa * dsl) * (ll[i] + ada * dll);
                    cub[lane + 64 * i] = dsu * dlu; clb[lane + 64 * i] = dsl * dll;
                }
                mucur = dpp_sum(mucur) * pt[PT_INV2NF]; muaff = dpp_sum(muaff) * pt[PT_INV2NF];
                const T r3 = muaff * rcp_(mucur);
                mu = fmax(cs[C_MUF], fmin((T)1, r3 * r3 * r3) * mucur);
                mu = fmax(mu, fmin(mucur, cs[C_RDS] * (T)KMPC_IKRD));  // no barrier target far below the dual infeasibility
                corr_active = true;
                STAMP(7);
            }
            } else {  // RESTEP: same factor, corrector term dropped
#pragma unroll
                for (int i = 0; i < NF; ++i) cub[lane + 64 * i] = clb[lane + 64 * i] = (T)0;
                corr_active = false;
            }
            // centering (+ corrector) part of the step: du = K^{-1}(-sc*g - A^T((mu - corr)/s_u - (mu - corr)/s_l))
#pragma unroll
            for (int i = 0; i < NF; ++i) w[i] = -((mu - cub[lane + 64 * i]) * isu[i] - (mu - clb[lane + 64 * i]) * isl[i]);
            du = back_subst((lane < n ? Lc[offc_rt(lane) + n] : (T)0) + diag_solve(fwd_subst(forms_applyT(w))));
            STAMP(15);
            forms_apply(du, aut);
            const T tau = fmax(tau_min, (T)1 - mu);
            T rp = 0, rq = 0, gw = 0;
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if (fv[i]) {
                    const T su = sup[i], sl = slo[i], dsu = -aut[i], dsl = aut[i];
                    const T dlu = (mu - cub[lane + 64 * i] - lu[i] * su) * isu[i] - lu[i] * isu[i] * dsu;
                    const T dll = (mu - clb[lane + 64 * i] - ll[i] * sl) * isl[i] - ll[i] * isl[i] * dsl;
                    gw += mu * (isu[i] - isl[i]) * aut[i];
                    rp = fmax(rp, fmax(-dsu * isu[i], -dsl * isl[i]));
                    rq = fmax(rq, fmax(-dlu * rcp_(lu[i]), -dll * rcp_(ll[i])));
                }
            // fraction to the boundary: alpha = min(1, tau * min(-s/ds)) = tau / max(tau, max(-ds/s))
            const T ap = tau * rcp_(fmax(tau, dpp_max_nn(rp)));
            cs[C_AD] = tau * rcp_(fmax(tau, dpp_max_nn(rq)));
            cs[C_PHI0] = sc * cs[C_J] - mu * cs[C_LGS];
            cs[C_DPHI] = dpp_sum((lane < n ? sc * gb[lane] * du : (T)0) + gw);  // d/dalpha of phi_mu: (sc*g + A^T(mu/s_u - mu/s_l))^T du
            alpha = ap; ls = 0;
            Ut = U + alpha * du;
            mode = TRIAL;
            STAMP(8);
        }
        STAMP(10);
        // ---- outputs (St / Jt are the evaluation of the returned U) ------------------------------------
        forms_apply(U, w);
        T viol = -(T)1e30;
#pragma unroll
        for (int i = 0; i < NF; ++i)
            if (fv[i]) {
                const int f = lane + 64 * i;
                T bu_, bl_;
                form_bounds(f, bu_, bl_);
                viol = fmax(viol, fmax(w[i] - (bu_ - form_relax(f, true)), -w[i] - (bl_ - form_relax(f, false))));
            }
        viol = dpp_max(viol);
        if (lane < n) {
            if (io.outU) io.outU[(size_t)b * n + lane] = U;
            if (io.warmU) io.warmU[(size_t)b * n + lane] = U;
            if (lane < 2) io.u0[(size_t)b * 2 + lane] = U;
        }
        if (io.outX && lane <= N) {
            T *o = io.outX + ((size_t)b * (N + 1) + lane) * 4;
            o[0] = St.x + xoff; o[1] = St.y + yoff; o[2] = St.psi; o[3] = St.v;
        }
        STAMP(11);
        STAMP_OUT(io.stamps, b);
        if (lane == 0) {
            io.status[b] = status;
            if (io.cost) io.cost[b] = Jt;
            if (io.viol) io.viol[b] = viol;
            if (io.iters) io.iters[b] = iters;
        }
    }
};

// waves per SIMD: the fp64 kernel needs ~250 VGPRs to run without scratch spills (measured: at 128 VGPRs the
// spills moved 1.6 GB of HBM traffic per 4096-problem launch against 2.4 MB of algorithmic bytes); the kernel is
// issue-bound, not occupancy-bound, so 2 waves/SIMD without spills beats 4 with.  fp32 fits 4 waves spill-free up to N = 20 (128 VGPRs; +10 % over 3
// waves at N = 16 / 20) and 3 waves beyond.  fp64 at N <= 12: the LDS footprint (10.9 / 13.1 KB) admits 12 waves per CU and a wave issues at most
// one vector instruction per 6.5 cycles (tools/calib/issue_probe.hip), so a third wave per SIMD is worth the 36 / 60 B of scratch that 168
// VGPRs cost: +21 % / +16 % at B = 262 144, +7 % / +3 % at B = 4096.  From N = 16 the LDS footprint allows 9 waves per CU or fewer.
// fp32 at N <= 12 likewise takes a fifth wave (96 VGPRs, 0 / 8 B of scratch): +7 % / +6 % at B = 262 144.
// The spills cost single-wave latency (N = 8 closed loop, B = 1: 80 -> 86 us p50), so the denser build (kmpc_solve_fast_dense_kernel) is
// launched only for batches that fill the chip (B > 2048); small batches and the B = 1 latency path keep the spill-free build.
template <typename T, int N> DEV void kmpc_solve_fast_body(const KP &P, const KIO<T> &io)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[FastSolver<T, N>::lds_elems() * sizeof(T)];
    if ((int)blockIdx.x >= P.B) return;
    const int b = io.perm ? io.perm[blockIdx.x] : (int)blockIdx.x;
#ifdef KMPC_POISON  // diagnostic build (make poison): every LDS word starts as NaN, so a read of a word nobody wrote shows up in the results
    for (int e = threadIdx.x; e < FastSolver<T, N>::lds_elems(); e += 64) reinterpret_cast<T *>(smem)[e] = (T)NAN;
    __syncthreads();
#endif
    FastSolver<T, N> sv(P, smem);
    sv.load_problem(io.z0, io.ref, io.vt, io.up, b);
    sv.solve(io, b);
}
template <typename T, int N>
__global__ __launch_bounds__(64, sizeof(T) == 8 ? 2 : (N <= 20 ? 4 : 3)) void kmpc_solve_fast_kernel(KP P, KIO<T> io)
{
    kmpc_solve_fast_body<T, N>(P, io);
}
// the same solve at one more wave per SIMD (N <= 12, batches that fill the chip)
template <typename T, int N>
__global__ __launch_bounds__(64, sizeof(T) == 8 ? 3 : 5) void kmpc_solve_fast_dense_kernel(KP P, KIO<T> io)
{
    kmpc_solve_fast_body<T, N>(P, io);
}

// diagnostics (tests/test_gpu_kernels.py): the KKT pipeline of THIS kernel -- roll-out, costates, condensing, in-register KKT
// assembly, blocked Cholesky, block substitutions -- at a given point, form weights, scaling and shift
template <typename T, int N>
__global__ __launch_bounds__(64, sizeof(T) == 8 ? 2 : (N <= 20 ? 4 : 3)) void kmpc_fast_kkt_kernel(KP P, KDbgK<T> io)
{
    typedef FastSolver<T, N> SV;
    typedef typename SV::acc_t acc_t;
    constexpr int n = SV::n, nf = SV::nf;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SV::lds_elems() * sizeof(T)];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= P.B) return;
    SV sv(P, smem);
    sv.load_problem(io.z0, io.ref, io.vt, io.up, b);
    const T sc = (T)io.sc, reg = (T)io.reg;
    const T U = lane < n ? io.U[(size_t)b * n + lane] : (T)0;
    StageF<T> St;
    sv.eval(U, St);
    const T g = sv.linearize(St, P.hessian == 1);
    sv.gb[lane] = g;
    T w[SV::NF];
#pragma unroll
    for (int i = 0; i < SV::NF; ++i) { const int f = lane + 64 * i; w[i] = f < nf ? io.w[(size_t)b * nf + f] : (T)0; }
    sv.stage_form_weights(w);
    acc_t kt[SV::NTTF];
    acc_t (&acc)[SV::NTT] = reinterpret_cast<acc_t (&)[SV::NTT]>(kt);
    sv.condense(sc, acc);
    sv.build_tiles(sc, reg, kt);
    T *K = io.K + (size_t)b * n * n;
#pragma unroll
    for (int ti = 0; ti < SV::NTF; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + Real<T>::row_of(lane, r), col = 16 * tj + (lane & 15);
                if (row < n && col <= row) { K[row * n + col] = kt[ti * (ti + 1) / 2 + tj][r]; K[col * n + row] = kt[ti * (ti + 1) / 2 + tj][r]; }
            }
    const bool okf = sv.factor(kt);
    T x = (T)0;
    if (okf) x = sv.back_subst((lane < n ? sv.Lc[SV::offc_rt(lane) + n] : (T)0) + sv.diag_solve(sv.fwd_subst(lane < n ? io.b[(size_t)b * n + lane] : (T)0)));
    if (lane < n) { io.g[(size_t)b * n + lane] = g; io.x[(size_t)b * n + lane] = x; }
    if (lane == 0) io.ok[b] = okf ? 1 : 0;
}
template <typename T, int N>
static hipError_t launch_fast_kkt_n(const KP &P, const KDbgK<T> &io, hipStream_t st)
{
    hipLaunchKernelGGL((kmpc_fast_kkt_kernel<T, N>), dim3(P.B), dim3(64), 0, st, P, io);
    return hipGetLastError();
}
template <typename T> hipError_t kmpc_launch_fast_kkt(const KP &P, const KDbgK<T> &io, hipStream_t st)
{
    switch (P.N) {
        case 8: return launch_fast_kkt_n<T, 8>(P, io, st);
        case 12: return launch_fast_kkt_n<T, 12>(P, io, st);
        case 16: return launch_fast_kkt_n<T, 16>(P, io, st);
        case 20: return launch_fast_kkt_n<T, 20>(P, io, st);
        case 24: return launch_fast_kkt_n<T, 24>(P, io, st);
        case 28: return launch_fast_kkt_n<T, 28>(P, io, st);
        default: return hipErrorInvalidValue;
    }
}
template hipError_t kmpc_launch_fast_kkt<double>(const KP &, const KDbgK<double> &, hipStream_t);
template hipError_t kmpc_launch_fast_kkt<float>(const KP &, const KDbgK<float> &, hipStream_t);

template <typename T, int N>
static hipError_t launch_fast_n(const KP &P, const KIO<T> &io, hipStream_t st)
{
    if constexpr (N <= 12) {
        if (P.B > 2048) {
            hipLaunchKernelGGL((kmpc_solve_fast_dense_kernel<T, N>), dim3(P.B), dim3(64), 0, st, P, io);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((kmpc_solve_fast_kernel<T, N>), dim3(P.B), dim3(64), 0, st, P, io);
    return hipGetLastError();
}

// Frenet-frame functor (kmpc_config.model = 1): io.ref carries k_poly [B,4]  (fp64 at N = 28: 34.5 KB of LDS per wave leave one wave per
// SIMD anyway, so the bound says so and the allocator may use all 512 registers)
template <typename T, int N>
__global__ __launch_bounds__(64, sizeof(T) == 8 ? (N >= 28 ? 1 : (N <= 8 ? 3 : 2)) : (N <= 20 ? 4 : 3)) void kmpc_solve_fast_frenet_kernel(KP P, KIO<T> io)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[FastSolver<T, N, 1>::lds_elems() * sizeof(T)];
    if ((int)blockIdx.x >= P.B) return;
    const int b = io.perm ? io.perm[blockIdx.x] : (int)blockIdx.x;
#ifdef KMPC_POISON
    for (int e = threadIdx.x; e < FastSolver<T, N, 1>::lds_elems(); e += 64) reinterpret_cast<T *>(smem)[e] = (T)NAN;
    __syncthreads();
#endif
    FastSolver<T, N, 1> sv(P, smem);
    sv.load_problem(io.z0, io.ref, io.vt, io.up, b);
    sv.solve(io, b);
}
template <typename T, int N>
static hipError_t launch_fast_frenet_n(const KP &P, const KIO<T> &io, hipStream_t st)
{
    hipLaunchKernelGGL((kmpc_solve_fast_frenet_kernel<T, N>), dim3(P.B), dim3(64), 0, st, P, io);
    return hipGetLastError();
}
template <typename T> hipError_t kmpc_launch_solve_fast_frenet(const KP &P, const KIO<T> &io, hipStream_t st)
{
    switch (P.N) {
        case 8: return launch_fast_frenet_n<T, 8>(P, io, st);
        case 12: return launch_fast_frenet_n<T, 12>(P, io, st);
        case 16: return launch_fast_frenet_n<T, 16>(P, io, st);
        case 20: return launch_fast_frenet_n<T, 20>(P, io, st);
        case 24: return launch_fast_frenet_n<T, 24>(P, io, st);
        case 28: return launch_fast_frenet_n<T, 28>(P, io, st);
        default: return hipErrorInvalidValue;
    }
}
template hipError_t kmpc_launch_solve_fast_frenet<double>(const KP &, const KIO<double> &, hipStream_t);
template hipError_t kmpc_launch_solve_fast_frenet<float>(const KP &, const KIO<float> &, hipStream_t);

// horizons with a compiled fast kernel; everything else runs the generic kernel
// compile-time horizons: N % 4 == 0 and 2N + 1 <= 64
template <typename T> bool kmpc_fast_available(int N) { return N == 8 || N == 12 || N == 16 || N == 20 || N == 24 || N == 28; }
template <typename T> hipError_t kmpc_launch_solve_fast(const KP &P, const KIO<T> &io, hipStream_t st)
{
    switch (P.N) {
        case 8: return launch_fast_n<T, 8>(P, io, st);
        case 12: return launch_fast_n<T, 12>(P, io, st);
        case 16: return launch_fast_n<T, 16>(P, io, st);
        case 20: return launch_fast_n<T, 20>(P, io, st);
        case 24: return launch_fast_n<T, 24>(P, io, st);
        case 28: return launch_fast_n<T, 28>(P, io, st);
        default: return hipErrorInvalidValue;
    }
}
template bool kmpc_fast_available<double>(int);
template bool kmpc_fast_available<float>(int);
template hipError_t kmpc_launch_solve_fast<double>(const KP &, const KIO<double> &, hipStream_t);
template hipError_t kmpc_launch_solve_fast<float>(const KP &, const KIO<float> &, hipStream_t);
