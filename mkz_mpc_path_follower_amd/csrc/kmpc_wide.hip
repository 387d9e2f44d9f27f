// kmpc_wide.hip -- long-horizon solver kernel for gfx950: ONE WORKGROUP OF FOUR WAVES PER PROBLEM (BASELINE config 5, N = 50; also
// instantiated for N = 48 and, with six / five tile rows instead of seven, N = 44, 40 / 36, 32).
//
// Same algorithm and same results as the one-wave kernels (kmpc_fast.hip / kmpc_kernels.hip; reference NLP:
// scripts/mpc_utils/MKZMPCPathFollower.jl:65-123).  At N = 50 the condensed KKT matrix has n = 100 columns (+ the rhs row): 28
// lower 16x16 tiles = 224 fp64 accumulator registers -- one wave cannot hold them, and with the matrix in LDS (100 KB) only one
// wave fits on a CU (the generic kernel: 1.4 % of the fp64 peak).  Here
//   * the 28 tiles live in REGISTERS, spread over the four waves of a workgroup by tile ROW: wave w owns rows 6-w and w-1
//     (7 tiles = 56 VGPRs each).  Every loop over tiles has compile-time tile COLUMNS and a wave-uniform (SGPR) tile row, so all
//     waves run the same code;
//   * LDS holds only what crosses waves: the packed image of sc*H / the block-LDL^T factor L~ (n(n+3)/2 words, 41 KB), the
//     double-buffered 4-column Cholesky panel, the stage records, the terminal sensitivities and a few n-vectors: 72 KB -> two
//     workgroups (8 waves) per CU, 2 waves per SIMD, 256 VGPRs;
//   * condensing in O(N^2): thread j < n carries column j of the Hessian through the adjoint recursion of kmpc_fast.hip (no
//     barrier inside, no matrix product); the matrix cores do the trailing updates of the factorisation;
//   * one thread per linear form (nf = 5N-2 = 248 <= 256) and per input (n = 100): every elementwise pass of the interior-point
//     method is ONE instruction stream of NF = 1; stage vectors (roll-out, costates) are evaluated redundantly by every wave
//     (lane k = stage k), so their results need no exchange;
//   * workgroup barriers only where data crosses waves: one per Cholesky block-step, one per reduction; the three triangular
//     substitutions run in wave 0 with two slots per lane (v_readlane broadcasts, no barrier inside).
#include "kmpc_math.h"

#define WGSYNC() __syncthreads()

template <typename T> struct StageW {  // lane k: state k / input k at the evaluated point
    T a, d, v, x, y, psi, c, s, sinb, cosb, b1, b2, ex, ey, ep, ev;
};
constexpr int WLIN = 16;  // stage record stride: A02 A03 A12 A13 A23 Bdx Bdy Bdp mpp mpv mpd mvd mdd

#ifndef KMPC_IKRD
#define KMPC_IKRD 1e-3
#endif

template <typename T, int N> struct WideSolver {
    static constexpr int n = 2 * N, R = 2 * (N - 1), nf = 5 * N - 2;
    static constexpr int NTF = (n + 1 + 15) / 16;   // tile rows of K with the rhs row n
    static constexpr int NP = 16 * NTF;              // padded dimension
    static constexpr int SROWS = (N + 1 + 15) / 16;  // 16-lane rows that carry stage data
    static constexpr int LC = n * (n + 3) / 2;       // packed lower triangle + rhs row, column-major
    static constexpr int NB = n / 4;                 // 4-column block-steps
    static_assert(NTF >= 5 && NTF <= 7, "tile rows are dealt to the four waves as (6), (5,0), (4,1), (3,2) -- (5), (4,0), (3,1), (2) with six, (4), (3,0), (2), (1) with five");
    static_assert(nf <= 256 && n % 4 == 0 && N + 1 <= 64, "one thread per form, 4-column panels, one lane per stage");
    typedef typename Real<T>::acc_t acc_t;
    static constexpr int offc(int j) { return j * (n + 1) - j * (j - 1) / 2 - j; }
    static DEV int offc_rt(int j) { return j * (n + 1) - ((j * (j - 1)) >> 1) - j; }
    // LDS map (elements of T)
    static constexpr int O_LC = 0, O_OPB = (LC + 1) & ~1, O_LIN = O_OPB + 2 * 4 * NP + 256 + 3 * 128, O_XB = O_LIN + WLIN * (N + 1), O_WB = O_XB + 128,
                         O_CBW = O_WB + 256, O_RED = O_CBW + 4 * 64, O_SINV = O_RED + 2 * 32, O_X2 = O_SINV + 16 * NB, O_X3 = O_X2 + 128,
                         O_GB = O_X3 + 128, O_UB = O_GB + 128, O_CS = O_UB + 128, O_PT = O_CS + 4 * 16, O_KC = O_PT + 32, O_END = O_KC + (sizeof(T) == 8 ? KC_COUNT : 0);
    static constexpr int lds_elems() { return O_END; }

    STAMP_MEMBERS
    const KP &P;
    int tid, lane, wv;
    T *Lc, *opb, *pan, *dgs, *sbs, *gnb, *lin, *xb, *wb, *cb, *red, *sinvb, *x2, *x3, *gbl, *ubl, *cs;
    Coef<T> kc;
    const T *pt, *cwt;
    int rpar;
    T psi0, v0, vt, rx, ry, rp;   // (x0 = y0 = 0 in vehicle-centred coordinates; u_prev and the offsets are re-read where they are used)
    const T *z0p, *upp_;
    enum { PT_DT = 0, PT_DTC, PT_RR, PT_DT2, PT_DTL, PT_LB, PT_TOL, PT_GAP_TOL, PT_TOL_X100, PT_TOL_X1000, PT_TOL_D100, PT_TOL_D10,
           PT_STEER_MAX, PT_A_MAX, PT_STEER_DMAX, PT_A_DMAX, PT_W = 16, PT_V_MIN = 24, PT_V_MAX, PT_RELAX, PT_WARM_PUSH, PT_WARM_MU, PT_MU_INIT,
           PT_INV2NF };

    DEV WideSolver(const KP &p, unsigned char *smem) : P(p), tid(threadIdx.x), lane(threadIdx.x & 63), rpar(0)
    {
        wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        T *base = reinterpret_cast<T *>(smem);
        Lc = base + O_LC; opb = base + O_OPB; lin = base + O_LIN; xb = base + O_XB; wb = base + O_WB;
        cb = base + O_CBW + 64 * wv;  // every wave keeps its OWN copy of the stage prefix / suffix sums: no barrier to share them
        red = base + O_RED; sinvb = base + O_SINV; x2 = base + O_X2; x3 = base + O_X3; gbl = base + O_GB; ubl = base + O_UB;
        cs = base + O_CS + 16 * wv;    // wave-uniform scalars that are read once or twice per iteration: every wave parks its own copy
        pan = opb;                      // the double-buffered Cholesky panel (2 x 4 NP)
        dgs = opb + 2 * 4 * NP; sbs = dgs + 128;  // 2 x n staging of build_tiles
        gnb = sbs + 128;                // terminal sensitivities G_N [3][128], written by linearize, read by every condense of that linearisation
        for (int e = tid; e < 16 * NB; e += 256) sinvb[e] = (T)0;
        kc.tab = base + O_KC;
        if (sizeof(T) == 8 && tid < KC_COUNT) const_cast<T *>(kc.tab)[tid] = (T)kmpc_coef[tid];
        pt = base + O_PT; cwt = pt + PT_W;
        if (tid == 0) {
            T *q = base + O_PT;
            q[PT_DT] = (T)p.dt; q[PT_DTC] = (T)p.dtc; q[PT_RR] = (T)p.r; q[PT_DT2] = (T)p.dt2; q[PT_DTL] = (T)p.dt_over_Lb; q[PT_LB] = (T)p.L_b;
            q[PT_TOL] = (T)p.tol; q[PT_GAP_TOL] = (T)p.gap_tol; q[PT_TOL_X100] = (T)p.tol_x100; q[PT_TOL_X1000] = (T)p.tol_x1000;
            q[PT_TOL_D100] = (T)p.tol_d100; q[PT_TOL_D10] = (T)p.tol_d10;
            q[PT_STEER_MAX] = (T)p.steer_max; q[PT_A_MAX] = (T)p.a_max; q[PT_STEER_DMAX] = (T)p.steer_dmax; q[PT_A_DMAX] = (T)p.a_dmax;
            for (int i = 0; i < 8; ++i) q[PT_W + i] = (T)p.C2[i];
            q[PT_V_MIN] = (T)p.v_min; q[PT_V_MAX] = (T)p.v_max; q[PT_RELAX] = (T)p.relax; q[PT_WARM_PUSH] = (T)p.warm_push;
            q[PT_WARM_MU] = (T)p.warm_mu; q[PT_MU_INIT] = (T)p.mu_init; q[PT_INV2NF] = (T)(1.0 / (2 * nf));
        }
        WGSYNC();
    }

    DEV void load_problem(const T *z0, const T *ref, const T *vtp, const T *upp, int b)
    {
        // vehicle-centred coordinates (the NLP is translation-invariant; see kmpc_fast.hip)
        z0p = z0 + 4 * (size_t)b; upp_ = upp + 2 * (size_t)b;
        const T xoff = z0p[0], yoff = z0p[1];
        psi0 = z0p[2]; v0 = z0p[3];
        vt = vtp[b];
        rx = ry = rp = (T)0;
        if (lane <= N) {  // every wave keeps the reference at stage `lane`
            const T *r = ref + ((size_t)b * (N + 1) + lane) * 3;
            rx = r[0] - xoff; ry = r[1] - yoff; rp = r[2];
        }
    }

    // ---- workgroup reductions: NS sums and NM maxima in one exchange (one barrier; the scratch alternates) -------------------
    template <int NS, int NM> DEV void wg_reduce(T (&s)[NS < 1 ? 1 : NS], T (&m)[NM < 1 ? 1 : NM])
    {
        static_assert(NS + NM <= 8, "8 slots per wave");
        T *r = red + 32 * (rpar & 1);
        ++rpar;
#pragma unroll
        for (int i = 0; i < NS; ++i) s[i] = dpp_sum(s[i]);
#pragma unroll
        for (int i = 0; i < NM; ++i) m[i] = dpp_max(m[i]);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NS; ++i) r[8 * wv + i] = s[i];
#pragma unroll
            for (int i = 0; i < NM; ++i) r[8 * wv + NS + i] = m[i];
        }
        WGSYNC();
#pragma unroll
        for (int i = 0; i < NS; ++i) s[i] = (r[i] + r[8 + i]) + (r[16 + i] + r[24 + i]);  // same order in every wave: bit-identical
#pragma unroll
        for (int i = 0; i < NM; ++i) m[i] = fmax(fmax(r[NS + i], r[8 + NS + i]), fmax(r[16 + NS + i], r[24 + NS + i]));
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
            const T u = kk == 0 ? upp_[jj] : (T)0;
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

    // y_f = a_f^T x   (thread j < n holds x_j; thread f holds y_f)
    DEV T forms_apply(T x)
    {
        if (tid < n) xb[tid] = x;
        WGSYNC();
        T a = lane < N ? xb[2 * lane] : (T)0;
        a = dpp_scan_prefix<SROWS>(a);
        cb[lane] = a;
        WFENCE();
        const int f = tid;
        T v = (T)0;
        if (f < n) v = xb[f];
        else if (f < n + R) { const int r = f - n; v = r < 2 ? xb[r] : xb[r + 2] - xb[r]; }
        else if (f < nf) v = pt[PT_DT] * cb[f - n - R];
        WGSYNC();
        return v;
    }
    DEV void stage_form_weights(T w)
    {
        if (tid < nf) wb[tid] = w;
        WGSYNC();
        T s = lane < N ? wb[n + R + lane] : (T)0;
        s = dpp_scan_suffix<SROWS>(s, lane);
        cb[lane] = s;
        WFENCE();
    }
    DEV T forms_applyT(T w)  // returns (A^T w)_j in thread j
    {
        stage_form_weights(w);
        T o = (T)0;
        const int j = tid;
        if (j < n) {
            o = wb[j];
            if (j < 2) o += wb[n + j];
            if (j >= 4) o += wb[n + j - 2];
            if (j >= 2 && j < R) o -= wb[n + j];
            if (!(j & 1)) o += pt[PT_DT] * cb[j >> 1];
        }
        WGSYNC();
        return o;
    }

    // roll-out (MKZMPCPathFollower.jl:115-122 as prefix scans) + objective (:97-103) at U (thread j: U_j); every wave evaluates
    // all stages (lane k = stage k), so the stage data and the cost are in every wave without an exchange
    DEV T eval(T U, StageW<T> &S)
    {
        const T dt = pt[PT_DT], rr_ = pt[PT_RR], dtL = pt[PT_DTL];
        if (tid < n) xb[tid] = U;
        WGSYNC();
        const int k = lane;
        const bool st = k < N;
        const T a = st ? xb[2 * k] : (T)0, d = st ? xb[2 * k + 1] : (T)0;
        const T an = (k + 1 < N) ? xb[2 * k + 2] : a, dn = (k + 1 < N) ? xb[2 * k + 3] : d;
        S.a = a; S.d = d;
        const T ia = dpp_scan_prefix<SROWS>(a);
        const T v = v0 + dt * (ia - a);
        T sd, cd;
        sincos_small(d, &sd, &cd, kc);
        const T Dn = cd * cd + rr_ * rr_ * sd * sd;
        const T rs = rsqrt_(Dn);
        S.sinb = rr_ * sd * rs;
        S.cosb = cd * rs;
        const T iD = rs * rs;
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
        S.x = dt * (ix - wx);
        S.y = dt * (iy - wy);
        S.v = v; S.psi = psi;
        const bool cs = (k >= 1 && k <= N);
        S.ex = cs ? S.x - rx : (T)0;
        S.ey = cs ? S.y - ry : (T)0;
        S.ep = cs ? psi - rp : (T)0;
        S.ev = (k >= 1 && k <= N - 1) ? v - vt : (T)0;
        const T Cx2 = cwt[0], Cy2 = cwt[1], Cp2 = cwt[2], Cv2 = cwt[3], Cda2 = cwt[4], Cdd2 = cwt[5], Ca2 = cwt[6], Cd2 = cwt[7];
        T Jl = Cx2 * S.ex * S.ex + Cy2 * S.ey * S.ey + Cp2 * S.ep * S.ep + Cv2 * S.ev * S.ev;
        if (st) Jl += Ca2 * a * a + Cd2 * d * d;
        if (k < N - 1) Jl += Cda2 * (an - a) * (an - a) + Cdd2 * (dn - d) * (dn - d);
        Jl *= (T)0.5;
        const T J = dpp_sum(Jl);
        WGSYNC();  // xb is free again
        return J;
    }

    // costates by suffix scans -> gradient (returned, thread j: g_j; also left in gbl); stage records go to LDS (wave 0 writes)
    DEV T linearize(const StageW<T> &S, bool exact)
    {
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
        // terminal sensitivities for condense_adjoint (closed form in suffix sums of the unit upper-triangular stage Jacobians: kmpc_fast.hip)
        const T Rs = dpp_scan_suffix<SROWS>(A23, lane);
        const T ux = A03 - A02 * Rs, uy = A13 - A12 * Rs;
        const T X2 = dpp_scan_suffix<SROWS>(A02, lane) - A02, Y2 = dpp_scan_suffix<SROWS>(A12, lane) - A12;
        const T zx = dpp_scan_suffix<SROWS>(ux, lane) - ux, zy = dpp_scan_suffix<SROWS>(uy, lane) - uy;
        const T P3 = Rs - A23;
        if (wv == 0) {
            if (st) {
                T *q = gnb + 2 * k;
                q[0] = dt * fma(P3, X2, zx); q[128] = dt * fma(P3, Y2, zy); q[256] = dt * P3;   // acceleration column: B = (0, 0, 0, dt)
                q[1] = fma(X2, Bdp, Bdx); q[129] = fma(Y2, Bdp, Bdy); q[257] = Bdp;               // steering column
            }
            if (st) { gbl[2 * k] = ga; gbl[2 * k + 1] = gd; }
            if (k <= N) {
                T *q = lin + WLIN * k;
                q[0] = A02; q[1] = A03; q[2] = A12; q[3] = A13; q[4] = A23; q[5] = Bdx; q[6] = Bdy; q[7] = Bdp;
                q[8] = mpp; q[9] = mpv; q[10] = mpd; q[11] = mvd; q[12] = mdd; q[13] = (T)0;
            }
        }
        WGSYNC();
        return tid < n ? gbl[tid] : (T)0;
    }
    DEV void drop_second_order()
    {
        if (tid <= N) { T *q = lin + WLIN * tid; q[8] = (T)0; q[9] = (T)0; q[10] = (T)0; q[11] = (T)0; q[12] = (T)0; }
        WGSYNC();
    }

    // ---- condensing ----------------------------------------------------------------------------------------------------------------------
    struct Rec { T a02, a03, a12, a13, a23, bx, by, bp, mpp, mpv, mpd, mvd, mdd; };
    DEV void load_rec(Rec &r, int s) const
    {
        const T *q = lin + WLIN * s;  // record N is all zero (linearize)
        r.a02 = q[0]; r.a03 = q[1]; r.a12 = q[2]; r.a13 = q[3]; r.a23 = q[4]; r.bx = q[5]; r.by = q[6]; r.bp = q[7];
        r.mpp = q[8]; r.mpv = q[9]; r.mpd = q[10]; r.mvd = q[11]; r.mdd = q[12];
    }
    // Condensing in O(N^2): thread j < n carries column j of sc * H through the adjoint recursion (kmpc_fast.hip, condense_adjoint) and
    // writes rows >= j of it into the packed image; no barrier inside (the stage records and the G_N table were published by linearize).
    DEV void condense_adjoint(T sc)
    {
        if (tid < n) {
            const T dtv = pt[PT_DT];
            const T Cx2 = cwt[0], Cy2 = cwt[1], Cp2 = cwt[2], Cv2 = cwt[3];
            T gx = sc * gnb[tid], gy = sc * gnb[128 + tid], gp = sc * gnb[256 + tid], gv = (tid & 1) ? (T)0 : sc * dtv;
            T px = Cx2 * gx, py = Cy2 * gy, pp = Cp2 * gp, pv = (T)0;   // p(N) = W_N G_N
            T *colK = Lc + offc_rt(tid);
            Rec cur;
            load_rec(cur, N - 1);
#pragma unroll 2
            for (int s = N - 1; s >= 0; --s) {
                Rec nxt;
                load_rec(nxt, s > 0 ? s - 1 : 0);
                const T ra = dtv * pv;
                T rd = fma(cur.bp, pp, fma(cur.by, py, cur.bx * px));
                gp = fma(-cur.a23, gv, gp);   // G_s from G_{s+1}: exact inverse of the unit upper-triangular A_s
                gx = fma(-cur.a03, gv, fma(-cur.a02, gp, gx));
                gy = fma(-cur.a13, gv, fma(-cur.a12, gp, gy));
                const T cross = fma(cur.mvd, gv, cur.mpd * gp);
                rd += tid < 2 * s ? cross : (T)0;
                if (tid <= 2 * s + 1) { colK[2 * s] = ra; colK[2 * s + 1] = rd; }  // thread 2s+1's row 2s lands on (row n, column 2s): unread until the factor writes it
                pv = fma(cur.a23, pp, fma(cur.a13, py, fma(cur.a03, px, pv)));
                pp = fma(cur.a12, py, fma(cur.a02, px, pp));
                px = fma(Cx2, gx, px);
                py = fma(Cy2, gy, py);
                pp = fma(cur.mpv, gv, fma(Cp2 + cur.mpp, gp, pp));
                pv = fma(cur.mpv, gp, fma(Cv2, gv, pv));
                cur = nxt;
            }
        }
        WGSYNC();
    }
    // Tile rows of wave W (compile-time in everything below: each wave runs its own specialisation, selected once per
    // factorisation by a switch on the wave number; tile indices, liveness tests and register arrays are then all static)
    template <int W> struct Rows {   // wave W owns tile row NTF-1-W and, of the rows 0 .. NTF-5 that are nobody's first row, row W-1
        static constexpr int R0 = NTF - 1 - W, R1 = (W - 1 < NTF - 4) ? W - 1 : -1, N0 = R0 + 1, N1 = R1 >= 0 ? R1 + 1 : 1;
    };
    // ---- KKT tiles, in place: K = sc*(H + input Hessian) + A^T W A + reg*I, rhs -sc*g as row n ------------------------------------
    // (needs stage_form_weights(w) done: wb = form weights, cb = suffix sums of the speed weights)
    template <int NTL> DEV void build_row(T sc, int ti, acc_t (&kt)[NTL])
    {
        const int c = lane & 15;
        const T dt2 = pt[PT_DT2];
#pragma unroll
        for (int tj = 0; tj < NTL; ++tj) {
            if (tj <= ti) {
                const int col = 16 * tj + c;
                const bool colok = col < n;
                const int cs_ = colok ? col : 0;
                const T dgv = dgs[cs_], sbv = sbs[cs_], rhv = -sc * gbl[cs_];
                const T *colK = Lc + offc_rt(cs_);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + Real<T>::row_of(lane, r);
                    T v = (T)0;
                    if (colok && row < n && col <= row) {
                        const T evm = (!(row & 1) && !(col & 1)) ? dt2 : (T)0;
                        v = fma(evm, cb[row >> 1], colK[row]);
                        if (row == col) v += dgv;
                        if (row == col + 2) v += sbv;
                    } else if (colok && row == n) v = rhv;
                    kt[tj][r] = v;
                }
            }
        }
    }
    template <int W> DEV void build_tiles(T sc, T reg, acc_t (&k0)[Rows<W>::N0], acc_t (&k1)[Rows<W>::N1])
    {
        constexpr int R0 = Rows<W>::R0, R1 = Rows<W>::R1;
        if (tid < n) {
            const int j = tid, jj = j & 1, k = j >> 1;
            const T Cu2 = cwt[jj ? 7 : 6], Cdl2 = cwt[jj ? 5 : 4];
            T dg = wb[j] + sc * (Cu2 + Cdl2 * (T)((k > 0) + (k < N - 1))) + reg;
            if (jj) dg += sc * lin[WLIN * k + 12];  // m_dd of stage k: the second-order (d_f, d_f) entry
            if (j < 2) dg += wb[n + j];
            if (j >= 4) dg += wb[n + j - 2];
            const bool rate = j >= 2 && j < R;
            const T wr = rate ? wb[n + j] : (T)0;
            dgs[j] = dg + wr;
            sbs[j] = -wr - sc * Cdl2;
        }
        WGSYNC();
        build_row<Rows<W>::N0>(sc, R0, k0);
        if (R1 >= 0) build_row<Rows<W>::N1>(sc, R1, k1);
        WGSYNC();  // dgs / sbs / the odd rows of the image have been consumed: the panel and the factor may overwrite them
    }

    // ---- blocked Cholesky on the matrix cores (block-LDL^T form, see kmpc_fast.hip), one barrier per 4-column block-step ---------
    template <int W> DEV void extract_panel(int jb, const acc_t (&k0)[Rows<W>::N0], const acc_t (&k1)[Rows<W>::N1])
    {
        constexpr int R0 = Rows<W>::R0, R1 = Rows<W>::R1;
        const int c = lane & 15, j0 = 4 * jb, tcol = j0 >> 4, kp = c - (j0 & 15);
        T *pn = pan + (jb & 1) * 4 * NP;
        // the tile of column tcol is picked by VALUE selects: a branch per tile column ends, after the optimiser's block merging, in a
        // phi over the addresses of the tile registers, which keeps every tile in scratch memory
        acc_t v0 = k0[0], v1 = k1[0];
#pragma unroll
        for (int tc = 1; tc <= R0; ++tc) {
            const bool hit = tcol == tc;
#pragma unroll
            for (int r = 0; r < 4; ++r) v0[r] = hit ? k0[tc][r] : v0[r];
        }
#pragma unroll
        for (int tc = 1; tc <= R1; ++tc) {
            const bool hit = tcol == tc;
#pragma unroll
            for (int r = 0; r < 4; ++r) v1[r] = hit ? k1[tc][r] : v1[r];
        }
        if (kp >= 0 && kp < 4) {
            if (tcol <= R0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pn[4 * (16 * R0 + Real<T>::row_of(lane, r)) + kp] = v0[r];
            }
            if (R1 >= 0 && tcol <= R1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pn[4 * (16 * R1 + Real<T>::row_of(lane, r)) + kp] = v1[r];
            }
        }
    }
    template <int W> DEV bool factor(acc_t (&k0)[Rows<W>::N0], acc_t (&k1)[Rows<W>::N1])
    {
        constexpr int R0 = Rows<W>::R0, R1 = Rows<W>::R1;
        const int c = lane & 15, kk = lane >> 4;
        extract_panel<W>(0, k0, k1);
        WGSYNC();
        bool ok = true;
#pragma nounroll
        for (int jb = 0; jb < NB; ++jb) {
            const int j0 = 4 * jb, tcol = j0 >> 4;
            const T *pn = pan + (jb & 1) * 4 * NP;
            const T *pd = pn + 4 * j0;
            const T d00 = pd[0], d10 = pd[4], d11 = pd[5], d20 = pd[8], d21 = pd[9], d22 = pd[10];
            const T d30 = pd[12], d31 = pd[13], d32 = pd[14], d33 = pd[15];
            const T r0 = rsqrt_(d00);
            const T l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
            const T e11 = fma(-l10, l10, d11), r1 = rsqrt_(e11);
            const T l21 = fma(-l20, l10, d21) * r1, l31 = fma(-l30, l10, d31) * r1;
            const T e22 = fma(-l21, l21, fma(-l20, l20, d22)), r2 = rsqrt_(e22);
            const T l32 = fma(-l31, l21, fma(-l30, l20, d32)) * r2;
            const T e33 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33))), r3 = rsqrt_(e33);
            const T dmin = fmin(fmin(d00, e11), fmin(e22, e33)), dmax = fmax(fmax(d00, e11), fmax(e22, e33));
            if (!(dmin > Real<T>::tiny() && dmax < (T)1e300)) { ok = false; break; }  // not positive definite (same data in every wave)
            const T i10 = -l10 * r0 * r1;
            const T i21 = -l21 * r1 * r2, i20 = -fma(l21, i10, l20 * r0) * r2;
            const T i32 = -l32 * r2 * r3, i31 = -fma(l32, i21, l31 * r1) * r3, i30 = -fma(l32, i20, fma(l31, i10, l30 * r0)) * r3;
            if (tid == 0) {  // D_j^-1, row-major 4x4 (the strict upper part stays zero from construction)
                T *sq = sinvb + 16 * jb;
                sq[0] = r0; sq[4] = i10; sq[5] = r1; sq[8] = i20; sq[9] = i21; sq[10] = r2; sq[12] = i30; sq[13] = i31; sq[14] = i32; sq[15] = r3;
            }
            // column kk of D_j^-1 for this lane's component of L~
            const T c0 = kk == 0 ? r0 : (T)0;
            const T c1 = kk == 0 ? i10 : (kk == 1 ? r1 : (T)0);
            const T c2 = kk == 0 ? i20 : (kk == 1 ? i21 : (kk == 2 ? r2 : (T)0));
            const T c3 = kk == 0 ? i30 : (kk == 1 ? i31 : (kk == 2 ? i32 : r3));
            const int jc = j0 + kk;
            T *colL = Lc + offc_rt(jc < n ? jc : 0);
            T pf[NTF];
#pragma unroll
            for (int t = 0; t < NTF; ++t) {
                const int row = 16 * t + c;
                pf[t] = (T)0;
                if (!(t <= R0 || (t & 3) == W)) continue;   // neither a B fragment of this wave's tiles nor a tile row whose L~ entries it stores
                if (t < tcol) continue;                      // finished tile rows (wave-uniform)
                const T *ar = pn + 4 * row;
                const T x0_ = ar[0] * r0;
                const T x1_ = fma(-x0_, l10, ar[1]) * r1;
                const T x2_ = fma(-x1_, l21, fma(-x0_, l20, ar[2])) * r2;
                const T x3_ = fma(-x2_, l32, fma(-x1_, l31, fma(-x0_, l30, ar[3]))) * r3;
                const T xs = kk == 0 ? x0_ : (kk == 1 ? x1_ : (kk == 2 ? x2_ : x3_));
                const bool live = row >= jc && row <= n;
                pf[t] = live ? xs : (T)0;                       // component kk of L (B fragment of the trailing update)
                if (live && (t & 3) == W) colL[row] = fma(x3_, c3, fma(x2_, c2, fma(x1_, c1, x0_ * c0)));  // component kk of L~ = L D^-1
            }
            // the A fragment of tile row R is the B fragment of tile column R (same rows of the solved panel)
            const T pa0 = pf[R0], pa1 = pf[R1 >= 0 ? R1 : 0];
            if (j0 + 4 < n) {
                const int tmin = (j0 + 4) >> 4;  // first tile column that still has live entries
                if (R0 >= tmin) {
#pragma unroll
                    for (int t = 0; t <= R0; ++t)
                        if (t >= tmin) k0[t] = Real<T>::mfma(pa0, -pf[t], k0[t]);
                }
                if (R1 >= tmin) {
#pragma unroll
                    for (int t = 0; t <= R1; ++t)
                        if (t >= tmin) k1[t] = Real<T>::mfma(pa1, -pf[t], k1[t]);
                }
                extract_panel<W>(jb + 1, k0, k1);
            }
            WGSYNC();
        }
        return ok;
    }

    // condense + (max |sc H_jj|) + KKT assembly + factorisation for the tile rows of wave W
    template <int W> DEV bool assemble_factor(T sc, T reg, bool want_hmax, T &hmax, T *Kdump = nullptr)
    {
        constexpr int R0 = Rows<W>::R0, R1 = Rows<W>::R1;
        acc_t k0[Rows<W>::N0], k1[Rows<W>::N1];
        build_tiles<W>(sc, reg, k0, k1);
        STAMP(4);
        if (Kdump) {  // diagnostics only (kmpc_debug_kkt): the assembled matrix, full symmetric n x n
#pragma unroll
            for (int tj = 0; tj <= R0; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * R0 + Real<T>::row_of(lane, r), col = 16 * tj + (lane & 15);
                    if (row < n && col <= row) { Kdump[row * n + col] = k0[tj][r]; Kdump[col * n + row] = k0[tj][r]; }
                }
#pragma unroll
            for (int tj = 0; tj <= R1; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * R1 + Real<T>::row_of(lane, r), col = 16 * tj + (lane & 15);
                    if (col <= row) { Kdump[row * n + col] = k1[R1 >= 0 ? tj : 0][r]; Kdump[col * n + row] = k1[R1 >= 0 ? tj : 0][r]; }
                }
        }
        return factor<W>(k0, k1);
    }

    // ---- substitutions on the block-LDL^T factor, in wave 0 with two slots per lane (j = lane, 64 + lane) ----------------------------
    DEV void fwd_subst2(T &w0, T &w1)  // L~ y = b
    {
        const bool v1 = 64 + lane < n;
#pragma nounroll
        for (int jb = 0; jb < 16; ++jb) {  // blocks in slot 0
            const int j0 = 4 * jb;
            T l0[4], l1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { l0[k] = Lc[offc_rt(j0 + k) + lane]; l1[k] = Lc[offc_rt(j0 + k) + (v1 ? 64 + lane : n)]; }
            const T t0 = readlane_(w0, j0), t1 = readlane_(w0, j0 + 1), t2 = readlane_(w0, j0 + 2), t3 = readlane_(w0, j0 + 3);
            const T u0 = fma(l0[3], t3, l0[2] * t2) + fma(l0[1], t1, l0[0] * t0);
            const T u1 = fma(l1[3], t3, l1[2] * t2) + fma(l1[1], t1, l1[0] * t0);
            w0 = lane >= j0 + 4 ? w0 - u0 : w0;
            w1 = v1 ? w1 - u1 : w1;
        }
#pragma nounroll
        for (int jb = 16; jb < NB - 1; ++jb) {  // blocks in slot 1
            const int j0 = 4 * jb;
            T l1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) l1[k] = Lc[offc_rt(j0 + k) + (v1 ? 64 + lane : n)];
            const int q = j0 - 64;
            const T t0 = readlane_(w1, q), t1 = readlane_(w1, q + 1), t2 = readlane_(w1, q + 2), t3 = readlane_(w1, q + 3);
            const T u1 = fma(l1[3], t3, l1[2] * t2) + fma(l1[1], t1, l1[0] * t0);
            w1 = (v1 && 64 + lane >= j0 + 4) ? w1 - u1 : w1;
        }
    }
    DEV void back_subst2(T &w0, T &w1)  // L~^T x = z
    {
        const bool v1 = 64 + lane < n;
        const T *pc0 = Lc + offc_rt(lane), *pc1 = Lc + offc_rt(v1 ? 64 + lane : 0);
#pragma nounroll
        for (int jb = NB - 1; jb >= 16; --jb) {  // blocks in slot 1: rows j0 .. j0+3 of columns i < j0
            const int j0 = 4 * jb, q = j0 - 64;
            T l0[4], l1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { l0[k] = pc0[j0 + k]; l1[k] = pc1[j0 + k]; }
            const T t0 = readlane_(w1, q), t1 = readlane_(w1, q + 1), t2 = readlane_(w1, q + 2), t3 = readlane_(w1, q + 3);
            const T u0 = fma(l0[3], t3, l0[2] * t2) + fma(l0[1], t1, l0[0] * t0);
            const T u1 = fma(l1[3], t3, l1[2] * t2) + fma(l1[1], t1, l1[0] * t0);
            w0 -= u0;
            w1 = (v1 && 64 + lane < j0) ? w1 - u1 : w1;
        }
#pragma nounroll
        for (int jb = 15; jb >= 1; --jb) {  // blocks in slot 0
            const int j0 = 4 * jb;
            T l0[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) l0[k] = pc0[j0 + k];
            const T t0 = readlane_(w0, j0), t1 = readlane_(w0, j0 + 1), t2 = readlane_(w0, j0 + 2), t3 = readlane_(w0, j0 + 3);
            const T u0 = fma(l0[3], t3, l0[2] * t2) + fma(l0[1], t1, l0[0] * t0);
            w0 = lane < j0 ? w0 - u0 : w0;
        }
    }
    DEV T diag_solve1(T y, int j)  // S^-1 y = D^-T (D^-1 y) for the 4x4 block of component j (quad of lanes)
    {
        const int a = lane & 3;
        const T *blk = sinvb + 16 * ((j < n ? j : 0) >> 2);
        const T *dr = blk + 4 * a, *dc = blk + a;
        const T y0_ = dpp_mov0<0x00, 0xf>(y), y1_ = dpp_mov0<0x55, 0xf>(y), y2_ = dpp_mov0<0xaa, 0xf>(y), y3_ = dpp_mov0<0xff, 0xf>(y);
        const T u = fma(dr[3], y3_, dr[2] * y2_) + fma(dr[1], y1_, dr[0] * y0_);
        const T u0 = dpp_mov0<0x00, 0xf>(u), u1 = dpp_mov0<0x55, 0xf>(u), u2 = dpp_mov0<0xaa, 0xf>(u), u3 = dpp_mov0<0xff, 0xf>(u);
        const T zz = fma(dc[12], u3, dc[8] * u2) + fma(dc[4], u1, dc[0] * u0);
        return j < n ? zz : (T)0;
    }
    // K^-1 (rhs + b) per thread j, where rhs = -sc*g rides in row n of the factor image (already through L~ and S):
    // x = L~^-T (row_n + S^-1 L~^-1 b); with_b = false gives the affine-scaling direction
    DEV T solve_dir(T b, bool with_b)
    {
        if (with_b) { if (tid < 128) x2[tid] = tid < n ? b : (T)0; }
        WGSYNC();
        if (wv == 0) {
            const bool v1 = 64 + lane < n;
            T z0 = Lc[offc_rt(lane) + n], z1 = v1 ? Lc[offc_rt(64 + lane) + n] : (T)0;
            if (with_b) {
                T b0 = x2[lane], b1 = x2[64 + lane];
                fwd_subst2(b0, b1);
                z0 += diag_solve1(b0, lane);
                z1 += diag_solve1(b1, 64 + lane);
            }
            back_subst2(z0, z1);
            x3[lane] = z0; x3[64 + lane] = v1 ? z1 : (T)0;
        }
        WGSYNC();
        return tid < n ? x3[tid] : (T)0;
    }

    DEV bool interior_point(T &Uf)
    {
        const T relax = pt[PT_RELAX], dt = pt[PT_DT], dtc = pt[PT_DTC];
        const T steer_max = pt[PT_STEER_MAX], a_max = pt[PT_A_MAX], steer_dmax = pt[PT_STEER_DMAX], a_dmax = pt[PT_A_DMAX];
        const T v_min = pt[PT_V_MIN], v_max = pt[PT_V_MAX];
        // first guess of the solution inside the bounds: same rule as kmpc_fast.hip / the CPU checker
        const T frac = (T)0.6, rr = pt[PT_RR];
        T len, kap;
        {
            const T rxn = __shfl_down(rx, 1), ryn = __shfl_down(ry, 1);
            const T seg = (lane >= 1 && lane < N) ? sqrt((rxn - rx) * (rxn - rx) + (ryn - ry) * (ryn - ry)) : (T)0;
            len = dpp_sum(seg);
            kap = (readlane_(rp, N) - readlane_(rp, 1)) / fmax(len, (T)1e-6);
        }
        const T vref = len / ((T)(N - 1) * dt);
        const T sb = fmin(fmax(pt[PT_LB] * kap, (T)-0.9), (T)0.9);
        const T dff = fmin(fmax(atan(sb * rsqrt_((T)1 - sb * sb) / rr), -frac * steer_max), frac * steer_max);
        const T aff = fmin(fmax(vref - v0, -frac * a_max), frac * a_max);
        T u0[2];
        bool ok = v0 >= v_min - relax * fmax((T)1, fabs(v_min)) && v0 <= v_max + relax * fmax((T)1, fabs(v_max));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const T ub = j ? steer_max : a_max;
            const T d0 = (j ? steer_dmax : a_dmax) * dtc;
            const T up = upp_[j];
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
        Uf = tid == 0 ? u0[0] : (tid == 1 ? u0[1] : (T)0);
#pragma nounroll
        for (int k = 1; k < N; ++k) {  // uniform scalar recurrence
            T a = fmin(fmax(vref - v, -frac * a_max), frac * a_max);
            a = fmin(fmax(a, ap - astep), ap + astep);
            if (v + dt * a < v_min + vm) a = fmin(v_min + vm - v, acap);
            else if (v + dt * a > v_max - vm) a = fmax(v_max - vm - v, -acap);
            const T d = fmin(fmax(dff, dp - dstep), dp + dstep);
            if (tid == 2 * k) Uf = a;
            if (tid == 2 * k + 1) Uf = d;
            v += dt * a; ap = a; dp = d;
        }
        return ok;
    }

    // The solve: the state machine of kmpc_fast.hip (FIRST / TRIAL / REFACTOR / RESTEP / FINAL), every decision taken on
    // workgroup-uniform values (identical in all four waves), so every barrier is reached by every thread.
    DEV void solve(const KIO<T> &io, int b)
    {
        const T kappa_eps = 10, kappa_mu = (T)0.2, tau_min = (T)0.99, kappa_sigma = (T)1e10, eta_phi = (T)1e-8, s_max = 100;
        const int max_ls = P.max_ls, max_iter = P.max_iter, indef_cfg = P.indef_strategy;
        const bool warm = P.warm != 0;
        const bool exact = P.hessian == 1;
        const bool fv = tid < nf;
        T U, Ut, du = 0;
        T sup = 0, slo = 0, isu = 0, isl = 0, lu = 0, ll = 0, aut = 0, cu_ = 0, cl_ = 0;  // one form per thread; cu_/cl_: corrector terms
        int status = 1, iters = 0, ls = 0, attempt = 0, n_polish = 0, n_accept = 0, gn_hold = 0;
        // wave-uniform scalars that are read once or twice per iteration live in this wave's LDS slots, not in VGPRs (every lane
        // stores the same value; a wave's LDS operations execute in program order)
        enum { C_ERR = 0, C_RDS, C_DWL, C_DWS, C_HMAX, C_MUF, C_PHI0, C_DPHI, C_AD, C_J, C_LGS, C_JP };
        cs[C_ERR] = (T)1e30; cs[C_RDS] = 0; cs[C_DWL] = 0; cs[C_DWS] = 0; cs[C_HMAX] = 0; cs[C_MUF] = 0; cs[C_PHI0] = 0; cs[C_DPHI] = 0;
        cs[C_AD] = 0; cs[C_J] = 0; cs[C_LGS] = 0; cs[C_JP] = (T)1e30;
        int indef = indef_cfg == 2 ? 0 : indef_cfg, n_fail = 0;
        bool have_best = false;
        T mu = pt[warm ? PT_WARM_MU : PT_MU_INIT], sc = 1, Jt = 0, alpha = 0, reg = 0;
        bool use_exact = exact;
        enum { FIRST = 0, TRIAL = 1, REFACTOR = 2, FINAL = 3, RESTEP = 4 };
        const bool pc = P.mu_strategy == 1;
        bool corr_active = false, first_attempt = true, tiny_stop = false;
        int n_tiny = 0, n_flat = 0;
        int mode = FIRST;
        StageW<T> St;
        STAMP_DECL
        {
            T Uf;
            const bool feas = interior_point(Uf);
            if (!feas) {
                status = 2;
                const T ub = pt[(tid & 1) ? PT_STEER_MAX : PT_A_MAX];
                U = tid < n ? fmin(fmax(upp_[tid & 1], -ub), ub) : (T)0;
                mode = FINAL;
            } else if (warm && io.warmU) {
                const T dw = tid < n ? io.warmU[(size_t)b * n + tid] - Uf : (T)0;
                const T w0 = forms_apply(Uf);
                const T a0 = forms_apply(dw);
                T th[1] = {(T)-1}, dummy[1] = {(T)0};
                if (fv) {
                    T bu_, bl_;
                    form_bounds(tid, bu_, bl_);
                    T t = 1;
                    if (a0 > 0) t = fmin(t, (bu_ - w0) / a0);
                    if (a0 < 0) t = fmin(t, (bl_ + w0) / -a0);
                    th[0] = -t;
                }
                wg_reduce<0, 1>(dummy, th);
                U = Uf + (-th[0]) * ((T)1 - pt[PT_WARM_PUSH]) * dw;
            } else U = Uf;
        }
        Ut = U;
        STAMP(0);
#pragma nounroll
        for (;;) {
            // re-materialised (opaque) at the top of every iteration: stops LICM from hoisting the index / mask / address arithmetic of
            // every phase out of the loop into registers that then live (and spill) across the whole solve
            asm volatile("" : "+v"(tid));
            lane = tid & 63;
            if (mode == FINAL && have_best && !tiny_stop && !(status == 0 && cs[C_ERR] <= pt[PT_TOL])) {
                Ut = ubl[tid & 127]; U = Ut; status = 0;   // (threads >= 128 pick up copies: their U is never read as an input -- every use is guarded by tid < n)
            }
            if (mode != REFACTOR && mode != RESTEP) Jt = eval(Ut, St);
            STAMP(9);
            if (mode == FINAL) break;
            if (mode == TRIAL) {
                const T a_ = sup - alpha * aut, b_ = slo + alpha * aut;
                const bool pos = a_ > 0 && b_ > 0;  // (unused threads carry s = 1, ds = 0)
                T sm[1] = {log_pos(pos ? a_ * b_ : (T)1, kc)};
                T mx[3] = {pos ? (T)0 : (T)1, fabs(alpha * du), fabs(U)};
                wg_reduce<1, 3>(sm, mx);
                const bool okp = mx[0] == (T)0;
                const T slg = sm[0];
                const T phi = sc * Jt - mu * slg;
                if (!(okp && phi - cs[C_PHI0] - (T)10 * Real<T>::eps() * fabs(cs[C_PHI0]) <= eta_phi * alpha * cs[C_DPHI])) {
                    if (corr_active) { mode = RESTEP; Ut = U; continue; }
                    if (++ls >= max_ls) { status = cs[C_ERR] <= pt[PT_TOL_X100] ? 0 : 3; mode = FINAL; Ut = U; continue; }
                    alpha *= (T)0.5;
                    Ut = U + alpha * du;
                    continue;
                }
                {
                    const T stepn = mx[1], umax = fmax((T)1, mx[2]);
                    n_tiny = stepn <= (T)10 * Real<T>::eps() * umax ? n_tiny + 1 : 0;
                    if (n_tiny >= 2) { U = Ut; status = cs[C_ERR] <= pt[PT_TOL_X1000] ? 0 : 3; tiny_stop = true; mode = FINAL; continue; }
                }
                cs[C_LGS] = slg;
                {
                    const T su = sup, sl = slo;
                    lu += cs[C_AD] * ((mu - cu_ - lu * su) * isu + lu * isu * aut);
                    ll += cs[C_AD] * ((mu - cl_ - ll * sl) * isl - ll * isl * aut);
                    sup = su - alpha * aut;
                    slo = sl + alpha * aut;
                    isu = fv ? rcp_(sup) : (T)0; isl = fv ? rcp_(slo) : (T)0;
                }
            }
            const bool restep = mode == RESTEP;
            if (!restep) {
                if (mode != REFACTOR) {
                    U = Ut; cs[C_J] = Jt;
                    const T g = linearize(St, exact && gn_hold == 0);
                    STAMP(1);
                    if (mode == FIRST) {
                        const T w0 = forms_apply(U);
                        T bu_, bl_;
                        form_bounds(tid, bu_, bl_);
                        sup = bu_ - w0; slo = bl_ + w0;
                        isu = fv ? (T)1 / sup : (T)0; isl = fv ? (T)1 / slo : (T)0;
                        T sm[1] = {fv ? log_pos(sup * slo, kc) : (T)0};
                        T mx[1] = {fabs(g)};
                        wg_reduce<1, 1>(sm, mx);
                        cs[C_LGS] = sm[0];
                        const T gm = mx[0];
                        sc = gm > (T)100 ? (T)100 / gm : (T)1;  // Ipopt nlp_scaling_max_gradient
                        lu = mu * isu; ll = mu * isl;
                    } else {
                        lu = fmax(fmin(lu, kappa_sigma * mu * isu), mu * isu * ((T)1 / kappa_sigma));
                        ll = fmax(fmin(ll, kappa_sigma * mu * isl), mu * isl * ((T)1 / kappa_sigma));
                    }
                    if (iters >= max_iter) { mode = FINAL; Ut = U; continue; }
                    ++iters;
                    const T rd = sc * g + forms_applyT(lu - ll);
                    const T cuv = sup * lu, clv = slo * ll;
                    T sm[2] = {lu + ll, cuv + clv};
                    T mx[2] = {fabs(rd), fmax(cuv, clv)};
                    wg_reduce<2, 2>(sm, mx);
                    const T lsum = sm[0], gap = sm[1], rdm = mx[0], cm0 = mx[1];
                    const T inv2nf = pt[PT_INV2NF];
                    const T isd = s_max * rcp_(fmax(s_max, lsum * inv2nf));  // 1 / s_d
                    const T err0 = fmax(rdm, cm0) * isd;
                    const T tol = pt[PT_TOL];
                    const T gap_lim = pt[PT_GAP_TOL] * fmax((T)1, fabs(Jt));
                    cs[C_ERR] = err0; cs[C_RDS] = rdm * isd;
                    bool done = false;
                    if (err0 <= tol) { if (tid < 128) ubl[tid] = U; have_best = true; }
                    if (err0 <= tol) {
                        if (gap <= gap_lim * sc || n_polish >= 1) done = true; else ++n_polish;
                    } else if (n_polish > 0 && ++n_polish > 1) done = true;
                    n_accept = err0 <= pt[PT_TOL_X100] ? n_accept + 1 : 0;
                    n_flat = fabs(Jt - cs[C_JP]) <= (T)20 * Real<T>::eps() * fmax((T)1, fabs(Jt)) ? n_flat + 1 : 0;
                    cs[C_JP] = Jt;
                    if (n_flat >= 12 && err0 <= pt[PT_TOL_X1000]) done = true;
                    if (done || n_accept >= 15) { status = 0; mode = FINAL; Ut = U; continue; }
                    const T mu_min = fmax(pt[PT_TOL_D100], fmin(pt[PT_TOL_D10], (T)0.1 * gap_lim * sc * inv2nf));
                    cs[C_MUF] = mu_min;
#pragma nounroll
                    for (; !pc;) {  // monotone barrier update (mu_strategy 0)
                        T dm[1] = {(T)0}, cm[1] = {fv ? fmax(fabs(sup * lu - mu), fabs(slo * ll - mu)) : (T)0};
                        wg_reduce<0, 1>(dm, cm);
                        if (fmax(rdm, cm[0]) * isd <= kappa_eps * mu && mu > mu_min) mu = fmax(mu_min, fmin(kappa_mu * mu, mu * sqrt(mu)));
                        else break;
                    }
                    use_exact = exact && gn_hold == 0; reg = 0; attempt = 0;
                    if (gn_hold > 0) --gn_hold;
                    if (use_exact && indef == 1 && cs[C_DWS] > (T)0) { reg = cs[C_DWS] / (T)3; if (reg < (T)1e-9 * cs[C_HMAX]) reg = 0; }
                    first_attempt = true;
                    STAMP(2);
                }
                // K = sc*H + A^T Sigma A with the affine right-hand side -sc*g riding along as row n
                stage_form_weights(lu * isu + ll * isl);
                STAMP(6);
                bool factored;
                {
                    const bool want_hmax = use_exact && indef == 1 && first_attempt;
                    T hmax = cs[C_HMAX];
                    condense_adjoint(sc);
                    STAMP(3);
                    if (want_hmax) {  // max |sc * H_jj|: scale of the delta_w shift
                        T dm[1] = {(T)0}, hx[1] = {tid < n ? fabs(Lc[offc_rt(tid) + tid]) : (T)0};
                        wg_reduce<0, 1>(dm, hx);
                        hmax = hx[0];
                    }
                    switch (wv) {   // every wave runs the specialisation for its tile rows; the barriers inside pair up across them
                        case 0: factored = assemble_factor<0>(sc, reg, want_hmax, hmax); break;
                        case 1: factored = assemble_factor<1>(sc, reg, want_hmax, hmax); break;
                        case 2: factored = assemble_factor<2>(sc, reg, want_hmax, hmax); break;
                        default: factored = assemble_factor<3>(sc, reg, want_hmax, hmax); break;
                    }
                    cs[C_HMAX] = hmax;
                    first_attempt = false;
                }
                STAMP(5);
                if (!factored) {
                    if (++attempt >= 40) { status = 3; mode = FINAL; Ut = U; continue; }
                    if (use_exact && indef == 1) {
                        const T hmax = cs[C_HMAX], dw_last = cs[C_DWL];
                        if (reg == (T)0) reg = dw_last > (T)0 ? fmax((T)1e-10 * hmax, dw_last / (T)3) : (T)1e-2 * hmax;
                        else reg *= dw_last > (T)0 ? (T)8 : (T)10;
                        if (reg > (T)1e2 * hmax) { use_exact = false; reg = 0; drop_second_order(); }
                    } else if (use_exact) {
                        use_exact = false; gn_hold = 2; drop_second_order();
                        if (indef_cfg == 2 && ++n_fail >= 2) { indef = 1; gn_hold = 0; }
                    } else reg = reg == (T)0 ? (T)1e-8 : reg * (T)100;
                    mode = REFACTOR; Ut = U;
                    continue;
                }
                if (use_exact && reg > (T)0) cs[C_DWL] = reg;
                if (use_exact) cs[C_DWS] = reg;
                cu_ = cl_ = (T)0;
                corr_active = false;
                if (pc) {
                    // Mehrotra predictor: affine-scaling step on the same factor -> this iteration's barrier target
                    const T dua = solve_dir((T)0, false);
                    aut = forms_apply(dua);
                    const T qu = aut * isu, ql = aut * isl;  // -ds_u/s_u, ds_l/s_l
                    T sm[1] = {sup * lu + slo * ll};
                    T mx[2] = {fmax((T)1, fmax(qu, -ql)), fmax((T)1, fmax((T)1 - qu, (T)1 + ql))};
                    wg_reduce<1, 2>(sm, mx);
                    const T apa = rcp_(mx[0]), ada = rcp_(mx[1]);
                    const T su = sup, sl = slo, dsu = -aut, dsl = aut;
                    const T dlu = -lu - lu * isu * dsu, dll = -ll - ll * isl * dsl;
                    T sa[1] = {(su + apa * dsu) * (lu + ada * dlu) + (sl + apa * dsl) * (ll + ada * dll)}, dm[1] = {(T)0};
                    cu_ = dsu * dlu; cl_ = dsl * dll;
                    wg_reduce<1, 0>(sa, dm);
                    const T mucur = sm[0] * pt[PT_INV2NF], muaff = sa[0] * pt[PT_INV2NF];
                    const T r3 = muaff * rcp_(mucur);
                    mu = fmax(cs[C_MUF], fmin((T)1, r3 * r3 * r3) * mucur);
                    mu = fmax(mu, fmin(mucur, cs[C_RDS] * (T)KMPC_IKRD));  // no barrier target far below the dual infeasibility
                    corr_active = true;
                    STAMP(7);
                }
            } else {  // RESTEP: same factor, corrector term dropped
                cu_ = cl_ = (T)0;
                corr_active = false;
            }
            // centering (+ corrector) part of the step: du = K^{-1}(-sc*g - A^T((mu - corr)/s_u - (mu - corr)/s_l))
            du = solve_dir(forms_applyT(-((mu - cu_) * isu - (mu - cl_) * isl)), true);
            STAMP(15);
            aut = forms_apply(du);
            const T tau = fmax(tau_min, (T)1 - mu);
            {
                T rp_ = 0, rq_ = 0, gw = 0;
                if (fv) {
                    const T su = sup, sl = slo, dsu = -aut, dsl = aut;
                    const T dlu = (mu - cu_ - lu * su) * isu - lu * isu * dsu;
                    const T dll = (mu - cl_ - ll * sl) * isl - ll * isl * dsl;
                    gw = mu * (isu - isl) * aut;
                    rp_ = fmax(-dsu * isu, -dsl * isl);
                    rq_ = fmax(-dlu * rcp_(lu), -dll * rcp_(ll));
                }
                T sm[1] = {(tid < n ? sc * gbl[tid] * du : (T)0) + gw};
                T mx[2] = {fmax(rp_, (T)0), fmax(rq_, (T)0)};
                wg_reduce<1, 2>(sm, mx);
                // fraction to the boundary: alpha = min(1, tau * min(-s/ds)) = tau / max(tau, max(-ds/s))
                const T ap = tau * rcp_(fmax(tau, mx[0]));
                cs[C_AD] = tau * rcp_(fmax(tau, mx[1]));
                cs[C_PHI0] = sc * cs[C_J] - mu * cs[C_LGS];
                cs[C_DPHI] = sm[0];
                alpha = ap; ls = 0;
            }
            Ut = U + alpha * du;
            mode = TRIAL;
            STAMP(8);
        }
        STAMP(10);
        // ---- outputs (St / Jt are the evaluation of the returned U) ------------------------------------
        const T w0 = forms_apply(U);
        T vi[1] = {-(T)1e30}, dm[1] = {(T)0};
        if (fv) {
            T bu_, bl_;
            form_bounds(tid, bu_, bl_);
            vi[0] = fmax(w0 - (bu_ - form_relax(tid, true)), -w0 - (bl_ - form_relax(tid, false)));
        }
        wg_reduce<0, 1>(dm, vi);
        if (tid < n) {
            if (io.outU) io.outU[(size_t)b * n + tid] = U;
            if (io.warmU) io.warmU[(size_t)b * n + tid] = U;
            if (tid < 2) io.u0[(size_t)b * 2 + tid] = U;
        }
        if (io.outX && tid <= N) {
            T *o = io.outX + ((size_t)b * (N + 1) + tid) * 4;
            o[0] = St.x + z0p[0]; o[1] = St.y + z0p[1]; o[2] = St.psi; o[3] = St.v;
        }
        STAMP(11);
        STAMP_OUT(io.stamps, b);
        if (tid == 0) {
            io.status[b] = status;
            if (io.cost) io.cost[b] = Jt;
            if (io.viol) io.viol[b] = vi[0];
            if (io.iters) io.iters[b] = iters;
        }
    }
};

// Workgroups per CU.  The workgroup is latency-bound between its barriers and the SIMDs' issue ports are mostly idle (tools/calib/issue_probe.hip), so
// throughput follows the number of resident workgroups almost 1 : 1.  fp64 at N >= 40: 56 ... 74 KB of LDS allow two (256 VGPRs, no scratch); at
// N = 32 / 36 the 44 / 50 KB allow three, which is worth the scratch that 168 VGPRs cost: 1.62 -> 2.14 and 1.39 -> 1.85 M solves/s at B = 262 144.
// fp32: four (128 VGPRs), five up to N = 40 (+8 / +13 / +2 % at N = 32 / 36 / 40).
template <typename T, int N>
__global__ __launch_bounds__(256, sizeof(T) == 8 ? (N <= 36 ? 3 : 2) : (N <= 40 ? 5 : 4)) void kmpc_solve_wide_kernel(KP P, KIO<T> io)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[WideSolver<T, N>::lds_elems() * sizeof(T)];
    if ((int)blockIdx.x >= P.B) return;
    const int b = io.perm ? io.perm[blockIdx.x] : (int)blockIdx.x;
#ifdef KMPC_POISON  // diagnostic build (make poison): every LDS word starts as NaN, so a read of a word nobody wrote shows up in the results
    for (int e = threadIdx.x; e < WideSolver<T, N>::lds_elems(); e += 256) reinterpret_cast<T *>(smem)[e] = (T)NAN;
    __syncthreads();
#endif
    WideSolver<T, N> sv(P, smem);
    sv.load_problem(io.z0, io.ref, io.vt, io.up, b);
    sv.solve(io, b);
}

// diagnostics (tests/test_gpu_kernels.py): the KKT pipeline of this kernel at a given point, form weights, scaling and shift
template <typename T, int N>
__global__ __launch_bounds__(256, 2) void kmpc_wide_kkt_kernel(KP P, KDbgK<T> io)
{
    typedef WideSolver<T, N> SV;
    constexpr int n = SV::n, nf = SV::nf;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SV::lds_elems() * sizeof(T)];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= P.B) return;
    SV sv(P, smem);
    sv.load_problem(io.z0, io.ref, io.vt, io.up, b);
    const T sc = (T)io.sc, reg = (T)io.reg;
    const T U = tid < n ? io.U[(size_t)b * n + tid] : (T)0;
    StageW<T> St;
    sv.eval(U, St);
    const T g = sv.linearize(St, P.hessian == 1);
    sv.stage_form_weights(tid < nf ? io.w[(size_t)b * nf + tid] : (T)0);
    T hm = 0;
    T *K = io.K + (size_t)b * n * n;
    bool okf;
    sv.condense_adjoint(sc);
    switch (sv.wv) {
        case 0: okf = sv.template assemble_factor<0>(sc, reg, false, hm, K); break;
        case 1: okf = sv.template assemble_factor<1>(sc, reg, false, hm, K); break;
        case 2: okf = sv.template assemble_factor<2>(sc, reg, false, hm, K); break;
        default: okf = sv.template assemble_factor<3>(sc, reg, false, hm, K); break;
    }
    T x = (T)0;
    if (okf) x = sv.solve_dir(tid < n ? io.b[(size_t)b * n + tid] : (T)0, true);
    if (tid < n) { io.g[(size_t)b * n + tid] = g; io.x[(size_t)b * n + tid] = x; }
    if (tid == 0) io.ok[b] = okf ? 1 : 0;
}
template <typename T> hipError_t kmpc_launch_wide_kkt(const KP &P, const KDbgK<T> &io, hipStream_t st)
{
    if (P.N == 50) hipLaunchKernelGGL((kmpc_wide_kkt_kernel<T, 50>), dim3(P.B), dim3(256), 0, st, P, io);
    else if (P.N == 48) hipLaunchKernelGGL((kmpc_wide_kkt_kernel<T, 48>), dim3(P.B), dim3(256), 0, st, P, io);
    else if (P.N == 44) hipLaunchKernelGGL((kmpc_wide_kkt_kernel<T, 44>), dim3(P.B), dim3(256), 0, st, P, io);
    else if (P.N == 40) hipLaunchKernelGGL((kmpc_wide_kkt_kernel<T, 40>), dim3(P.B), dim3(256), 0, st, P, io);
    else if (P.N == 36) hipLaunchKernelGGL((kmpc_wide_kkt_kernel<T, 36>), dim3(P.B), dim3(256), 0, st, P, io);
    else if (P.N == 32) hipLaunchKernelGGL((kmpc_wide_kkt_kernel<T, 32>), dim3(P.B), dim3(256), 0, st, P, io);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
template hipError_t kmpc_launch_wide_kkt<double>(const KP &, const KDbgK<double> &, hipStream_t);
template hipError_t kmpc_launch_wide_kkt<float>(const KP &, const KDbgK<float> &, hipStream_t);

// horizons whose n + 1 rows make 7 (N = 48, the BASELINE's 50), 6 (N = 40, 44) or 5 (N = 32, 36) tile rows, with one thread per form (5N - 2 <= 256)
template <typename T> bool kmpc_wide_available(int N) { return N == 50 || N == 48 || N == 44 || N == 40 || N == 36 || N == 32; }
template <typename T> hipError_t kmpc_launch_solve_wide(const KP &P, const KIO<T> &io, hipStream_t st)
{
    if (P.N == 50) hipLaunchKernelGGL((kmpc_solve_wide_kernel<T, 50>), dim3(P.B), dim3(256), 0, st, P, io);
    else if (P.N == 48) hipLaunchKernelGGL((kmpc_solve_wide_kernel<T, 48>), dim3(P.B), dim3(256), 0, st, P, io);
    else if (P.N == 44) hipLaunchKernelGGL((kmpc_solve_wide_kernel<T, 44>), dim3(P.B), dim3(256), 0, st, P, io);
    else if (P.N == 40) hipLaunchKernelGGL((kmpc_solve_wide_kernel<T, 40>), dim3(P.B), dim3(256), 0, st, P, io);
    else if (P.N == 36) hipLaunchKernelGGL((kmpc_solve_wide_kernel<T, 36>), dim3(P.B), dim3(256), 0, st, P, io);
    else if (P.N == 32) hipLaunchKernelGGL((kmpc_solve_wide_kernel<T, 32>), dim3(P.B), dim3(256), 0, st, P, io);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
template bool kmpc_wide_available<double>(int);
template bool kmpc_wide_available<float>(int);
template hipError_t kmpc_launch_solve_wide<double>(const KP &, const KIO<double> &, hipStream_t);
template hipError_t kmpc_launch_solve_wide<float>(const KP &, const KIO<float> &, hipStream_t);
