// kmpc_kernels.hip -- CDNA4 (gfx950) kernels of the batched kinematic-bicycle MPC solver.
//
// One 64-lane wavefront solves one problem (workgroup = 1 wave, grid = batch).  The NLP is the
// one built in the reference's scripts/mpc_utils/MKZMPCPathFollower.jl:65-123; states are
// eliminated by forward simulation so every inequality is linear in the 2N inputs, and the
// problem is solved by a primal-dual interior-point Newton method (Ipopt's published
// algorithm restricted to this structure):
//
//   per iteration   (a) roll the bicycle model out over the horizon -- lane k owns stage k,
//                       the recurrences v, psi, x, y are wave prefix scans (:115-122);
//                   (b) costates by suffix scans -> exact gradient and stage Hessians;
//                   (c) condense: stream the 4 x 2k sensitivity block G_k through the stages
//                       and accumulate H += G_k^T P_k G_k on the matrix cores
//                       (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, K = nx = 4);
//                   (d) KKT matrix K = sc*H + A^T Sigma A staged in LDS, blocked Cholesky on the
//                       matrix cores (left-looking by 16-column tile column), triangular solves;
//                   (e) Mehrotra predictor-corrector, fraction-to-the-boundary + Armijo
//                       back-tracking on the barrier function.
// This is the generic kernel: any horizon N <= 56 at run time, matrices in LDS; it also carries the
// Frenet-frame model (MODEL = 1) as a second dynamics functor.  Horizons with a compile-time kernel
// (kmpc_fast.hip, kmpc_wide.hip, kmpc_quad.hip) run there instead.  The model code (a)-(c) and the KKT code (d)
// below are this kernel's own; the interior-point iteration (e) is ipm::solve of kmpc_ipm.h -- the ONE state
// machine all four solve kernels instantiate (up to two input slots per lane here: NV).
//
// Layouts
//   n-vector  (length n = 2N, element j = 2k + {0: acc_k, 1: d_f_k}):  lane j%64, slot j/64
//   form-vector (the 5N-2 two-sided linear forms a_f^T U, f = lane + 64*slot):
//        f in [0,n) box on u_f | [n, n+R) rate forms, R = 2(N-1) | [n+R, nf) speed prefix sums
//   stage data: lane k <-> stage / state k (k = 0..N)
//   H tiles: lower-triangular 16x16 tiles in MFMA C/D layout (col = lane&15, row from Real<T>)
#include "kmpc_ipm.h"

// per-lane stage record: lane k holds state k (k = 0..N) and input k (k < N)
template <typename T> struct Stage {
    T a, d, v, x, y, psi, c, s, sinb, cosb, b1, b2, ex, ey, ep, ev;
    T K, Kp, iden, dsdt;  // Frenet functor only: curvature, dK/ds, 1/(1 - e_y K), ds/dt at the stage
};

// MODEL 0: Cartesian kinematic bicycle (MKZMPCPathFollower.jl); MODEL 1: Frenet-frame variant (MKZMPCPathFollowerFrenet.jl:112-123),
// states (s, e_y, e_psi, v) in the (x, y, psi, v) slots, zero cost references, Gauss-Newton Hessian only.
template <typename T, int NT, int MODEL = 0> struct Solver {
    static constexpr int NV = (16 * NT + 63) / 64;       // n-vector slots per lane
    static constexpr int NF = (40 * NT - 2 + 63) / 64;   // form-vector slots per lane
    static constexpr int NTT = NT * (NT + 1) / 2;        // lower-triangular tiles
    // From NT = 5 on the condensed Hessian no longer fits in registers next to the solver state (28 tiles = 224 fp64 VGPRs at NT = 7:
    // 1146 spilled registers, measured): its tiles are accumulated in LDS, in place in the K matrix, one tile row at a time.
    static constexpr bool LDSACC = NT >= 5;
    static constexpr int NACC = LDSACC ? 1 : NTT;
    typedef typename Real<T>::acc_t acc_t;

    STAMP_MEMBERS
    const KP &P;
    const int lane, N, n, R, nf, ld;
    T *Km, *stg, *xb, *wb, *cb, *dinv, *pan;
    // ---- what the shared interior-point state machine (ipm::solve, kmpc_ipm.h) reads from a back-end; this kernel instantiates ONLY the
    // state machine -- its model code (run-time horizon, up to two inputs per lane) and its KKT code (matrix in LDS) are its own
    typedef T real;
    typedef Stage<T> stage_t;
    static constexpr int NTH = 64;
    int vid;                    // = lane
    T *cs;                      // solve scalars and the parameter table: caller-owned register arrays here (constant indices; kept outside the object so
    const T *pt;                // that it holds no pointer into itself), LDS in the compile-time-horizon kernels
    Coef<T> kc;
    T cu_[NF], cl_[NF];         // corrector terms
    T gk[NV], ubest_[NV];       // gradient of the current linearisation, last iterate that passed Ipopt's test
    acc_t acc[NACC];            // condensed-Hessian tiles (kept across re-factorisations unless they live in the matrix the factorisation overwrites)
    bool need_condense, exact_now;
    T sc_;
    // problem data
    T x0, y0, psi0, v0, vt, up0, up1, xoff_, yoff_;
    T rx, ry, rp;  // reference at stage `lane`
    T kp0, kp1, kp2, kp3;  // Frenet: K(s) = kp0 s^3 + kp1 s^2 + kp2 s + kp3
    T dt, dtc, Lb, rr_;
    T Cx, Cy, Cp, Cv, Cda, Cdd, Ca, Cd;

    DEV Solver(const KP &p, unsigned char *smem, T *scalars /* [16] */, T *params /* [32] */)
        : P(p), lane(threadIdx.x), N(p.N), n(2 * p.N), R(2 * (p.N - 1)), nf(5 * p.N - 2), ld(NT >= 5 ? 16 * NT + 1 : 2 * p.N + 1), vid(threadIdx.x)
    {
        cs = scalars; pt = params;
        ipm::fill_param_table(params, p, nf);
        if constexpr (sizeof(T) == 8) kc.tab = kmpc_coef; else kc.tab = nullptr;   // (fp64: the polynomial coefficients straight from constant memory; fp32: literals)
        need_condense = true; exact_now = false; sc_ = (T)1;
        Km = reinterpret_cast<T *>(smem);
        stg = Km + (NT >= 5 ? 16 * NT : n) * ld;
        xb = stg + KMPC_STG * (N + 1);
        wb = xb + 16 * NT;
        cb = wb + 64 * NF;
        dinv = cb + 64;
        pan = dinv + 16 * NT;
        dt = (T)p.dt; dtc = (T)p.dtc; Lb = (T)p.L_b; rr_ = (T)p.r;
        Cx = (T)p.C[0]; Cy = (T)p.C[1]; Cp = (T)p.C[2]; Cv = (T)p.C[3];
        Cda = (T)p.C[4]; Cdd = (T)p.C[5]; Ca = (T)p.C[6]; Cd = (T)p.C[7];
    }

    DEV void load_problem(const T *z0, const T *ref, const T *vtp, const T *upp, int b)
    {
        vt = vtp[b];
        up0 = upp ? upp[2 * (size_t)b] : (T)0; up1 = upp ? upp[2 * (size_t)b + 1] : (T)0;
        rx = ry = rp = (T)0;
        kp0 = kp1 = kp2 = kp3 = (T)0;
        psi0 = z0[4 * (size_t)b + 2]; v0 = z0[4 * (size_t)b + 3];
        if (MODEL == 1) {  // `ref` carries k_poly [B,4]; (s, e_y) are not translation-invariant (K depends on s)
            xoff_ = yoff_ = (T)0; x0 = z0[4 * (size_t)b]; y0 = z0[4 * (size_t)b + 1];
            const T *kp = ref + 4 * (size_t)b;
            kp0 = kp[0]; kp1 = kp[1]; kp2 = kp[2]; kp3 = kp[3];
            return;
        }
        // the NLP is invariant under a translation of (x, y): solve it in vehicle-centred coordinates (recorded paths live hundreds
        // of metres from their origin; positions would carry ~1e-13 m of rounding = ~1e-12 in the cost, above the Armijo
        // decrease of the last iterations); predictions are shifted back on output
        xoff_ = z0[4 * (size_t)b]; yoff_ = z0[4 * (size_t)b + 1]; x0 = (T)0; y0 = (T)0;
        if (lane <= N) {
            const T *r = ref + ((size_t)b * (N + 1) + lane) * 3;
            rx = r[0] - xoff_; ry = r[1] - yoff_; rp = r[2];
        }
    }

    // ---- two-sided bounds of form f (relaxed like Ipopt's bound_relax_factor) -----------------
    DEV void form_bounds(int f, T &bu, T &bl, T &rlx) const
    {
        const T relax = (T)P.relax;
        if (f < n) {
            const T ub = (f & 1) ? (T)P.steer_max : (T)P.a_max;
            rlx = relax * fmax((T)1, ub);
            bu = ub + rlx; bl = ub + rlx;
        } else if (f < n + R) {
            const int r = f - n, jj = r & 1, kk = r >> 1;
            const T d = (jj ? (T)P.steer_dmax : (T)P.a_dmax) * (kk == 0 ? dtc : dt);
            rlx = relax * fmax((T)1, d);
            const T u = kk == 0 ? (jj ? up1 : up0) : (T)0;
            bu = d + rlx + u; bl = d + rlx - u;
        } else if (f < nf) {
            const T ru = relax * fmax((T)1, fabs((T)P.v_max)), rl = relax * fmax((T)1, fabs((T)P.v_min));
            bu = (T)P.v_max + ru - v0; bl = -(T)P.v_min + rl + v0;
            rlx = fmax(ru, rl);
        } else { bu = bl = (T)1; rlx = (T)0; }
    }

    // ---- y_f = a_f^T x ------------------------------------------------------------------------
    DEV void forms_apply(const T (&x)[NV], T (&y)[NF])
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int j = lane + 64 * i; if (j < n) xb[j] = x[i]; }
        WSYNC();
        T a = lane < N ? xb[2 * lane] : (T)0;
        a = scan_prefix(a, lane);
        if (lane < N) cb[lane] = a;
        WSYNC();
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = lane + 64 * i;
            T v = (T)0;
            if (f < n) v = xb[f];
            else if (f < n + R) { const int r = f - n; v = r < 2 ? xb[r] : xb[r + 2] - xb[r]; }
            else if (f < nf) v = dt * cb[f - n - R];
            y[i] = v;
        }
        WSYNC();
    }

    // stage the form weights w_f in LDS (wb) together with the suffix sums of the speed weights (cb)
    DEV void stage_form_weights(const T (&w)[NF])
    {
#pragma unroll
        for (int i = 0; i < NF; ++i) { const int f = lane + 64 * i; if (f < nf) wb[f] = w[i]; }
        WSYNC();
        T s = lane < N ? wb[n + R + lane] : (T)0;
        s = scan_suffix(s, lane);
        if (lane < N) cb[lane] = s;
        WSYNC();
    }
    // out_j += sum_f w_f a_f[j]
    DEV void forms_applyT_add(const T (&w)[NF], T (&out)[NV])
    {
        stage_form_weights(w);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = lane + 64 * i;
            if (j < n) {
                T o = wb[j];
                if (j < 2) o += wb[n + j];
                if (j >= 4) o += wb[n + j - 2];
                if (j >= 2 && j < R) o -= wb[n + j];
                if (!(j & 1)) o += dt * cb[j >> 1];
                out[i] += o;
            }
        }
        WSYNC();
    }
    // (A^T W A)(row, col), row >= col; needs stage_form_weights() done
    DEV T gram_entry(int row, int col) const
    {
        T g = (T)0;
        if (row == col) {
            g = wb[row];
            if (row < 2) g += wb[n + row];
            if (row >= 4) g += wb[n + row - 2];
            if (row >= 2 && row < R) g += wb[n + row];
        } else if (row == col + 2 && col >= 2 && col < R) {
            g = -wb[n + col];
        }
        if (!((row | col) & 1)) g += dt * dt * cb[row >> 1];
        return g;
    }
    // Hessian of the input-cost terms (MKZMPCPathFollower.jl:99-102), row >= col
    DEV T input_hess(int row, int col) const
    {
        const int jj = row & 1, k = row >> 1;
        const T Cu = jj ? Cd : Ca, Cdl = jj ? Cdd : Cda;
        if (row == col) return (T)2 * Cu + (T)2 * Cdl * (T)((k > 0) + (k < N - 1));
        if (row == col + 2) return -(T)2 * Cdl;
        return (T)0;
    }

    // ---- (a) roll-out + objective at U ----------------------------------------------------------
    DEV T eval(const T (&U)[NV], Stage<T> &S)
    {
        if (MODEL == 1) return eval_frenet(U, S);
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int j = lane + 64 * i; if (j < n) xb[j] = U[i]; }
        WSYNC();
        const int k = lane;
        const bool st = k < N;
        T a = st ? xb[2 * k] : (T)0, d = st ? xb[2 * k + 1] : (T)0;
        T an = (k + 1 < N) ? xb[2 * k + 2] : a, dn = (k + 1 < N) ? xb[2 * k + 3] : d;
        WSYNC();
        S.a = a; S.d = d;
        const T ia = scan_prefix(a, lane);
        const T v = v0 + dt * (ia - a);  // v_k = v0 + dt*sum_{j<k} acc_j   (:122)
        T sd, cd;
        Real<T>::sincos_(d, &sd, &cd);
        const T Dn = cd * cd + rr_ * rr_ * sd * sd;
        const T rs = (T)1 / sqrt(Dn);
        S.sinb = rr_ * sd * rs;  // sin(atan(r tan d))   (:115)
        S.cosb = cd * rs;
        S.b1 = rr_ / Dn;
        S.b2 = rr_ * ((T)1 - rr_ * rr_) * ((T)2 * sd * cd) / (Dn * Dn);
        const T wp = st ? v * S.sinb : (T)0;
        const T ip = scan_prefix(wp, lane);
        const T psi = psi0 + (dt / Lb) * (ip - wp);  // (:121)
        T sp, cp;
        Real<T>::sincos_(psi, &sp, &cp);
        S.c = cp * S.cosb - sp * S.sinb;  // cos(psi + beta)
        S.s = sp * S.cosb + cp * S.sinb;
        const T wx = st ? v * S.c : (T)0, wy = st ? v * S.s : (T)0;
        const T ix = scan_prefix(wx, lane), iy = scan_prefix(wy, lane);
        S.x = x0 + dt * (ix - wx);  // (:119)
        S.y = y0 + dt * (iy - wy);  // (:120)
        S.v = v; S.psi = psi;
        const bool cs = (k >= 1 && k <= N);
        S.ex = cs ? S.x - rx : (T)0;
        S.ey = cs ? S.y - ry : (T)0;
        S.ep = cs ? psi - rp : (T)0;
        S.ev = (k >= 1 && k <= N - 1) ? v - vt : (T)0;
        T Jl = Cx * S.ex * S.ex + Cy * S.ey * S.ey + Cp * S.ep * S.ep + Cv * S.ev * S.ev;  // (:97-98)
        if (st) Jl += Ca * a * a + Cd * d * d;                                             // (:99-100)
        if (k < N - 1) Jl += Cda * (an - a) * (an - a) + Cdd * (dn - d) * (dn - d);        // (:101-102)
        return uniform_(wave_sum(Jl));
    }

    // ---- (b) costates, gradient (n-vector g) and per-stage scalars for the condensing loop -------
    DEV void linearize_model(const Stage<T> &S, bool exact, T (&g)[NV])
    {
        if (MODEL == 1) { linearize_frenet(S, exact, g); return; }
        const int k = lane;
        const bool st = k < N;
        const T lx = (T)2 * Cx * S.ex, ly = (T)2 * Cy * S.ey, lp = (T)2 * Cp * S.ep, lv = (T)2 * Cv * S.ev;
        const T px = scan_suffix(lx, lane), py = scan_suffix(ly, lane);
        const T px1 = __shfl_down(px, 1), py1 = __shfl_down(py, 1);
        const T A02 = st ? -dt * S.v * S.s : (T)0, A12 = st ? dt * S.v * S.c : (T)0;
        const T A03 = st ? dt * S.c : (T)0, A13 = st ? dt * S.s : (T)0, A23 = st ? dt / Lb * S.sinb : (T)0;
        const T tp = lp + (st ? A02 * px1 + A12 * py1 : (T)0);
        const T pp = scan_suffix(tp, lane);
        const T pp1 = __shfl_down(pp, 1);
        const T tv = lv + (st ? A03 * px1 + A13 * py1 + A23 * pp1 : (T)0);
        const T pv = scan_suffix(tv, lane);
        const T pv1 = __shfl_down(pv, 1);
        const T Bdx = st ? -dt * S.v * S.s * S.b1 : (T)0, Bdy = st ? dt * S.v * S.c * S.b1 : (T)0,
                Bdp = st ? dt * S.v / Lb * S.cosb * S.b1 : (T)0;
        const T aprev = __shfl_up(S.a, 1), dprev = __shfl_up(S.d, 1);
        const T anext = __shfl_down(S.a, 1), dnext = __shfl_down(S.d, 1);
        T ga = dt * pv1 + (T)2 * Ca * S.a, gd = Bdx * px1 + Bdy * py1 + Bdp * pp1 + (T)2 * Cd * S.d;
        if (k >= 1) { ga += (T)2 * Cda * (S.a - aprev); gd += (T)2 * Cdd * (S.d - dprev); }
        if (k < N - 1) { ga -= (T)2 * Cda * (anext - S.a); gd -= (T)2 * Cdd * (dnext - S.d); }
        if (st) { xb[2 * k] = ga; xb[2 * k + 1] = gd; }
        // second derivatives of the Euler step contracted with the costate of state k+1
        T mpp = 0, mpv = 0, mpd = 0, mvd = 0, mdd = 0;
        if (exact && st) {
            const T v = S.v, c = S.c, s = S.s, b1 = S.b1, b2 = S.b2;
            mpp = px1 * (-dt * v * c) + py1 * (-dt * v * s);
            mpv = px1 * (-dt * s) + py1 * (dt * c);
            mpd = px1 * (-dt * v * c * b1) + py1 * (-dt * v * s * b1);
            mvd = px1 * (-dt * s * b1) + py1 * (dt * c * b1) + pp1 * (dt / Lb * S.cosb * b1);
            mdd = px1 * (-dt * v * (c * b1 * b1 + s * b2)) + py1 * (dt * v * (-s * b1 * b1 + c * b2)) +
                  pp1 * (dt * v / Lb * (-S.sinb * b1 * b1 + S.cosb * b2));
        }
        if (k <= N) {
            T *q = stg + KMPC_STG * k;
            q[0] = A02; q[1] = A03; q[2] = A12; q[3] = A13; q[4] = A23; q[5] = Bdx; q[6] = Bdy; q[7] = Bdp;
            q[8] = mpp; q[9] = mpv; q[10] = mpd; q[11] = mvd; q[12] = mdd;
        }
        WSYNC();
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int j = lane + 64 * i; g[i] = j < n ? xb[j] : (T)0; }
        WSYNC();
    }

    // ---- (c) condensing on the matrix cores ------------------------------------------------------
    // acc[tile(ti,tj)] accumulates  sum_s G_s^T (2 Q_s + M_s) G_s  + delta-row terms  (unscaled)
    // tile (ti, tj) of the tile-padded LDS matrix in MFMA C layout (LDSACC only: the matrix has 16 NT rows, so no bounds checks);
    // the four per-lane element offsets are computed once (tofs), a tile adds one wave-uniform offset
    int tofs[4];
    DEV void tile_init()
    {
#pragma unroll
        for (int r = 0; r < 4; ++r) tofs[r] = Real<T>::row_of(lane, r) * ld + (lane & 15);
    }
    DEV acc_t tile_load(int ti, int tj) const
    {
        const T *base = Km + (16 * ti) * ld + 16 * tj;
        acc_t a;
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = base[tofs[r]];
        return a;
    }
    DEV void tile_store(int ti, int tj, const acc_t &a)
    {
        T *base = Km + (16 * ti) * ld + 16 * tj;
#pragma unroll
        for (int r = 0; r < 4; ++r) base[tofs[r]] = a[r];
    }
    DEV void condense(bool exact, acc_t (&acc)[NACC])
    {
        if constexpr (MODEL == 1) { condense_dense(exact, acc); return; }
        const int kk = lane >> 4, c = lane & 15;
        if constexpr (LDSACC) {
            tile_init();
            for (int e = lane; e < 16 * NT * ld; e += 64) Km[e] = (T)0;
            WSYNC();
        } else {
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = acc_t{0, 0, 0, 0};
        }
        T own[NT], gps[NT], gv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) own[t] = gps[t] = gv[t] = (T)0;
        for (int s = 0; s <= N; ++s) {
            if (s >= 1) {  // G_s = [A_{s-1} G_{s-1} | B_{s-1}]
                const T *q = stg + KMPC_STG * (s - 1);
                const T A02 = q[0], A03 = q[1], A12 = q[2], A13 = q[3], A23 = q[4], Bdx = q[5], Bdy = q[6], Bdp = q[7];
                const T cA = kk == 0 ? A02 : (kk == 1 ? A12 : (T)0);
                const T cB = kk == 0 ? A03 : (kk == 1 ? A13 : (kk == 2 ? A23 : (T)0));
                const T bo = kk == 0 ? Bdx : (kk == 1 ? Bdy : (kk == 2 ? Bdp : (T)0));
                const int col0 = 2 * (s - 1);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    own[t] += cA * gps[t] + cB * gv[t];
                    gps[t] += A23 * gv[t];
                    const int col = 16 * t + c;
                    if (col == col0) { own[t] = kk == 3 ? dt : (T)0; gps[t] = (T)0; gv[t] = dt; }
                    if (col == col0 + 1) { own[t] = bo; gps[t] = Bdp; gv[t] = (T)0; }
                }
                const T *qs = stg + KMPC_STG * s;
                const T mpp = (exact && s < N) ? qs[8] : (T)0, mpv = (exact && s < N) ? qs[9] : (T)0;
                const T Cvs = s <= N - 1 ? Cv : (T)0;
                const T dco = kk == 0 ? (T)2 * Cx : (kk == 1 ? (T)2 * Cy : (kk == 2 ? (T)2 * Cp + mpp : (T)2 * Cvs));
                const T oco = kk >= 2 ? mpv : (T)0;
                T bop[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bop[t] = dco * own[t] + oco * (kk == 2 ? gv[t] : gps[t]);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
                    if (16 * ti < 2 * s) {
                        if constexpr (LDSACC) {  // one tile row at a time through registers
                            acc_t row[NT];
#pragma unroll
                            for (int tj = 0; tj <= ti; ++tj) row[tj] = tile_load(ti, tj);
#pragma unroll
                            for (int tj = 0; tj <= ti; ++tj) row[tj] = Real<T>::mfma(own[ti], bop[tj], row[tj]);
#pragma unroll
                            for (int tj = 0; tj <= ti; ++tj) tile_store(ti, tj, row[tj]);
                        } else {
#pragma unroll
                            for (int tj = 0; tj <= ti; ++tj)
                                acc[(LDSACC ? 0 : ti * (ti + 1) / 2 + tj)] = Real<T>::mfma(own[ti], bop[tj], acc[(LDSACC ? 0 : ti * (ti + 1) / 2 + tj)]);
                        }
                    }
            }
            if (LDSACC && exact && s < N) {  // row 2s+1 of the second-order term, straight into the LDS matrix (lanes kk == 0: one per column)
                const T *qs = stg + KMPC_STG * s;
                const T mpd = qs[10], mvd = qs[11], mdd = qs[12];
                const int rho = 2 * s + 1;
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    const int col = 16 * tj + c;
                    if (kk == 0 && col <= rho) {
                        T val = mpd * gps[tj] + mvd * gv[tj];
                        if (col == rho) val += mdd;
                        Km[rho * ld + col] += val;
                    }
                }
            }
            if (!LDSACC && exact && s < N) {  // row 2s+1 (d_f of stage s) of the second-order term
                const T *qs = stg + KMPC_STG * s;
                const T mpd = qs[10], mvd = qs[11], mdd = qs[12];
                const int rho = 2 * s + 1, rt = rho >> 4, rr = rho & 15;
                const bool mine = kk == Real<T>::q_of_row(rr);
                const int reg = Real<T>::reg_of_row(rr);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
                    if (ti == rt) {
#pragma unroll
                        for (int tj = 0; tj <= ti; ++tj) {
                            T val = mpd * gps[tj] + mvd * gv[tj];
                            if (tj == ti && c == rr) val += mdd;
                            if (!mine) val = (T)0;
                            acc_t &A = acc[LDSACC ? 0 : ti * (ti + 1) / 2 + tj];
                            A[0] += reg == 0 ? val : (T)0;
                            A[1] += reg == 1 ? val : (T)0;
                            A[2] += reg == 2 ? val : (T)0;
                            A[3] += reg == 3 ? val : (T)0;
                        }
                    }
            }
        }
    }

    // ================= Frenet functor (MODEL 1), MKZMPCPathFollowerFrenet.jl:112-123 =================================
    // The dynamics couple s, e_y, e_psi through K(s) and 1/(1 - e_y K): no triangular cascade, so roll-out and costates are
    // serial recursions over the stages on wave-uniform values (every lane runs them; lane k keeps stage k).
    DEV T eval_frenet(const T (&U)[NV], Stage<T> &S)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int j = lane + 64 * i; if (j < n) xb[j] = U[i]; }
        WSYNC();
        const int k = lane;
        const bool st = k < N;
        T a = st ? xb[2 * k] : (T)0, d = st ? xb[2 * k + 1] : (T)0;
        T an = (k + 1 < N) ? xb[2 * k + 2] : a, dn = (k + 1 < N) ? xb[2 * k + 3] : d;
        S.a = a; S.d = d;
        T sd, cd;
        Real<T>::sincos_(d, &sd, &cd);
        const T Dn = cd * cd + rr_ * rr_ * sd * sd;
        const T rs = (T)1 / sqrt(Dn);
        S.sinb = rr_ * sd * rs;  // sin(atan(r tan d))   (:113)
        S.cosb = cd * rs;
        S.b1 = rr_ / Dn;         // d beta / d d_f
        S.b2 = rr_ * ((T)1 - rr_ * rr_) * ((T)2 * sd * cd) / (Dn * Dn);  // d2 beta / d d_f2
        if (st) { T *q = stg + KMPC_STG * k; q[13] = a; q[14] = S.sinb; q[15] = S.cosb; }  // slots 0..12 hold the stage Jacobians
        WSYNC();
        T s_ = x0, ey_ = y0, ep_ = psi0, v_ = v0;
        S.x = S.y = S.psi = S.v = S.c = S.s = S.K = S.Kp = S.iden = S.dsdt = (T)0;
        for (int kk = 0; kk <= N; ++kk) {
            const T K = ((kp0 * s_ + kp1) * s_ + kp2) * s_ + kp3;               // :112
            const T Kp = ((T)3 * kp0 * s_ + (T)2 * kp1) * s_ + kp2;
            if (lane == kk) { S.x = s_; S.y = ey_; S.psi = ep_; S.v = v_; S.K = K; S.Kp = Kp; }
            if (kk == N) break;
            const T *q = stg + KMPC_STG * kk;
            const T ak = q[13], sb = q[14], cbt = q[15];
            T sp, cp;
            Real<T>::sincos_(ep_, &sp, &cp);
            const T c = cp * cbt - sp * sb, sn = sp * cbt + cp * sb;            // cos / sin(e_psi + beta)
            const T iden = (T)1 / ((T)1 - ey_ * K), dsdt = v_ * c * iden;        // :114
            if (lane == kk) { S.c = c; S.s = sn; S.iden = iden; S.dsdt = dsdt; }
            s_ += dt * dsdt;                                                     // :118
            ey_ += dt * (v_ * sn);                                               // :119
            ep_ += dt * (v_ / Lb * sb - dsdt * K);                               // :120
            v_ += dt * ak;                                                       // :121
        }
        WSYNC();
        const bool cs = (k >= 1 && k <= N);
        S.ex = cs ? S.x - rx : (T)0;
        S.ey = cs ? S.y - ry : (T)0;
        S.ep = cs ? S.psi - rp : (T)0;
        S.ev = (k >= 1 && k <= N - 1) ? S.v - vt : (T)0;
        T Jl = Cx * S.ex * S.ex + Cy * S.ey * S.ey + Cp * S.ep * S.ep + Cv * S.ev * S.ev;  // Frenet.jl:97-98 (C_x = 0)
        if (st) Jl += Ca * a * a + Cd * d * d;
        if (k < N - 1) Jl += Cda * (an - a) * (an - a) + Cdd * (dn - d) * (dn - d);
        return uniform_(wave_sum(Jl));
    }

    // stage record of the Frenet functor: A00 A01 A02 A03 A12 A13 A20 A21 A22 A23 Bs Bey Bep  (A11 = A33 = 1, B_v,acc = dt)
    DEV void linearize_frenet(const Stage<T> &S, bool exact, T (&g)[NV])
    {
        const int k = lane;
        const bool st = k < N;
        if (k <= N) {
            T *q = stg + KMPC_STG * k;
            const T v = S.v, c = S.c, sn = S.s, K = S.K, Kp = S.Kp, iden = S.iden, dsdt = S.dsdt, ey = S.y, b1 = S.b1;
            const T ds_s = v * c * ey * Kp * iden * iden, ds_ey = v * c * K * iden * iden, ds_ep = -v * sn * iden, ds_v = c * iden,
                    ds_d = -v * sn * iden * b1;
            q[0] = st ? (T)1 + dt * ds_s : (T)0; q[1] = st ? dt * ds_ey : (T)0; q[2] = st ? dt * ds_ep : (T)0; q[3] = st ? dt * ds_v : (T)0;
            q[4] = st ? dt * v * c : (T)0; q[5] = st ? dt * sn : (T)0;
            q[6] = st ? dt * (-ds_s * K - dsdt * Kp) : (T)0; q[7] = st ? -dt * ds_ey * K : (T)0;
            q[8] = st ? (T)1 - dt * ds_ep * K : (T)0; q[9] = st ? dt * (S.sinb / Lb - ds_v * K) : (T)0;
            q[10] = st ? dt * ds_d : (T)0; q[11] = st ? dt * v * c * b1 : (T)0; q[12] = st ? dt * (v / Lb * S.cosb * b1 - ds_d * K) : (T)0;
            T *l = wb + 4 * k;  // stage cost gradient (wb is scratch here; stage_form_weights rewrites it later)
            l[0] = (T)2 * Cx * S.ex; l[1] = (T)2 * Cy * S.ey; l[2] = (T)2 * Cp * S.ep; l[3] = (T)2 * Cv * S.ev;
        }
        WSYNC();
        T l0 = wb[4 * N], l1 = wb[4 * N + 1], l2 = wb[4 * N + 2], l3 = wb[4 * N + 3];  // costate of state N
        if (lane == 0) { T *ql = stg + KMPC_STG * N + 16; ql[0] = l0; ql[1] = l1; ql[2] = l2; ql[3] = l3; }
        for (int kk = N - 1; kk >= 0; --kk) {
            const T *q = stg + KMPC_STG * kk;
            if (lane == 0) { xb[2 * kk] = dt * l3; xb[2 * kk + 1] = q[10] * l0 + q[11] * l1 + q[12] * l2; }  // B_k^T lambda_{k+1}
            const T t0 = q[0] * l0 + q[6] * l2;
            const T t1 = q[1] * l0 + l1 + q[7] * l2;
            const T t2 = q[2] * l0 + q[4] * l1 + q[8] * l2;
            const T t3 = q[3] * l0 + q[5] * l1 + q[9] * l2 + l3;
            const T *l = wb + 4 * kk;
            l0 = t0 + l[0]; l1 = t1 + l[1]; l2 = t2 + l[2]; l3 = t3 + l[3];  // (unused after kk = 0)
            if (lane == 0) { T *ql = stg + KMPC_STG * kk + 16; ql[0] = l0; ql[1] = l1; ql[2] = l2; ql[3] = l3; }
        }
        WSYNC();
        if (exact && st) {
            // second derivatives of the Euler step wrt (s, e_y, e_psi, v, d_f), contracted with the costate of state k+1 (slots 16..18
            // of the next record): M = dt [ (l0 - l2 K) Hess(g) - l2 (grad g grad K^T + grad K grad g^T) - l2 g K'' e_s e_s^T
            //                              + l1 Hess(v sin phi) + l2 Hess(v sin(beta) / L_b) ],  g = ds/dt = v cos(phi) D
            const T *ln = stg + KMPC_STG * (k + 1) + 16;
            const T m0 = ln[0], m1 = ln[1], m2 = ln[2];
            const T s_ = S.x, ey = S.y, v = S.v, C = S.c, Sn = S.s, K = S.K, K1 = S.Kp, K2 = (T)6 * kp0 * s_ + (T)2 * kp1;
            const T D = S.iden, b1 = S.b1, b2 = S.b2, gq = S.dsdt;
            const T Ds = ey * K1 * D * D, De = K * D * D;
            const T Dss = ey * K2 * D * D + (T)2 * ey * K1 * D * Ds, Dse = K1 * D * D + (T)2 * ey * K1 * D * De, Dee = (T)2 * K * D * De;
            const T g_s = v * C * Ds, g_e = v * C * De, g_p = -v * Sn * D, g_v = C * D, g_d = -v * Sn * b1 * D;
            const T w = m0 - m2 * K, a2 = m2 * K1;
            T *q = stg + KMPC_STG * k + 20;  // upper triangle, row-major: ss se sp sv sd | ee ep ev ed | pp pv pd | vv vd | dd
            q[0] = dt * (w * (v * C * Dss) - (T)2 * a2 * g_s - m2 * gq * K2);
            q[1] = dt * (w * (v * C * Dse) - a2 * g_e);
            q[2] = dt * (w * (-v * Sn * Ds) - a2 * g_p);
            q[3] = dt * (w * (C * Ds) - a2 * g_v);
            q[4] = dt * (w * (-v * Sn * b1 * Ds) - a2 * g_d);
            q[5] = dt * (w * (v * C * Dee));
            q[6] = dt * (w * (-v * Sn * De));
            q[7] = dt * (w * (C * De));
            q[8] = dt * (w * (-v * Sn * b1 * De));
            q[9] = dt * (w * (-v * C * D) + m1 * (-v * Sn));
            q[10] = dt * (w * (-Sn * D) + m1 * C);
            q[11] = dt * (w * (-v * C * b1 * D) + m1 * (-v * Sn * b1));
            q[12] = (T)0;
            q[13] = dt * (w * (-Sn * b1 * D) + m1 * (C * b1) + m2 * (S.cosb * b1 / Lb));
            q[14] = dt * (w * (v * D * (-C * b1 * b1 - Sn * b2)) + m1 * (v * (-Sn * b1 * b1 + C * b2)) + m2 * (v * (-S.sinb * b1 * b1 + S.cosb * b2) / Lb));
        }
        WSYNC();
        // input-cost terms, Frenet.jl:99-102 (same as the Cartesian model)
        const T aprev = __shfl_up(S.a, 1), dprev = __shfl_up(S.d, 1);
        const T anext = __shfl_down(S.a, 1), dnext = __shfl_down(S.d, 1);
        if (st) {
            T ga = xb[2 * k] + (T)2 * Ca * S.a, gd = xb[2 * k + 1] + (T)2 * Cd * S.d;
            if (k >= 1) { ga += (T)2 * Cda * (S.a - aprev); gd += (T)2 * Cdd * (S.d - dprev); }
            if (k < N - 1) { ga -= (T)2 * Cda * (anext - S.a); gd -= (T)2 * Cdd * (dnext - S.d); }
            xb[2 * k] = ga; xb[2 * k + 1] = gd;
        }
        WSYNC();
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int j = lane + 64 * i; g[i] = j < n ? xb[j] : (T)0; }
        WSYNC();
    }

    // Condensing with dense stage Jacobians: every lane keeps all four components of G for its column of each tile (replicated over
    // the four kk groups); the MFMA fragment is component kk.  exact: + the second-order term (5x5 stage blocks M of linearize_frenet):
    // its state part joins the weights of the contraction, its d_f row is added to row 2s+1 of the tiles.
    DEV void condense_dense(bool exact, acc_t (&acc)[NACC])
    {
        static_assert(!(MODEL == 1 && LDSACC), "the Frenet functor keeps its tiles in registers (NT <= 4)");
        const int kk = lane >> 4, c = lane & 15;
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = acc_t{0, 0, 0, 0};
        T g0[NT], g1[NT], g2[NT], g3[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) g0[t] = g1[t] = g2[t] = g3[t] = (T)0;
        for (int s = 0; s <= N; ++s) {
            if (s >= 1) {  // G_s = [A_{s-1} G_{s-1} | B_{s-1}]
                const T *q = stg + KMPC_STG * (s - 1);
                const T A00 = q[0], A01 = q[1], A02 = q[2], A03 = q[3], A12 = q[4], A13 = q[5], A20 = q[6], A21 = q[7], A22 = q[8], A23 = q[9];
                const T Bs = q[10], Bey = q[11], Bep = q[12];
                const int col0 = 2 * (s - 1);
                const T Cvs = s <= N - 1 ? Cv : (T)0;
                // row kk of the weight 2 Q_s + M_s^{zz}
                T w0 = kk == 0 ? (T)2 * Cx : (T)0, w1 = kk == 1 ? (T)2 * Cy : (T)0, w2 = kk == 2 ? (T)2 * Cp : (T)0, w3 = kk == 3 ? (T)2 * Cvs : (T)0;
                if (exact && s < N) {
                    const T *m = stg + KMPC_STG * s + 20;
                    w0 += kk == 0 ? m[0] : (kk == 1 ? m[1] : (kk == 2 ? m[2] : m[3]));
                    w1 += kk == 0 ? m[1] : (kk == 1 ? m[5] : (kk == 2 ? m[6] : m[7]));
                    w2 += kk == 0 ? m[2] : (kk == 1 ? m[6] : (kk == 2 ? m[9] : m[10]));
                    w3 += kk == 0 ? m[3] : (kk == 1 ? m[7] : (kk == 2 ? m[10] : m[12]));
                }
                T own[NT], bop[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const T n0 = A00 * g0[t] + A01 * g1[t] + A02 * g2[t] + A03 * g3[t];
                    const T n1 = g1[t] + A12 * g2[t] + A13 * g3[t];
                    const T n2 = A20 * g0[t] + A21 * g1[t] + A22 * g2[t] + A23 * g3[t];
                    g0[t] = n0; g1[t] = n1; g2[t] = n2;
                    const int col = 16 * t + c;
                    if (col == col0) { g0[t] = (T)0; g1[t] = (T)0; g2[t] = (T)0; g3[t] = dt; }
                    if (col == col0 + 1) { g0[t] = Bs; g1[t] = Bey; g2[t] = Bep; g3[t] = (T)0; }
                    own[t] = kk == 0 ? g0[t] : (kk == 1 ? g1[t] : (kk == 2 ? g2[t] : g3[t]));
                    bop[t] = w0 * g0[t] + w1 * g1[t] + w2 * g2[t] + w3 * g3[t];
                }
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
                    if (16 * ti < 2 * s) {
#pragma unroll
                        for (int tj = 0; tj <= ti; ++tj)
                            acc[LDSACC ? 0 : ti * (ti + 1) / 2 + tj] = Real<T>::mfma(own[ti], bop[tj], acc[LDSACC ? 0 : ti * (ti + 1) / 2 + tj]);
                    }
            }
            if (exact && s < N) {  // row 2s+1 (d_f of stage s): M^{z d}^T G_s, M^{dd} on the diagonal
                const T *m = stg + KMPC_STG * s + 20;
                const T msd = m[4], med = m[8], mpd = m[11], mvd = m[13], mdd = m[14];
                const int rho = 2 * s + 1, rt = rho >> 4, rr = rho & 15;
                const bool mine = kk == Real<T>::q_of_row(rr);
                const int reg = Real<T>::reg_of_row(rr);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
                    if (ti == rt) {
#pragma unroll
                        for (int tj = 0; tj <= ti; ++tj) {
                            T val = msd * g0[tj] + med * g1[tj] + mpd * g2[tj] + mvd * g3[tj];
                            if (tj == ti && c == rr) val += mdd;
                            if (!mine) val = (T)0;
                            acc_t &A = acc[LDSACC ? 0 : ti * (ti + 1) / 2 + tj];
                            A[0] += reg == 0 ? val : (T)0;
                            A[1] += reg == 1 ? val : (T)0;
                            A[2] += reg == 2 ? val : (T)0;
                            A[3] += reg == 3 ? val : (T)0;
                        }
                    }
            }
        }
    }

    // ---- (d) K (lower triangle, LDS) = sc*(H + input Hessian) + A^T W A + reg*I -----------------
    DEV void build_K(const acc_t (&acc)[NACC], T sc, T reg)
    {
        const int c = lane & 15;
        if constexpr (LDSACC) WSYNC();
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + Real<T>::row_of(lane, r), col = 16 * tj + c;
                    if (row < n && col <= row) {
                        const T h = LDSACC ? Km[row * ld + col] : acc[LDSACC ? 0 : ti * (ti + 1) / 2 + tj][r];  // in place when the tiles live in LDS
                        T v = sc * (h + input_hess(row, col)) + gram_entry(row, col);
                        if (row == col) v += reg;
                        Km[row * ld + col] = v;
                    }
                }
        WSYNC();
    }

    // Blocked Cholesky of the lower triangle in LDS on the matrix cores; false on a non-positive pivot.  Left-looking over
    // 16-column tile columns: the tiles of column TJ are loaded into MFMA C-layout registers, updated with every finished 4-column
    // panel to their left (fragments straight from the L already in LDS), then factored by four 4-column block steps -- the 4x4
    // diagonal block redundantly in every lane, lane (kk, c) solving panel row 16t + c and keeping component kk (the MFMA fragment
    // layout), trailing tiles of the same column updated by one MFMA each.  (The scalar left-looking version this replaces
    // spent 0.8 M cycles per factorisation at N = 50.)
    DEV bool cholesky()
    {
        const int c = lane & 15, kk = lane >> 4;
        for (int TJ = 0; TJ < NT; ++TJ) {
            const int nrt = NT - TJ, cb0 = 16 * TJ;
            acc_t kt[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = cb0 + 16 * t + Real<T>::row_of(lane, r), col = cb0 + c;
                    const int rr = row > col ? row : col, cc = row > col ? col : row;   // diagonal tile loaded symmetric
                    const bool valid = t < nrt && rr < n;
                    const T v = Km[valid ? rr * ld + cc : 0];
                    kt[t][r] = valid ? v : (row == col ? (T)1 : (T)0);                   // padding rows/cols: identity
                }
            // left update with the finished panels
            for (int kc = 0; kc < 4 * TJ; ++kc) {
                const int colk = 4 * kc + kk;
                T fr[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int row = cb0 + 16 * t + c;
                    const bool valid = t < nrt && row < n;
                    const T v = Km[valid ? row * ld + colk : 0];
                    fr[t] = valid ? v : (T)0;
                }
                const T nb = -fr[0];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (t < nrt) kt[t] = Real<T>::mfma(fr[t], nb, kt[t]);
            }
            // four block steps inside the tile column
            for (int q4 = 0; q4 < 4; ++q4) {
                const int j0 = cb0 + 4 * q4;
                if (j0 >= n) break;
                const int kp = c - 4 * q4;
                const bool holder = kp >= 0 && kp < 4;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rl = 16 * t + Real<T>::row_of(lane, r);
                        if (holder && t < nrt) pan[4 * rl + kp] = kt[t][r];
                    }
                WSYNC();
                const T *pd = pan + 4 * (4 * q4);
                const T d00 = pd[0], d10 = pd[4], d11 = pd[5], d20 = pd[8], d21 = pd[9], d22 = pd[10];
                const T d30 = pd[12], d31 = pd[13], d32 = pd[14], d33 = pd[15];
                const T r0 = (T)1 / sqrt(d00);
                const T l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
                const T e11 = d11 - l10 * l10, r1 = (T)1 / sqrt(e11);
                const T l21 = (d21 - l20 * l10) * r1, l31 = (d31 - l30 * l10) * r1;
                const T e22 = d22 - l20 * l20 - l21 * l21, r2 = (T)1 / sqrt(e22);
                const T l32 = (d32 - l30 * l20 - l31 * l21) * r2;
                const T e33 = d33 - l30 * l30 - l31 * l31 - l32 * l32, r3 = (T)1 / sqrt(e33);
                const T dmin = fmin(fmin(d00, e11), fmin(e22, e33)), dmax = fmax(fmax(d00, e11), fmax(e22, e33));
                if (!(dmin > Real<T>::tiny() && dmax < (T)1e300)) return false;  // wave-uniform
                if (lane == 0) { dinv[j0] = r0; dinv[j0 + 1] = r1; dinv[j0 + 2] = r2; dinv[j0 + 3] = r3; }
                const int jc = j0 + kk;
                T pf[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    pf[t] = (T)0;
                    if (t < nrt) {
                        const int rl = 16 * t + c, row = cb0 + rl;
                        const T *pa = pan + 4 * rl;
                        const T x0 = pa[0] * r0;
                        const T x1 = (pa[1] - x0 * l10) * r1;
                        const T x2 = (pa[2] - x0 * l20 - x1 * l21) * r2;
                        const T x3 = (pa[3] - x0 * l30 - x1 * l31 - x2 * l32) * r3;
                        const T xs = kk == 0 ? x0 : (kk == 1 ? x1 : (kk == 2 ? x2 : x3));
                        const bool live = row >= jc && row < n && jc < n;
                        if (live) { pf[t] = xs; Km[row * ld + jc] = xs; }
                    }
                }
                WSYNC();
                if (q4 < 3) {
                    const T nb = -pf[0];
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (t < nrt) kt[t] = Real<T>::mfma(pf[t], nb, kt[t]);
                }
            }
        }
        WSYNC();
        return true;
    }

    // x <- K^{-1} x with the factor in LDS: forward and backward substitution, one column per step.  The pivot component is
    // broadcast with v_readlane (wave-uniform column index) and the column entries of the next step are fetched while the current
    // FMA chain runs -- the version with __shfl (an LDS round trip per column) took 110 k cycles per solve at N = 50.
    DEV void chol_solve(T (&x)[NV])
    {
        T cn[NV], dn = dinv[0];
#pragma unroll
        for (int ii = 0; ii < NV; ++ii) { const int i = lane + 64 * ii; cn[ii] = i < n ? Km[i * ld] : (T)0; }
        for (int j = 0; j < n; ++j) {
            T c[NV];
            const T dj = dn;
            const int jn = j + 1 < n ? j + 1 : j;
            dn = dinv[jn];
#pragma unroll
            for (int ii = 0; ii < NV; ++ii) { c[ii] = cn[ii]; const int i = lane + 64 * ii; cn[ii] = Km[(i < n ? i : 0) * ld + jn]; }
            T xj = readlane_(x[0], j & 63);
            if (NV > 1) { const T x1 = readlane_(x[NV - 1], j & 63); if (j >= 64) xj = x1; }
            xj *= dj;
#pragma unroll
            for (int ii = 0; ii < NV; ++ii) {
                const int i = lane + 64 * ii;
                const T upd = x[ii] - c[ii] * xj;
                x[ii] = i == j ? xj : ((i > j && i < n) ? upd : x[ii]);
            }
        }
        dn = dinv[n - 1];
#pragma unroll
        for (int ii = 0; ii < NV; ++ii) { const int i = lane + 64 * ii; cn[ii] = Km[(n - 1) * ld + (i < n ? i : 0)]; }
        for (int j = n - 1; j >= 0; --j) {
            T c[NV];
            const T dj = dn;
            const int jp = j >= 1 ? j - 1 : 0;
            dn = dinv[jp];
#pragma unroll
            for (int ii = 0; ii < NV; ++ii) { c[ii] = cn[ii]; const int i = lane + 64 * ii; cn[ii] = Km[jp * ld + (i < n ? i : 0)]; }
            T xj = readlane_(x[0], j & 63);
            if (NV > 1) { const T x1 = readlane_(x[NV - 1], j & 63); if (j >= 64) xj = x1; }
            xj *= dj;
#pragma unroll
            for (int ii = 0; ii < NV; ++ii) {
                const int i = lane + 64 * ii;
                const T upd = x[ii] - c[ii] * xj;
                x[ii] = i == j ? xj : (i < j ? upd : x[ii]);
            }
        }
    }

    // well-centred strictly interior point: first inputs = point of the first-step interval closest to 0,
    // a quarter of its width inside; later accelerations steer v_k off a speed bound (u_1-u_0 is rate-free, Q1);
    // returns false when the first-step / speed bounds are inconsistent (Q5)
    DEV bool interior_point(T (&Uf)[NV])
    {
        const T relax = (T)P.relax;
        // first guess of the solution inside the bounds (same rule as the CPU checker): accelerations approach
        // the reference speed (time constant 1 s), steering the kinematic feed-forward of the reference's mean curvature;
        // reference points 1..N only -- point 0 is a dead input (Q3)
        const T frac = (T)0.6, rr = (T)P.r;
        T len, kap;
        {
            const T rxn = __shfl_down(rx, 1), ryn = __shfl_down(ry, 1);
            const T seg = (lane >= 1 && lane < N) ? sqrt((rxn - rx) * (rxn - rx) + (ryn - ry) * (ryn - ry)) : (T)0;
            len = uniform_(wave_sum(seg));
            kap = (readlane_(rp, N) - readlane_(rp, 1)) / fmax(len, (T)1e-6);
        }
        if (MODEL == 1) kap = ((kp0 * x0 + kp1) * x0 + kp2) * x0 + kp3;  // Frenet: curvature of the polynomial at s0
        const T vref = MODEL == 1 ? vt : len / ((T)(N - 1) * dt);
        const T sb = fmin(fmax((T)P.L_b * kap, (T)-0.9), (T)0.9);
        const T ffw = P.start == 1 ? (T)0 : (T)1;  // start = 1: the reference's all-zero start (MKZMPCPathFollower.jl:65-72), no feed-forward (kmpc_ipm.h)
        const T dff = ffw * fmin(fmax(atan(sb / sqrt((T)1 - sb * sb) / rr)  /* tan(asin(sb)) = sb / sqrt(1 - sb^2), |sb| <= 0.9 */, -frac * (T)P.steer_max), frac * (T)P.steer_max);
        const T aff = ffw * fmin(fmax(vref - v0, -frac * (T)P.a_max), frac * (T)P.a_max);
        T u0[2];
        // Q5: v[1] = v0 is itself bounded in the reference model -> any v0 outside the (relaxed) speed bounds is infeasible
        bool ok = v0 >= (T)P.v_min - relax * fmax((T)1, fabs((T)P.v_min)) && v0 <= (T)P.v_max + relax * fmax((T)1, fabs((T)P.v_max));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const T ub = j ? (T)P.steer_max : (T)P.a_max;
            const T d0 = (j ? (T)P.steer_dmax : (T)P.a_dmax) * dtc;
            const T up = j ? up1 : up0;
            T lo = fmax(-ub - relax * fmax((T)1, ub), up - d0 - relax * fmax((T)1, d0));
            T hi = fmin(ub + relax * fmax((T)1, ub), up + d0 + relax * fmax((T)1, d0));
            if (j == 0) {
                lo = fmax(lo, ((T)P.v_min - relax * fmax((T)1, fabs((T)P.v_min)) - v0) / dt);
                hi = fmin(hi, ((T)P.v_max + relax * fmax((T)1, fabs((T)P.v_max)) - v0) / dt);
            }
            if (!(lo < hi)) ok = false;
            const T push = (T)0.25 * (hi - lo);
            u0[j] = fmin(fmax(j ? dff : aff, lo + push), hi - push);
        }
        // later inputs: strictly feasible by construction (round 4; the rule and its argument: ipm::interior_point in kmpc_ipm.h)
        const T vm = fmin((T)1, (T)0.25 * ((T)P.v_max - (T)P.v_min));
        const T dstep = frac * (T)P.steer_dmax * dt;
        const T vstar = fmin(fmax(vref, (T)P.v_min + vm), (T)P.v_max - vm);
        T v = v0 + dt * u0[0], dp = u0[1];
#pragma unroll
        for (int i = 0; i < NV; ++i) Uf[i] = (T)0;
        if (lane == 0) Uf[0] = u0[0];
        if (lane == 1) Uf[0] = u0[1];
        for (int k = 1; k < N; ++k) {  // uniform scalar recurrence, N steps
            const T a = ffw * fmin(fmax(vstar - v, -frac * (T)P.a_max), frac * (T)P.a_max);
            const T d = fmin(fmax(dff, dp - dstep), dp + dstep);
            const int j = 2 * k;
#pragma unroll
            for (int i = 0; i < NV; ++i) { if (lane + 64 * i == j) Uf[i] = a; if (lane + 64 * i == j + 1) Uf[i] = d; }
            v += dt * a; dp = d;
        }
        return ok;
    }

    // ---- hooks of the shared interior-point state machine ---------------------------------------------------------------------------------
    DEV int dim_N() const { return N; }
    DEV int dim_n() const { return n; }
    DEV int dim_nf() const { return nf; }
    DEV T up(int j) const { return j ? up1 : up0; }
    DEV T xoff() const { return xoff_; }
    DEV T yoff() const { return yoff_; }
    DEV void refresh_ids() {}
    DEV T &cu(int i) { return cu_[i]; }
    DEV T &cl(int i) { return cl_[i]; }
    DEV void save_best(const T (&U)[NV])
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) ubest_[i] = U[i];
    }
    DEV void load_best(T (&U)[NV]) const
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) U[i] = ubest_[i];
    }
    template <int NS, int NM> DEV void reduce(T (&sm)[NS < 1 ? 1 : NS], T (&mx)[NM < 1 ? 1 : NM]) const
    {
#pragma unroll
        for (int i = 0; i < NS; ++i) sm[i] = uniform_(wave_sum(sm[i]));   // (the butterfly leaves bit-identical values in every lane)
#pragma unroll
        for (int i = 0; i < NM; ++i) mx[i] = uniform_(wave_max(mx[i]));
    }
    template <int NS, int NM> DEV void reduce_flag(T (&sm)[NS < 1 ? 1 : NS], T (&mx)[NM < 1 ? 1 : NM], bool &all_true) const
    {
        all_true = __all(all_true);
        reduce<NS, NM>(sm, mx);
    }
    DEV T max_any(T x) const { return uniform_(wave_max(x)); }
    DEV void form_bounds(int f, T &bu, T &bl) const { T rlx; form_bounds(f, bu, bl, rlx); }
    DEV T form_relax(int f, bool upper) const   // the relaxation contained in form_bounds (speed forms relax upper / lower separately)
    {
        const T relax = (T)P.relax;
        if (f < n) return relax * fmax((T)1, (f & 1) ? (T)P.steer_max : (T)P.a_max);
        if (f < n + R) { const int r = f - n; return relax * fmax((T)1, ((r & 1) ? (T)P.steer_dmax : (T)P.a_dmax) * ((r >> 1) == 0 ? dtc : dt)); }
        return relax * fmax((T)1, fabs(upper ? (T)P.v_max : (T)P.v_min));
    }
    DEV void forms_applyT(const T (&w)[NF], T (&o)[NV])
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) o[i] = (T)0;
        forms_applyT_add(w, o);
    }
    DEV void linearize(const Stage<T> &S, bool exact, T (&g)[NV])
    {
        linearize_model(S, exact, g);
#pragma unroll
        for (int i = 0; i < NV; ++i) gk[i] = g[i];
        exact_now = exact; need_condense = true;
    }
    DEV T grad(int i) const { return gk[i]; }
    DEV void drop_second_order() { exact_now = false; need_condense = true; }   // (the condensing takes the Gauss-Newton / exact choice as a flag)
    // condense (kept across re-factorisations) + K = sc*(H + input Hessian) + A^T W A + reg*I in LDS + blocked Cholesky; needs stage_form_weights done
    DEV bool kkt_factor(T sc, T reg, bool want_hmax)
    {
        sc_ = sc;
        if (need_condense) {
            condense(exact_now, acc);
            need_condense = false;
            if (want_hmax) {  // max |sc * H_jj|: scale of the delta_w shift
                T hm = 0;
                if constexpr (LDSACC) {
                    WSYNC();
                    for (int j = lane; j < n; j += 64) hm = fmax(hm, fabs(sc * Km[j * ld + j]));
                } else {
#pragma unroll
                    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (Real<T>::row_of(lane, r) == (lane & 15)) hm = fmax(hm, fabs(sc * acc[LDSACC ? 0 : ti * (ti + 1) / 2 + ti][r]));
                }
                cs[C_HMAX] = uniform_(wave_max(hm));
            }
        }
        STAMP(3);
        build_K(acc, sc, reg);
        STAMP(4);
        const bool ok = cholesky();
        if (!ok && LDSACC) need_condense = true;  // the tiles lived in the matrix the factorisation just overwrote
        return ok;
    }
    DEV void kkt_affine(T (&x)[NV])                          // K^-1 (-sc g)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) x[i] = -sc_ * gk[i];
        chol_solve(x);
    }
    DEV void kkt_direction(const T (&b)[NV], T (&x)[NV])     // K^-1 (-sc g + b)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) x[i] = -sc_ * gk[i] + b[i];
        chol_solve(x);
    }

    DEV void solve(const KIO<T> &io, int b) { ipm::solve(*this, io, b); }
};

template <typename T, int NT>
__global__ __launch_bounds__(64) void kmpc_solve_kernel(KP P, KIO<T> io)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= P.B) return;
    const int b = io.perm ? io.perm[blockIdx.x] : (int)blockIdx.x;
#ifdef KMPC_POISON  // diagnostic build (make poison): every LDS word starts as NaN
    { T *w_ = reinterpret_cast<T *>(smem); const int ne_ = (int)(kmpc_lds_bytes<T>(P.N, NT) / sizeof(T));
      for (int e = threadIdx.x; e < ne_; e += 64) w_[e] = (T)NAN;
      __syncthreads(); }
#endif
    T scalars[16], params[32];
    Solver<T, NT> sv(P, smem, scalars, params);
    ipm::load_problem_io(sv, io, b);
    sv.solve(io, b);
}

template <typename T, int NT>
__global__ __launch_bounds__(64) void kmpc_solve_frenet_kernel(KP P, KIO<T> io)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= P.B) return;
    const int b = (int)blockIdx.x;
#ifdef KMPC_POISON  // diagnostic build (make poison): every LDS word starts as NaN
    { T *w_ = reinterpret_cast<T *>(smem); const int ne_ = (int)(kmpc_lds_bytes<T>(P.N, NT) / sizeof(T));
      for (int e = threadIdx.x; e < ne_; e += 64) w_[e] = (T)NAN;
      __syncthreads(); }
#endif
    T scalars[16], params[32];
    Solver<T, NT, 1> sv(P, smem, scalars, params);
    sv.load_problem(io.z0, io.ref, io.vt, io.up, b);  // io.ref = k_poly [B,4]
    sv.solve(io, b);
}

// diagnostics: condensed Hessian / gradient / cost at a given U (used by the parity tests)
template <typename T, int NT>
__global__ __launch_bounds__(64) void kmpc_condense_kernel(KP P, KDbg<T> io)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef Solver<T, NT> SV;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= P.B) return;
    T scalars[16], params[32];
    SV sv(P, smem, scalars, params);
    sv.load_problem(io.z0, io.ref, io.vt, nullptr, b);
    const int n = 2 * P.N;
    T U[SV::NV], g[SV::NV];
#pragma unroll
    for (int i = 0; i < SV::NV; ++i) { const int j = lane + 64 * i; U[i] = j < n ? io.U[(size_t)b * n + j] : (T)0; }
    Stage<T> S;
    const T J = sv.eval(U, S);
    sv.linearize_model(S, P.hessian == 1, g);
    typename SV::acc_t acc[SV::NACC];
    sv.condense(P.hessian == 1, acc);
    if (SV::LDSACC) WSYNC();
    T *H = io.H + (size_t)b * n * n;
    const int c = lane & 15;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + Real<T>::row_of(lane, r), col = 16 * tj + c;
                if (row < n && col <= row) {
                    const T v = (SV::LDSACC ? sv.Km[row * sv.ld + col] : acc[SV::LDSACC ? 0 : ti * (ti + 1) / 2 + tj][r]) + sv.input_hess(row, col);
                    H[row * n + col] = v;
                    H[col * n + row] = v;
                }
            }
#pragma unroll
    for (int i = 0; i < SV::NV; ++i) { const int j = lane + 64 * i; if (j < n) io.g[(size_t)b * n + j] = g[i]; }
    if (lane == 0) io.J[b] = J;
}

template <typename T>
__global__ __launch_bounds__(64) void kmpc_mfma_probe_kernel(const T *a, const T *b, T *d)
{
    const int lane = threadIdx.x;
    typename Real<T>::acc_t acc = {0, 0, 0, 0};
    acc = Real<T>::mfma(a[lane], b[lane], acc);
    for (int r = 0; r < 4; ++r) d[lane * 4 + r] = acc[r];
}

// ---- launchers (called from kmpc_api.hip) ------------------------------------------------------
// The dynamic-LDS limit of a kernel is a property of the (function, device) pair: it is raised once, the first time a launch
// needs more than what was set before (horizons differ in their LDS need), not on every launch.
template <typename K> static hipError_t ensure_dynamic_lds(K kernel, size_t lds, size_t (&set_for_device)[64])
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (lds <= set_for_device[dev]) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) set_for_device[dev] = lds;
    return e;
}
template <typename T, int NT>
static hipError_t launch_solve_nt(const KP &P, const KIO<T> &io, hipStream_t st)
{
    const size_t lds = kmpc_lds_bytes<T>(P.N, NT);
    static size_t lds_set[64] = {0};
    hipError_t e = ensure_dynamic_lds(&kmpc_solve_kernel<T, NT>, lds, lds_set);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((kmpc_solve_kernel<T, NT>), dim3(P.B), dim3(64), lds, st, P, io);
    return hipGetLastError();
}
template <typename T, int NT>
static hipError_t launch_solve_frenet_nt(const KP &P, const KIO<T> &io, hipStream_t st)
{
    const size_t lds = kmpc_lds_bytes<T>(P.N, NT);
    static size_t lds_set[64] = {0};
    hipError_t e = ensure_dynamic_lds(&kmpc_solve_frenet_kernel<T, NT>, lds, lds_set);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((kmpc_solve_frenet_kernel<T, NT>), dim3(P.B), dim3(64), lds, st, P, io);
    return hipGetLastError();
}
template <typename T, int NT>
static hipError_t launch_condense_nt(const KP &P, const KDbg<T> &io, hipStream_t st)
{
    const size_t lds = kmpc_lds_bytes<T>(P.N, NT);
    static size_t lds_set[64] = {0};
    hipError_t e = ensure_dynamic_lds(&kmpc_condense_kernel<T, NT>, lds, lds_set);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((kmpc_condense_kernel<T, NT>), dim3(P.B), dim3(64), lds, st, P, io);
    return hipGetLastError();
}

#define KMPC_DISPATCH_NT(FN, ...)                                   \
    switch (NT) {                                                   \
        case 1: return FN<T, 1>(__VA_ARGS__);                       \
        case 2: return FN<T, 2>(__VA_ARGS__);                       \
        case 3: return FN<T, 3>(__VA_ARGS__);                       \
        case 4: return FN<T, 4>(__VA_ARGS__);                       \
        case 5: return FN<T, 5>(__VA_ARGS__);                       \
        case 6: return FN<T, 6>(__VA_ARGS__);                       \
        case 7: return FN<T, 7>(__VA_ARGS__);                       \
        default: return hipErrorInvalidValue;                       \
    }

template <typename T> hipError_t kmpc_launch_solve(const KP &P, const KIO<T> &io, hipStream_t st)
{
    const int NT = (2 * P.N + 15) / 16;
    KMPC_DISPATCH_NT(launch_solve_nt, P, io, st)
}
// Frenet functor: built for horizons up to N = 24 (the reference's is N = 8)
template <typename T> hipError_t kmpc_launch_solve_frenet(const KP &P, const KIO<T> &io, hipStream_t st)
{
    switch ((2 * P.N + 15) / 16) {
        case 1: return launch_solve_frenet_nt<T, 1>(P, io, st);
        case 2: return launch_solve_frenet_nt<T, 2>(P, io, st);
        case 3: return launch_solve_frenet_nt<T, 3>(P, io, st);
        default: return hipErrorInvalidValue;
    }
}
template <typename T> hipError_t kmpc_launch_condense(const KP &P, const KDbg<T> &io, hipStream_t st)
{
    const int NT = (2 * P.N + 15) / 16;
    KMPC_DISPATCH_NT(launch_condense_nt, P, io, st)
}
template <typename T> hipError_t kmpc_launch_probe(const T *a, const T *b, T *d, hipStream_t st)
{
    hipLaunchKernelGGL((kmpc_mfma_probe_kernel<T>), dim3(1), dim3(64), 0, st, a, b, d);
    return hipGetLastError();
}

template hipError_t kmpc_launch_solve<double>(const KP &, const KIO<double> &, hipStream_t);
template hipError_t kmpc_launch_solve<float>(const KP &, const KIO<float> &, hipStream_t);
template hipError_t kmpc_launch_solve_frenet<double>(const KP &, const KIO<double> &, hipStream_t);
template hipError_t kmpc_launch_solve_frenet<float>(const KP &, const KIO<float> &, hipStream_t);
template hipError_t kmpc_launch_condense<double>(const KP &, const KDbg<double> &, hipStream_t);
template hipError_t kmpc_launch_condense<float>(const KP &, const KDbg<float> &, hipStream_t);
template hipError_t kmpc_launch_probe<double>(const double *, const double *, double *, hipStream_t);
template hipError_t kmpc_launch_probe<float>(const float *, const float *, float *, hipStream_t);
