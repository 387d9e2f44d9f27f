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
#include "kmpc_ipm.h"


template <typename T> using StageF = StageV<T>;
// linearisation scalars of stage k, parked in LDS (16 words per stage):
// A02 A03 A12 A13 A23 Bdx Bdy Bdp mpp mpv mpd mvd mdd 0
constexpr int LIN_STRIDE = 16;

// MODEL 0: Cartesian kinematic bicycle (MKZMPCPathFollower.jl); MODEL 1: Frenet-frame functor (MKZMPCPathFollowerFrenet.jl:112-123): states
// (s, e_y, e_psi, v) in the (x, y, psi, v) slots, zero cost references, curvature polynomial K(s); only roll-out, costates and the
// sensitivity recursion differ (the dynamics couple s, e_y, e_psi, so they are serial recursions on wave-uniform values and the stage
// Jacobians are dense, and the condensing is a matrix-core contraction instead of the adjoint recursion) -- forms, barrier method, KKT
// assembly, factorisation and substitutions are shared.
template <typename T, int N, int MODEL = 0> struct FastSolver {
    static constexpr int LSTR = MODEL == 1 ? KMPC_STG : LIN_STRIDE;  // stage record stride (Frenet: 13 Jacobian + 3 roll-out + 4 costate + 15 Hessian)
    KMPC_HORIZON_CONSTANTS(N)
    static constexpr int NF = (nf + 63) / 64;
    static constexpr int NT = (n + 15) / 16, NTT = NT * (NT + 1) / 2;
    static constexpr int NROWS = (n + 1 + 15) / 16;  // rows that carry n-vector data
    static_assert(n + 1 <= 64 && n % 8 == 0, "fast kernel needs 2N + 1 <= 64 and N % 4 == 0");
    typedef typename Real<T>::acc_t acc_t;
    typedef T real;
    static constexpr int N_ = N, NTH = 64, GS = 64, MODEL_ID = MODEL;   // what kmpc_ipm.h reads: horizon, threads per problem, stride of the G_N table, functor
    // Split adjoint recursion (kmpc_ipm.h, condense_adjoint): lanes n .. n + 2M - 1 run the stages below M for the columns < 2M while lanes
    // 0 .. n - 1 run the stages from M up -- M = N / 2 where 3N lanes exist (N <= 20: both halves N / 2 trips), else what the spare lanes allow
    static constexpr int MSPLIT = MODEL == 1 ? 0 : ((64 - n) / 2 < N / 2 ? (64 - n) / 2 : N / 2), LOW0 = n, GMS = 2 * MSPLIT;
    static constexpr int TM = (2 * MSPLIT + 15) / 16;   // 16 x 16 tiles per side that the rank-4 correction touches
    static_assert(3 * GMS <= 64 && n + 6 * MSPLIT <= 64 * NF, "G_M lives in the spare row of opb, p(M) and the diagonal shares behind the staging rows of cub / clb");
    static constexpr int lds_elems() { return ((LC + 1) & ~1) + 64 + 64 * NF + 64 + LSTR * (N + 1) + 8 * 64 + 16 + 64 + 64 + 2 * 64 * NF + 16 * (n / 4) + (sizeof(T) == 8 ? KC_COUNT : 0); }

    STAMP_MEMBERS
    const KP &P;
    int lane, vid;  // stage index / input-form slot (the same lane here); re-materialised (opaque) at the top of every iteration: stops
                    // LICM from hoisting the lane-derived index / mask arithmetic of every phase out of the loop into long-lived VGPRs
    T *Lc, *xb, *wb, *cb, *lin, *opb, *gnb, *gmb, *hm, *cs, *ubest, *gb, *cub, *clb, *sinvb;
    Coef<T> kc;  // polynomial coefficients (LDS table in fp64)
    T x0, y0, psi0, v0, vt, up0, up1, rx, ry, rp, xoff_, yoff_;
    T kp0, kp1, kp2, kp3;  // Frenet: K(s) = kp0 s^3 + kp1 s^2 + kp2 s + kp3
    // (the scalar table pt[] / cwt[]: kmpc_ipm.h)
    const T *pt, *cwt;

    DEV FastSolver(const KP &p, unsigned char *smem) : P(p), lane(threadIdx.x), vid(threadIdx.x)
    {
        Lc = reinterpret_cast<T *>(smem);
        xb = Lc + ((LC + 1) & ~1);
        wb = xb + 64;
        cb = wb + 64 * NF;
        lin = cb + 64;
        opb = lin + LSTR * (N + 1);
        gnb = opb + 4 * 64;     // terminal sensitivities G_N [3][64] (Cartesian model): the Cholesky panel scratch uses opb[0 .. 4 * 64) only
        gmb = opb + 7 * 64;     // mid-horizon sensitivities G_M [3][2M] of the split recursion
        cs = opb + 8 * 64;      // wave-uniform scalars that are read once or twice per iteration live here, not in VGPRs
        ubest = cs + 16;        // last iterate that passed Ipopt's test
        gb = ubest + 64;        // gradient of the current linearisation (lane j: g_j)
        cub = gb + 64; clb = cub + 64 * NF;  // corrector terms
        hm = cub + n + 4 * MSPLIT;  // (split recursion) p_j(M): rows 0, 1 behind the n staging entries of cub, rows 2, 3 behind those of clb; then the diagonal shares
        sinvb = clb + 64 * NF;  // D_j^-1 of the factor's 4x4 diagonal blocks (row-major, 16 per 4-column panel)
        for (int e = lane; e < 16 * (n / 4); e += 64) sinvb[e] = (T)0;
        kc.tab = sinvb + 16 * (n / 4);
        if (sizeof(T) == 8 && lane < KC_COUNT) const_cast<T *>(kc.tab)[lane] = (T)kmpc_coef[lane];
        static_assert(N <= 32, "the scalar table shares the 64-entry cb buffer with the N suffix sums");
        pt = cb + 32; cwt = pt + PT_W;
        if (lane == 0) ipm::fill_param_table(cb + 32, p, nf);
        // the tables are written by a few lanes and read by all: order the reads behind the writes (found with poisoned LDS in the
        // Frenet instantiation, whose first reads of the table were scheduled ahead of lane 0's stores -- invisible whenever the
        // previous occupant of the LDS was the same kernel, because the constants it left behind are the same)
        WSYNC();
    }

    DEV void load_problem(const T *z0, const T *ref, const T *vtp, const T *upp, int b)
    {
        if (MODEL == 1) {  // `ref` carries k_poly [B,4]; (s, e_y) are not translation-invariant (K depends on s); zero cost references
            xoff_ = yoff_ = (T)0; x0 = z0[4 * (size_t)b]; y0 = z0[4 * (size_t)b + 1];
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
        xoff_ = z0[4 * (size_t)b]; yoff_ = z0[4 * (size_t)b + 1]; x0 = (T)0; y0 = (T)0;
        psi0 = z0[4 * (size_t)b + 2]; v0 = z0[4 * (size_t)b + 3];
        vt = vtp[b];
        up0 = upp[2 * (size_t)b]; up1 = upp[2 * (size_t)b + 1];
        rx = ry = rp = (T)0;
        if (lane <= N) {
            const T *r = ref + ((size_t)b * (N + 1) + lane) * 3;
            rx = r[0] - xoff_; ry = r[1] - yoff_; rp = r[2];
        }
    }

    // ---- hooks of the shared interior-point code (kmpc_ipm.h) ------------------------------------------------------------------------
    DEV T up(int j) const { return j ? up1 : up0; }
    DEV T xoff() const { return xoff_; }
    DEV T yoff() const { return yoff_; }
    DEV bool rec_writer() const { return true; }
    DEV void refresh_ids() { asm volatile("" : "+v"(lane)); vid = lane; }
    DEV T *pm(int c) const { return (c < 2 ? cub : clb) + n + (c & 1) * 2 * MSPLIT; }   // component c of p_j(M), j < 2M (the corrector buffers are dead while K is built)
    DEV T &cu(int i) { return cub[lane + 64 * i]; }   // corrector terms live in LDS
    DEV T &cl(int i) { return clb[lane + 64 * i]; }
    DEV void save_best1(T U) { ubest[lane] = U; }
    DEV T load_best1() const { return ubest[lane]; }
    KMPC_IPM_ONE_SLOT_HOOKS
    template <int NS, int NM> DEV void reduce(T (&sm)[NS < 1 ? 1 : NS], T (&mx)[NM < 1 ? 1 : NM])   // sums and NON-NEGATIVE maxima, wave-uniform results
    {
#pragma unroll
        for (int i = 0; i < NS; ++i) sm[i] = dpp_sum(sm[i]);
#pragma unroll
        for (int i = 0; i < NM; ++i) mx[i] = dpp_max_nn(mx[i]);
    }
    template <int NS, int NM> DEV void reduce_flag(T (&sm)[NS < 1 ? 1 : NS], T (&mx)[NM < 1 ? 1 : NM], bool &all_true)
    {
        all_true = __all(all_true);
        reduce<NS, NM>(sm, mx);
    }
    DEV T sum_stages(T x) const { return dpp_sum(x); }
    DEV T stage_bcast(T x, int k) const { return readlane_(x, k); }
    DEV T max_any(T x) const { return dpp_max(x); }
    DEV void forms_apply(T x, T (&y)[NF]) { ipm::forms_apply(*this, x, y); }   // (scalar forms for the Frenet functor's own code)
    DEV T forms_applyT(const T (&w)[NF]) { return ipm::forms_applyT(*this, w); }
    // roll-out + objective at U (lane j: U_j)
    DEV T eval1(T U, StageF<T> &S)
    {
        if constexpr (MODEL == 1) return eval_frenet(U, S);
        else return ipm::eval_cartesian(*this, U, S);
    }
    // costates -> gradient (returned, lane j: g_j, and left in gb); per-stage scalars go to the LDS records
    DEV T linearize1(const StageF<T> &S, bool exact)
    {
        if constexpr (MODEL == 1) { const T g = linearize_frenet(S, exact); gb[lane] = g; return g; }
        else return ipm::linearize_cartesian(*this, S, exact);
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

    // The second-order entries of the stage records decide between the exact and the Gauss-Newton matrix: linearize writes them only
    // when the exact Hessian is wanted, and a fallback inside an iteration clears them.
    DEV void drop_second_order()
    {
        if constexpr (MODEL == 1) {
            T z = (T)0;
            pin(z);  // materialised here: hoisted out of the iteration loop this zero would occupy (and spill) a VGPR pair for the whole solve
            if (lane <= N) { T *q = lin + LSTR * lane + 20; for (int i = 0; i < 15; ++i) q[i] = z; }
            WSYNC();
        } else ipm::drop_second_order_cartesian(*this);
    }
    DEV void condense(T sc, acc_t (&acc)[NTT])
    {
        if constexpr (MODEL == 1) condense_frenet(sc, acc);   // dense stage Jacobians: MFMA contraction into `acc`
        else ipm::condense_adjoint(*this, sc);                 // Cartesian model: O(N^2) adjoint recursion into the packed image, `acc` untouched
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
        // split recursion: rows and columns < 2M of the image lack G_M^T P_M -- a rank-4 product, one matrix-core instruction per tile, issued
        // first so that the matrix pipe's latency passes behind the staging exchange
        constexpr int NCM = TM * (TM + 1) / 2;
        acc_t cmt[NCM < 1 ? 1 : NCM];
        if constexpr (ADJ && MSPLIT > 0) {
            T fa[TM], fb[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) { fa[t] = ipm::split_fragment_a(*this, t, c, lane >> 4); fb[t] = ipm::split_fragment_b(*this, t, c, lane >> 4); }
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                for (int tj = 0; tj <= ti; ++tj) cmt[ti * (ti + 1) / 2 + tj] = Real<T>::mfma(fa[ti], fb[tj], acc_t{0, 0, 0, 0});
        }
        ipm::kkt_diag_staging(*this, sc, reg, ADJ, dgs, sbs);
#pragma unroll
        for (int tj = 0; tj < NTF; ++tj) {
            const int col = 16 * tj + c;
            const bool colok = col < n;
            const int cs_ = colok ? col : 0;
            const T dgv = dgs[cs_], sbv = sbs[cs_], rhv = -sc * gb[cs_];
            const T *colK = Lc + offc_rt(cs_);
#pragma unroll
            for (int ti = tj; ti < NTF; ++ti) {
                acc_t cm = acc_t{0, 0, 0, 0};
                if constexpr (ADJ && MSPLIT > 0) { if (ti < TM) cm = cmt[ti < TM ? ti * (ti + 1) / 2 + tj : 0]; }
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
                        if (ADJ && MSPLIT > 0 && ti < TM) v += cm[r];
                        if (ti == tj) v += row == col ? dgv : (T)0;
                        if (ti <= tj + 1) v += row == col + 2 ? sbv : (T)0;
                    }
                    if (ti == NTF - 1) { v = row == n ? rhv : v; v = row <= n ? v : (T)0; }
                    if (tj == NTF - 1) v = colok ? v : (T)0;
                    kt[ti * (ti + 1) / 2 + tj][r] = v;
                }
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
            const ipm::Diag4<T> dd = ipm::load_diag4(pan + 4 * j0);
            // panel rows are independent of the diagonal factor until the solve: issue their loads now
            T a[NTF][4];
#pragma unroll
            for (int t = TJ; t < NTF; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) a[t][k] = pan[4 * (16 * t + c) + k];
            const ipm::Chol4<T> c4 = ipm::factor_diag4(dd);
            if (!c4.ok) return false;  // not positive definite (wave-uniform)
            if (lane == 0) c4.store_inv(sinvb + 16 * jb);
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
                T x[4];
                c4.solve_row(a[t], x);
                const T xs = kk == 0 ? x[0] : (kk == 1 ? x[1] : (kk == 2 ? x[2] : x[3]));
                const bool live = row >= jc && row <= n;
                pf[t] = live ? xs : (T)0;                       // component kk of L (MFMA fragment of the trailing update)
                if (live) colL[row] = fma(x[3], c3, fma(x[2], c2, fma(x[1], c1, x[0] * c0)));  // component kk of L~ = L D^-1
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
    DEV T diag_solve(T y) { return ipm::diag_solve4(sinvb, y, lane, lane, n); }  // S^-1 y, every 4x4 block at once

    // ---- KKT hooks of the shared solve ---------------------------------------------------------------------------------------------------
    // condense + KKT assembly + factorisation (needs stage_form_weights(w) done); max |sc * H_jj|, the scale of the delta_w shift, on request
    DEV bool kkt_factor(T sc, T reg, bool want_hmax)
    {
        acc_t kt[NTTF];  // condense accumulates into the first NTT tiles; build_tiles turns them into K in place
        acc_t (&acc)[NTT] = reinterpret_cast<acc_t (&)[NTT]>(kt);
        condense(sc, acc);
        if (ADJ) {
            if (want_hmax) {
                T dj = lane < n ? Lc[offc_rt(lane) + lane] : (T)0;
                if constexpr (MSPLIT > 0) dj += lane < 2 * MSPLIT ? hm[lane] : (T)0;   // the diagonal's share of G_M^T P_M
                cs[C_HMAX] = dpp_max_nn(fabs(dj));
            }
        } else if (want_hmax) {  // max |sc * H_jj| over the diagonal of the tiles
            T hm = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (Real<T>::row_of(lane, r) == (lane & 15)) hm = fmax(hm, fabs(sc * acc[ti * (ti + 1) / 2 + ti][r]));
            cs[C_HMAX] = dpp_max_nn(hm);
        }
        STAMP(3);
        build_tiles(sc, reg, kt);
        STAMP(4);
        return factor(kt);
    }
    // S^-1 L~^-1 (-sc*g) sits in row n of the factor image; re-read where needed rather than held in registers
    DEV T kkt_affine1() { return back_subst(lane < n ? Lc[offc_rt(lane) + n] : (T)0); }                                        // K^-1 (-sc g)
    DEV T kkt_direction1(T b) { return back_subst((lane < n ? Lc[offc_rt(lane) + n] : (T)0) + diag_solve(fwd_subst(b))); }    // K^-1 (-sc g + b)

    DEV void solve(const KIO<T> &io, int b) { ipm::solve(*this, io, b); }
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
    ipm::run_solver<FastSolver<T, N>>(P, io, smem);
}
template <typename T, int N>
__global__ __launch_bounds__(64, sizeof(T) == 8 ? (N >= 28 ? 1 : 2) : (N <= 20 ? 4 : 3)) void kmpc_solve_fast_kernel(KP P, KIO<T> io)
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
    StageF<T> St;
    const T g = ipm::debug_linearize_at(sv, P, io, b, St);
    const T sc = (T)io.sc, reg = (T)io.reg;
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
// SIMD anyway, so the bound says so and the allocator may use all 512 registers; fp64 at N = 8 ran at three waves per SIMD with 42 spilled registers
// until the degenerate-pair rule moved one of the spill stores into a divergent region -- tools/spill_exec_check.py -- two waves, no scratch)
template <typename T, int N>
__global__ __launch_bounds__(64, sizeof(T) == 8 ? (N >= 28 ? 1 : 2) : (N <= 20 ? 4 : (N >= 28 ? 2 : 3))) void kmpc_solve_fast_frenet_kernel(KP P, KIO<T> io)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[FastSolver<T, N, 1>::lds_elems() * sizeof(T)];
    ipm::run_solver<FastSolver<T, N, 1>>(P, io, smem);
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
