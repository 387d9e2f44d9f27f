// kmpc_quad.hip -- FOUR PROBLEMS PER WAVEFRONT at the reference's own horizon N = 8 (MKZMPCPathFollower.jl:34, mpc_cmd_pub.jl:51), gfx950 only.
//
// At N = 8 the condensed problem has n = 16 inputs, 38 linear forms and 9 stages: the one-wave-per-problem kernel (kmpc_fast.hip) keeps 17 of
// its 64 lanes busy.  Here a 16-lane DPP ROW is one problem: lane r of a row holds input r, forms r, r + 16, r + 32 and stage r.
//   * The model code and the interior-point state machine are the shared ones (kmpc_ipm.h) with NTH = 16 "threads per problem": the stage
//     scans (row_shr / row_shl steps) never leave a row, every LDS buffer is per row, and what the one-wave kernels keep as wave-uniform
//     scalars (mode, mu, alpha, iteration counters, ...) are per-row values here -- the rows of a wave branch independently (a row whose
//     line search back-tracks or whose factorisation is repeated runs that path under its own exec mask; a finished row idles until the
//     wave's last row is done).  The start order (kmpc_schedule.hip) puts problems of similar predicted difficulty into one wave.
//   * Reductions are row-local butterflies on DPP row rotations (row_ror 8, 4, 2, 1): every lane of the row ends with bit-identical
//     results (each level adds the same two partial sums in either order), which is what keeps a row's 16 lanes on one control path.
//   * KKT: the 16 x 16 matrix is one row (i) per lane, in registers; Cholesky is 16 right-looking column steps whose only exchange is one
//     per-row LDS all-gather of the pivot column; the right-hand side -sc*g rides along (forward substitution inside the factorisation).
//     The substitutions are FOUR dependent block steps each: every quad of lanes inverts its 4x4 diagonal block of L once per factorisation
//     (quad-permute DPP), a block step is one quad-local product, a row-local broadcast of the four finished components (ds_swizzle) and four
//     FMAs with entries of the lane's row / column read from the packed image; at this size the matrix cores have nothing to offer -- a
//     trailing update is n^3/3 = 1.4 k flops per problem.
// Results agree with the one-wave kernel to rounding (different summation trees), not bit for bit: tests/test_quad.py.
#include "kmpc_ipm.h"

template <typename T> DEV T row_sum(T x) {   // sum over the 16 lanes of a DPP row, identical bits in every lane
    x += dpp_mov0<0x128, 0xf>(x); x += dpp_mov0<0x124, 0xf>(x); x += dpp_mov0<0x122, 0xf>(x); x += dpp_mov0<0x121, 0xf>(x);
    return x;
}
template <typename T> DEV T row_max(T x) {   // maximum over the row (any sign; NaN operands dropped as by fmax)
    x = max_raw(x, dpp_mov0<0x128, 0xf>(x)); x = max_raw(x, dpp_mov0<0x124, 0xf>(x));
    x = max_raw(x, dpp_mov0<0x122, 0xf>(x)); x = max_raw(x, dpp_mov0<0x121, 0xf>(x));
    return x;
}

// value of lane J of the own 16-lane row, through the LDS crossbar only (ds_swizzle, bit-mask mode: lane' = (lane & 0x10) | J inside each
// group of 32): no LDS memory, no address register, no write -> fence -> read round trip
template <int J> DEV float row_bcast(float x) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), (J << 5) | 0x10)); }
template <int J> DEV double row_bcast(double x)
{
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), (J << 5) | 0x10), hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), (J << 5) | 0x10);
    return __hiloint2double(hi, lo);
}

template <int R, typename T> DEV T quad_bcast(T x) { return dpp_mov0<R | (R << 2) | (R << 4) | (R << 6), 0xf>(x); }   // value of lane R of the own quad

template <typename T> struct QuadSolver {
    static constexpr int N = 8;
    KMPC_HORIZON_CONSTANTS(8)
    typedef T real;
    // what kmpc_ipm.h reads: horizon, threads per problem, forms per thread, stride of the G_N table, record stride, functor
    static constexpr int N_ = 8, NTH = 16, NF = 3, GS = 16, LSTR = 16, MODEL_ID = 0, MSPLIT = 0 /* every lane of a row carries a column: no spare lanes to split the recursion over */;
    static_assert(n == 16 && nf <= 48, "one DPP row per problem");
    // per-row LDS map (elements of T)
    static constexpr int O_LC = 0, O_XB = (LC + 1) & ~1, O_WB = O_XB + 16, O_CB = O_WB + 48, O_LIN = O_CB + 16, O_GNB = O_LIN + LSTR * (N + 1),
                         O_GB = O_GNB + 3 * GS, O_CS = O_GB + 16, O_UB = O_CS + 16, O_CU = O_UB + 16, O_CL = O_CU + 48, O_EX = O_CL + 48,
                         ROW = O_EX + 32;
    static constexpr int O_PT = 4 * ROW, O_KC = O_PT + 32;
    static constexpr int lds_elems() { return O_KC + (sizeof(T) == 8 ? KC_COUNT : 0); }

    STAMP_MEMBERS
    const KP &P;
    int lane, vid, row;   // lane = vid = position in the row (stage / input / first form slot), row = which of the wave's four problems
    T *Lc, *xb, *wb, *cb, *lin, *gnb, *gb, *cs, *ubest, *cub, *clb, *ex;
    const T *pt, *cwt;
    Coef<T> kc;
    T psi0, v0, vt, up0, up1, rx, ry, rp, xoff_, yoff_;
    T dinv[4]; // row (lane & 3) of D_q^-1, D_q = the 4x4 diagonal block of L this lane's row runs through (q = lane >> 2)
    T dcol[4]; // column (lane & 3) of D_q^-1
    T yv;      // (L^-1 (-sc g))[lane]

    DEV QuadSolver(const KP &p, unsigned char *smem) : P(p), lane(threadIdx.x & 15), vid(threadIdx.x & 15), row(threadIdx.x >> 4)
    {
        T *base = reinterpret_cast<T *>(smem);
        T *rb = base + ROW * row;
        Lc = rb + O_LC; xb = rb + O_XB; wb = rb + O_WB; cb = rb + O_CB; lin = rb + O_LIN; gnb = rb + O_GNB; gb = rb + O_GB; cs = rb + O_CS;
        ubest = rb + O_UB; cub = rb + O_CU; clb = rb + O_CL; ex = rb + O_EX;
        kc.tab = base + O_KC;
        if (sizeof(T) == 8 && threadIdx.x < KC_COUNT) const_cast<T *>(kc.tab)[threadIdx.x] = (T)kmpc_coef[threadIdx.x];
        pt = base + O_PT; cwt = pt + PT_W;
        if (threadIdx.x == 0) ipm::fill_param_table(base + O_PT, p, nf);
        WSYNC();
    }

    DEV void load_problem(const T *z0, const T *ref, const T *vtp, const T *upp, int b)   // b: this row's problem
    {
        // vehicle-centred coordinates (the NLP is translation-invariant; see kmpc_fast.hip)
        xoff_ = z0[4 * (size_t)b]; yoff_ = z0[4 * (size_t)b + 1];
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
    DEV T &cu(int i) { return cub[lane + 16 * i]; }
    DEV T &cl(int i) { return clb[lane + 16 * i]; }
    DEV void save_best1(T U) { ubest[lane] = U; }
    DEV T load_best1() const { return ubest[lane]; }
    KMPC_IPM_ONE_SLOT_HOOKS
    template <int NS, int NM> DEV void reduce(T (&sm)[NS < 1 ? 1 : NS], T (&mx)[NM < 1 ? 1 : NM])
    {
#pragma unroll
        for (int i = 0; i < NS; ++i) sm[i] = row_sum(sm[i]);
#pragma unroll
        for (int i = 0; i < NM; ++i) mx[i] = row_max(mx[i]);
    }
    template <int NS, int NM> DEV void reduce_flag(T (&sm)[NS < 1 ? 1 : NS], T (&mx)[NM < 1 ? 1 : NM], bool &all_true)
    {
        all_true = row_max(all_true ? (T)0 : (T)1) == (T)0;
        reduce<NS, NM>(sm, mx);
    }
    DEV T max_any(T x) const { return row_max(x); }
    DEV T sum_stages(T x) const { return row_sum(x); }
    DEV T stage_bcast(T x, int k)   // value of the row's lane k (start-up only: through the row's exchange buffer)
    {
        ex[lane] = x;
        WFENCE();
        const T v = ex[k];
        WFENCE();
        return v;
    }
    DEV T eval1(T U, StageV<T> &S) { return ipm::eval_cartesian(*this, U, S); }
    DEV T linearize1(const StageV<T> &S, bool exact) { return ipm::linearize_cartesian(*this, S, exact); }
    DEV void drop_second_order() { ipm::drop_second_order_cartesian(*this); }

    // ---- KKT: K = sc*(H + input Hessian) + A^T W A + reg*I, one row per lane, Cholesky + the affine right-hand side in one sweep ------------
    // (needs stage_form_weights(w) done: wb = form weights, cb = suffix sums of the speed weights)
    DEV bool kkt_factor(T sc, T reg, bool want_hmax)
    {
        ipm::condense_adjoint(*this, sc);   // column j (rows >= j) of sc*H into the row's packed image
        STAMP(3);
        if (want_hmax) cs[C_HMAX] = row_max(fabs(Lc[offc_rt(lane) + lane]));   // max |sc * H_jj|: scale of the delta_w shift
        T *dgs = cub, *sbs = clb;   // the corrector buffers are dead between the accepted step and the end of the factorisation
        ipm::kkt_diag_staging(*this, sc, reg, true, dgs, sbs);
        const int i = lane;
        T a[16];   // row i of the KKT matrix, then of its Cholesky factor L (registers only while the factorisation runs)
        const T dt2 = pt[PT_DT2];
        const T spd = (i & 1) ? (T)0 : dt2 * cb[i >> 1];   // speed rows: dt^2 * S[max(i,k)/2] on the (even, even) entries; max = i in the lower triangle
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            // (branch-free: a lane above the diagonal reads the column's diagonal entry and discards it)
            const bool low = k <= i;
            T v = Lc[offc(k) + (low ? i : k)];
            if (!(k & 1)) v += spd;
            v += k == i ? dgs[k] : (T)0;
            if (k < 14) v += k == i - 2 ? sbs[k] : (T)0;
            a[k] = low ? v : (T)0;
        }
        T b = -sc * gb[i];
        WFENCE();   // image, dgs / sbs consumed: the exchange buffer and (later) the image may be overwritten
        STAMP(4);
        bool ok = true;
        T rd = (T)0;   // 1 / L[i][i]
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            ex[i] = a[j];        // column j of the current Schur complement (rows < j: stale values nobody reads)
            ex[16 + i] = b;
            WFENCE();
            const T d = ex[j];
            ok = ok && d > Real<T>::tiny() && d < (T)1e300;
            const T rinv = rsqrt_(d);
            const T lij = a[j] * rinv;          // L[i][j] (i >= j; L[j][j] = d * rinv = sqrt(d))
            const T yj = ex[16 + j] * rinv;     // y_j = b_j / L[j][j]
            const T t = lij * rinv;             // a[k] -= L[i][j] L[k][j] = t * colbuf[k]
#pragma unroll
            for (int k = j + 1; k < 16; ++k) a[k] = fma(-t, ex[k], a[k]);   // (entries k > i are never used)
            b = i > j ? fma(-lij, yj, b) : (i == j ? yj : b);   // rows above j hold finished components of y
            a[j] = lij;
            if (i == j) rd = rinv;
            WFENCE();
        }
        yv = b;
        STAMP(12);
        // L goes to the row's packed image (sc*H there is dead; a refactorisation re-condenses): the substitutions read the lane's row
        // (forward) and column (backward) of it from there instead of holding 32 more registers across the whole iteration
#pragma unroll
        for (int k = 0; k < 16; ++k) Lc[k <= i ? offc(k) + i : O_EX - O_LC + i] = a[k];   // (entries above the diagonal go to a scratch slot of the exchange buffer)
        // Inverses of the four 4x4 diagonal blocks of L, one per quad of lanes, for substitutions in FOUR dependent block steps instead of
        // sixteen: lane (q, r) gathers its quad's block through quad-permute DPP, inverts it redundantly and keeps row r and column r
        {
            const int q = i >> 2, r = i & 3;
            T blk[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const T v = q == 0 ? a[c] : (q == 1 ? a[4 + c] : (q == 2 ? a[8 + c] : a[12 + c]));   // L[i][4q + c]
                blk[c] = c < r ? v : (T)0;
            }
            const T r0 = quad_bcast<0>(rd), r1 = quad_bcast<1>(rd), r2 = quad_bcast<2>(rd), r3 = quad_bcast<3>(rd);
            const T D10 = quad_bcast<1>(blk[0]), D20 = quad_bcast<2>(blk[0]), D21 = quad_bcast<2>(blk[1]);
            const T D30 = quad_bcast<3>(blk[0]), D31 = quad_bcast<3>(blk[1]), D32 = quad_bcast<3>(blk[2]);
            const T i10 = -D10 * r0 * r1;
            const T i21 = -D21 * r1 * r2, i20 = -fma(D21, i10, D20 * r0) * r2;
            const T i32 = -D32 * r2 * r3, i31 = -fma(D32, i21, D31 * r1) * r3, i30 = -fma(D32, i20, fma(D31, i10, D30 * r0)) * r3;
            // inverse (lower triangular): rows (r0, 0, 0, 0), (i10, r1, 0, 0), (i20, i21, r2, 0), (i30, i31, i32, r3)
            dinv[0] = r == 0 ? r0 : (r == 1 ? i10 : (r == 2 ? i20 : i30));
            dinv[1] = r == 0 ? (T)0 : (r == 1 ? r1 : (r == 2 ? i21 : i31));
            dinv[2] = r <= 1 ? (T)0 : (r == 2 ? r2 : i32);
            dinv[3] = r <= 2 ? (T)0 : r3;
            dcol[0] = r == 0 ? r0 : (T)0;
            dcol[1] = r == 0 ? i10 : (r == 1 ? r1 : (T)0);
            dcol[2] = r == 0 ? i20 : (r == 1 ? i21 : (r == 2 ? r2 : (T)0));
            dcol[3] = r == 0 ? i30 : (r == 1 ? i31 : (r == 2 ? i32 : r3));
        }
        WFENCE();
        return ok;   // (not positive definite: NaNs may have been produced above; nothing is used in that case)
    }
    // block step Q of L^T x = z (Q = 3 .. 0): the quad solves its 4x4 block (all quads do, on whatever they hold; quad Q's is final), its four
    // components are broadcast inside the row, and the lanes of the earlier quads subtract them with the four entries of their column
    template <int Q> DEV void back_block(T &z) const
    {
        const T z0 = quad_bcast<0>(z), z1 = quad_bcast<1>(z), z2 = quad_bcast<2>(z), z3 = quad_bcast<3>(z);
        const T x = fma(dcol[3], z3, dcol[2] * z2) + fma(dcol[1], z1, dcol[0] * z0);     // (D^-T z)_r = sum_c inv[c][r] z_c
        if (Q > 0) {
            const T x0 = row_bcast<4 * Q>(x), x1 = row_bcast<4 * Q + 1>(x), x2 = row_bcast<4 * Q + 2>(x), x3 = row_bcast<4 * Q + 3>(x);
            const bool up = lane < 4 * Q;
            const T *col = Lc + offc_rt(lane) + (up ? 4 * Q : lane);   // L[4Q + c][lane], c = 0..3 (lanes >= 4Q: their own diagonal, discarded)
            const T l0 = col[0], l1 = up ? col[1] : (T)0, l2 = up ? col[2] : (T)0, l3 = up ? col[3] : (T)0;
            const T upd = fma(l3, x3, l2 * x2) + fma(l1, x1, (up ? l0 : (T)0) * x0);
            z = (lane >> 2) == Q ? x : (up ? z - upd : z);
        } else z = (lane >> 2) == 0 ? x : z;
        __builtin_amdgcn_sched_barrier(0);
    }
    DEV T back_subst(T z) const { back_block<3>(z); back_block<2>(z); back_block<1>(z); back_block<0>(z); return z; }   // L^T x = z
    template <int Q> DEV void fwd_block(T &b) const
    {
        const T b0 = quad_bcast<0>(b), b1 = quad_bcast<1>(b), b2 = quad_bcast<2>(b), b3 = quad_bcast<3>(b);
        const T w = fma(dinv[3], b3, dinv[2] * b2) + fma(dinv[1], b1, dinv[0] * b0);     // (D^-1 b)_r
        if (Q < 3) {
            const T w0 = row_bcast<4 * Q>(w), w1 = row_bcast<4 * Q + 1>(w), w2 = row_bcast<4 * Q + 2>(w), w3 = row_bcast<4 * Q + 3>(w);
            const bool dn = lane >= 4 * Q + 4;
            // L[lane][4Q + c]: row `lane` of L (lanes above the block read the column's diagonal entry and discard it)
            const T l0 = Lc[offc(4 * Q) + (dn ? lane : 4 * Q)], l1 = Lc[offc(4 * Q + 1) + (dn ? lane : 4 * Q + 1)];
            const T l2 = Lc[offc(4 * Q + 2) + (dn ? lane : 4 * Q + 2)], l3 = Lc[offc(4 * Q + 3) + (dn ? lane : 4 * Q + 3)];
            const T upd = fma(l3, w3, l2 * w2) + fma(l1, w1, l0 * w0);
            b = (lane >> 2) == Q ? w : (dn ? b - upd : b);
        } else b = (lane >> 2) == 3 ? w : b;
        __builtin_amdgcn_sched_barrier(0);
    }
    DEV T fwd_subst(T b) const { fwd_block<0>(b); fwd_block<1>(b); fwd_block<2>(b); fwd_block<3>(b); return b; }   // L w = b
    DEV T kkt_affine1() { return back_subst(yv); }                           // K^-1 (-sc g)
    DEV T kkt_direction1(T b) { return back_subst(yv + fwd_subst(b)); }      // K^-1 (-sc g + b)

    DEV void solve(const KIO<T> &io, int b) { ipm::solve(*this, io, b); }
};

template <typename T>
__global__ __launch_bounds__(64, sizeof(T) == 8 ? 2 : 4) void kmpc_solve_quad_kernel(KP P, KIO<T> io)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[QuadSolver<T>::lds_elems() * sizeof(T)];
    const int slot = 4 * (int)blockIdx.x + ((int)threadIdx.x >> 4);
    if (4 * (int)blockIdx.x >= P.B) return;
    // a wave's four rows take four consecutive entries of the start order (similar predicted difficulty); the rows past the end of a batch
    // whose size is not a multiple of four repeat its last problem (same inputs, same stores)
    const int sl = slot < P.B ? slot : P.B - 1;
    const int b = io.perm ? io.perm[sl] : sl;
#ifdef KMPC_POISON
    for (int e = threadIdx.x; e < QuadSolver<T>::lds_elems(); e += 64) reinterpret_cast<T *>(smem)[e] = (T)NAN;
    __syncthreads();
#endif
    QuadSolver<T> sv(P, smem);
    ipm::load_problem_io(sv, io, b);
    sv.solve(io, b);
}

template <typename T> bool kmpc_quad_available(int N) { return N == 8; }
template <typename T> hipError_t kmpc_launch_solve_quad(const KP &P, const KIO<T> &io, hipStream_t st)
{
    if (P.N != 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL((kmpc_solve_quad_kernel<T>), dim3((P.B + 3) / 4), dim3(64), 0, st, P, io);
    return hipGetLastError();
}
template bool kmpc_quad_available<double>(int);
template bool kmpc_quad_available<float>(int);
template hipError_t kmpc_launch_solve_quad<double>(const KP &, const KIO<double> &, hipStream_t);
template hipError_t kmpc_launch_solve_quad<float>(const KP &, const KIO<float> &, hipStream_t);
