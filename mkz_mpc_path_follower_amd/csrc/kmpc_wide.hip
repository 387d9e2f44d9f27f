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
//   * workgroup barriers only where data crosses waves: one per block-step of the factorisation, one per reduction; the three triangular
//     substitutions run in wave 0 with two slots per lane (v_readlane broadcasts, no barrier inside).
#include "kmpc_ipm.h"

#define WGSYNC() __syncthreads()

template <typename T> using StageW = StageV<T>;
constexpr int WLIN = 16;  // stage record stride: A02 A03 A12 A13 A23 Bdx Bdy Bdp mpp mpv mpd mvd mdd

template <typename T, int N> struct WideSolver {
    KMPC_HORIZON_CONSTANTS(N)
    static constexpr int NTF = (n + 1 + 15) / 16;   // tile rows of K with the rhs row n
    static constexpr int NP = 16 * NTF;              // padded dimension
    static constexpr int NB = n / 4;                 // 4-column block-steps
    static_assert(NTF >= 5 && NTF <= 7, "tile rows are dealt to the four waves as (6), (5,0), (4,1), (3,2) -- (5), (4,0), (3,1), (2) with six, (4), (3,0), (2), (1) with five");
    static_assert(nf <= 256 && n % 4 == 0 && N + 1 <= 64, "one thread per form, 4-column panels, one lane per stage");
    typedef typename Real<T>::acc_t acc_t;
    typedef T real;
    // what kmpc_ipm.h reads: horizon, threads per problem, forms per thread (one), stride of the G_N table, record stride, functor
    static constexpr int N_ = N, NTH = 256, NF = 1, GS = 128, LSTR = WLIN, MODEL_ID = 0;
    // split adjoint recursion (kmpc_ipm.h, condense_adjoint): threads 128 + j (waves 2 and 3, idle in the unsplit recursion) run the stages below
    // M = N / 2 for the columns j < 2M while threads j < n run the stages from M up: N / 2 trips instead of N
    static constexpr int MSPLIT = N / 2, LOW0 = 128, GMS = 64, TM = (2 * MSPLIT + 15) / 16;
    static_assert(2 * MSPLIT <= GMS && LOW0 + 2 * MSPLIT <= 256 && N % 2 == 0, "one thread per lower column, G_M rows of 64");
    // LDS map (elements of T)
    static constexpr int O_LC = 0, O_OPB = (LC + 1) & ~1, O_LIN = O_OPB + 2 * 4 * NP + 256 + 3 * 128 + 3 * GMS + 64, O_XB = O_LIN + WLIN * (N + 1), O_WB = O_XB + 128,
                         O_CBW = O_WB + 256, O_RED = O_CBW + 4 * 64, O_SINV = O_RED + 2 * 32, O_X2 = O_SINV + 16 * NB, O_X3 = O_X2 + 128,
                         O_GB = O_X3 + 128, O_UB = O_GB + 128, O_CS = O_UB + 128, O_PT = O_CS + 4 * 16, O_KC = O_PT + 32, O_END = O_KC + (sizeof(T) == 8 ? KC_COUNT : 0);
    static constexpr int lds_elems() { return O_END; }

    STAMP_MEMBERS
    const KP &P;
    int tid, lane, wv, vid;   // vid = tid: the input / form slot of this thread (kmpc_ipm.h)
    T *Lc, *opb, *pan, *dgs, *sbs, *gnb, *gmb, *hm, *lin, *xb, *wb, *cb, *red, *sinvb, *x2, *x3, *gbl, *gb, *ubl, *cs;
    T cu_[1], cl_[1];         // corrector terms of this thread's form
    Coef<T> kc;
    const T *pt, *cwt;
    int rpar;
    T psi0, v0, vt, rx, ry, rp;   // (x0 = y0 = 0 in vehicle-centred coordinates; u_prev and the offsets are re-read where they are used)
    const T *z0p, *upp_;
    DEV WideSolver(const KP &p, unsigned char *smem) : P(p), tid(threadIdx.x), lane(threadIdx.x & 63), vid(threadIdx.x), rpar(0)
    {
        wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        T *base = reinterpret_cast<T *>(smem);
        Lc = base + O_LC; opb = base + O_OPB; lin = base + O_LIN; xb = base + O_XB; wb = base + O_WB;
        cb = base + O_CBW + 64 * wv;  // every wave keeps its OWN copy of the stage prefix / suffix sums: no barrier to share them
        red = base + O_RED; sinvb = base + O_SINV; x2 = base + O_X2; x3 = base + O_X3; gbl = base + O_GB; gb = gbl; ubl = base + O_UB;
        cs = base + O_CS + 16 * wv;    // wave-uniform scalars that are read once or twice per iteration: every wave parks its own copy
        pan = opb;                      // the double-buffered Cholesky panel (2 x 4 NP)
        dgs = opb + 2 * 4 * NP; sbs = dgs + 128;  // 2 x n staging of build_tiles
        gnb = sbs + 128;                // terminal sensitivities G_N [3][128], written by linearize, read by every condense of that linearisation
        gmb = gnb + 3 * 128;            // mid-horizon sensitivities G_M [3][64] of the split recursion, likewise
        hm = gmb + 3 * GMS;             // what the diagonal entries j < 2M of the image lack (read before the tiles exist: max |sc H_jj|)
        for (int e = tid; e < 16 * NB; e += 256) sinvb[e] = (T)0;
        kc.tab = base + O_KC;
        if (sizeof(T) == 8 && tid < KC_COUNT) const_cast<T *>(kc.tab)[tid] = (T)kmpc_coef[tid];
        pt = base + O_PT; cwt = pt + PT_W;
        if (tid == 0) ipm::fill_param_table(base + O_PT, p, nf);
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

    // ---- hooks of the shared interior-point code (kmpc_ipm.h) ------------------------------------------------------------------------
    DEV T up(int j) const { return upp_[j]; }
    DEV T xoff() const { return z0p[0]; }
    DEV T yoff() const { return z0p[1]; }
    DEV bool rec_writer() const { return wv == 0; }   // every wave holds the same stage data; wave 0 publishes the records
    DEV void refresh_ids() { asm volatile("" : "+v"(tid)); lane = tid & 63; vid = tid; }
    DEV T *pm(int c) const { return x2 + 64 * c; }     // component c of p_j(M), j < 2M: the substitution buffers x2 | x3 are dead while K is built
    DEV T &cu(int) { return cu_[0]; }                  // corrector terms live in registers (one form per thread)
    DEV T &cl(int) { return cl_[0]; }
    DEV void save_best1(T U) { if (tid < 128) ubl[tid] = U; }
    DEV T load_best1() const { return ubl[tid & 127]; }  // (threads >= 128 pick up copies: their U is never read as an input -- every use is guarded by tid < n)
    template <int NS, int NM> DEV void reduce(T (&sm)[NS < 1 ? 1 : NS], T (&mx)[NM < 1 ? 1 : NM]) { wg_reduce<NS, NM>(sm, mx); }
    template <int NS, int NM> DEV void reduce_flag(T (&sm)[NS < 1 ? 1 : NS], T (&mx)[NM < 1 ? 1 : NM], bool &all_true)
    {
        T m2[NM + 1];
        m2[0] = all_true ? (T)0 : (T)1;
#pragma unroll
        for (int i = 0; i < NM; ++i) m2[i + 1] = mx[i];
        wg_reduce<NS, NM + 1>(sm, m2);
        all_true = m2[0] == (T)0;
#pragma unroll
        for (int i = 0; i < NM; ++i) mx[i] = m2[i + 1];
    }
    DEV T sum_stages(T x) const { return dpp_sum(x); }
    DEV T stage_bcast(T x, int k) const { return readlane_(x, k); }
    DEV T max_any(T x) { T dm[1] = {(T)0}, m[1] = {x}; wg_reduce<0, 1>(dm, m); return m[0]; }
    DEV void stage_form_weights(T w) { const T w1[1] = {w}; ipm::stage_form_weights(*this, w1); }   // (scalar form for the diagnostics kernel)
    DEV T eval1(T U, StageW<T> &S) { return ipm::eval_cartesian(*this, U, S); }
    DEV T linearize1(const StageW<T> &S, bool exact) { return ipm::linearize_cartesian(*this, S, exact); }
    KMPC_IPM_ONE_SLOT_HOOKS
    DEV void drop_second_order() { ipm::drop_second_order_cartesian(*this); }
    DEV void condense_adjoint(T sc) { ipm::condense_adjoint(*this, sc); }

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
        // split recursion: rows and columns < 2M of the image lack G_M^T P_M -- a rank-4 product, one matrix-core instruction per tile
        T fa = (T)0;
        if (ti < TM) fa = ipm::split_fragment_a(*this, ti, c, lane >> 4);
#pragma unroll
        for (int tj = 0; tj < NTL; ++tj) {
            if (tj <= ti) {
                acc_t cm = acc_t{0, 0, 0, 0};
                if (ti < TM) cm = Real<T>::mfma(fa, ipm::split_fragment_b(*this, tj, c, lane >> 4), cm);
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
                        v = fma(evm, cb[row >> 1], colK[row]) + cm[r];
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
        ipm::kkt_diag_staging(*this, sc, reg, true, dgs, sbs);
        build_row<Rows<W>::N0>(sc, R0, k0);
        if (R1 >= 0) build_row<Rows<W>::N1>(sc, R1, k1);
        WGSYNC();  // dgs / sbs / the odd rows of the image have been consumed: the panel and the factor may overwrite them
    }

    // ---- blocked Cholesky on the matrix cores (block-LDL^T form, see kmpc_fast.hip), ONE barrier per 4-column block-step ---------
    // The panel of block-step jb (columns 4 jb .. 4 jb + 3 of the current Schur complement) goes to LDS from the tile column TC that holds it.
    // TC is a TEMPLATE parameter: the factorisation is unrolled over the tile columns (four block-steps each in a rolled loop), so tile indices,
    // liveness tests and the panel's source registers are all static.  (Until round 3 the tile column was a run-time value and the panel's
    // tile was picked by value selects -- 48 v_cndmask per block-step in wave 0; a branch per tile column had ended in scratch-resident tiles.)
    template <int W, int TC> DEV void extract_panel(int jb, const acc_t (&k0)[Rows<W>::N0], const acc_t (&k1)[Rows<W>::N1])
    {
        constexpr int R0 = Rows<W>::R0, R1 = Rows<W>::R1;
        const int c = lane & 15, j0 = 4 * jb, kp = c - (j0 & 15);
        T *pn = pan + (jb & 1) * 4 * NP;
        if (kp >= 0 && kp < 4) {
            if constexpr (TC <= R0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pn[4 * (16 * R0 + Real<T>::row_of(lane, r)) + kp] = k0[TC][r];
            }
            if constexpr (R1 >= 0 && TC <= R1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pn[4 * (16 * R1 + Real<T>::row_of(lane, r)) + kp] = k1[TC][r];
            }
        }
    }
    // One block-step: every wave factors the 4x4 diagonal block (same data), solves the panel rows of ITS OWN tile rows against it (x L_dd^T = a), forms
    // their L~ = x D^-1 = a A_jj^-1 entries, stores them in the factor image and runs its trailing updates in the form  K -= L~ A_.j^T  -- A fragment: its
    // own L~ rows, B fragment: the RAW panel rows of the tile column, which every wave reads from the panel buffer.  Nothing a wave computes in a
    // block-step is needed by another wave before the next panel: ONE workgroup barrier per block-step.  (Until round 4 the update was L L^T with
    // L = A_.j L_dd^-T on both sides: the solved rows of the other waves' tile rows had to be published first -- a second barrier per block-step; 48 % of
    // the wave lifetime was barrier / s_waitcnt wait.  L~ comes out of two triangular solves as before, so the update is as accurate as the factor image.)
    DEV T raw_fragment(const T *pn, int t, int c, int kk, int jc) const   // component kk of raw panel row 16 t + c, negated, zero on dead rows
    {
        const int row = 16 * t + c;
        const T v = pn[4 * row + kk];
        return (row >= jc && row <= n) ? -v : (T)0;
    }
    template <int W, int TC> DEV bool factor_column(acc_t (&k0)[Rows<W>::N0], acc_t (&k1)[Rows<W>::N1])
    {
        constexpr int R0 = Rows<W>::R0, R1 = Rows<W>::R1;
        constexpr int STEPS = NB - 4 * TC < 4 ? NB - 4 * TC : 4;   // block-steps inside this tile column (the last one may be partial)
        const int c = lane & 15, kk = lane >> 4;
#pragma nounroll
        for (int s4 = 0; s4 < STEPS; ++s4) {
            const int jb = 4 * TC + s4, j0 = 4 * jb;
            const T *pn = pan + (jb & 1) * 4 * NP;
            const ipm::Diag4<T> dd = ipm::load_diag4(pn + 4 * j0);
            // the panel rows of this wave's tile rows and the raw fragments of the tile columns it updates do not depend on the diagonal factor: request them now
            T av[2][4];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int t = o == 0 ? R0 : R1;
                const T *ar = pn + 4 * (16 * (t >= TC ? t : TC) + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) av[o][k] = ar[k];
            }
            const int jc = j0 + kk;
            T bf[NTF];   // raw fragments of tile rows TC .. R0 (R1 < R0: a prefix of them serves the second tile row)
#pragma unroll
            for (int t = 0; t < NTF; ++t) bf[t] = (t >= TC && t <= R0) ? raw_fragment(pn, t, c, kk, jc) : (T)0;
            const ipm::Chol4<T> c4 = ipm::factor_diag4(dd);
            if (!c4.ok) return false;  // not positive definite (same data in every wave)
            if (tid == 0) c4.store_inv(sinvb + 16 * jb);
            // column kk of D_j^-1 for this lane's component of L~
            const T c0 = kk == 0 ? c4.r0 : (T)0;
            const T c1 = kk == 0 ? c4.i10 : (kk == 1 ? c4.r1 : (T)0);
            const T c2 = kk == 0 ? c4.i20 : (kk == 1 ? c4.i21 : (kk == 2 ? c4.r2 : (T)0));
            const T c3 = kk == 0 ? c4.i30 : (kk == 1 ? c4.i31 : (kk == 2 ? c4.i32 : c4.r3));
            T *colL = Lc + offc_rt(jc < n ? jc : 0);
            T own[2] = {(T)0, (T)0};
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int t = o == 0 ? R0 : R1;
                if (t < TC) continue;                         // no second row (R1 = -1) / finished tile row: static after unrolling
                const int row = 16 * t + c;
                T x[4];
                c4.solve_row(av[o], x);
                const T lt = fma(x[3], c3, fma(x[2], c2, fma(x[1], c1, x[0] * c0)));   // component kk of L~ = L D^-1
                const bool live = row >= jc && row <= n;
                own[o] = live ? lt : (T)0;                    // A fragment of the trailing update
                if (live) colL[row] = lt;
            }
            if (j0 + 4 < n) {
                const bool same = s4 < 3;   // the next panel still lies in tile column TC (and that column still has live entries)
                if constexpr (R0 >= TC) {
                    if (same) k0[TC] = Real<T>::mfma(own[0], bf[TC], k0[TC]);
#pragma unroll
                    for (int t = TC + 1; t <= R0; ++t) k0[t] = Real<T>::mfma(own[0], bf[t], k0[t]);
                }
                if constexpr (R1 >= TC) {
                    if (same) k1[TC] = Real<T>::mfma(own[1], bf[TC], k1[TC]);
#pragma unroll
                    for (int t = TC + 1; t <= R1; ++t) k1[t] = Real<T>::mfma(own[1], bf[t], k1[t]);
                }
                if (same) extract_panel<W, TC>(jb + 1, k0, k1);
                else extract_panel<W, (TC + 1 < NTF ? TC + 1 : TC)>(jb + 1, k0, k1);
            }
            WGSYNC();   // the next panel is complete (and this one has been read)
        }
        return true;
    }
    template <int W, int TC> DEV bool factor_from(acc_t (&k0)[Rows<W>::N0], acc_t (&k1)[Rows<W>::N1])
    {
        if constexpr (4 * TC < NB) {
            if (!factor_column<W, TC>(k0, k1)) return false;
            return factor_from<W, TC + 1>(k0, k1);
        } else return true;
    }
    template <int W> DEV bool factor(acc_t (&k0)[Rows<W>::N0], acc_t (&k1)[Rows<W>::N1])
    {
        extract_panel<W, 0>(0, k0, k1);
        WGSYNC();
        return factor_from<W, 0>(k0, k1);
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
    // (the coefficients of the NEXT block are requested before this block's v_readlane -> FMA chain runs: a block step is then that chain, not an LDS round trip)
    DEV void fwd_subst2(T &w0, T &w1)  // L~ y = b
    {
        const bool v1 = 64 + lane < n;
        const int r1 = v1 ? 64 + lane : n;
        T l0[4], l1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { l0[k] = Lc[offc_rt(k) + lane]; l1[k] = Lc[offc_rt(k) + r1]; }
#pragma unroll 2
        for (int jb = 0; jb < 16; ++jb) {  // blocks in slot 0
            const int j0 = 4 * jb;
            T n0[4], n1[4];
            const int jn = j0 + 4 < n ? j0 + 4 : j0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { n0[k] = Lc[offc_rt(jn + k) + lane]; n1[k] = Lc[offc_rt(jn + k) + r1]; }
            const T t0 = readlane_(w0, j0), t1 = readlane_(w0, j0 + 1), t2 = readlane_(w0, j0 + 2), t3 = readlane_(w0, j0 + 3);
            const T u0 = fma(l0[3], t3, l0[2] * t2) + fma(l0[1], t1, l0[0] * t0);
            const T u1 = fma(l1[3], t3, l1[2] * t2) + fma(l1[1], t1, l1[0] * t0);
            w0 = lane >= j0 + 4 ? w0 - u0 : w0;
            w1 = v1 ? w1 - u1 : w1;
#pragma unroll
            for (int k = 0; k < 4; ++k) { l0[k] = n0[k]; l1[k] = n1[k]; }
        }
#pragma unroll 2
        for (int jb = 16; jb < NB - 1; ++jb) {  // blocks in slot 1
            const int j0 = 4 * jb;
            T n1[4];
            const int jn = j0 + 4 < n ? j0 + 4 : j0;
#pragma unroll
            for (int k = 0; k < 4; ++k) n1[k] = Lc[offc_rt(jn + k) + r1];
            const int q = j0 - 64;
            const T t0 = readlane_(w1, q), t1 = readlane_(w1, q + 1), t2 = readlane_(w1, q + 2), t3 = readlane_(w1, q + 3);
            const T u1 = fma(l1[3], t3, l1[2] * t2) + fma(l1[1], t1, l1[0] * t0);
            w1 = (v1 && 64 + lane >= j0 + 4) ? w1 - u1 : w1;
#pragma unroll
            for (int k = 0; k < 4; ++k) l1[k] = n1[k];
        }
    }
    DEV void back_subst2(T &w0, T &w1)  // L~^T x = z
    {
        const bool v1 = 64 + lane < n;
        const T *pc0 = Lc + offc_rt(lane), *pc1 = Lc + offc_rt(v1 ? 64 + lane : 0);
        T l0[4], l1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { l0[k] = pc0[4 * (NB - 1) + k]; l1[k] = pc1[4 * (NB - 1) + k]; }
#pragma unroll 2
        for (int jb = NB - 1; jb >= 16; --jb) {  // blocks in slot 1: rows j0 .. j0+3 of columns i < j0
            const int j0 = 4 * jb, q = j0 - 64;
            T n0[4], n1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { n0[k] = pc0[j0 - 4 + k]; n1[k] = pc1[j0 - 4 + k]; }
            const T t0 = readlane_(w1, q), t1 = readlane_(w1, q + 1), t2 = readlane_(w1, q + 2), t3 = readlane_(w1, q + 3);
            const T u0 = fma(l0[3], t3, l0[2] * t2) + fma(l0[1], t1, l0[0] * t0);
            const T u1 = fma(l1[3], t3, l1[2] * t2) + fma(l1[1], t1, l1[0] * t0);
            w0 -= u0;
            w1 = (v1 && 64 + lane < j0) ? w1 - u1 : w1;
#pragma unroll
            for (int k = 0; k < 4; ++k) { l0[k] = n0[k]; l1[k] = n1[k]; }
        }
#pragma unroll 2
        for (int jb = 15; jb >= 1; --jb) {  // blocks in slot 0
            const int j0 = 4 * jb;
            T n0[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) n0[k] = pc0[j0 - 4 + k];
            const T t0 = readlane_(w0, j0), t1 = readlane_(w0, j0 + 1), t2 = readlane_(w0, j0 + 2), t3 = readlane_(w0, j0 + 3);
            const T u0 = fma(l0[3], t3, l0[2] * t2) + fma(l0[1], t1, l0[0] * t0);
            w0 = lane < j0 ? w0 - u0 : w0;
#pragma unroll
            for (int k = 0; k < 4; ++k) l0[k] = n0[k];
        }
    }
    DEV T diag_solve1(T y, int j) { return ipm::diag_solve4(sinvb, y, j, lane, n); }  // S^-1 y for the 4x4 block of component j (quad of lanes)
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

    // ---- KKT hooks of the shared solve: every wave runs the specialisation for its tile rows; the barriers inside pair up across them ------
    DEV bool kkt_factor(T sc, T reg, bool want_hmax)
    {
        T hmax = cs[C_HMAX];
        condense_adjoint(sc);
        STAMP(3);
        if (want_hmax) hmax = max_any(tid < n ? fabs(Lc[offc_rt(tid) + tid] + (tid < 2 * MSPLIT ? hm[tid] : (T)0)) : (T)0);   // max |sc * H_jj|: scale of the delta_w shift
        bool factored;
        switch (wv) {
            case 0: factored = assemble_factor<0>(sc, reg, want_hmax, hmax); break;
            case 1: factored = assemble_factor<1>(sc, reg, want_hmax, hmax); break;
            case 2: factored = assemble_factor<2>(sc, reg, want_hmax, hmax); break;
            default: factored = assemble_factor<3>(sc, reg, want_hmax, hmax); break;
        }
        cs[C_HMAX] = hmax;
        return factored;
    }
    DEV T kkt_affine1() { return solve_dir((T)0, false); }      // K^-1 (-sc g)
    DEV T kkt_direction1(T b) { return solve_dir(b, true); }    // K^-1 (-sc g + b)

    DEV void solve(const KIO<T> &io, int b) { ipm::solve(*this, io, b); }
};

// Workgroups per CU.  The workgroup is latency-bound between its barriers and the SIMDs' issue ports are mostly idle (tools/calib/issue_probe.hip), so
// throughput follows the number of resident workgroups almost 1 : 1.  fp64 at N >= 40: 56 ... 74 KB of LDS allow two (256 VGPRs, no scratch); at
// N = 32 / 36 the 44 / 50 KB allow three, which is worth the scratch that 168 VGPRs cost: 1.62 -> 2.14 and 1.39 -> 1.85 M solves/s at B = 262 144.
// fp32: four (128 VGPRs), five up to N = 36 (+8 / +13 % at N = 32 / 36; N = 40 gained 2 % until round 4, when the raw fragments of the one-barrier block-step
// raised its register pressure and the allocator put spill code ahead of an exec-mask restore: tools/spill_exec_check.py -- four, no scratch).
template <typename T, int N>
__global__ __launch_bounds__(256, sizeof(T) == 8 ? (N <= 36 ? 3 : 2) : (N <= 36 ? 5 : 4)) void kmpc_solve_wide_kernel(KP P, KIO<T> io)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[WideSolver<T, N>::lds_elems() * sizeof(T)];
    ipm::run_solver<WideSolver<T, N>>(P, io, smem);
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
    StageW<T> St;
    const T g = ipm::debug_linearize_at(sv, P, io, b, St);
    const T sc = (T)io.sc, reg = (T)io.reg;
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
