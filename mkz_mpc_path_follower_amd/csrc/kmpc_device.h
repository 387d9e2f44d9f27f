// kmpc_device.h -- kernel-side parameter blocks shared by kmpc_kernels.hip and kmpc_api.hip.
// gfx950 (MI355X) only.
#pragma once
#include <stdint.h>

// model + solver parameters, passed to kernels by value
struct KP {
    int N, B, max_iter, hessian, warm, max_ls, mu_strategy, indef_strategy;
    int start;  // start point: 0 feed-forward, 1 the reference's all-zero start (kmpc_ipm.h)
    double dt, dtc, L_b, r;  // r = L_b / (L_a + L_b)  (MKZMPCPathFollower.jl:115)
    double steer_max, steer_dmax, a_max, a_dmax, v_min, v_max;
    double C[8];  // update_cost order: C_x, C_y, C_psi, C_v, C_dacc, C_ddf, C_acc, C_df
    double tol, mu_init, relax, warm_push, warm_mu, gap_tol;
    // derived wave-uniform constants, computed once on the host: there is no scalar fp64 ALU, so a product like 2*C_x or 100*tol
    // evaluated in the kernel occupies a VGPR pair for the whole solve once it is hoisted out of the loop (measured: ~30 spilled VGPRs)
    double C2[8];  // 2 * C[i]
    double dt2, dt_over_Lb, tol_x100, tol_x1000, tol_d100, tol_d10;
};

// device buffers of one batch (T = double / float)
template <typename T>
struct KIO {
    const T *z0, *ref, *vt, *up;
    T *warmU;
    T *u0;
    int32_t *status;
    T *cost, *viol;
    int32_t *iters;
    T *outU, *outX;
    unsigned long long *stamps;  // diagnostic builds only (KMPC_STAMPS), else NULL
    const int32_t *perm;         // start order: workgroup i solves problem perm[i] (NULL = index order)
    // element strides of the per-problem entries (ABI v8): the plain arrays of kmpc_solve_batch have (4, 3 (N + 1), 1, 2 | 2, 1, 1); with packed records
    // (kmpc_solve_batch_packed) the same pointers aim INTO the records -- z0 = rec, vt = rec + 4, up = rec + 5, ref = rec + 8; u0 = orec,
    // cost = orec + 2, viol = orec + 3, status / iters = the two int32 behind them -- and every stride is the record's (whole 64-B lines): no second code path
    int zs, rs, vs, us;      // inputs: z0, ref, v_target, u_prev
    int u0s, ss, is;         // outputs: u0, cost / viol, status / iters
    unsigned int *done;          // host-visible completion counter (pinned memory) of the small-batch host entry point, else NULL: every problem adds 1 after its outputs
};

template <typename T>
struct KDbg {
    const T *z0, *ref, *vt, *U;
    T *H, *g, *J;
};

// diagnostics of the KKT pipeline of the compile-time-horizon kernels (kmpc_debug_kkt)
template <typename T>
struct KDbgK {
    const T *z0, *ref, *vt, *up, *U, *w, *b;
    double sc, reg;
    T *K, *g, *x;
    int32_t *ok;
};

constexpr int KMPC_STG = 36;  // scalars stored per stage in LDS (Cartesian: 13; Frenet: 13 Jacobian + 3 roll-out + 4 costate + 15 Hessian)

// LDS bytes the solver kernel needs for horizon N with NT column tiles
#ifdef __HIPCC__
#define KMPC_HD __host__ __device__
#else
#define KMPC_HD
#endif
template <typename T>
KMPC_HD inline size_t kmpc_lds_bytes(int N, int NT)
{
    const int n = 2 * N;
    const int NF = (40 * NT - 2 + 63) / 64;
    // K (n x (n+1)), stage records, xb, wb, cb, dinv, panel scratch of the blocked Cholesky (64 NT)
    // from NT = 5 the matrix is padded to whole 16x16 tiles (the Hessian tiles are accumulated in it without bounds checks)
    const size_t kelems = NT >= 5 ? (size_t)(16 * NT) * (16 * NT + 1) : (size_t)n * (n + 1);
    size_t elems = kelems + (size_t)KMPC_STG * (N + 1) + 16 * NT + 64 * NF + 64 + 16 * NT + 64 * NT;
    elems = (elems + 1) & ~(size_t)1;
    return elems * sizeof(T);
}
