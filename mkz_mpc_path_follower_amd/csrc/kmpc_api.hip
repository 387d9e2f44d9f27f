// kmpc_api.hip -- host side of the C ABI declared in include/kmpc.h (gfx950 / MI355X only).
// The library has no CPU fallback: every entry point that computes launches the HIP kernels
// of kmpc_kernels.hip and fails with KMPC_ERR_NODEVICE / KMPC_ERR_HIP when it cannot.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "../../include/kmpc.h"
#include "kmpc_device.h"

template <typename T> hipError_t kmpc_launch_solve(const KP &, const KIO<T> &, hipStream_t);
template <typename T> hipError_t kmpc_launch_condense(const KP &, const KDbg<T> &, hipStream_t);
template <typename T> hipError_t kmpc_launch_probe(const T *, const T *, T *, hipStream_t);
template <typename T> bool kmpc_fast_available(int N);
template <typename T> hipError_t kmpc_launch_solve_fast(const KP &, const KIO<T> &, hipStream_t);
template <typename T> bool kmpc_wide_available(int N);
template <typename T> hipError_t kmpc_launch_solve_wide(const KP &, const KIO<T> &, hipStream_t);
template <typename T> bool kmpc_quad_available(int N);
template <typename T> hipError_t kmpc_launch_solve_quad(const KP &, const KIO<T> &, hipStream_t);
template <typename T> hipError_t kmpc_launch_solve_frenet(const KP &, const KIO<T> &, hipStream_t);
template <typename T> hipError_t kmpc_launch_solve_fast_frenet(const KP &, const KIO<T> &, hipStream_t);
template <typename T> hipError_t kmpc_launch_fast_kkt(const KP &, const KDbgK<T> &, hipStream_t);
template <typename T> hipError_t kmpc_launch_wide_kkt(const KP &, const KDbgK<T> &, hipStream_t);
hipError_t kmpc_launch_sim(int, double *, const double *, int, hipStream_t);
hipError_t kmpc_launch_command(int, const double *, const int32_t *, uint8_t *, double *, double *, hipStream_t);
template <typename T> hipError_t kmpc_launch_schedule(int, int, double, const T *, size_t, const T *, size_t, uint32_t *, uint32_t *, uint32_t *, int32_t *, hipStream_t);
template <typename T> hipError_t kmpc_launch_pack(int, int, int, const T *, const T *, const T *, const T *, T *, hipStream_t);

#ifndef KMPC_QUAD_MIN_BATCH
#define KMPC_HOST_ZERO_COPY_MAX 16   // kmpc_solve_batch_host: batches up to this size run on pinned host memory (no copies); above, staged through device memory
#define KMPC_QUAD_MIN_BATCH 1024   // below this the one-wave-per-problem kernel's shorter single-solve latency wins (measured: tools/quad_probe.py)
#endif

struct kmpc_handle {
    kmpc_config cfg;
    int device;
    hipStream_t stream;
    double cost[8];
    std::string err;
    // staging for the host-pointer entry point
    void *dbuf;
    size_t dbuf_bytes;
    // ... and for SMALL batches (B <= KMPC_HOST_ZERO_COPY_MAX: the reference's own B = 1 loop) one pinned, device-mapped host buffer the kernel reads its
    // inputs from and writes its outputs to directly: no copy launches at all (round 4: 13 small pageable copies cost 120 us of a 174 us control step)
    void *hbuf, *hbuf_dev;
    size_t hbuf_bytes;
    unsigned int *done_flag;   // device view of the completion counter inside hbuf while such a launch is being issued, else NULL
    // start-order workspace (kmpc_schedule.hip): perm[cap], tag[cap], hist[2][256]
    int32_t *perm;
    uint32_t *tag, *hist;
    size_t sched_cap;
    unsigned sched_parity;
};

static std::string g_create_err;
static unsigned long long *g_stamps = nullptr;  // set by kmpc_debug_set_stamps in diagnostic builds
#if defined(KMPC_STAMPS) || defined(KMPC_TRACE)
extern "C" int32_t kmpc_debug_set_stamps(void *p) { g_stamps = (unsigned long long *)p; return 0; }
#endif

static int fail(kmpc_handle *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return fail(h, KMPC_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

extern "C" int32_t kmpc_abi_version(void) { return KMPC_ABI_VERSION; }

// MKZMPCPathFollower.jl:28-48
extern "C" int32_t kmpc_config_default(kmpc_config *c, int32_t N, int32_t dtype)
{
    if (!c) return KMPC_ERR_ARG;
    memset(c, 0, sizeof *c);
    c->N = N;
    c->dtype = dtype;
    c->dt = 0.20;
    c->dt_control = 0.10;
    c->L_a = 1.108;
    c->L_b = 1.742;
    c->steer_max = 0.5;
    c->steer_dmax = 0.5;
    c->a_max = 1.0;
    c->a_dmax = 1.5;
    c->v_min = 0.0;
    c->v_max = 20.0;
    c->max_iter = 200;
    c->hessian = 1;
    c->tol = dtype == KMPC_F32 ? 1e-4 : 1e-8;
    c->mu_init = 1.0;  // Ipopt default is 0.1; with the objective scaled to max-gradient 100, mu = 1 centres the first iterates better (mean iterations -5 %, thinner tail)
    c->bound_relax = dtype == KMPC_F32 ? 1e-5 : 1e-8;
    c->warm_push = 1e-5;  // (1e-5, 1e-6): closed loop on the recorded path 5.5 / 5.0 -> 4.4 / 4.4 mean iterations against (1e-4, 1e-6); a warm start from the solution of an UNRELATED problem is still always Optimal (tools/warm_probe.py, tools/closed_loop_probe.py)
    c->warm_mu = 1e-6;
    c->max_ls = 30;
    c->indef_strategy = 2;  // indefinite exact Hessian: hybrid (Gauss-Newton fallback, delta_w shift from the second failure on)
    c->schedule = 1;  // longest-predicted-first start order (kmpc_schedule.hip)
    c->model = 0;
    c->mu_strategy = 1;  // Mehrotra predictor-corrector (all Optimal on seeded draws at N = 8 ... 56; a third fewer iterations than the monotone rule)
    c->start = 0;        // feed-forward start (1 = the reference's all-zero start, MKZMPCPathFollower.jl:65-72)
    return KMPC_OK;
}

extern "C" int32_t kmpc_create(const kmpc_config *cfg, int32_t device, kmpc_handle **out)
{
    if (!cfg || !out) return fail(nullptr, KMPC_ERR_ARG, "kmpc_create: null argument");
    if (cfg->N < 2 || cfg->N > 56) return fail(nullptr, KMPC_ERR_ARG, "kmpc_create: horizon N=%d outside 2..56", cfg->N);
    if (cfg->dtype != KMPC_F64 && cfg->dtype != KMPC_F32) return fail(nullptr, KMPC_ERR_ARG, "kmpc_create: bad dtype %d", cfg->dtype);
    if (!(cfg->dt > 0) || !(cfg->dt_control > 0) || !(cfg->L_b > 0) || !(cfg->L_a + cfg->L_b > 0) ||
        !(cfg->v_max > cfg->v_min) || !(cfg->a_max > 0) || !(cfg->steer_max > 0) || !(cfg->steer_max < 1.5) ||
        !(cfg->a_dmax > 0) || !(cfg->steer_dmax > 0) || cfg->max_iter < 1 || cfg->max_ls < 1 || !(cfg->tol > 0) ||
        cfg->kernel_variant < 0 || cfg->kernel_variant > 2 || cfg->mu_strategy < 0 || cfg->mu_strategy > 1 ||
        cfg->indef_strategy < 0 || cfg->indef_strategy > 2 || cfg->schedule < 0 || cfg->schedule > 1 || cfg->model < 0 || cfg->model > 1 ||
        cfg->start < 0 || cfg->start > 1 ||
        (cfg->model == 1 && cfg->N > 24 && !(cfg->kernel_variant != 1 && cfg->N == 28)))  // Frenet: generic kernel up to N = 24, compile-time kernel also at 28
        return fail(nullptr, KMPC_ERR_ARG, "kmpc_create: invalid model / solver parameter");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, KMPC_ERR_NODEVICE, "kmpc_create: no HIP device");
    if (device < 0 || device >= ndev) return fail(nullptr, KMPC_ERR_ARG, "kmpc_create: device %d of %d", device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, KMPC_ERR_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, KMPC_ERR_NODEVICE, "kmpc_create: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
    kmpc_handle *h = new kmpc_handle();
    h->cfg = *cfg;
    h->device = device;
    h->dbuf = nullptr;
    h->dbuf_bytes = 0;
    h->hbuf = h->hbuf_dev = nullptr;
    h->hbuf_bytes = 0;
    h->done_flag = nullptr;
    h->perm = nullptr; h->tag = nullptr; h->hist = nullptr; h->sched_cap = 0; h->sched_parity = 0;
    const double w0[8] = {9.0, 9.0, 10.0, 0.0, 100.0, 1000.0, 0.0, 0.0};  // MKZMPCPathFollower.jl:51-59
    const double w1[8] = {0.0, 9.0, 10.0, 0.5, 100.0, 1000.0, 0.0, 0.0};  // MKZMPCPathFollowerFrenet.jl:51-59 (no x slot)
    memcpy(h->cost, cfg->model == 1 ? w1 : w0, sizeof w0);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(nullptr, KMPC_ERR_HIP, "kmpc_create: cannot create stream on device %d", device);
    }
    *out = h;
    return KMPC_OK;
}

extern "C" int32_t kmpc_destroy(kmpc_handle *h)
{
    if (!h) return KMPC_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->dbuf) (void)hipFree(h->dbuf);
    if (h->hbuf) (void)hipHostFree(h->hbuf);
    if (h->perm) (void)hipFree(h->perm);
    if (h->tag) (void)hipFree(h->tag);
    if (h->hist) (void)hipFree(h->hist);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return KMPC_OK;
}

extern "C" int32_t kmpc_set_cost(kmpc_handle *h, const double w[8])
{
    if (!h || !w) return KMPC_ERR_ARG;
    for (int i = 0; i < 8; ++i)
        if (!(w[i] >= 0.0)) return fail(h, KMPC_ERR_ARG, "kmpc_set_cost: weight %d is negative or NaN", i);
    memcpy(h->cost, w, 8 * sizeof(double));
    return KMPC_OK;
}

extern "C" int32_t kmpc_get_cost(kmpc_handle *h, double w[8])
{
    if (!h || !w) return KMPC_ERR_ARG;
    memcpy(w, h->cost, 8 * sizeof(double));
    return KMPC_OK;
}

extern "C" const char *kmpc_last_error(kmpc_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

static KP make_kp(const kmpc_handle *h, int B, int warm, int hessian_override)
{
    const kmpc_config &c = h->cfg;
    KP P;
    memset(&P, 0, sizeof P);
    P.N = c.N; P.B = B; P.max_iter = c.max_iter;
    P.hessian = hessian_override >= 0 ? hessian_override : c.hessian;
    P.warm = warm; P.max_ls = c.max_ls; P.mu_strategy = c.mu_strategy; P.indef_strategy = c.indef_strategy;
    P.start = c.start;
    P.dt = c.dt; P.dtc = c.dt_control; P.L_b = c.L_b; P.r = c.L_b / (c.L_a + c.L_b);
    P.steer_max = c.steer_max; P.steer_dmax = c.steer_dmax; P.a_max = c.a_max; P.a_dmax = c.a_dmax;
    P.v_min = c.v_min; P.v_max = c.v_max;
    memcpy(P.C, h->cost, sizeof P.C);
    P.tol = c.tol; P.mu_init = c.mu_init; P.relax = c.bound_relax; P.warm_push = c.warm_push; P.warm_mu = c.warm_mu;
    P.gap_tol = c.dtype == KMPC_F32 ? 1e-4 : 1e-7;  // unscaled duality-gap bound relative to max(1, |J|)
    for (int i = 0; i < 8; ++i) P.C2[i] = 2.0 * P.C[i];
    P.dt2 = P.dt * P.dt; P.dt_over_Lb = P.dt / P.L_b;
    P.tol_x100 = 100.0 * P.tol; P.tol_x1000 = 1e3 * P.tol; P.tol_d100 = P.tol * 1e-2; P.tol_d10 = P.tol / 10.0;
    return P;
}

// input record of kmpc_solve_batch_packed in scalars: z0[4], v_target, u_prev[2], pad, ref[(N+1)*3], rounded up to whole 64-B lines (72 scalars
// = 576 B at N = 20, fp64).  (Whole 128-B lines were measured too: same counter traffic, more padding -- profiles/README.md, round 4.)
static inline int record_scalars(int N, int dtype)
{
    const int per_line = dtype == KMPC_F64 ? 8 : 16;
    return (8 + 3 * (N + 1) + per_line - 1) / per_line * per_line;
}

template <typename T>
static int solve_dev(kmpc_handle *h, int B, const void *z0, const void *ref, const void *vt, const void *up,
                     void *warmU, int warm, void *u0, int32_t *status, void *cost, void *viol, int32_t *iters,
                     void *outU, void *outX, hipStream_t st, const void *rec = nullptr, void *orec = nullptr)
{
    KIO<T> io;
    const int rstride = record_scalars(h->cfg.N, h->cfg.dtype), per = (int)(64 / sizeof(T));
    io.zs = 4; io.rs = h->cfg.model == 1 ? 4 : 3 * (h->cfg.N + 1); io.vs = 1; io.us = 2; io.u0s = 2; io.ss = 1; io.is = 1;
    if (rec) {   // packed records: the same pointers aim into the records, every stride is the record's
        z0 = rec; ref = (const T *)rec + 8; vt = (const T *)rec + 4; up = (const T *)rec + 5;
        io.zs = io.rs = io.vs = io.us = rstride;
        u0 = orec; cost = (T *)orec + 2; viol = (T *)orec + 3; status = (int32_t *)((T *)orec + 4); iters = status + 1;
        io.u0s = io.ss = per; io.is = 16;
    }
    io.z0 = (const T *)z0; io.ref = (const T *)ref; io.vt = (const T *)vt; io.up = (const T *)up;
    io.warmU = (T *)warmU; io.u0 = (T *)u0; io.status = status; io.cost = (T *)cost; io.viol = (T *)viol;
    io.iters = iters; io.outU = (T *)outU; io.outX = (T *)outX;
    io.stamps = g_stamps;
    const KP P = make_kp(h, B, warm && warmU ? 1 : 0, -1);
    io.perm = nullptr;
    io.done = h->done_flag;   // (set only around the small-batch host entry point's launch)
    // start order: only matters once a launch no longer fits on the chip at once (2 waves x 4 SIMDs x 256 CUs)
    if (h->cfg.model == 1) {  // Frenet functor: `ref` carries k_poly [B,4]; index order (the start-order key reads reference points)
        if (h->cfg.kernel_variant != 1 && kmpc_fast_available<T>(P.N)) HIPCHK(h, kmpc_launch_solve_fast_frenet<T>(P, io, st));
        else HIPCHK(h, kmpc_launch_solve_frenet<T>(P, io, st));   // generic kernel: N <= 24
        return KMPC_OK;
    }
    if (h->cfg.schedule == 1 && B > 2048) {
        if ((size_t)B > h->sched_cap) {
            if (h->perm) (void)hipFree(h->perm);
            if (h->tag) (void)hipFree(h->tag);
            h->perm = nullptr; h->tag = nullptr; h->sched_cap = 0;
            const size_t cap = (size_t)B + (size_t)B / 4;
            HIPCHK(h, hipMalloc((void **)&h->perm, cap * sizeof(int32_t)));
            HIPCHK(h, hipMalloc((void **)&h->tag, cap * sizeof(uint32_t)));
            h->sched_cap = cap;
        }
        if (!h->hist) {
            HIPCHK(h, hipMalloc((void **)&h->hist, 2 * 256 * sizeof(uint32_t)));
            HIPCHK(h, hipMemsetAsync(h->hist, 0, 2 * 256 * sizeof(uint32_t), st));
            h->sched_parity = 0;
        }
        uint32_t *hc = h->hist + 256 * (h->sched_parity & 1), *hn = h->hist + 256 * ((h->sched_parity & 1) ^ 1);
        HIPCHK(h, kmpc_launch_schedule<T>(B, P.N, P.dt, io.z0, (size_t)io.zs, io.ref, (size_t)io.rs, hc, hn, h->tag, h->perm, st));
        h->sched_parity ^= 1;
        io.perm = h->perm;
    }
    // N = 8 (the reference's own horizon): four problems per wave once a batch has enough problems to fill the chip that way
    if (h->cfg.kernel_variant == 0 && kmpc_quad_available<T>(P.N) && B >= KMPC_QUAD_MIN_BATCH) HIPCHK(h, kmpc_launch_solve_quad<T>(P, io, st));
    else if (h->cfg.kernel_variant != 1 && kmpc_fast_available<T>(P.N)) HIPCHK(h, kmpc_launch_solve_fast<T>(P, io, st));   // one wave per problem
    else if (h->cfg.kernel_variant != 1 && kmpc_wide_available<T>(P.N)) HIPCHK(h, kmpc_launch_solve_wide<T>(P, io, st));  // four waves per problem
    else HIPCHK(h, kmpc_launch_solve<T>(P, io, st));
    return KMPC_OK;
}

extern "C" int32_t kmpc_solve_batch(kmpc_handle *h, int32_t B, const void *z0, const void *ref, const void *v_target,
                                    const void *u_prev, void *warm_U, int32_t warm, void *out_u0, int32_t *out_status,
                                    void *out_cost, void *out_viol, int32_t *out_iters, void *out_U, void *out_X,
                                    void *stream)
{
    if (!h) return KMPC_ERR_ARG;
    if (B < 0) return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch: B=%d", B);
    if (B == 0) return KMPC_OK;
    if (!z0 || !ref || !v_target || !u_prev || !out_u0 || !out_status)
        return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch: null required buffer");
    if (h->cfg.model != 0) return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch: handle was created for the Frenet model; use kmpc_solve_batch_frenet");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the device's default stream, as in HIP
    if (h->cfg.dtype == KMPC_F64)
        return solve_dev<double>(h, B, z0, ref, v_target, u_prev, warm_U, warm, out_u0, out_status, out_cost, out_viol,
                                 out_iters, out_U, out_X, st);
    return solve_dev<float>(h, B, z0, ref, v_target, u_prev, warm_U, warm, out_u0, out_status, out_cost, out_viol,
                            out_iters, out_U, out_X, st);
}

extern "C" int64_t kmpc_record_bytes(int32_t N, int32_t dtype)
{
    if (N < 2 || N > 56 || (dtype != KMPC_F64 && dtype != KMPC_F32)) return -1;
    return (int64_t)record_scalars(N, dtype) * (dtype == KMPC_F64 ? 8 : 4);
}

extern "C" int32_t kmpc_pack_records(kmpc_handle *h, int32_t B, const void *z0, const void *ref, const void *v_target, const void *u_prev,
                                     void *records, void *stream)
{
    if (!h) return KMPC_ERR_ARG;
    if (B < 0) return fail(h, KMPC_ERR_ARG, "kmpc_pack_records: B=%d", B);
    if (B == 0) return KMPC_OK;
    if (!z0 || !ref || !v_target || !u_prev || !records) return fail(h, KMPC_ERR_ARG, "kmpc_pack_records: null buffer");
    if (h->cfg.model != 0) return fail(h, KMPC_ERR_ARG, "kmpc_pack_records: Cartesian model only");
    HIPCHK(h, hipSetDevice(h->device));
    const int stride = record_scalars(h->cfg.N, h->cfg.dtype);
    if (h->cfg.dtype == KMPC_F64)
        HIPCHK(h, kmpc_launch_pack<double>(B, h->cfg.N, stride, (const double *)z0, (const double *)ref, (const double *)v_target, (const double *)u_prev, (double *)records, (hipStream_t)stream));
    else
        HIPCHK(h, kmpc_launch_pack<float>(B, h->cfg.N, stride, (const float *)z0, (const float *)ref, (const float *)v_target, (const float *)u_prev, (float *)records, (hipStream_t)stream));
    return KMPC_OK;
}

extern "C" int32_t kmpc_solve_batch_packed(kmpc_handle *h, int32_t B, const void *records, void *warm_U, int32_t warm, void *out_records,
                                           void *out_U, void *out_X, void *stream)
{
    if (!h) return KMPC_ERR_ARG;
    if (B < 0) return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch_packed: B=%d", B);
    if (B == 0) return KMPC_OK;
    if (!records || !out_records) return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch_packed: null required buffer");
    if (((uintptr_t)records | (uintptr_t)out_records) & 63) return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch_packed: records must be 64-byte aligned");
    if (h->cfg.model != 0) return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch_packed: Cartesian model only");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    if (h->cfg.dtype == KMPC_F64)
        return solve_dev<double>(h, B, nullptr, nullptr, nullptr, nullptr, warm_U, warm, nullptr, nullptr, nullptr, nullptr, nullptr, out_U, out_X, st, records, out_records);
    return solve_dev<float>(h, B, nullptr, nullptr, nullptr, nullptr, warm_U, warm, nullptr, nullptr, nullptr, nullptr, nullptr, out_U, out_X, st, records, out_records);
}

extern "C" int32_t kmpc_solve_batch_frenet(kmpc_handle *h, int32_t B, const void *z0, const void *k_poly, const void *v_target,
                                           const void *u_prev, void *warm_U, int32_t warm, void *out_u0, int32_t *out_status,
                                           void *out_cost, void *out_viol, int32_t *out_iters, void *out_U, void *out_X,
                                           void *stream)
{
    if (!h) return KMPC_ERR_ARG;
    if (B < 0) return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch_frenet: B=%d", B);
    if (B == 0) return KMPC_OK;
    if (!z0 || !k_poly || !v_target || !u_prev || !out_u0 || !out_status)
        return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch_frenet: null required buffer");
    if (h->cfg.model != 1) return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch_frenet: handle was created with cfg.model = 0");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    if (h->cfg.dtype == KMPC_F64)
        return solve_dev<double>(h, B, z0, k_poly, v_target, u_prev, warm_U, warm, out_u0, out_status, out_cost, out_viol,
                                 out_iters, out_U, out_X, st);
    return solve_dev<float>(h, B, z0, k_poly, v_target, u_prev, warm_U, warm, out_u0, out_status, out_cost, out_viol,
                            out_iters, out_U, out_X, st);
}

extern "C" int32_t kmpc_solve_batch_host(kmpc_handle *h, int32_t B, const void *z0, const void *ref,
                                         const void *v_target, const void *u_prev, void *warm_U, int32_t warm,
                                         void *out_u0, int32_t *out_status, void *out_cost, void *out_viol,
                                         int32_t *out_iters, void *out_U, void *out_X)
{
    if (!h) return KMPC_ERR_ARG;
    if (B < 0) return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch_host: B=%d", B);
    if (B == 0) return KMPC_OK;
    if (!z0 || !ref || !v_target || !u_prev || !out_u0 || !out_status)
        return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch_host: null required buffer");
    if (h->cfg.model != 0) return fail(h, KMPC_ERR_ARG, "kmpc_solve_batch_host: Cartesian model only (use kmpc_solve_batch_frenet with device buffers)");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t es = h->cfg.dtype == KMPC_F64 ? 8 : 4;
    const size_t N = h->cfg.N, b = (size_t)B;
    // carve one device allocation: inputs then outputs, each 256-B aligned
    const size_t sz[13] = {b * 4 * es, b * (N + 1) * 3 * es, b * es, b * 2 * es, b * N * 2 * es,  // z0 ref vt up warmU
                           b * 2 * es, b * 4, b * es, b * es, b * 4, b * N * 2 * es, b * (N + 1) * 4 * es, 0};
    size_t off[13], total = 0;
    for (int i = 0; i < 13; ++i) { off[i] = total; total += (sz[i] + 255) & ~(size_t)255; }
    hipStream_t st = h->stream;
    if (B <= KMPC_HOST_ZERO_COPY_MAX) {
        // small batch: the kernel works on pinned host memory (inputs: one burst of loads per problem when it starts; outputs: posted writes when it ends)
        if (total > h->hbuf_bytes) {
            if (h->hbuf) HIPCHK(h, hipHostFree(h->hbuf));
            h->hbuf = h->hbuf_dev = nullptr; h->hbuf_bytes = 0;
            size_t cap = 0;   // sized once for the largest zero-copy batch of this handle's horizon
            { const size_t bb = KMPC_HOST_ZERO_COPY_MAX;
              const size_t szm[12] = {bb * 4 * es, bb * (N + 1) * 3 * es, bb * es, bb * 2 * es, bb * N * 2 * es, bb * 2 * es, bb * 4, bb * es, bb * es, bb * 4, bb * N * 2 * es, bb * (N + 1) * 4 * es};
              for (int i = 0; i < 12; ++i) cap += (szm[i] + 255) & ~(size_t)255; }
            cap += 256;   // + the completion counter
            HIPCHK(h, hipHostMalloc(&h->hbuf, cap, hipHostMallocMapped));
            HIPCHK(h, hipHostGetDevicePointer(&h->hbuf_dev, h->hbuf, 0));
            h->hbuf_bytes = cap;
        }
        char *hp = (char *)h->hbuf, *dp = (char *)h->hbuf_dev;
        memcpy(hp + off[0], z0, sz[0]); memcpy(hp + off[1], ref, sz[1]); memcpy(hp + off[2], v_target, sz[2]); memcpy(hp + off[3], u_prev, sz[3]);
        if (warm_U && warm) memcpy(hp + off[4], warm_U, sz[4]);
        // completion: every problem adds 1 to a counter in the pinned buffer after its outputs (release, system scope); the host spins on it -- the runtime's
        // own completion path (interrupt / signal wait) is several microseconds slower -- and falls back to the stream after 2 ms (long solves, stalled device)
        volatile unsigned int *flag_h = (volatile unsigned int *)(hp + h->hbuf_bytes - 256);
        *flag_h = 0u;
        h->done_flag = (unsigned int *)(dp + h->hbuf_bytes - 256);
        int rc = kmpc_solve_batch(h, B, dp + off[0], dp + off[1], dp + off[2], dp + off[3], warm_U ? dp + off[4] : nullptr, warm,
                                  dp + off[5], (int32_t *)(dp + off[6]), dp + off[7], dp + off[8], (int32_t *)(dp + off[9]),
                                  out_U ? dp + off[10] : nullptr, out_X ? dp + off[11] : nullptr, st);
        h->done_flag = nullptr;
        if (rc != KMPC_OK) return rc;
        {
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            bool seen = false;
            for (;;) {
                if (__atomic_load_n((const unsigned int *)flag_h, __ATOMIC_ACQUIRE) >= (unsigned)B) { seen = true; break; }
                if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
            }
            if (!seen) HIPCHK(h, hipStreamSynchronize(st));
        }
        memcpy(out_u0, hp + off[5], sz[5]); memcpy(out_status, hp + off[6], sz[6]);
        if (out_cost) memcpy(out_cost, hp + off[7], sz[7]);
        if (out_viol) memcpy(out_viol, hp + off[8], sz[8]);
        if (out_iters) memcpy(out_iters, hp + off[9], sz[9]);
        if (out_U) memcpy(out_U, hp + off[10], sz[10]);
        if (out_X) memcpy(out_X, hp + off[11], sz[11]);
        if (warm_U) memcpy(warm_U, hp + off[4], sz[4]);
        return KMPC_OK;
    }
    if (total > h->dbuf_bytes) {
        if (h->dbuf) HIPCHK(h, hipFree(h->dbuf));
        h->dbuf = nullptr; h->dbuf_bytes = 0;
        HIPCHK(h, hipMalloc(&h->dbuf, total));
        h->dbuf_bytes = total;
    }
    char *d = (char *)h->dbuf;
    HIPCHK(h, hipMemcpyAsync(d + off[0], z0, sz[0], hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(d + off[1], ref, sz[1], hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(d + off[2], v_target, sz[2], hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(d + off[3], u_prev, sz[3], hipMemcpyHostToDevice, st));
    if (warm_U && warm) HIPCHK(h, hipMemcpyAsync(d + off[4], warm_U, sz[4], hipMemcpyHostToDevice, st));
    int rc = kmpc_solve_batch(h, B, d + off[0], d + off[1], d + off[2], d + off[3], warm_U ? d + off[4] : nullptr, warm,
                              d + off[5], (int32_t *)(d + off[6]), d + off[7], d + off[8], (int32_t *)(d + off[9]),
                              out_U ? d + off[10] : nullptr, out_X ? d + off[11] : nullptr, st);
    if (rc != KMPC_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(out_u0, d + off[5], sz[5], hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(out_status, d + off[6], sz[6], hipMemcpyDeviceToHost, st));
    if (out_cost) HIPCHK(h, hipMemcpyAsync(out_cost, d + off[7], sz[7], hipMemcpyDeviceToHost, st));
    if (out_viol) HIPCHK(h, hipMemcpyAsync(out_viol, d + off[8], sz[8], hipMemcpyDeviceToHost, st));
    if (out_iters) HIPCHK(h, hipMemcpyAsync(out_iters, d + off[9], sz[9], hipMemcpyDeviceToHost, st));
    if (out_U) HIPCHK(h, hipMemcpyAsync(out_U, d + off[10], sz[10], hipMemcpyDeviceToHost, st));
    if (out_X) HIPCHK(h, hipMemcpyAsync(out_X, d + off[11], sz[11], hipMemcpyDeviceToHost, st));
    if (warm_U) HIPCHK(h, hipMemcpyAsync(warm_U, d + off[4], sz[4], hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    return KMPC_OK;
}

extern "C" int32_t kmpc_debug_condense(kmpc_handle *h, int32_t B, const void *z0, const void *ref, const void *v_target,
                                       const void *U, int32_t hessian, void *H, void *g, void *J, void *stream)
{
    if (!h || B <= 0 || !z0 || !ref || !v_target || !U || !H || !g || !J) return h ? fail(h, KMPC_ERR_ARG, "kmpc_debug_condense: bad argument") : KMPC_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the device's default stream, as in HIP
    const KP P = make_kp(h, B, 0, hessian);
    if (h->cfg.dtype == KMPC_F64) {
        KDbg<double> io = {(const double *)z0, (const double *)ref, (const double *)v_target, (const double *)U,
                           (double *)H, (double *)g, (double *)J};
        HIPCHK(h, kmpc_launch_condense<double>(P, io, st));
    } else {
        KDbg<float> io = {(const float *)z0, (const float *)ref, (const float *)v_target, (const float *)U,
                          (float *)H, (float *)g, (float *)J};
        HIPCHK(h, kmpc_launch_condense<float>(P, io, st));
    }
    return KMPC_OK;
}

template <typename T>
static int debug_kkt(kmpc_handle *h, int B, const void *z0, const void *ref, const void *vt, const void *up, const void *U, const void *w,
                     const void *b, double sc, double reg, int hessian, void *K, void *g, void *x, int32_t *ok, hipStream_t st)
{
    const KP P = make_kp(h, B, 0, hessian);
    KDbgK<T> io = {(const T *)z0, (const T *)ref, (const T *)vt, (const T *)up, (const T *)U, (const T *)w, (const T *)b, sc, reg,
                   (T *)K, (T *)g, (T *)x, ok};
    if (kmpc_fast_available<T>(P.N)) HIPCHK(h, kmpc_launch_fast_kkt<T>(P, io, st));
    else if (kmpc_wide_available<T>(P.N)) HIPCHK(h, kmpc_launch_wide_kkt<T>(P, io, st));
    else return fail(h, KMPC_ERR_ARG, "kmpc_debug_kkt: no compile-time-horizon kernel for N=%d in this element type", P.N);
    return KMPC_OK;
}

extern "C" int32_t kmpc_debug_kkt(kmpc_handle *h, int32_t B, const void *z0, const void *ref, const void *v_target, const void *u_prev,
                                  const void *U, const void *w, const void *b, double sc, double reg, int32_t hessian, void *K_out,
                                  void *g, void *x, int32_t *ok, void *stream)
{
    if (!h || B <= 0 || !z0 || !ref || !v_target || !u_prev || !U || !w || !b || !K_out || !g || !x || !ok || !(sc > 0) || !(reg >= 0))
        return h ? fail(h, KMPC_ERR_ARG, "kmpc_debug_kkt: bad argument") : KMPC_ERR_ARG;
    if (h->cfg.model != 0) return fail(h, KMPC_ERR_ARG, "kmpc_debug_kkt: Cartesian model only");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    if (h->cfg.dtype == KMPC_F64) return debug_kkt<double>(h, B, z0, ref, v_target, u_prev, U, w, b, sc, reg, hessian, K_out, g, x, ok, st);
    return debug_kkt<float>(h, B, z0, ref, v_target, u_prev, U, w, b, sc, reg, hessian, K_out, g, x, ok, st);
}

extern "C" int32_t kmpc_debug_mfma_probe(kmpc_handle *h, const void *a, const void *b, void *d, void *stream)
{
    if (!h || !a || !b || !d) return KMPC_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the device's default stream, as in HIP
    if (h->cfg.dtype == KMPC_F64) HIPCHK(h, kmpc_launch_probe<double>((const double *)a, (const double *)b, (double *)d, st));
    else HIPCHK(h, kmpc_launch_probe<float>((const float *)a, (const float *)b, (float *)d, st));
    return KMPC_OK;
}

// ---- batched waypoint generation (scripts/gps_utils/ref_gps_traj.py) -------------------------------
struct WP {
    int M, B, H;
    int use_vtarget;
    double traj_dt;
    const double *t, *X, *Y, *psi, *s;
    const double *pose;
    const double *vt;
    double *ref;
    int32_t *stop;
    int32_t *closest;
};
hipError_t kmpc_launch_waypoints(const WP &w, hipStream_t st);

struct kmpc_path {
    int device, M;
    double *d;  // t | X | Y | psi | s, each M doubles
    std::string err;
};

extern "C" int32_t kmpc_path_create(int32_t device, int32_t M, const double *t, const double *X, const double *Y,
                                    const double *psi, const double *cdist, kmpc_path **out)
{
    if (!out || M < 2 || !t || !X || !Y || !psi || !cdist) return fail(nullptr, KMPC_ERR_ARG, "kmpc_path_create: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, KMPC_ERR_NODEVICE, "kmpc_path_create: no HIP device");
    if (device < 0 || device >= ndev) return fail(nullptr, KMPC_ERR_ARG, "kmpc_path_create: device %d of %d", device, ndev);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, KMPC_ERR_HIP, "kmpc_path_create: hipSetDevice failed");
    kmpc_path *p = new kmpc_path();
    p->device = device; p->M = M; p->d = nullptr;
    if (hipMalloc((void **)&p->d, (size_t)5 * M * sizeof(double)) != hipSuccess) { delete p; return fail(nullptr, KMPC_ERR_HIP, "kmpc_path_create: hipMalloc failed"); }
    const double *src[5] = {t, X, Y, psi, cdist};
    for (int i = 0; i < 5; ++i)
        if (hipMemcpy(p->d + (size_t)i * M, src[i], (size_t)M * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(p->d); delete p;
            return fail(nullptr, KMPC_ERR_HIP, "kmpc_path_create: hipMemcpy failed");
        }
    *out = p;
    return KMPC_OK;
}

extern "C" int32_t kmpc_path_destroy(kmpc_path *p)
{
    if (!p) return KMPC_OK;
    (void)hipSetDevice(p->device);
    (void)hipFree(p->d);
    delete p;
    return KMPC_OK;
}

extern "C" int32_t kmpc_waypoints_batch(kmpc_path *p, int32_t B, int32_t horizon, double traj_dt, const double *pose,
                                        const double *v_target, double *ref_out, int32_t *stop_out, int32_t *closest_out,
                                        void *stream)
{
    if (!p) return KMPC_ERR_ARG;
    if (B < 0 || horizon < 1 || horizon > 63 || !(traj_dt > 0)) { p->err = "kmpc_waypoints_batch: bad B / horizon / traj_dt"; return KMPC_ERR_ARG; }
    if (B == 0) return KMPC_OK;
    if (!pose || !ref_out || !stop_out) { p->err = "kmpc_waypoints_batch: null required buffer"; return KMPC_ERR_ARG; }
    if (hipSetDevice(p->device) != hipSuccess) { p->err = "hipSetDevice failed"; return KMPC_ERR_HIP; }
    WP w;
    w.M = p->M; w.B = B; w.H = horizon; w.use_vtarget = v_target ? 1 : 0; w.traj_dt = traj_dt;
    w.t = p->d; w.X = p->d + p->M; w.Y = p->d + 2 * (size_t)p->M; w.psi = p->d + 3 * (size_t)p->M; w.s = p->d + 4 * (size_t)p->M;
    w.pose = pose; w.vt = v_target; w.ref = ref_out; w.stop = stop_out; w.closest = closest_out;
    hipError_t e = kmpc_launch_waypoints(w, (hipStream_t)stream);
    if (e != hipSuccess) { p->err = std::string("waypoints launch failed: ") + hipGetErrorString(e); return KMPC_ERR_HIP; }
    return KMPC_OK;
}

extern "C" const char *kmpc_path_last_error(kmpc_path *p) { return p ? p->err.c_str() : g_create_err.c_str(); }

// ---- closed-loop simulator (kmpc_sim.hip) ------------------------------------------------------------------------
extern "C" int32_t kmpc_sim_advance_batch(int32_t device, int32_t B, void *state, const void *cmd, int32_t n_updates, void *stream)
{
    if (B < 0 || n_updates < 0 || (B > 0 && (!state || !cmd))) return fail(nullptr, KMPC_ERR_ARG, "kmpc_sim_advance_batch: bad argument (B=%d, n_updates=%d)", B, n_updates);
    if (B == 0 || n_updates == 0) return KMPC_OK;
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, KMPC_ERR_HIP, "kmpc_sim_advance_batch: hipSetDevice(%d) failed", device);
    const hipError_t e = kmpc_launch_sim(B, (double *)state, (const double *)cmd, n_updates, (hipStream_t)stream);
    if (e != hipSuccess) return fail(nullptr, KMPC_ERR_HIP, "kmpc_sim_advance_batch: %s", hipGetErrorString(e));
    return KMPC_OK;
}

extern "C" int32_t kmpc_command_batch(int32_t device, int32_t B, const void *u0, const int32_t *stop, uint8_t *stop_latch, void *u_prev, void *cmd, void *stream)
{
    if (B < 0 || (B > 0 && (!u0 || !stop || !stop_latch || !u_prev || !cmd))) return fail(nullptr, KMPC_ERR_ARG, "kmpc_command_batch: bad argument (B=%d)", B);
    if (B == 0) return KMPC_OK;
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, KMPC_ERR_HIP, "kmpc_command_batch: hipSetDevice(%d) failed", device);
    const hipError_t e = kmpc_launch_command(B, (const double *)u0, stop, stop_latch, (double *)u_prev, (double *)cmd, (hipStream_t)stream);
    if (e != hipSuccess) return fail(nullptr, KMPC_ERR_HIP, "kmpc_command_batch: %s", hipGetErrorString(e));
    return KMPC_OK;
}
