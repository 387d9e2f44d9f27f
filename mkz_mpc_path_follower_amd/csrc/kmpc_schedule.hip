// kmpc_schedule.hip -- longest-predicted-first start order for a batch (gfx950 / MI355X).
//
// One wave solves one problem and problems differ 3x and more in iteration count (mean ~10, tail 25-40 at N = 20), so
// a launch of a few thousand problems ends with a few waves still running: its time is (start time of the slowest
// problem) + (that problem's solve time).  Workgroups are dispatched in index order, so starting the problems that
// are expected to run longest FIRST takes the first term to zero (LPT list scheduling).  The predictor is two terms
// of the inputs that a least-squares fit on the iteration counts of synthetic batches singled out:
//     key = |v0 - v_ref| + 0.3 * |psi_ref[N] - psi_ref[0]|,   v_ref = |p_ref[1] - p_ref[0]| / dt
// (speed mismatch against the reference sampling -- the cause of long active-constraint phases and, when the car is
// too fast for its reference, of the non-convex "swerve to lose distance" optima that take 20+ iterations -- and, as
// a tie-breaker, how much the reference turns).  Mean launch time of 4096 problems over 8 synthetic batches, index
// order -> this key: see DESIGN.md section 2; a turn weight of 1.33 (an earlier fit) is 0-11 % slower at N = 12..28.
// Results do not depend on the order.
//
// Implementation: a 256-bucket counting sort on the quantised key in two small kernels -- keys + histogram ranks
// (atomics), then prefix + scatter -- into a permutation the solve kernels index through (KIO::perm).
#include <hip/hip_runtime.h>
#include "kmpc_device.h"

template <typename T>
__device__ __forceinline__ uint32_t sched_bucket(int i, int N, double dt, const T *z0, size_t zs, const T *ref, size_t rs)
{
    // three reference points are enough for a predictor: the spacing of the first segment (references are sampled at
    // constant arclength, ref_gps_traj.py:172-179) and the net heading change between the ends (24 + 16 + 8 B instead of the whole
    // 24 (N+1) B row per problem: 9.3 -> ~3 us at B = 4096)
    const T *r = ref + (size_t)i * rs;   // (rs = 3 (N + 1) for the plain arrays, the record stride for packed records)
    const double dx = (double)r[3] - (double)r[0], dy = (double)r[4] - (double)r[1];
    const double v_ref = sqrt(dx * dx + dy * dy) / dt, turn = fabs((double)r[3 * N + 2] - (double)r[2]);
    const double key = fabs((double)z0[zs * (size_t)i + 3] - v_ref) + 0.3 * turn;
    int q = (int)(key * 48.0);                 // 1/48 m/s resolution; everything above 5.3 shares the first bucket
    q = q < 0 || !(key == key) ? 0 : (q > 255 ? 255 : q);
    return 255u - (uint32_t)q;  // bucket 0 = longest
}

template <typename T>
__global__ __launch_bounds__(256) void kmpc_sched_keys(int B, int N, double dt, const T *z0, size_t zs, const T *ref, size_t rs,
                                                       uint32_t *hist, uint32_t *tag)
{
    // ranks inside a bucket come from a workgroup-local histogram (LDS atomics) plus ONE global atomic per bucket and workgroup: with a
    // global atomic per problem the 256 counters serialised the pre-pass (287 us at B = 262 144 -- 6 % of the N = 8 launch it precedes)
    __shared__ uint32_t lh[256], lbase[256];
    lh[threadIdx.x] = 0u;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bucket = 0u, pos = 0u;
    if (i < B) {
    bucket = sched_bucket(i, N, dt, z0, zs, ref, rs);
    pos = atomicAdd(&lh[bucket], 1u);
    }
    __syncthreads();
    const uint32_t cnt = lh[threadIdx.x];
    lbase[threadIdx.x] = cnt ? atomicAdd(&hist[threadIdx.x], cnt) : 0u;
    __syncthreads();
    if (i < B) tag[i] = bucket | ((lbase[bucket] + pos) << 8);
}

__global__ __launch_bounds__(256) void kmpc_sched_scatter(int B, const uint32_t *hist, uint32_t *hist_next, const uint32_t *tag,
                                                          int32_t *perm)
{
    __shared__ uint32_t pre[256];
    const int t = threadIdx.x;
    pre[t] = hist[t];
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // inclusive scan
        const uint32_t v = t >= d ? pre[t - d] : 0u;
        __syncthreads();
        pre[t] += v;
        __syncthreads();
    }
    if (blockIdx.x == 0) hist_next[t] = 0u;  // the other parity buffer is the next call's histogram
    const int i = blockIdx.x * blockDim.x + t;
    if (i >= B) return;
    const uint32_t g = tag[i], bucket = g & 255u, pos = g >> 8;
    const uint32_t off = bucket ? pre[bucket - 1] : 0u;
    perm[off + pos] = i;
}

template <typename T>
hipError_t kmpc_launch_schedule(int B, int N, double dt, const T *z0, size_t zs, const T *ref, size_t rs, uint32_t *hist, uint32_t *hist_next,
                                uint32_t *tag, int32_t *perm, hipStream_t st)
{
    // (a one-workgroup, one-launch version for batches of a few thousand was measured: 16.5 us against 6.2 + 4.6 us for the two kernels)
    const int nb = (B + 255) / 256;
    hipLaunchKernelGGL((kmpc_sched_keys<T>), dim3(nb), dim3(256), 0, st, B, N, dt, z0, zs, ref, rs, hist, tag);
    hipLaunchKernelGGL(kmpc_sched_scatter, dim3(nb), dim3(256), 0, st, B, (const uint32_t *)hist, hist_next, (const uint32_t *)tag, perm);
    return hipGetLastError();
}
template hipError_t kmpc_launch_schedule<double>(int, int, double, const double *, size_t, const double *, size_t, uint32_t *, uint32_t *, uint32_t *, int32_t *, hipStream_t);
template hipError_t kmpc_launch_schedule<float>(int, int, double, const float *, size_t, const float *, size_t, uint32_t *, uint32_t *, uint32_t *, int32_t *, hipStream_t);

// SoA arrays -> packed records (kmpc_pack_records): thread e of problem b copies scalar e of the record (coalesced writes of whole records)
template <typename T>
__global__ __launch_bounds__(256) void kmpc_pack_kernel(int B, int N, int stride, const T *z0, const T *ref, const T *vt, const T *up, T *rec)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = g / (size_t)stride;
    const int e = (int)(g - b * (size_t)stride);
    if (b >= (size_t)B) return;
    T v = (T)0;
    if (e < 4) v = z0[4 * b + e];
    else if (e == 4) v = vt[b];
    else if (e < 7) v = up[2 * b + (e - 5)];
    else if (e >= 8 && e < 8 + 3 * (N + 1)) v = ref[b * 3 * (N + 1) + (e - 8)];
    rec[g] = v;
}
template <typename T> hipError_t kmpc_launch_pack(int B, int N, int stride, const T *z0, const T *ref, const T *vt, const T *up, T *rec, hipStream_t st)
{
    const size_t tot = (size_t)B * stride;
    hipLaunchKernelGGL((kmpc_pack_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, B, N, stride, z0, ref, vt, up, rec);
    return hipGetLastError();
}
template hipError_t kmpc_launch_pack<double>(int, int, int, const double *, const double *, const double *, const double *, double *, hipStream_t);
template hipError_t kmpc_launch_pack<float>(int, int, int, const float *, const float *, const float *, const float *, float *, hipStream_t);
