// tools/calib/issue_probe.hip -- what ONE fp64 wave (and two waves sharing a SIMD) pays per instruction on gfx950: the numbers DESIGN.md section 4c
// reasons with.  Not part of the product or the tests.  Build + run: bash tools/calib/issue_run.sh (on the GPU box).
// Every probe runs REP unrolled copies of a short pattern between two s_memtime reads in wave 0 and reports cycles per pattern.
// waves_per_simd = 1: a 64-thread block; 2: a 512-thread block (8 waves: two per SIMD, all running the same code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 16   // patterns per trip of the outer (rolled) loop; the slope between two trip counts cancels every fixed cost
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }

enum { P_FMA_DEP = 0, P_FMA_IND4, P_MUL_DEP, P_F32_DEP, P_F32_IND4, P_DPP_ADD, P_READLANE_FMA, P_LDS_RT, P_LDS_RT_UNIFORM, P_MFMA_DEP, P_MFMA_IND4,
       P_MFMA_THEN_VALU8, P_VALU8, P_RSQ_NEWTON, P_RCP, P_CNDMASK_DEP, P_COUNT };
static const char *NAMES[P_COUNT] = {
    "v_fma_f64, dependent chain (per instruction)", "v_fma_f64, 4 independent chains (per instruction)", "v_mul_f64, dependent chain",
    "v_fma_f32, dependent chain", "v_fma_f32, 4 independent chains (per instruction)", "fp64 DPP row_shr:1 + v_add_f64 (one scan step: 2 v_mov_dpp + add)",
    "fp64 v_readlane x2 -> v_fma_f64 with the SGPR pair (one substitution operand)", "LDS write b64 -> read b64 of another lane's word -> dependent use (round trip)",
    "LDS uniform-address read b64 with a dependent address (v_cvt + readfirstlane + read + use)", "v_mfma_f64_16x16x4, dependent accumulator chain (per MFMA)",
    "v_mfma_f64_16x16x4, 4 independent accumulators (per MFMA)", "1 MFMA + 8 dependent v_fma_f64 behind it (independent of the MFMA)",
    "8 dependent v_fma_f64 alone", "rsqrt_: v_rsq_f64 + cubic step (6 instructions, dependent chain on the result)", "v_rcp_f64 + 2 Newton steps (5 instructions)",
    "v_cndmask_b32 x2 (one fp64 select), dependent chain"};

template <int which> __global__ void probe(int reps, double *out, unsigned long long *cyc, double seed)
{
    __shared__ double lds[512];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = seed + threadIdx.x;
    __syncthreads();
    double a = seed + 1e-3 * lane, b = 1.0 + 1e-9 * lane, c = 0.5, d = 0.25, e = 0.125;
    float fa = (float)a, fb = (float)b, fc = 0.5f, fd = 0.25f, fe = 0.125f;
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    pin(a); pin(b); pin(c); pin(d); pin(e);
    unsigned long long t0 = now();
#pragma nounroll
    for (int rr = 0; rr < reps; ++rr) {
    switch (which) {   // compile-time: one kernel per pattern, nothing but the pattern in the loop
    case P_FMA_DEP:
#pragma unroll
        for (int i = 0; i < REP; ++i) { a = fma(a, b, c); }
        break;
    case P_FMA_IND4:
#pragma unroll
        for (int i = 0; i < REP / 4; ++i) { a = fma(a, b, 1.0); c = fma(c, b, 1.0); d = fma(d, b, 1.0); e = fma(e, b, 1.0); }
        break;
    case P_MUL_DEP:
#pragma unroll
        for (int i = 0; i < REP; ++i) { a = a * b; }
        break;
    case P_F32_DEP:
#pragma unroll
        for (int i = 0; i < REP; ++i) { fa = fmaf(fa, fb, fc); }
        break;
    case P_F32_IND4:
#pragma unroll
        for (int i = 0; i < REP / 4; ++i) {
            fa = fmaf(fa, fb, 1.0f); fc = fmaf(fc, fb, 1.0f); fd = fmaf(fd, fb, 1.0f); fe = fmaf(fe, fb, 1.0f);
        }
        break;
    case P_DPP_ADD:
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            int lo = __double2loint(a), hi = __double2hiint(a);
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true);
            a = a + __hiloint2double(hi, lo);
        }
        break;
    case P_READLANE_FMA:
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            const double t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), i & 63), __builtin_amdgcn_readlane(__double2loint(a), i & 63));
            a = fma(b, t, a);
        }
        break;
    case P_LDS_RT:
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            lds[threadIdx.x] = a;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            a = lds[threadIdx.x ^ 1] * b;
        }
        break;
    case P_LDS_RT_UNIFORM:
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            int idx = (int)(a * 0.0) + (i & 63);        // dependent address, uniform across the wave
            idx = __builtin_amdgcn_readfirstlane(idx);
            a = a * 0.5 + lds[idx];
        }
        break;
    case P_MFMA_DEP:
#pragma unroll
        for (int i = 0; i < REP; ++i) { acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0); }
        break;
    case P_MFMA_IND4:
#pragma unroll
        for (int i = 0; i < REP / 4; ++i) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
        }
        break;
    case P_MFMA_THEN_VALU8:
#pragma unroll
        for (int i = 0; i < REP / 4; ++i) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d, e, acc0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) { a = fma(a, b, c); }
        }
        break;
    case P_VALU8:
#pragma unroll
        for (int i = 0; i < REP / 4; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { a = fma(a, b, c); }
        }
        break;
    case P_RSQ_NEWTON:
#pragma unroll
        for (int i = 0; i < REP / 4; ++i) {
            double y = __builtin_amdgcn_rsq(a);
            const double er = fma(-a * y, y, 1.0);
            y = fma(y * er, fma(er, 0.375, 0.5), y);
            a = y + 1.0;
        }
        break;
    case P_RCP:
#pragma unroll
        for (int i = 0; i < REP / 4; ++i) {
            double y = __builtin_amdgcn_rcp(a);
            double er = fma(-a, y, 1.0);
            y = fma(y, er, y);
            er = fma(-a, y, 1.0);
            a = fma(y, er, y) + 1.0;
        }
        break;
    case P_CNDMASK_DEP:
#pragma unroll
        for (int i = 0; i < REP; ++i) { a = ((lane >> (i & 3)) & 1) ? a * 1.0 : b; }
        break;
    }
    }
    unsigned long long t1 = now();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d + e + fa + fc + fd + fe + acc0[0] + acc1[1] + acc2[2] + acc3[3] + lds[(threadIdx.x + 1) & 511];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W> static void launch_one(int threads, int reps, double *out, unsigned long long *cyc)
{
    hipLaunchKernelGGL((probe<W>), dim3(1), dim3(threads), 0, 0, reps, out, cyc, 1.25);
}
static void launch(int w, int threads, int reps, double *out, unsigned long long *cyc)
{
    switch (w) {
        case 0: launch_one<0>(threads, reps, out, cyc); break; case 1: launch_one<1>(threads, reps, out, cyc); break;
        case 2: launch_one<2>(threads, reps, out, cyc); break; case 3: launch_one<3>(threads, reps, out, cyc); break;
        case 4: launch_one<4>(threads, reps, out, cyc); break; case 5: launch_one<5>(threads, reps, out, cyc); break;
        case 6: launch_one<6>(threads, reps, out, cyc); break; case 7: launch_one<7>(threads, reps, out, cyc); break;
        case 8: launch_one<8>(threads, reps, out, cyc); break; case 9: launch_one<9>(threads, reps, out, cyc); break;
        case 10: launch_one<10>(threads, reps, out, cyc); break; case 11: launch_one<11>(threads, reps, out, cyc); break;
        case 12: launch_one<12>(threads, reps, out, cyc); break; case 13: launch_one<13>(threads, reps, out, cyc); break;
        case 14: launch_one<14>(threads, reps, out, cyc); break; default: launch_one<15>(threads, reps, out, cyc); break;
    }
}

int main()
{
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 512 * sizeof(double)); (void)hipMalloc(&cyc, sizeof(unsigned long long));
    static const int per[P_COUNT] = {REP, REP, REP, REP, REP, REP, REP, REP, REP, REP, REP, REP / 4, REP / 4, REP / 4, REP / 4, REP};
    printf("%-95s %12s %12s\n", "pattern (cycles per pattern, wave 0; slope between 8 and 40 trips)", "1 wave/SIMD", "2 waves/SIMD");
    for (int w = 0; w < P_COUNT; ++w) {
        double res[2];
        for (int m = 0; m < 2; ++m) {
            unsigned long long t[2];
            for (int k = 0; k < 2; ++k) {
                unsigned long long best = ~0ull, h;
                for (int r = 0; r < 5; ++r) {
                    launch(w, m == 0 ? 64 : 512, k == 0 ? 8 : 40, out, cyc);
                    (void)hipDeviceSynchronize();
                    (void)hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
                    if (h < best) best = h;
                }
                t[k] = best;
            }
            res[m] = (double)(t[1] - t[0]) / (32.0 * per[w]);
        }
        printf("%-95s %12.1f %12.1f\n", NAMES[w], res[0], res[1]);
    }
    return 0;
}
