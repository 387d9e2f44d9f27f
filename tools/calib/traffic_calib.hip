// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on this solver's access widths (MI355X_MICROARCH.md, HBM section: "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Three kernels over a buffer far larger
// than L2 + Infinity Cache: 8 B per lane coalesced reads, 8 B per lane coalesced writes (the width of the scratch traffic and of
// most input loads), and 24 B per lane strided reads (the `ref` rows).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void read8(const double *p, double *sink, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (; i < n; i += stride) acc += p[i];
    if (acc == 1.2345e300) sink[0] = acc;
}
__global__ void write8(double *p, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (double)i;
}
__global__ void read24(const double *p, double *sink, size_t nrec)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (; i < nrec; i += stride) { const double *r = p + 3 * i; acc += r[0] + r[1] + r[2]; }
    if (acc == 1.2345e300) sink[0] = acc;
}
int main()
{
    const size_t n = (size_t)1 << 28;  // 2 GiB of doubles
    double *p, *sink;
    if (hipMalloc(&p, n * 8) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(p, 0, n * 8);
    hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, p, n);
    hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, p, sink, n);
    hipLaunchKernelGGL(read24, dim3(4096), dim3(256), 0, 0, p, sink, n / 3);
    hipDeviceSynchronize();
    printf("bytes: write8 %zu read8 %zu read24 %zu\n", n * 8, n * 8, (n / 3) * 24);
    return 0;
}
