// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: NACC independent accumulator tiles per wave, W waves per SIMD, all CUs busy.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_rate.hip -o tools/mfma_f64_rate ; run: tools/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double *out, int iters, long long *ticks, long long *cycles)
{
    v4f64 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = v4f64{0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = wall_clock64(), c1 = __builtin_readcyclecounter();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { *ticks = t1 - t0; *cycles = c1 - c0; }
}
int main()
{
    double *out; long long *t, *c;
    hipMalloc(&out, sizeof(double) * 256 * 1024); hipMalloc(&t, 8); hipMalloc(&c, 8);
    const int iters = 20000;
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k<8>, dim3(256), dim3(threads), 0, 0, out, iters, t, c);
            hipDeviceSynchronize();
        }
        long long ht, hc; hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        const double us = ht * 0.01, n = (double)iters * 8;
        printf("threads/block %d (waves/SIMD %d): %.1f us for %.0f MFMA per wave -> %.1f ns per MFMA per wave, %.1f s_memtime cycles per MFMA; chip rate %.1f TFLOP/s\n",
               threads, threads / 256, us, n, us * 1e3 / n, (double)hc / n, 256.0 * (threads / 64) * n * 2048 / (us * 1e-6) / 1e12);
    }
    return 0;
}
