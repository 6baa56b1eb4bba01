// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950, closer to a real kernel: NACC accumulators, A operands from NA distinct
// registers (NA = 1: the same register every time), optionally re-read from LDS, two waves per SIMD, all CUs busy.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_rate2.hip -o tools/mfma_f64_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC, int NA, int LDS>
__global__ __launch_bounds__(512) void k(double *out, int iters, long long *ticks)
{
    __shared__ double sh[64 * 32];
    for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) sh[i] = 1.0 + i * 1e-5;
    __syncthreads();
    v4f64 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = v4f64{0.0, 0.0, 0.0, 0.0};
    const int lane = threadIdx.x & 63;
    double a[NA], b = 1.0 + threadIdx.x * 1e-4;
#pragma unroll
    for (int i = 0; i < NA; ++i) a[i] = threadIdx.x * 1e-3 + i;
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
#pragma unroll
            for (int i = 0; i < NA; ++i) a[i] = sh[((it + i) & 31) * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i % NA], b, acc[i], 0, 0, 0);
        if (LDS == 2) __syncthreads();
    }
    const long long t1 = wall_clock64();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = t1 - t0;
}
// NV independent vector instructions (KIND 0: v_cndmask_b32 pairs = a 64-bit select, 1: v_add_f64, 2: ds_read_b64) after every product
template <int NACC, int NV, int KIND>
__global__ __launch_bounds__(512) void kv(double *out, int iters, long long *ticks, int flag)
{
    __shared__ double sh[64 * 32];
    for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) sh[i] = 1.0 + i * 1e-5;
    __syncthreads();
    v4f64 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = v4f64{0.0, 0.0, 0.0, 0.0};
    const int lane = threadIdx.x & 63;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, c = 2.0 + threadIdx.x;
    double f[NV > 0 ? NV : 1];
#pragma unroll
    for (int j = 0; j < NV; ++j) f[j] = j + lane;
    const unsigned long long selm = __builtin_amdgcn_ballot_w64(flag != 0);
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (KIND == 0) { asm volatile("v_cndmask_b32_e64 %0, %2, %3, %4\n v_cndmask_b32_e64 %1, %5, %6, %4" : "+v"(((int *)&f[j])[0]), "+v"(((int *)&f[j])[1]) : "v"(((int *)&b)[0]), "v"(((int *)&c)[0]), "s"(selm), "v"(((int *)&b)[1]), "v"(((int *)&c)[1])); }
                else if (KIND == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[j]) : "v"(c));
                else asm volatile("ds_read_b64 %0, %1" : "=v"(f[j]) : "v"((int)(lane * 8 + ((it + j) & 31) * 512)));
            }
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = wall_clock64();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int j = 0; j < NV; ++j) s += f[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = t1 - t0;
}
template <int NACC, int NV, int KIND>
void runv(const char *what, double *out, long long *t, int threads, int blocks)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((kv<NACC, NV, KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters, t, 1);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double us = ms * 1e3, n = (double)iters * NACC;
    printf("%-44s blocks %4d threads %d: kernel %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", what, blocks, threads, us * 1e3 / n / (threads / 256),
           us * 1e3 / n / (threads / 256) * 2.4);
}
template <int NACC, int NA, int LDS>
void run(const char *what, double *out, long long *t, int threads, int blocks)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<NACC, NA, LDS>), dim3(blocks), dim3(threads), 0, 0, out, iters, t);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    long long ht; (void)hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
    // wave 0's own clock flatters a two-waves-per-SIMD launch: the OLDER wave of a SIMD wins the matrix pipe and finishes early
    // while its partner is still running — the kernel's duration (events) is what counts
    const double us0 = ht * 0.01, us = ms * 1e3, n = (double)iters * NACC;
    printf("%-44s blocks %4d threads %d: wave 0 %.1f ns per MFMA, kernel %.1f ns per MFMA per wave (%.1f per SIMD), chip %.1f TFLOP/s\n", what, blocks,
           threads, us0 * 1e3 / n, us * 1e3 / n, us * 1e3 / n / (threads / 256), (double)blocks * (threads / 64) * n * 2048 / (us * 1e-6) / 1e12);
}
int main()
{
    double *out; long long *t;
    (void)hipMalloc(&out, sizeof(double) * 512 * 1024); (void)hipMalloc(&t, 8);
    for (int blocks : {256}) {
        run<8, 1, 0>("8 acc, one A register", out, t, 512, blocks);
        run<13, 1, 0>("13 acc, one A register", out, t, 512, blocks);
        run<13, 13, 0>("13 acc, 13 A registers", out, t, 512, blocks);
        run<13, 13, 1>("13 acc, 13 A registers re-read from LDS", out, t, 512, blocks);
        run<13, 13, 2>("13 acc, 13 A from LDS, barrier per 13", out, t, 512, blocks);
        run<13, 13, 0>("13 acc, 13 A registers, one wave per SIMD", out, t, 256, blocks);
        for (int threads : {256, 512}) {
            runv<13, 1, 0>("+ 1 64-bit select (2 v_cndmask_b32) per MFMA", out, t, threads, blocks);
            runv<13, 2, 0>("+ 2 64-bit selects per MFMA", out, t, threads, blocks);
            runv<13, 4, 0>("+ 4 64-bit selects per MFMA", out, t, threads, blocks);
            runv<13, 1, 1>("+ 1 v_add_f64 per MFMA", out, t, threads, blocks);
            runv<13, 4, 1>("+ 4 v_add_f64 per MFMA", out, t, threads, blocks);
            runv<13, 2, 2>("+ 2 ds_read_b64 per MFMA", out, t, threads, blocks);
            runv<13, 4, 2>("+ 4 ds_read_b64 per MFMA", out, t, threads, blocks);
        }
    }
    return 0;
}
