// Standalone probe: what read bandwidth can a streaming kernel reach on this MI355X?  (ceiling for k_sweep)
// hipcc --offload-arch=gfx950 -O3 tools/hbm_read_bw.hip -o tools/hbm_read_bw && tools/hbm_read_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v2f64 __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const v2f64 *__restrict__ x, size_t nvec, double *out)
{
    double acc = 0.0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < nvec; i += U * stride) {
        v2f64 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&x[i + u * stride]) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
    for (; i < nvec; i += stride) acc += x[i].x + x[i].y;
    if (acc == 123.456) out[0] = acc;
}
template <int U, bool NT>
void run(const v2f64 *x, size_t nvec, double *out, int blocks, const char *name)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_read<U, NT>), dim3(blocks), dim3(256), 0, 0, x, nvec, out);
    hipEventRecord(a);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_read<U, NT>), dim3(blocks), dim3(256), 0, 0, x, nvec, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-18s blocks=%5d  %.1f GB/s\n", name, blocks, nvec * 16.0 * reps / (ms * 1e-3) / 1e9);
}
int main()
{
    const size_t bytes = 4ull << 30, nvec = bytes / 16;
    v2f64 *x; double *out;
    hipMalloc(&x, bytes); hipMalloc(&out, 8);
    hipMemset(x, 1, bytes);
    for (int bpc : {2, 3, 4, 8}) {
        run<4, false>(x, nvec, out, 256 * bpc, "U=4");
        run<8, false>(x, nvec, out, 256 * bpc, "U=8");
        run<16, false>(x, nvec, out, 256 * bpc, "U=16");
        run<8, true>(x, nvec, out, 256 * bpc, "U=8 nt");
        run<16, true>(x, nvec, out, 256 * bpc, "U=16 nt");
    }
    return 0;
}
