// Harness utilities: device-side synthetic input generator (README.md:79-94 `rand(n,p)` stand-in).
#include "jch_internal.h"

__device__ __forceinline__ double sm64_u01(uint64_t seed, uint64_t k)
{
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

__global__ __launch_bounds__(256) void k_fill_uniform(double *__restrict__ out, int64_t n, int64_t p, int64_t ld, int64_t row0,
                                                      int64_t n_total, uint64_t seed)
{
    const int64_t total = n * p;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t j = e / n, i = e - j * n;
        out[(size_t)i + (size_t)j * (size_t)ld] = sm64_u01(seed, (uint64_t)(row0 + i) + (uint64_t)j * (uint64_t)n_total);
    }
}

int32_t jch_launch_fill(jch_ctx *ctx, double *out, int64_t n, int64_t p, int64_t ld, int64_t row0, int64_t n_total,
                        uint64_t seed)
{
    int64_t nb = (n * p + 255) / 256;
    if (nb > (int64_t)ctx->cus * 16) nb = (int64_t)ctx->cus * 16;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_fill_uniform, dim3((unsigned)nb), dim3(256), 0, ctx->stream, out, n, p, ld, row0, n_total, seed);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

extern "C" int32_t jch_fill_uniform(jch_ctx *ctx, double *dev_out, int64_t n, int64_t p, int64_t ld, int64_t row0,
                                    int64_t n_total, uint64_t seed)
{
    if (!ctx) return JCH_EINVAL;
    if (!dev_out || n < 0 || p < 0 || ld < n) return jch_fail(ctx, JCH_EINVAL, "jch_fill_uniform: bad arguments");
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0 || p == 0) return JCH_OK;
    JCH_TRY(jch_launch_fill(ctx, dev_out, n, p, ld, row0, n_total, seed));
    JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JCH_OK;
}
