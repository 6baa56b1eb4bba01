// Transposing wave reductions (gfx950: v_permlane32_swap / v_permlane16_swap + DPP): the sums of R per-lane partials over
// the 64 lanes for R rows at once (sweep.hip K4 v2, lwplsr.hip K8).
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void jch_fold32(double &a, double &b)   // a += other half of a (lanes 0-31), b's halves folded into lanes 32-63
{
    const v2u32 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const v2u32 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ void jch_fold16(double &a, double &b)
{
    const v2u32 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const v2u32 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
template <int CTRL>
__device__ __forceinline__ double jch_dpp(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double jch_readlane(double v, int srclane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), srclane), __builtin_amdgcn_readlane(__double2loint(v), srclane));
}
// Sums of R (4 or 8) per-lane partials over the 64 lanes.  On return every lane of the 8-lane group g = lane >> 3 holds
// the total of row jch_rowsum_row(g); jch_rowsum_lane(rr) names a lane that holds row rr.
template <int R>
__device__ __forceinline__ double jch_rowsums(double (&s)[R], int lane)
{
    static_assert(R == 4 || R == 8, "R");
#pragma unroll
    for (int i = 0; i < R; i += 2) jch_fold32(s[i], s[i + 1]);          // s[i]: lanes 0-31 row i, lanes 32-63 row i+1
#pragma unroll
    for (int i = 0; i < R; i += 4) jch_fold16(s[i], s[i + 2]);          // s[i]: 16-lane rows hold rows i, i+2, i+1, i+3
    double h;
    if (R == 8) {
        const bool up = (lane & 8) != 0;
        const double w = up ? s[4] : s[0], z = up ? s[0] : s[4];
        h = w + jch_dpp<0x128>(z);                                      // row_ror:8 : lanes with bit 3 clear keep s[0], set keep s[4]
    } else {
        h = s[0] + jch_dpp<0x128>(s[0]);
    }
    h += jch_dpp<0x141>(h);                                             // row_half_mirror
    h += jch_dpp<0xB1>(h);                                              // quad_perm [1,0,3,2]
    h += jch_dpp<0x4E>(h);                                              // quad_perm [2,3,0,1]
    return h;
}
template <int R>
__device__ __forceinline__ constexpr int jch_rowsum_lane(int rr)
{
    // 16-lane row order after the two folds: rows (i, i+2, i+1, i+3); R == 8: bit 3 selects rows 4..7
    return 16 * (((rr & 3) == 1) ? 2 : ((rr & 3) == 2) ? 1 : (rr & 3)) + (R == 8 ? 8 * (rr >> 2) : 0);
}

