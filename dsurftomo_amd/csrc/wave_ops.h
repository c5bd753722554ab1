// Wave-wide scans and reductions of the solve kernels (fim_kernel.hip, bundle_kernel.hip) on the VALU's data-parallel primitives.
#pragma once

#include <hip/hip_runtime.h>

namespace dsa {

// Wave-wide scans and reductions on the VALU's data-parallel primitives (row shifts inside the four rows of 16
// lanes, then the two row broadcasts): six dependent VALU operations instead of six LDS permutes.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_zero(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_scan_incl(int v)
{
    v += dpp_or_zero<0x111, 0xf>(v);      // row_shr:1
    v += dpp_or_zero<0x112, 0xf>(v);      // row_shr:2
    v += dpp_or_zero<0x114, 0xf>(v);      // row_shr:4
    v += dpp_or_zero<0x118, 0xf>(v);      // row_shr:8
    v += dpp_or_zero<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int wave_last(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ unsigned wave_sum(unsigned v) { return (unsigned)wave_last(wave_scan_incl((int)v)); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or_inf(float v)
{
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0x7f800000, (int)__float_as_uint(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_min(float v)
{
    v = fminf(v, dpp_or_inf<0x111, 0xf>(v));
    v = fminf(v, dpp_or_inf<0x112, 0xf>(v));
    v = fminf(v, dpp_or_inf<0x114, 0xf>(v));
    v = fminf(v, dpp_or_inf<0x118, 0xf>(v));
    v = fminf(v, dpp_or_inf<0x142, 0xa>(v));
    v = fminf(v, dpp_or_inf<0x143, 0xc>(v));
    return __uint_as_float((unsigned)wave_last((int)__float_as_uint(v)));
}


}  // namespace dsa
