// Wave-wide float64 reductions without the LDS crossbar: inside a row of sixteen lanes by DPP moves (quad_perm, row_half_mirror,
// row_mirror), across the four rows by v_readlane - a fixed order, the result in every lane.  (__shfl_xor on a double is two
// ds_bpermute_b32 per step, ~120 cycles each and in line with every other LDS access of the CU: ~0.6 us per reduction.)
#pragma once
#include "common.h"

namespace dgcn {

template <int CTRL>
__device__ __forceinline__ double wave_dpp_f64(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double wave_readlane_f64(double x, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}

__device__ __forceinline__ double wave_sum_f64(double x) {
    x += wave_dpp_f64<0xB1>(x);   // quad_perm [1, 0, 3, 2]
    x += wave_dpp_f64<0x4E>(x);   // quad_perm [2, 3, 0, 1]
    x += wave_dpp_f64<0x141>(x);  // row_half_mirror
    x += wave_dpp_f64<0x140>(x);  // row_mirror: every lane holds its row's sum
    return ((wave_readlane_f64(x, 0) + wave_readlane_f64(x, 16)) + wave_readlane_f64(x, 32)) + wave_readlane_f64(x, 48);
}

__device__ __forceinline__ double wave_max_f64(double x) {
    x = fmax(x, wave_dpp_f64<0xB1>(x));
    x = fmax(x, wave_dpp_f64<0x4E>(x));
    x = fmax(x, wave_dpp_f64<0x141>(x));
    x = fmax(x, wave_dpp_f64<0x140>(x));
    return fmax(fmax(wave_readlane_f64(x, 0), wave_readlane_f64(x, 16)), fmax(wave_readlane_f64(x, 32), wave_readlane_f64(x, 48)));
}

}  // namespace dgcn
