// Device-side helpers: 4-element vector access for fp32/fp64 factor rows and
// wave64 lane-group reductions built on DPP (gfx950).
#pragma once

#include <hip/hip_runtime.h>

#include "pmf_internal.h"

template <typename T>
struct Vec4 {
    T v[4];
};

// 16-byte (fp32) / 2 x 16-byte (fp64) row-chunk load and store.
__device__ __forceinline__ Vec4<float> load4(const float *p) {
    float4 t = *reinterpret_cast<const float4 *>(p);
    Vec4<float> r;
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    return r;
}
__device__ __forceinline__ Vec4<double> load4(const double *p) {
    double2 a = *reinterpret_cast<const double2 *>(p);
    double2 b = *reinterpret_cast<const double2 *>(p + 2);
    Vec4<double> r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
    return r;
}
__device__ __forceinline__ void store4(float *p, const Vec4<float> &r) {
    *reinterpret_cast<float4 *>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
}
__device__ __forceinline__ void store4(double *p, const Vec4<double> &r) {
    *reinterpret_cast<double2 *>(p) = make_double2(r.v[0], r.v[1]);
    *reinterpret_cast<double2 *>(p + 2) = make_double2(r.v[2], r.v[3]);
}
template <typename T>
__device__ __forceinline__ Vec4<T> zero4() {
    Vec4<T> r;
    r.v[0] = r.v[1] = r.v[2] = r.v[3] = (T)0;
    return r;
}

// ---- DPP moves --------------------------------------------------------------
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140.  After the two quad steps every lane of a quad holds the
// quad's sum, so the mirrors (which pair lane l with 7-l / 15-l) combine whole
// quads / half-rows.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float x) {
    int i = __builtin_bit_cast(int, x);
    i = __builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(float, i);
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
    long long q = __builtin_bit_cast(long long, x);
    int lo = (int)(q & 0xFFFFFFFFll), hi = (int)(q >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    q = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(double, q);
}

// Sum over the aligned group of LANES consecutive lanes that contains this
// lane; every lane of the group receives the total.  LANES is a power of two
// in [1, 64].  All lanes of a group must be active together.
template <int LANES, typename T>
__device__ __forceinline__ T group_sum(T x) {
    if constexpr (LANES >= 2) x += dpp_move<0xB1>(x);
    if constexpr (LANES >= 4) x += dpp_move<0x4E>(x);
    if constexpr (LANES >= 8) x += dpp_move<0x141>(x);
    if constexpr (LANES >= 16) x += dpp_move<0x140>(x);
    if constexpr (LANES >= 32) x += __shfl_xor(x, 16, 64);
    if constexpr (LANES >= 64) x += __shfl_xor(x, 32, 64);
    return x;
}

template <typename T>
__device__ __forceinline__ T vmax(T a, T b) {
    return a > b ? a : b;
}
