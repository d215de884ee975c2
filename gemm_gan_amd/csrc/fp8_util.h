// OCP e4m3 helpers shared by the fp8 instantiations of tlin.hip and wst.hip (device code only).
#pragma once
#include "gg_common.h"

namespace gg {
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned fp8_u32x4 __attribute__((ext_vector_type(4)));

// four floats -> four e4m3 bytes (x * sc, clamped to the finite range: the conversion does not saturate by itself)
__device__ __forceinline__ unsigned cvt4_fp8(float a, float b, float c, float d, float sc) {
    a = __builtin_amdgcn_fmed3f(a * sc, -448.f, 448.f);
    b = __builtin_amdgcn_fmed3f(b * sc, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c * sc, -448.f, 448.f);
    d = __builtin_amdgcn_fmed3f(d * sc, -448.f, 448.f);
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (unsigned)r;
}
// a lane's 32-byte fragment of the block-scaled matrix instructions (32 consecutive k)
__device__ __forceinline__ i32x8 lds_frag32(const unsigned char* q) {
    const fp8_u32x4 lo = *reinterpret_cast<const fp8_u32x4*>(q), hi = *reinterpret_cast<const fp8_u32x4*>(q + 16);
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}
__device__ __forceinline__ float exp2i(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }
}  // namespace gg
