// Pieces shared by the LDS-tiled MFMA GEMM kernels (gemm_tile.hip, gemm_decode.hip): the 16-byte
// global -> LDS copy, the LDS chunk swizzle of 128-byte rows and the MFMA step over 128 bytes of K.
#pragma once
#include "common.h"

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// 16 bytes per lane global -> LDS; the LDS side is `lds_wave_base` (wave-uniform) + 16 * lane
__device__ __forceinline__ void glds16(const uint8_t* src, uint8_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int swz(int r) { return ((r >> 1) & 7) ^ ((((r >> 2) ^ (r >> 3)) & 1) << 1); }

__device__ __forceinline__ f32x4 mma(i32x4 a0, i32x4 a1, i32x4 b0, i32x4 b1, f32x4 c, bf16_t) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(i32x4 a0, i32x4 a1, i32x4 b0, i32x4 b1, f32x4 c, f16_t) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b1), c, 0, 0, 0);
}
// e4m3 x e4m3, 128 deep, unit block scales (E8M0 byte 127 = 2^0 in every scale lane)
__device__ __forceinline__ f32x4 mma(i32x4 a0, i32x4 a1, i32x4 b0, i32x4 b1, f32x4 c, uint8_t) {
  const i32x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  const i32x8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

