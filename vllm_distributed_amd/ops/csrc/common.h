// Shared device helpers for the gfx950 op library.
// Wave size is 64 everywhere; 16-bit activations are bf16 or fp16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VDA_EXPORT extern "C" __attribute__((visibility("default")))

// dtype codes shared with ops/__init__.py
enum VdaDtype { VDA_BF16 = 0, VDA_F16 = 1, VDA_F32 = 2, VDA_FP8 = 3 };

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct Vec;
template <> struct Vec<bf16_t> { using v8 = bf16x8; using v4 = bf16x4; };
template <> struct Vec<f16_t> { using v8 = f16x8; using v4 = f16x4; };

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// 16-byte global load / store of 8 halfwords
template <typename V> __device__ __forceinline__ V ld16(const void* p) {
  return *reinterpret_cast<const V*>(p);
}
template <typename V> __device__ __forceinline__ void st16(void* p, V v) {
  *reinterpret_cast<V*>(p) = v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide reductions for blocks of up to 1024 threads (16 waves). `red` is
// an LDS scratch of at least 16 floats; the result is returned to every thread.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = lane < nw ? red[lane] : 0.f;
  return wave_sum(t);
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = lane < nw ? red[lane] : -INFINITY;
  return wave_max(t);
}

// OCP e4m3fn conversion (gfx950 hardware converts; saturating to +-448)
__device__ __forceinline__ uint8_t f32_to_fp8(float x) {
  x = fminf(fmaxf(x, -448.f), 448.f);
  uint32_t r = __builtin_amdgcn_cvt_pk_fp8_f32(x, x, 0, false);
  return (uint8_t)(r & 0xff);
}
__device__ __forceinline__ float fp8_to_f32(uint8_t v) {
  return __builtin_amdgcn_cvt_f32_fp8((uint32_t)v, 0);
}

__device__ __forceinline__ float clamp_fp8(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
// four floats -> four packed e4m3 bytes (element 0 in the low byte)
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
  uint32_t r = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_fp8(a), clamp_fp8(b), 0, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(clamp_fp8(c), clamp_fp8(d), r, true);
}

// Weight loaders: 8 consecutive K elements of one weight row as an MFMA operand.
template <typename T> struct WLoad {
  using V8 = typename Vec<T>::v8;
  static constexpr int kElemBytes = sizeof(T);
  static __device__ __forceinline__ V8 ld(const void* base, long elem_off) {
    return __builtin_nontemporal_load(reinterpret_cast<const V8*>(reinterpret_cast<const T*>(base) + elem_off));
  }
};
// e4m3 weights, widened in registers (W8A16): half the HBM bytes of the 16-bit path
template <typename T> struct WLoadFp8 {
  using V8 = typename Vec<T>::v8;
  static constexpr int kElemBytes = 1;
  static __device__ __forceinline__ V8 ld(const void* base, long elem_off) {
    const u32x2 raw = __builtin_nontemporal_load(
        reinterpret_cast<const u32x2*>(reinterpret_cast<const uint8_t*>(base) + elem_off));
    V8 o;
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(raw[0], false), b = __builtin_amdgcn_cvt_pk_f32_fp8(raw[0], true);
    const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8(raw[1], false), d = __builtin_amdgcn_cvt_pk_f32_fp8(raw[1], true);
    o[0] = (T)a[0]; o[1] = (T)a[1]; o[2] = (T)b[0]; o[3] = (T)b[1];
    o[4] = (T)c[0]; o[5] = (T)c[1]; o[6] = (T)d[0]; o[7] = (T)d[1];
    return o;
  }
};

// KV-cache element access: 16-bit caches hold T, fp8 caches hold e4m3 bytes (value = byte * scale).
template <typename T, bool F8> struct KVElem { using type = T; };
template <typename T> struct KVElem<T, true> { using type = uint8_t; };

// 8 consecutive cache elements -> MFMA operand
template <typename T, bool F8> struct KVLoad {
  using V8 = typename Vec<T>::v8;
  static __device__ __forceinline__ V8 ld(const T* p) { return ld16<V8>(p); }
};
template <typename T> struct KVLoad<T, true> {
  using V8 = typename Vec<T>::v8;
  static __device__ __forceinline__ V8 ld(const uint8_t* p) {
    const u32x2 raw = *reinterpret_cast<const u32x2*>(p);
    V8 o;
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(raw[0], false), b = __builtin_amdgcn_cvt_pk_f32_fp8(raw[0], true);
    const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8(raw[1], false), d = __builtin_amdgcn_cvt_pk_f32_fp8(raw[1], true);
    o[0] = (T)a[0]; o[1] = (T)a[1]; o[2] = (T)b[0]; o[3] = (T)b[1];
    o[4] = (T)c[0]; o[5] = (T)c[1]; o[6] = (T)d[0]; o[7] = (T)d[1];
    return o;
  }
};
// store 8 consecutive elements (inv = 1 / scale for fp8) and one element
template <typename T, bool F8> struct KVStore {
  using V8 = typename Vec<T>::v8;
  static __device__ __forceinline__ void st8(T* p, V8 v, float) { st16<V8>(p, v); }
  static __device__ __forceinline__ void st1(T* p, T v, float) { *p = v; }
};
template <typename T> struct KVStore<T, true> {
  using V8 = typename Vec<T>::v8;
  static __device__ __forceinline__ void st8(uint8_t* p, V8 v, float inv) {
    u32x2 o;
    o[0] = pack_fp8x4((float)v[0] * inv, (float)v[1] * inv, (float)v[2] * inv, (float)v[3] * inv);
    o[1] = pack_fp8x4((float)v[4] * inv, (float)v[5] * inv, (float)v[6] * inv, (float)v[7] * inv);
    *reinterpret_cast<u32x2*>(p) = o;
  }
  static __device__ __forceinline__ void st1(uint8_t* p, T v, float inv) { *p = f32_to_fp8((float)v * inv); }
};

// Activation of the gate half of a gated MLP inside a GEMM epilogue, chosen at run time (wave-uniform branch).
// Codes follow ops/native.py ACT: 0 silu, 1 gelu (erf), 2 gelu (tanh), 3 relu, 8 relu squared.
__device__ __forceinline__ float gate_act(float x, int kind) {
  switch (kind) {
    case 1: return 0.5f * x * (1.f + erff(x * 0.7071067811865476f));
    case 2: return 0.5f * x * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
    case 3: return fmaxf(x, 0.f);
    case 8: { const float r = fmaxf(x, 0.f); return r * r; }
    default: return x / (1.f + __expf(-x));
  }
}

#define VDA_CHECK_LAUNCH() \
  do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
