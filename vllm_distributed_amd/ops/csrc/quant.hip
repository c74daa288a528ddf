// fp8 (OCP e4m3fn, the only flavour gfx950 MFMA takes) quantisation kernels.
// Reference ops: csrc/quantization/fp8/common.cu (static / dynamic per-tensor /
// dynamic per-token), csrc/cache_kernels.cu convert_fp8.
#include "common.h"

// scale = max(|x|) / 448, kept in *scale by an atomic max on the float bits (x >= 0).
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* __restrict__ in, long n8, float* __restrict__ scale) {
  using V8 = typename Vec<T>::v8;
  __shared__ float red[16];
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const V8 a = ld16<V8>(in + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf((float)a[j]));
  }
  m = block_max(m, red);
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned int*>(scale), __float_as_uint(m / 448.f));
}

template <typename T>
__global__ __launch_bounds__(256) void quant_static_kernel(uint8_t* __restrict__ out, const T* __restrict__ in,
                                                           const float* __restrict__ scale, long n8) {
  using V8 = typename Vec<T>::v8;
  const float inv = 1.f / fmaxf(*scale, 1e-12f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const V8 a = ld16<V8>(in + i * 8);
    u32x2 pk;
    pk[0] = pack_fp8x4((float)a[0] * inv, (float)a[1] * inv, (float)a[2] * inv, (float)a[3] * inv);
    pk[1] = pack_fp8x4((float)a[4] * inv, (float)a[5] * inv, (float)a[6] * inv, (float)a[7] * inv);
    *reinterpret_cast<u32x2*>(out + i * 8) = pk;
  }
}

// one workgroup per token row: scale[row] = max(|row|)/448 (optionally capped), then quantise
template <typename T, int MAXV>
__global__ __launch_bounds__(1024) void quant_per_token_kernel(uint8_t* __restrict__ out, float* __restrict__ scales,
                                                               const T* __restrict__ in, const float* scale_ub,
                                                               int H, long in_stride, long out_stride) {
  using V8 = typename Vec<T>::v8;
  __shared__ float red[16];
  const long row = blockIdx.x;
  const int nvec = H >> 3;
  V8 x[MAXV];
  float m = 0.f;
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int i = threadIdx.x + v * blockDim.x;
    if (i < nvec) {
      x[v] = ld16<V8>(in + row * in_stride + 8 * i);
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf((float)x[v][j]));
    }
  }
  m = block_max(m, red);
  if (scale_ub) m = fminf(m, *scale_ub);
  const float sc = fmaxf(m / 448.f, 1.f / (448.f * 512.f));
  if (threadIdx.x == 0) scales[row] = sc;
  const float inv = 1.f / sc;
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int i = threadIdx.x + v * blockDim.x;
    if (i < nvec) {
      u32x2 pk;
      pk[0] = pack_fp8x4((float)x[v][0] * inv, (float)x[v][1] * inv, (float)x[v][2] * inv, (float)x[v][3] * inv);
      pk[1] = pack_fp8x4((float)x[v][4] * inv, (float)x[v][5] * inv, (float)x[v][6] * inv, (float)x[v][7] * inv);
      *reinterpret_cast<u32x2*>(out + row * out_stride + 8 * i) = pk;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dequant_kernel(T* __restrict__ out, const uint8_t* __restrict__ in,
                                                      const float* __restrict__ scale, long n8) {
  using V8 = typename Vec<T>::v8;
  const float s = *scale;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const u32x2 pk = *reinterpret_cast<const u32x2*>(in + i * 8);
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t w = pk[j >> 2];
      float f;
      switch (j & 3) {
        case 0: f = __builtin_amdgcn_cvt_f32_fp8(w, 0); break;
        case 1: f = __builtin_amdgcn_cvt_f32_fp8(w, 1); break;
        case 2: f = __builtin_amdgcn_cvt_f32_fp8(w, 2); break;
        default: f = __builtin_amdgcn_cvt_f32_fp8(w, 3); break;
      }
      o[j] = (T)(f * s);
    }
    st16<V8>(out + i * 8, o);
  }
}

static inline int q_grid(long n8) {
  long b = (n8 + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

// mode 0: static (scale given); mode 1: dynamic per-tensor (scale written; must be zeroed by the caller)
VDA_EXPORT int vda_scaled_fp8_quant(void* out, const void* in, float* scale, long n, int mode, int dtype,
                                    hipStream_t st) {
  if (n % 8) return -1;
  if (n == 0) return 0;
  const long n8 = n / 8;
  dim3 grid(q_grid(n8)), block(256);
  if (dtype == VDA_BF16) {
    if (mode == 1) hipLaunchKernelGGL(absmax_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)in, n8, scale);
    hipLaunchKernelGGL(quant_static_kernel<bf16_t>, grid, block, 0, st, (uint8_t*)out, (const bf16_t*)in, scale, n8);
  } else if (dtype == VDA_F16) {
    if (mode == 1) hipLaunchKernelGGL(absmax_kernel<f16_t>, grid, block, 0, st, (const f16_t*)in, n8, scale);
    hipLaunchKernelGGL(quant_static_kernel<f16_t>, grid, block, 0, st, (uint8_t*)out, (const f16_t*)in, scale, n8);
  } else return -3;
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_per_token_fp8_quant(void* out, float* scales, const void* in, const float* scale_ub, int rows,
                                       int H, long in_stride, long out_stride, int dtype, hipStream_t st) {
  if (H % 8) return -1;
  if (rows == 0) return 0;
  const int nvec = H / 8;
  int threads = ((nvec + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  const int need = (nvec + threads - 1) / threads;
  if (need > 8) return -2;
  dim3 grid(rows), block(threads);
#define L(TT, MV) hipLaunchKernelGGL((quant_per_token_kernel<TT, MV>), grid, block, 0, st, (uint8_t*)out, scales, \
                                     (const TT*)in, scale_ub, H, in_stride, out_stride)
#define BY(TT) do { if (need <= 1) L(TT, 1); else if (need <= 2) L(TT, 2); else if (need <= 4) L(TT, 4); else L(TT, 8); } while (0)
  if (dtype == VDA_BF16) BY(bf16_t); else if (dtype == VDA_F16) BY(f16_t); else return -3;
#undef BY
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

// e4m3 with one dynamic scale per `group` consecutive elements of a row (group % 8 == 0, a power of two
// times 8 up to 512): group / 8 lanes share a group, each lane 8 elements, scale = absmax / 448.  The
// activation side of block-scaled fp8 GEMMs (1 x 128 groups).  Role of the reference's Triton
// per_token_group_quant_fp8 (vllm/model_executor/layers/quantization/utils/fp8_utils.py:357).
template <typename T>
__global__ __launch_bounds__(256) void quant_per_group_kernel(uint8_t* __restrict__ out, float* __restrict__ scales,
                                                              const T* __restrict__ in, int rows, int H, int group,
                                                              long in_stride, long out_stride) {
  using V8 = typename Vec<T>::v8;
  const int lpg = group >> 3;                      // lanes per group (a power of two <= 64)
  const int gpr = H / group;                       // groups per row
  const long vec = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)rows * gpr * lpg;
  const bool live = vec < total;
  const long gidx = vec / lpg;
  const int l = (int)(vec % lpg);
  const long row = gidx / gpr;
  const int gi = (int)(gidx % gpr);
  V8 x = V8{};
  if (live) x = ld16<V8>(in + row * in_stride + (long)gi * group + 8 * l);
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf((float)x[j]));
  for (int o = lpg >> 1; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  const float scale = fmaxf(m / 448.f, 1.f / (448.f * 512.f));
  const float inv = 1.f / scale;
  if (!live) return;
  u32x2 o;
  o[0] = pack_fp8x4((float)x[0] * inv, (float)x[1] * inv, (float)x[2] * inv, (float)x[3] * inv);
  o[1] = pack_fp8x4((float)x[4] * inv, (float)x[5] * inv, (float)x[6] * inv, (float)x[7] * inv);
  *reinterpret_cast<u32x2*>(out + row * out_stride + (long)gi * group + 8 * l) = o;
  if (l == 0) scales[row * gpr + gi] = scale;
}

VDA_EXPORT int vda_per_group_fp8_quant(void* out, float* scales, const void* in, int rows, int H, int group,
                                       long in_stride, long out_stride, int dtype, hipStream_t st) {
  if (group < 8 || group > 512 || (group & (group - 1)) || H % group) return -1;
  if (rows == 0) return 0;
  const long total = (long)rows * (H / 8);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == VDA_BF16)
    hipLaunchKernelGGL(quant_per_group_kernel<bf16_t>, grid, block, 0, st, (uint8_t*)out, scales, (const bf16_t*)in,
                       rows, H, group, in_stride, out_stride);
  else if (dtype == VDA_F16)
    hipLaunchKernelGGL(quant_per_group_kernel<f16_t>, grid, block, 0, st, (uint8_t*)out, scales, (const f16_t*)in,
                       rows, H, group, in_stride, out_stride);
  else return -3;
  VDA_CHECK_LAUNCH();
  return 0;
}

// int8: one workgroup per token row.  Dynamic: scale[row] = max(|row|)/127 and the scale is written
// out; static: the given scale is used for every row.  Values are rounded to nearest and saturated.
// Role of the reference's csrc/quantization/compressed_tensors/int8_quant_kernels.cu:109-338
// (symmetric forms; the asymmetric zero-point forms are not built).
template <typename T, int MAXV>
__global__ __launch_bounds__(1024) void quant_int8_kernel(int8_t* __restrict__ out, float* __restrict__ scales,
                                                          const T* __restrict__ in, const float* static_scale,
                                                          int H, long in_stride, long out_stride) {
  using V8 = typename Vec<T>::v8;
  __shared__ float red[16];
  const long row = blockIdx.x;
  const int nvec = H >> 3;
  V8 x[MAXV];
  float m = 0.f;
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int i = threadIdx.x + v * blockDim.x;
    if (i < nvec) {
      x[v] = ld16<V8>(in + row * in_stride + 8 * i);
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf((float)x[v][j]));
    }
  }
  float sc;
  if (static_scale) {
    sc = *static_scale;
  } else {
    m = block_max(m, red);
    sc = fmaxf(m / 127.f, 1e-10f);
    if (threadIdx.x == 0) scales[row] = sc;
  }
  const float inv = 1.f / sc;
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int i = threadIdx.x + v * blockDim.x;
    if (i < nvec) {
      u32x2 pk;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int q = __float2int_rn((float)x[v][h * 4 + j] * inv);
          q = q < -128 ? -128 : (q > 127 ? 127 : q);
          w |= (uint32_t)(q & 0xFF) << (8 * j);
        }
        pk[h] = w;
      }
      *reinterpret_cast<u32x2*>(out + row * out_stride + 8 * i) = pk;
    }
  }
}

// Asymmetric int8 (zero point): q = clamp(rn(x / scale) + azp, -128, 127).  Dynamic per token:
// scale = (max - min) / 255, azp = rn(-128 - min / scale); static: one scale and one zero point.
// Reference: csrc/quantization/compressed_tensors/int8_quant_kernels.cu (the *_azp_* kernels).
template <typename T, int MAXV>
__global__ __launch_bounds__(1024) void quant_int8_azp_kernel(int8_t* __restrict__ out, float* __restrict__ scales,
                                                              int32_t* __restrict__ azps, const T* __restrict__ in,
                                                              const float* static_scale, const int32_t* static_azp,
                                                              int H, long in_stride, long out_stride) {
  using V8 = typename Vec<T>::v8;
  __shared__ float red[16];
  const long row = blockIdx.x;
  const int nvec = H >> 3;
  V8 x[MAXV];
  float mx = -INFINITY, mn = INFINITY;
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int i = threadIdx.x + v * blockDim.x;
    if (i < nvec) {
      x[v] = ld16<V8>(in + row * in_stride + 8 * i);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)x[v][j]; mx = fmaxf(mx, f); mn = fminf(mn, f); }
    }
  }
  float sc;
  int azp;
  if (static_scale) {
    sc = *static_scale;
    azp = static_azp ? *static_azp : 0;
  } else {
    mx = block_max(mx, red);
    mn = -block_max(-mn, red);
    sc = fmaxf((mx - mn) / 255.f, 1e-10f);
    azp = __float2int_rn(-128.f - mn / sc);
    if (threadIdx.x == 0) { scales[row] = sc; azps[row] = azp; }
  }
  const float inv = 1.f / sc;
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int i = threadIdx.x + v * blockDim.x;
    if (i < nvec) {
      u32x2 pk;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int q = __float2int_rn((float)x[v][h * 4 + j] * inv) + azp;
          q = q < -128 ? -128 : (q > 127 ? 127 : q);
          w |= (uint32_t)(q & 0xFF) << (8 * j);
        }
        pk[h] = w;
      }
      *reinterpret_cast<u32x2*>(out + row * out_stride + 8 * i) = pk;
    }
  }
}

VDA_EXPORT int vda_int8_quant_azp(void* out, float* scales, int32_t* azps, const void* in, const float* static_scale,
                                  const int32_t* static_azp, int rows, int H, long in_stride, long out_stride,
                                  int dtype, hipStream_t st) {
  if (H % 8) return -1;
  if (rows == 0) return 0;
  const int nvec = H / 8;
  int threads = ((nvec + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  const int need = (nvec + threads - 1) / threads;
  if (need > 8) return -2;
  dim3 grid(rows), block(threads);
#define L(TT, MV) hipLaunchKernelGGL((quant_int8_azp_kernel<TT, MV>), grid, block, 0, st, (int8_t*)out, scales, azps, \
                                     (const TT*)in, static_scale, static_azp, H, in_stride, out_stride)
#define BY(TT) do { if (need <= 1) L(TT, 1); else if (need <= 2) L(TT, 2); else if (need <= 4) L(TT, 4); else L(TT, 8); } while (0)
  if (dtype == VDA_BF16) BY(bf16_t); else if (dtype == VDA_F16) BY(f16_t); else return -3;
#undef BY
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

// out[r, c] = in[r, perm[c]] for 16-bit elements (activation reordering in front of act-order GPTQ layers).
// Reference: csrc/permute_cols.cu.  One workgroup per row; 8 output columns (16 bytes) per thread and step.
__global__ __launch_bounds__(256) void permute_cols_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ in,
                                                           const int* __restrict__ perm, int cols, long in_stride,
                                                           long out_stride) {
  const long row = blockIdx.x;
  const uint16_t* ip = in + row * in_stride;
  uint16_t* op = out + row * out_stride;
  const int nvec = cols >> 3;
  for (int i = threadIdx.x; i < nvec; i += blockDim.x) {
    u32x4 w;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const uint32_t lo = ip[perm[8 * i + 2 * h]], hi = ip[perm[8 * i + 2 * h + 1]];
      w[h] = lo | (hi << 16);
    }
    *reinterpret_cast<u32x4*>(op + 8 * i) = w;
  }
  for (int c = (nvec << 3) + threadIdx.x; c < cols; c += blockDim.x) op[c] = ip[perm[c]];
}

VDA_EXPORT int vda_permute_cols(void* out, const void* in, const int* perm, int rows, int cols, long in_stride,
                                long out_stride, hipStream_t st) {
  if (rows == 0 || cols == 0) return 0;
  if (out_stride % 8) return -1;           // 16-byte stores
  hipLaunchKernelGGL(permute_cols_kernel, dim3(rows), dim3(256), 0, st, (uint16_t*)out, (const uint16_t*)in, perm, cols,
                     in_stride, out_stride);
  VDA_CHECK_LAUNCH();
  return 0;
}

// scales: [rows] written when static_scale is null, ignored otherwise.
VDA_EXPORT int vda_int8_quant(void* out, float* scales, const void* in, const float* static_scale, int rows, int H,
                              long in_stride, long out_stride, int dtype, hipStream_t st) {
  if (H % 8) return -1;
  if (rows == 0) return 0;
  const int nvec = H / 8;
  int threads = ((nvec + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  const int need = (nvec + threads - 1) / threads;
  if (need > 8) return -2;
  dim3 grid(rows), block(threads);
#define L(TT, MV) hipLaunchKernelGGL((quant_int8_kernel<TT, MV>), grid, block, 0, st, (int8_t*)out, scales, \
                                     (const TT*)in, static_scale, H, in_stride, out_stride)
#define BY(TT) do { if (need <= 1) L(TT, 1); else if (need <= 2) L(TT, 2); else if (need <= 4) L(TT, 4); else L(TT, 8); } while (0)
  if (dtype == VDA_BF16) BY(bf16_t); else if (dtype == VDA_F16) BY(f16_t); else return -3;
#undef BY
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_fp8_dequant(void* out, const void* in, const float* scale, long n, int dtype, hipStream_t st) {
  if (n % 8) return -1;
  if (n == 0) return 0;
  const long n8 = n / 8;
  dim3 grid(q_grid(n8)), block(256);
  if (dtype == VDA_BF16) hipLaunchKernelGGL(dequant_kernel<bf16_t>, grid, block, 0, st, (bf16_t*)out, (const uint8_t*)in, scale, n8);
  else if (dtype == VDA_F16) hipLaunchKernelGGL(dequant_kernel<f16_t>, grid, block, 0, st, (f16_t*)out, (const uint8_t*)in, scale, n8);
  else return -3;
  VDA_CHECK_LAUNCH();
  return 0;
}
