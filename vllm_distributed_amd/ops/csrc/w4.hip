// 4-bit weight-only GEMM (W4A16) for GPTQ / AWQ checkpoints.
//
// Weights are repacked at load time into a layout chosen for this kernel, not for the file format:
//   qweight uint32 [N, K/8]   word j of row n holds K elements 8j..8j+7; element e (0..7) sits in
//                             nibble e/2 + 4*(e%2): `(w >> 4 i) & 0x000F000F` is the pair (2i, 2i+1) as two
//                             16-bit fields (an MFMA operand register once a float's exponent is or-ed in,
//                             see gemm_wq.hip), and `w & 0x0F0F0F0F` / `(w >> 4) & 0x0F0F0F0F` yield elements
//                             0, 4, 1, 5 / 2, 6, 3, 7 as bytes (byte -> float is one v_cvt_f32_ubyteN each)
//   scales  T [N, K/G], zeros T [N, K/G]   value = (q - zero) * scale, G a multiple of 32
// Decode (M <= 64): the skinny-GEMM scheme of gemv.hip, with a quarter of its HBM traffic: 16 weight
// rows per workgroup, each lane streams 32 consecutive K elements (one 16-byte load) per 128-wide
// chunk, widens them in registers and feeds them to MFMA as the A operand; the activation rows ride
// on the MFMA column dimension.  Larger M: `vda_w4_dequant` expands the weights to 16 bit once per
// call and the library GEMM does the rest (compute bound there).
// Role of the reference's csrc/quantization/gptq/q_gemm.cu and csrc/quantization/awq/gemm_kernels.cu.
#include "common.h"

template <typename T>
__device__ __forceinline__ typename Vec<T>::v8 w4_expand(uint32_t w, float s, float nzs) {
  const uint32_t lo = w & 0x0F0F0F0Fu, hi = (w >> 4) & 0x0F0F0F0Fu;
  typename Vec<T>::v8 o;
  o[0] = (T)fmaf((float)(lo & 0xff), s, nzs);
  o[4] = (T)fmaf((float)((lo >> 8) & 0xff), s, nzs);
  o[1] = (T)fmaf((float)((lo >> 16) & 0xff), s, nzs);
  o[5] = (T)fmaf((float)(lo >> 24), s, nzs);
  o[2] = (T)fmaf((float)(hi & 0xff), s, nzs);
  o[6] = (T)fmaf((float)((hi >> 8) & 0xff), s, nzs);
  o[3] = (T)fmaf((float)((hi >> 16) & 0xff), s, nzs);
  o[7] = (T)fmaf((float)(hi >> 24), s, nzs);
  return o;
}

template <typename T, int MT>
__global__ __launch_bounds__(256) void w4_skinny_gemm_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                             const uint32_t* __restrict__ qw,
                                                             const T* __restrict__ scales, const T* __restrict__ zeros,
                                                             const T* __restrict__ bias, int M, int N, int K, int G,
                                                             long x_stride, long y_stride) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int ngroups = K / G;
  const uint32_t* wrow = qw + (long)(n0 + r) * (K / 8) + g * 4;          // 4 words = 32 elements per chunk
  const T* srow = scales + (long)(n0 + r) * ngroups;
  const T* zrow = zeros + (long)(n0 + r) * ngroups;
  const T* xp[MT];
  bool xv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + r;
    xv[mt] = m < M;
    xp[mt] = x + (long)(xv[mt] ? m : 0) * x_stride + g * 32;
  }
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0, 0, 0, 0};

  const int nchunk = K >> 7;
  // UN chunks per iteration: their weight, scale and zero loads are all issued before the first
  // widening, so a lane keeps UN x 16 bytes of packed weights in flight (the loop is latency bound
  // otherwise: one 16-byte load per lane per trip)
  constexpr int UN = 4;
  int c = wave;
  for (; c + 4 * (UN - 1) < nchunk; c += 4 * UN) {
    u32x4 raw[UN];
    float sv[UN], nz[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int cc = c + 4 * u;
      raw[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (long)cc * 16));
      const int grp = (cc * 128 + g * 32) / G;
      sv[u] = (float)srow[grp];
      nz[u] = (float)zrow[grp];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int cc = c + 4 * u;
      const float nzs = -nz[u] * sv[u];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const V8 wa = w4_expand<T>(raw[u][q], sv[u], nzs);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const V8 xa = xv[mt] ? ld16<V8>(xp[mt] + (long)cc * 128 + 8 * q) : V8{};
          acc[mt] = mfma16(wa, xa, acc[mt]);
        }
      }
    }
  }
  for (; c < nchunk; c += 4) {
    const u32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (long)c * 16));
    const int grp = (c * 128 + g * 32) / G;
    const float s = (float)srow[grp];
    const float nzs = -(float)zrow[grp] * s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const V8 wa = w4_expand<T>(raw[q], s, nzs);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const V8 xa = xv[mt] ? ld16<V8>(xp[mt] + (long)c * 128 + 8 * q) : V8{};
        acc[mt] = mfma16(wa, xa, acc[mt]);
      }
    }
  }

  __shared__ f32x4 red[3][MT][64];
  if (wave > 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[wave - 1][mt][lane] = acc[mt];
  }
  __syncthreads();
  if (wave != 0) return;
  V4 bv = V4{};
  if (bias) bv = *reinterpret_cast<const V4*>(bias + n0 + 4 * g);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 sum = acc[mt] + red[0][mt][lane] + red[1][mt][lane] + red[2][mt][lane];
    if (xv[mt]) {
      V4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (T)(sum[i] + (float)bv[i]);
      *reinterpret_cast<V4*>(y + (long)(mt * 16 + r) * y_stride + n0 + 4 * g) = o;
    }
  }
}

VDA_EXPORT int vda_w4_skinny_gemm(void* y, const void* x, const void* qw, const void* scales, const void* zeros,
                                  const void* bias, int M, int N, int K, int G, long x_stride, long y_stride,
                                  int dtype, hipStream_t st) {
  if (M < 1 || M > 64 || N % 16 || K % 128 || G % 32 || G <= 0 || K % G) return -1;
  dim3 grid(N / 16), block(256);
  const int mt = (M + 15) / 16;
#define L(TT, MM) hipLaunchKernelGGL((w4_skinny_gemm_kernel<TT, MM>), grid, block, 0, st, (TT*)y, (const TT*)x,        \
                                     (const uint32_t*)qw, (const TT*)scales, (const TT*)zeros, (const TT*)bias, M, N, \
                                     K, G, x_stride, y_stride)
#define BY(TT) do { if (mt == 1) L(TT, 1); else if (mt == 2) L(TT, 2); else if (mt == 3) L(TT, 3); else L(TT, 4); } while (0)
  if (dtype == VDA_BF16) BY(bf16_t);
  else if (dtype == VDA_F16) BY(f16_t);
  else return -3;
#undef BY
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

// out[n, 8j .. 8j+7] = expand(qw[n, j]); one thread per packed word, 16-byte stores
template <typename T>
__global__ __launch_bounds__(256) void w4_dequant_kernel(T* __restrict__ out, const uint32_t* __restrict__ qw,
                                                         const T* __restrict__ scales, const T* __restrict__ zeros,
                                                         long nwords, int K, int G) {
  using V8 = typename Vec<T>::v8;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nwords) return;
  const int wpr = K / 8;
  const long n = i / wpr;
  const int j = (int)(i - n * wpr);
  const int grp = (j * 8) / G;
  const float s = (float)scales[n * (K / G) + grp];
  const float nzs = -(float)zeros[n * (K / G) + grp] * s;
  st16<V8>(out + i * 8, w4_expand<T>(qw[i], s, nzs));
}

// 8-bit weights (unsigned bytes [N, K]): out[n, 8j .. 8j+7] = (q - zero) * scale; one thread per 8 bytes
template <typename T>
__global__ __launch_bounds__(256) void w8_dequant_kernel(T* __restrict__ out, const uint8_t* __restrict__ qw,
                                                         const T* __restrict__ scales, const T* __restrict__ zeros,
                                                         long nchunks, int K, int G) {
  using V8 = typename Vec<T>::v8;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunks) return;
  const int cpr = K / 8;
  const long n = i / cpr;
  const int j = (int)(i - n * cpr);
  const int grp = (j * 8) / G;
  const float s = (float)scales[n * (K / G) + grp];
  const float nzs = -(float)zeros[n * (K / G) + grp] * s;
  const u32x2 raw = *reinterpret_cast<const u32x2*>(qw + i * 8);
  V8 o;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    o[4 * h + 0] = (T)fmaf((float)(raw[h] & 0xff), s, nzs);
    o[4 * h + 1] = (T)fmaf((float)((raw[h] >> 8) & 0xff), s, nzs);
    o[4 * h + 2] = (T)fmaf((float)((raw[h] >> 16) & 0xff), s, nzs);
    o[4 * h + 3] = (T)fmaf((float)(raw[h] >> 24), s, nzs);
  }
  st16<V8>(out + i * 8, o);
}

VDA_EXPORT int vda_w8_dequant(void* out, const void* qw, const void* scales, const void* zeros, int N, int K, int G,
                              int dtype, hipStream_t st) {
  if (K % 8 || G % 8 || G <= 0 || K % G) return -1;
  const long nchunks = (long)N * (K / 8);
  dim3 grid((unsigned)((nchunks + 255) / 256)), block(256);
  if (dtype == VDA_BF16)
    hipLaunchKernelGGL(w8_dequant_kernel<bf16_t>, grid, block, 0, st, (bf16_t*)out, (const uint8_t*)qw,
                       (const bf16_t*)scales, (const bf16_t*)zeros, nchunks, K, G);
  else if (dtype == VDA_F16)
    hipLaunchKernelGGL(w8_dequant_kernel<f16_t>, grid, block, 0, st, (f16_t*)out, (const uint8_t*)qw,
                       (const f16_t*)scales, (const f16_t*)zeros, nchunks, K, G);
  else return -3;
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_w4_dequant(void* out, const void* qw, const void* scales, const void* zeros, int N, int K, int G,
                              int dtype, hipStream_t st) {
  if (K % 8 || G % 8 || G <= 0 || K % G) return -1;
  const long nwords = (long)N * (K / 8);
  dim3 grid((unsigned)((nwords + 255) / 256)), block(256);
  if (dtype == VDA_BF16)
    hipLaunchKernelGGL(w4_dequant_kernel<bf16_t>, grid, block, 0, st, (bf16_t*)out, (const uint32_t*)qw,
                       (const bf16_t*)scales, (const bf16_t*)zeros, nwords, K, G);
  else if (dtype == VDA_F16)
    hipLaunchKernelGGL(w4_dequant_kernel<f16_t>, grid, block, 0, st, (f16_t*)out, (const uint32_t*)qw,
                       (const f16_t*)scales, (const f16_t*)zeros, nwords, K, G);
  else return -3;
  VDA_CHECK_LAUNCH();
  return 0;
}
