// Skinny GEMM for decode: y[M, N] = x[M, K] . W[N, K]^T (+ bias), M <= 64.
// The weight matrix is streamed exactly once straight from HBM into MFMA A
// operands (16 weight rows per workgroup, 64 contiguous bytes per lane per
// 128-wide K chunk); the few activation rows ride on the MFMA column dimension,
// so one kernel covers M = 1 .. 64 without a GEMV special case.  The 4 waves of
// a workgroup split K and are summed through LDS.
// Role of the reference's csrc/rocm/skinny_gemms.cu (LLMM1 / wvSplitK).
#include "common.h"

template <typename T, int MT, typename WL>   // MT = number of 16-row activation tiles, WL = weight loader
__global__ __launch_bounds__(256) void skinny_gemm_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                          const void* __restrict__ w, const T* __restrict__ bias,
                                                          const float* __restrict__ w_scale,
                                                          int M, int N, int K, long x_stride, long w_stride,
                                                          long y_stride) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  // K elements of MFMA k-step s for lane group g inside a 128-wide step.  The MFMA sums over K in any
  // order, so the order is chosen for the memory system: one load instruction reads 64 contiguous bytes
  // of each of its 16 weight rows (16-bit weights: 32 s + 8 g; 8-bit: 64 (s / 2) + 16 g + 8 (s % 2), the two
  // halves of a lane's 16 bytes being one load) rather than four pieces 64 bytes apart.
  auto koff = [&](int s_) { return WL::kElemBytes == 1 ? 64 * (s_ >> 1) + 16 * g + 8 * (s_ & 1) : 32 * s_ + 8 * g; };
  const long wp = (long)(n0 + r) * w_stride;            // element offset into w
  const T* xp[MT];
  bool xv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + r;
    xv[mt] = m < M;
    xp[mt] = x + (long)(xv[mt] ? m : 0) * x_stride;
  }
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0, 0, 0, 0};

  const int nchunk = K >> 7;
  // two chunks per iteration keep 8 weight loads per lane in flight
  int c = wave;
  for (; c + 4 < nchunk; c += 8) {
    V8 wa[4], wb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      wa[s] = WL::ld(w, wp + (long)c * 128 + koff(s));
      wb[s] = WL::ld(w, wp + (long)(c + 4) * 128 + koff(s));
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const V8 xa = xv[mt] ? ld16<V8>(xp[mt] + (long)c * 128 + koff(s)) : V8{};
        const V8 xb = xv[mt] ? ld16<V8>(xp[mt] + (long)(c + 4) * 128 + koff(s)) : V8{};
        acc[mt] = mfma16(wa[s], xa, acc[mt]);
        acc[mt] = mfma16(wb[s], xb, acc[mt]);
      }
    }
  }
  for (; c < nchunk; c += 4) {
    V8 wa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
      wa[s] = WL::ld(w, wp + (long)c * 128 + koff(s));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const V8 xa = xv[mt] ? ld16<V8>(xp[mt] + (long)c * 128 + koff(s)) : V8{};
        acc[mt] = mfma16(wa[s], xa, acc[mt]);
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
  // lane holds y[m = mt*16 + r][n0 + 4g + i]
  const float ws = w_scale ? w_scale[0] : 1.f;
  V4 bv = V4{};
  if (bias) bv = *reinterpret_cast<const V4*>(bias + n0 + 4 * g);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 s = acc[mt] + red[0][mt][lane] + red[1][mt][lane] + red[2][mt][lane];
    if (xv[mt]) {
      V4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (T)(s[i] * ws + (float)bv[i]);
      *reinterpret_cast<V4*>(y + (long)(mt * 16 + r) * y_stride + n0 + 4 * g) = o;
    }
  }
}

// wdtype VDA_FP8: w holds e4m3 bytes and w_scale (device, 1 float) dequantises the result (W8A16).
VDA_EXPORT int vda_skinny_gemm(void* y, const void* x, const void* w, const void* bias, const float* w_scale, int M,
                               int N, int K, long x_stride, long w_stride, long y_stride, int dtype, int wdtype,
                               hipStream_t st) {
  if (M < 1 || M > 64 || N % 16 || K % 128) return -1;
  dim3 grid(N / 16), block(256);
  const int mt = (M + 15) / 16;
#define L(TT, MM, WL) hipLaunchKernelGGL((skinny_gemm_kernel<TT, MM, WL>), grid, block, 0, st, (TT*)y, (const TT*)x, \
                                         w, (const TT*)bias, w_scale, M, N, K, x_stride, w_stride, y_stride)
#define BY(TT, WL) do { if (mt == 1) L(TT, 1, WL); else if (mt == 2) L(TT, 2, WL); else if (mt == 3) L(TT, 3, WL); \
                        else L(TT, 4, WL); } while (0)
  const bool f8 = wdtype == VDA_FP8;
  if (dtype == VDA_BF16) { if (f8) BY(bf16_t, WLoadFp8<bf16_t>); else BY(bf16_t, WLoad<bf16_t>); }
  else if (dtype == VDA_F16) { if (f8) BY(f16_t, WLoadFp8<f16_t>); else BY(f16_t, WLoad<f16_t>); }
  else return -3;
#undef BY
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}
