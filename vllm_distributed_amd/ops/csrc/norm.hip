// RMSNorm family: plain, fused residual-add, and the same two with fp8 output
// (static per-tensor scale, or dynamic per-token scale computed in the same pass).
// One workgroup per token row; the row stays in registers between the two
// passes so HBM is read once.  Reference ops: csrc/layernorm_kernels.cu,
// csrc/layernorm_quant_kernels.cu,
// csrc/quantization/fused_kernels/fused_layernorm_dynamic_per_token_quant.cu.
#include "common.h"

// QUANT: 0 = T out, 1 = fp8 static scale, 2 = fp8 dynamic per-token scale
template <typename T, bool ADD, int QUANT, int MAXV>
__global__ __launch_bounds__(1024) void rms_norm_kernel(
    void* __restrict__ out, const T* __restrict__ in, T* __restrict__ residual, const T* __restrict__ w,
    float eps, int H, long in_stride, long res_stride, long out_stride, float wbias,
    const float* __restrict__ static_scale, float* __restrict__ token_scale, const float* scale_ub) {
  using V8 = typename Vec<T>::v8;
  __shared__ float red[16];
  const long row = blockIdx.x;
  const int nvec = H >> 3;
  const T* ip = in + row * in_stride;
  T* rp = ADD ? residual + row * res_stride : nullptr;
  V8 x[MAXV];
  float ss = 0.f;
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int i = threadIdx.x + v * blockDim.x;
    if (i < nvec) {
      V8 a = ld16<V8>(ip + 8 * i);
      if (ADD) {
        const V8 b = ld16<V8>(rp + 8 * i);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (T)((float)a[j] + (float)b[j]);
        st16<V8>(rp + 8 * i, a);
      }
      x[v] = a;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)a[j]; ss += f * f; }
    }
  }
  ss = block_sum(ss, red);
  const float rstd = rsqrtf(ss / (float)H + eps);

  if (QUANT == 0) {
    T* op = (T*)out + row * out_stride;
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int i = threadIdx.x + v * blockDim.x;
      if (i < nvec) {
        const V8 ww = ld16<V8>(w + 8 * i);
        V8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          o[j] = (T)((float)(T)((float)x[v][j] * rstd) * ((float)ww[j] + wbias));
        st16<V8>(op + 8 * i, o);
      }
    }
  } else {
    float y[MAXV][8];
    float amax = 0.f;
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int i = threadIdx.x + v * blockDim.x;
      if (i < nvec) {
        const V8 ww = ld16<V8>(w + 8 * i);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          y[v][j] = (float)(T)((float)(T)((float)x[v][j] * rstd) * ((float)ww[j] + wbias));
          amax = fmaxf(amax, fabsf(y[v][j]));
        }
      }
    }
    float inv;
    if (QUANT == 2) {
      amax = block_max(amax, red);
      if (scale_ub) amax = fminf(amax, *scale_ub);
      const float sc = fmaxf(amax / 448.f, 1.f / (448.f * 512.f));
      if (threadIdx.x == 0) token_scale[row] = sc;
      inv = 1.f / sc;
    } else {
      inv = 1.f / *static_scale;
    }
    uint8_t* op = (uint8_t*)out + row * out_stride;
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int i = threadIdx.x + v * blockDim.x;
      if (i < nvec) {
        u32x2 pk;
        pk[0] = pack_fp8x4(y[v][0] * inv, y[v][1] * inv, y[v][2] * inv, y[v][3] * inv);
        pk[1] = pack_fp8x4(y[v][4] * inv, y[v][5] * inv, y[v][6] * inv, y[v][7] * inv);
        *reinterpret_cast<u32x2*>(op + 8 * i) = pk;
      }
    }
  }
}

template <typename T, bool ADD, int QUANT>
static int launch_norm(void* out, const void* in, void* res, const void* w, float eps, int rows, int H,
                       long in_stride, long res_stride, long out_stride, float wbias, const float* sscale,
                       float* tscale, const float* ub, hipStream_t st) {
  if (H % 8) return -1;
  const int nvec = H / 8;
  int threads = ((nvec + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  const int need = (nvec + threads - 1) / threads;
  if (need > 8) return -2;   // H up to 65536
  dim3 grid(rows), block(threads);
#define L(MV) hipLaunchKernelGGL((rms_norm_kernel<T, ADD, QUANT, MV>), grid, block, 0, st, out, (const T*)in, \
                                 (T*)res, (const T*)w, eps, H, in_stride, res_stride, out_stride, wbias, sscale, tscale, ub)
  if (need <= 1) L(1); else if (need <= 2) L(2); else if (need <= 4) L(4); else L(8);
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

// quant: 0 none, 1 static fp8, 2 dynamic per-token fp8.  residual may be null.
VDA_EXPORT int vda_rms_norm(void* out, const void* in, void* residual, const void* weight, float eps,
                            int rows, int H, long in_stride, long res_stride, long out_stride,
                            float weight_bias, int quant, const float* static_scale, float* token_scale,
                            const float* scale_ub, int dtype, hipStream_t st) {
#define GO(TT, A, Q) return launch_norm<TT, A, Q>(out, in, residual, weight, eps, rows, H, in_stride, res_stride, \
                                                  out_stride, weight_bias, static_scale, token_scale, scale_ub, st)
#define BYQ(TT, A) do { if (quant == 0) GO(TT, A, 0); else if (quant == 1) GO(TT, A, 1); else GO(TT, A, 2); } while (0)
  if (rows == 0) return 0;
  if (dtype == VDA_BF16) { if (residual) BYQ(bf16_t, true); else BYQ(bf16_t, false); }
  if (dtype == VDA_F16) { if (residual) BYQ(f16_t, true); else BYQ(f16_t, false); }
#undef BYQ
#undef GO
  return -3;
}

// LayerNorm (mean/variance, weight + bias) for OPT / GPT-2 style models.
template <typename T, int MAXV>
__global__ __launch_bounds__(1024) void layer_norm_kernel(T* __restrict__ out, const T* __restrict__ in,
                                                          const T* __restrict__ w, const T* __restrict__ b,
                                                          float eps, int H, long in_stride, long out_stride) {
  using V8 = typename Vec<T>::v8;
  __shared__ float red[16];
  const long row = blockIdx.x;
  const int nvec = H >> 3;
  const T* ip = in + row * in_stride;
  V8 x[MAXV];
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int i = threadIdx.x + v * blockDim.x;
    if (i < nvec) {
      x[v] = ld16<V8>(ip + 8 * i);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)x[v][j];
    }
  }
  const float mean = block_sum(s, red) / (float)H;
  float ss = 0.f;
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int i = threadIdx.x + v * blockDim.x;
    if (i < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)x[v][j] - mean; ss += d * d; }
    }
  }
  const float rstd = rsqrtf(block_sum(ss, red) / (float)H + eps);
  T* op = out + row * out_stride;
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int i = threadIdx.x + v * blockDim.x;
    if (i < nvec) {
      const V8 ww = ld16<V8>(w + 8 * i);
      V8 bb = V8{};
      if (b) bb = ld16<V8>(b + 8 * i);
      V8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (T)(((float)x[v][j] - mean) * rstd * (float)ww[j] + (float)bb[j]);
      st16<V8>(op + 8 * i, o);
    }
  }
}

VDA_EXPORT int vda_layer_norm(void* out, const void* in, const void* w, const void* b, float eps, int rows,
                              int H, long in_stride, long out_stride, int dtype, hipStream_t st) {
  if (H % 8) return -1;
  if (rows == 0) return 0;
  const int nvec = H / 8;
  int threads = ((nvec + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  const int need = (nvec + threads - 1) / threads;
  if (need > 8) return -2;
  dim3 grid(rows), block(threads);
#define L(TT, MV) hipLaunchKernelGGL((layer_norm_kernel<TT, MV>), grid, block, 0, st, (TT*)out, (const TT*)in, \
                                     (const TT*)w, (const TT*)b, eps, H, in_stride, out_stride)
#define BY(TT) do { if (need <= 1) L(TT, 1); else if (need <= 2) L(TT, 2); else if (need <= 4) L(TT, 4); else L(TT, 8); } while (0)
  if (dtype == VDA_BF16) BY(bf16_t); else if (dtype == VDA_F16) BY(f16_t); else return -3;
#undef BY
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}
