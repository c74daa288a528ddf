// Gated and plain activations, 8 elements (16 bytes) per lane, grid-stride.
// Reference ops: csrc/activation_kernels.cu, csrc/quantization/activation_kernels.cu.
#include "common.h"

enum Act { ACT_SILU = 0, ACT_GELU = 1, ACT_GELU_TANH = 2, ACT_RELU = 3, ACT_GELU_NEW = 4,
           ACT_GELU_FAST = 5, ACT_GELU_QUICK = 6, ACT_FATRELU = 7, ACT_RELU2 = 8 };

template <int A> __device__ __forceinline__ float act(float x, float thr) {
  if (A == ACT_SILU) return x / (1.f + __expf(-x));
  if (A == ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
  if (A == ACT_GELU_TANH || A == ACT_GELU_NEW) {
    const float c = 0.7978845608028654f;
    return 0.5f * x * (1.f + tanhf(c * (x + 0.044715f * x * x * x)));
  }
  if (A == ACT_RELU) return fmaxf(x, 0.f);
  if (A == ACT_GELU_FAST) return 0.5f * x * (1.f + tanhf(x * 0.79788456f * (1.f + 0.044715f * x * x)));
  if (A == ACT_GELU_QUICK) return x / (1.f + __expf(-1.702f * x));
  if (A == ACT_FATRELU) return x > thr ? x : 0.f;
  if (A == ACT_RELU2) { const float r = fmaxf(x, 0.f); return r * r; }
  return x;
}

// out[t, :d] = act(in[t, :d]) * in[t, d:2d]   (ACT_FIRST) or in[t,:d] * act(in[t,d:2d])
// QUANT: 0 = T out, 1 = fp8 with static scale
template <typename T, int A, bool ACT_FIRST, int QUANT>
__global__ __launch_bounds__(256) void act_and_mul_kernel(void* __restrict__ out, const T* __restrict__ in,
                                                          long rows, int d, long in_stride, long out_stride,
                                                          float thr, const float* __restrict__ scale) {
  using V8 = typename Vec<T>::v8;
  const int nv = d >> 3;
  const long total = rows * nv;
  const float inv = QUANT ? 1.f / *scale : 1.f;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long t = idx / nv;
    const int i = (int)(idx % nv) * 8;
    const T* ip = in + t * in_stride;
    const V8 a = ld16<V8>(ip + i), b = ld16<V8>(ip + d + i);
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (float)a[j], g = (float)b[j];
      y[j] = ACT_FIRST ? act<A>(x, thr) * g : x * act<A>(g, thr);
    }
    if (QUANT == 0) {
      V8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (T)y[j];
      st16<V8>((T*)out + t * out_stride + i, o);
    } else {
      u32x2 pk;
      pk[0] = pack_fp8x4(y[0] * inv, y[1] * inv, y[2] * inv, y[3] * inv);
      pk[1] = pack_fp8x4(y[4] * inv, y[5] * inv, y[6] * inv, y[7] * inv);
      *reinterpret_cast<u32x2*>((uint8_t*)out + t * out_stride + i) = pk;
    }
  }
}

template <typename T, int A>
__global__ __launch_bounds__(256) void act_kernel(T* __restrict__ out, const T* __restrict__ in, long n8, float thr) {
  using V8 = typename Vec<T>::v8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n8; idx += (long)gridDim.x * blockDim.x) {
    const V8 a = ld16<V8>(in + idx * 8);
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)act<A>((float)a[j], thr);
    st16<V8>(out + idx * 8, o);
  }
}

static inline int grid_for(long work_items) {
  long b = (work_items + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

template <typename T>
static int launch_gated(void* out, const void* in, long rows, int d, long is, long os, int a, bool act_first,
                        float thr, const float* scale, hipStream_t st) {
  dim3 grid(grid_for(rows * (d / 8))), block(256);
#define L(AA, F, Q) hipLaunchKernelGGL((act_and_mul_kernel<T, AA, F, Q>), grid, block, 0, st, out, (const T*)in, \
                                       rows, d, is, os, thr, scale)
#define BYQ(AA, F) do { if (scale) L(AA, F, 1); else L(AA, F, 0); } while (0)
  if (act_first) {
    switch (a) {
      case ACT_SILU: BYQ(ACT_SILU, true); break;
      case ACT_GELU: BYQ(ACT_GELU, true); break;
      case ACT_GELU_TANH: BYQ(ACT_GELU_TANH, true); break;
      case ACT_FATRELU: BYQ(ACT_FATRELU, true); break;
      case ACT_RELU: BYQ(ACT_RELU, true); break;
      default: return -2;
    }
  } else {
    if (a != ACT_SILU) return -2;
    BYQ(ACT_SILU, false);
  }
#undef BYQ
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_act_and_mul(void* out, const void* in, long rows, int d, long in_stride, long out_stride,
                               int act_kind, int act_first, float threshold, const float* fp8_scale, int dtype,
                               hipStream_t st) {
  if (d % 8) return -1;
  if (rows == 0) return 0;
  if (dtype == VDA_BF16)
    return launch_gated<bf16_t>(out, in, rows, d, in_stride, out_stride, act_kind, act_first, threshold, fp8_scale, st);
  if (dtype == VDA_F16)
    return launch_gated<f16_t>(out, in, rows, d, in_stride, out_stride, act_kind, act_first, threshold, fp8_scale, st);
  return -3;
}

template <typename T>
static int launch_plain(void* out, const void* in, long n, int a, float thr, hipStream_t st) {
  dim3 grid(grid_for(n / 8)), block(256);
#define L(AA) hipLaunchKernelGGL((act_kernel<T, AA>), grid, block, 0, st, (T*)out, (const T*)in, n / 8, thr)
  switch (a) {
    case ACT_SILU: L(ACT_SILU); break;
    case ACT_GELU: L(ACT_GELU); break;
    case ACT_GELU_TANH: L(ACT_GELU_TANH); break;
    case ACT_RELU: L(ACT_RELU); break;
    case ACT_GELU_NEW: L(ACT_GELU_NEW); break;
    case ACT_GELU_FAST: L(ACT_GELU_FAST); break;
    case ACT_GELU_QUICK: L(ACT_GELU_QUICK); break;
    case ACT_RELU2: L(ACT_RELU2); break;
    default: return -2;
  }
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_activation(void* out, const void* in, long n, int act_kind, float threshold, int dtype,
                              hipStream_t st) {
  if (n % 8) return -1;
  if (n == 0) return 0;
  if (dtype == VDA_BF16) return launch_plain<bf16_t>(out, in, n, act_kind, threshold, st);
  if (dtype == VDA_F16) return launch_plain<f16_t>(out, in, n, act_kind, threshold, st);
  return -3;
}
