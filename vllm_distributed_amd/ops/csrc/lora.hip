// Multi-LoRA low-rank update for decode-sized batches: every token row picks its adapter slot.
//   shrink: h[t, r]  = sum_k x[t, k] * A[slot_t, r, k]          (A [slots, R, K])
//   expand: y[t, n] += sum_r h[t, r] * B[slot_t, n, r]          (B [slots, N, R])
// Rows with slot < 0 are left alone.  One wave per (token, rank) dot product in shrink (16-byte loads,
// the few distinct adapters of a batch stay L2 resident); one thread per 8 outputs in expand.
// Large (prefill) batches group tokens by adapter and use library GEMMs instead (lora/manager.py).
// Role of the reference's vllm/lora/ops/triton_ops (bgmv / sgmv shrink and expand).
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void lora_shrink_kernel(float* __restrict__ h, const T* __restrict__ x,
                                                          const T* __restrict__ A, const int* __restrict__ tok_slot,
                                                          int R, int K, long x_stride) {
  using V8 = typename Vec<T>::v8;
  const int t = blockIdx.x, lane = threadIdx.x & 63;
  const int r = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int slot = tok_slot[t];
  if (slot < 0 || r >= R) return;
  const T* a = A + ((long)slot * R + r) * K;
  const T* xr = x + (long)t * x_stride;
  float acc = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const V8 av = ld16<V8>(a + k), xv = ld16<V8>(xr + k);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += (float)av[i] * (float)xv[i];
  }
  acc = wave_sum(acc);
  if (lane == 0) h[(long)t * R + r] = acc;
}

template <typename T, int RMAX>
__global__ __launch_bounds__(256) void lora_expand_kernel(T* __restrict__ y, const float* __restrict__ h,
                                                          const T* __restrict__ B, const int* __restrict__ tok_slot,
                                                          int R, int N, long y_stride) {
  using V8 = typename Vec<T>::v8;
  const int t = blockIdx.y;
  const int n = (blockIdx.x * 256 + threadIdx.x) * 8;
  const int slot = tok_slot[t];
  if (slot < 0) return;            // block-uniform: the whole block serves one token
  __shared__ float hs[RMAX];
  for (int i = threadIdx.x; i < R; i += 256) hs[i] = h[(long)t * R + i];
  __syncthreads();
  if (n >= N) return;
  const T* b = B + ((long)slot * N + n) * R;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int r0 = 0; r0 < R; r0 += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const V8 bv = ld16<V8>(b + (long)j * R + r0);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[j] += (float)bv[i] * hs[r0 + i];
    }
  }
  T* yp = y + (long)t * y_stride + n;
  V8 yv = ld16<V8>(yp);
#pragma unroll
  for (int j = 0; j < 8; ++j) yv[j] = (T)((float)yv[j] + acc[j]);
  st16<V8>(yp, yv);
}

VDA_EXPORT int vda_lora_shrink(float* h, const void* x, const void* A, const int* tok_slot, int T, int R, int K,
                               long x_stride, int dtype, hipStream_t st) {
  if (T <= 0) return 0;
  if (K % 8 || R < 1) return -1;
  dim3 grid(T, (R + 3) / 4), block(256);
  if (dtype == VDA_BF16)
    hipLaunchKernelGGL(lora_shrink_kernel<bf16_t>, grid, block, 0, st, h, (const bf16_t*)x, (const bf16_t*)A, tok_slot,
                       R, K, x_stride);
  else if (dtype == VDA_F16)
    hipLaunchKernelGGL(lora_shrink_kernel<f16_t>, grid, block, 0, st, h, (const f16_t*)x, (const f16_t*)A, tok_slot,
                       R, K, x_stride);
  else return -3;
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_lora_expand(void* y, const float* h, const void* B, const int* tok_slot, int T, int R, int N,
                               long y_stride, int dtype, hipStream_t st) {
  if (T <= 0) return 0;
  if (N % 8 || R % 8 || R > 256) return -1;
  dim3 grid((N / 8 + 255) / 256, T), block(256);
  if (dtype == VDA_BF16)
    hipLaunchKernelGGL((lora_expand_kernel<bf16_t, 256>), grid, block, 0, st, (bf16_t*)y, h, (const bf16_t*)B, tok_slot,
                       R, N, y_stride);
  else if (dtype == VDA_F16)
    hipLaunchKernelGGL((lora_expand_kernel<f16_t, 256>), grid, block, 0, st, (f16_t*)y, h, (const f16_t*)B, tok_slot,
                       R, N, y_stride);
  else return -3;
  VDA_CHECK_LAUNCH();
  return 0;
}
