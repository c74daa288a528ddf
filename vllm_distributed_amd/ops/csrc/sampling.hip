// Sampling on fp32 logits: one workgroup per row does temperature, top-k,
// top-p, min-p and the draw itself in a single launch, with no sort and no
// softmax round trip:
//   * top-k / top-p thresholds come from a radix descent over the order-preserving
//     integer image of the logits (4 passes of 8 bits, histogram in LDS);
//   * the draw is Gumbel-argmax (argmax of z_i + g_i, g_i ~ Gumbel(0,1)), which is
//     exactly a sample from softmax(z) restricted to the kept tokens.
// Reference behaviour: vllm/v1/sample/sampler.py, ops/topk_topp_sampler.py,
// csrc/sampler.cu (apply_repetition_penalties_).
#include "common.h"

__device__ __forceinline__ uint32_t fkey(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ uint32_t mix32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)(x >> 16);
}

struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax amax2(ArgMax a, ArgMax b) {
  // larger value wins; ties go to the smaller index (matches torch.argmax on unique maxima
  // and makes the result independent of thread scheduling)
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ArgMax block_argmax(ArgMax a, float* rv, int* ri) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax b{__shfl_xor(a.v, o, 64), __shfl_xor(a.i, o, 64)};
    a = amax2(a, b);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) { rv[wid] = a.v; ri[wid] = a.i; }
  __syncthreads();
  ArgMax t = lane < nw ? ArgMax{rv[lane], ri[lane]} : ArgMax{-INFINITY, 0x7fffffff};
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax b{__shfl_xor(t.v, o, 64), __shfl_xor(t.i, o, 64)};
    t = amax2(t, b);
  }
  return t;
}

template <typename T>
__global__ __launch_bounds__(1024) void argmax_kernel(const T* __restrict__ logits, long* __restrict__ out, int V,
                                                      long stride) {
  __shared__ float rv[16];
  __shared__ int ri[16];
  const T* row = logits + (long)blockIdx.x * stride;
  ArgMax a{-INFINITY, 0x7fffffff};
  for (int i = threadIdx.x; i < V; i += blockDim.x) a = amax2(a, ArgMax{(float)row[i], i});
  a = block_argmax(a, rv, ri);
  if (threadIdx.x == 0) out[blockIdx.x] = a.i;
}

// One radix descent.  weight(i) = 1 (COUNT) or exp(z_i - zmax) for keys >= floor_key; target = a number of
// entries (COUNT) or a fraction of the total weight.  Returns the key u* with  W{key >= u*} >= target  and
// W{key > u*} < target.  Weights are accumulated as 32.32 fixed-point integers: integer addition does not
// depend on the order in which the lanes' atomics land, so the same logits always give the same
// threshold (a float histogram could flip the boundary key between runs and between ranks that must
// sample identically).
template <bool COUNT>
__device__ uint32_t radix_threshold(const float* __restrict__ row, int V, float inv_t, float zmax, float target,
                                    uint32_t floor_key, unsigned long long* hist /*[256]*/, uint32_t* s_sel,
                                    unsigned long long* s_rem) {
  uint32_t prefix = 0;
  unsigned long long remaining = COUNT ? (unsigned long long)target : 0ull;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
      const float z = row[i] * inv_t;
      const uint32_t k = fkey(z);
      if (k < floor_key) continue;
      if (pass > 0 && (k >> (shift + 8)) != prefix) continue;
      const unsigned long long w = COUNT ? 1ull : (unsigned long long)(__expf(z - zmax) * 4294967296.f);
      if (w) atomicAdd(&hist[(k >> shift) & 255u], w);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (!COUNT && pass == 0) {
        unsigned long long total = 0ull;
        for (int b = 0; b < 256; ++b) total += hist[b];
        remaining = (unsigned long long)((double)target * (double)total);
        if (remaining < 1ull) remaining = 1ull;
      }
      unsigned long long cum = 0ull;
      int chosen = -1, lowest = 0;
      for (int b = 255; b >= 0; --b) {
        const unsigned long long w = hist[b];
        if (w > 0ull) lowest = b;
        if (chosen < 0 && cum + w >= remaining && w > 0ull) { chosen = b; *s_rem = remaining - cum; }
        cum += w;
      }
      if (chosen < 0) { chosen = lowest; *s_rem = 0ull; }   // rounding: keep everything
      *s_sel = (uint32_t)chosen;
    }
    __syncthreads();
    prefix = (prefix << 8) | *s_sel;
    remaining = *s_rem;
    __syncthreads();
  }
  return prefix;
}

// logits: fp32 [S, V].  Per row: temperature (0 => greedy), top_k (0 => off), top_p (1 => off),
// min_p (0 => off), seed.  out: sampled token per row.
__global__ __launch_bounds__(1024) void sample_kernel(const float* __restrict__ logits, long* __restrict__ out,
                                                      const float* __restrict__ temperature,
                                                      const int* __restrict__ top_k, const float* __restrict__ top_p,
                                                      const float* __restrict__ min_p,
                                                      const long* __restrict__ seeds, int V, long stride) {
  __shared__ float red[16];
  __shared__ float rv[16];
  __shared__ int ri[16];
  __shared__ unsigned long long hist[256];
  __shared__ uint32_t s_sel;
  __shared__ unsigned long long s_rem;
  const int rowi = blockIdx.x;
  const float* row = logits + (long)rowi * stride;
  const float temp = temperature ? temperature[rowi] : 1.f;
  if (temp < 1e-5f) {
    ArgMax a{-INFINITY, 0x7fffffff};
    for (int i = threadIdx.x; i < V; i += blockDim.x) a = amax2(a, ArgMax{row[i], i});
    a = block_argmax(a, rv, ri);
    if (threadIdx.x == 0) out[rowi] = a.i;
    return;
  }
  const float inv_t = 1.f / temp;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < V; i += blockDim.x) mx = fmaxf(mx, row[i] * inv_t);
  const float zmax = block_max(mx, red);

  uint32_t thr = 0;   // keep keys >= thr
  const int k = top_k ? top_k[rowi] : 0;
  const float p = top_p ? top_p[rowi] : 1.f;
  const float mp = min_p ? min_p[rowi] : 0.f;
  if (k > 0 && k < V) thr = radix_threshold<true>(row, V, inv_t, zmax, (float)k, 0u, hist, &s_sel, &s_rem);
  if (p < 1.f) {     // the fraction p of the weight that survived top-k (the descent sums it in its first pass)
    const uint32_t tp = radix_threshold<false>(row, V, inv_t, zmax, p, thr, hist, &s_sel, &s_rem);
    thr = max(thr, tp);
  }
  if (mp > 0.f) thr = max(thr, fkey(zmax + __logf(mp)));

  const uint64_t seed = (uint64_t)(seeds ? seeds[rowi] : 0);
  ArgMax a{-INFINITY, 0x7fffffff};
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    const float z = row[i] * inv_t;
    if (fkey(z) < thr) continue;
    const uint32_t rnd = mix32(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)i + 0x632BE59BD9B4E019ULL);
    const float u = ((float)(rnd >> 8) + 0.5f) * (1.f / 16777216.f);     // (0, 1)
    const float gmb = -__logf(-__logf(u));
    a = amax2(a, ArgMax{z + gmb, i});
  }
  a = block_argmax(a, rv, ri);
  if (threadIdx.x == 0) out[rowi] = a.i;
}

// In place: logits masked below the same thresholds the sampler would use (-inf),
// for callers that need the filtered distribution itself (logprobs of the
// processed distribution, rejection sampling).
__global__ __launch_bounds__(1024) void topk_topp_mask_kernel(float* __restrict__ logits,
                                                              const int* __restrict__ top_k,
                                                              const float* __restrict__ top_p,
                                                              const float* __restrict__ min_p, int V, long stride) {
  __shared__ float red[16];
  __shared__ unsigned long long hist[256];
  __shared__ uint32_t s_sel;
  __shared__ unsigned long long s_rem;
  const int rowi = blockIdx.x;
  float* row = logits + (long)rowi * stride;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < V; i += blockDim.x) mx = fmaxf(mx, row[i]);
  const float zmax = block_max(mx, red);
  uint32_t thr = 0;
  const int k = top_k ? top_k[rowi] : 0;
  const float p = top_p ? top_p[rowi] : 1.f;
  const float mp = min_p ? min_p[rowi] : 0.f;
  if (k > 0 && k < V) thr = radix_threshold<true>(row, V, 1.f, zmax, (float)k, 0u, hist, &s_sel, &s_rem);
  if (p < 1.f) {
    thr = max(thr, radix_threshold<false>(row, V, 1.f, zmax, p, thr, hist, &s_sel, &s_rem));
  }
  if (mp > 0.f) thr = max(thr, fkey(zmax + __logf(mp)));
  if (thr == 0) return;
  for (int i = threadIdx.x; i < V; i += blockDim.x)
    if (fkey(row[i]) < thr) row[i] = -INFINITY;
}

// Repetition / frequency / presence penalties on fp32 logits.
// prompt_mask [S, V] (uint8), out_counts [S, V] (int32).
__global__ __launch_bounds__(256) void penalties_kernel(float* __restrict__ logits,
                                                        const uint8_t* __restrict__ prompt_mask,
                                                        const int* __restrict__ out_counts,
                                                        const float* __restrict__ rep, const float* __restrict__ freq,
                                                        const float* __restrict__ pres, int V, long stride) {
  const int rowi = blockIdx.y;
  const float r = rep[rowi], f = freq[rowi], pz = pres[rowi];
  float* row = logits + (long)rowi * stride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
    const int c = out_counts[(long)rowi * V + i];
    const bool seen = c > 0 || prompt_mask[(long)rowi * V + i];
    float x = row[i];
    if (seen) x = x > 0.f ? x / r : x * r;
    x -= f * (float)c;
    if (c > 0) x -= pz;
    row[i] = x;
  }
}

VDA_EXPORT int vda_argmax(const void* logits, long* out, int S, int V, long stride, int dtype, hipStream_t st) {
  if (S == 0) return 0;
  dim3 grid(S), block(1024);
  if (dtype == VDA_F32) hipLaunchKernelGGL(argmax_kernel<float>, grid, block, 0, st, (const float*)logits, out, V, stride);
  else if (dtype == VDA_BF16) hipLaunchKernelGGL(argmax_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)logits, out, V, stride);
  else if (dtype == VDA_F16) hipLaunchKernelGGL(argmax_kernel<f16_t>, grid, block, 0, st, (const f16_t*)logits, out, V, stride);
  else return -3;
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_sample(const float* logits, long* out, const float* temperature, const int* top_k,
                          const float* top_p, const float* min_p, const long* seeds, int S, int V, long stride,
                          hipStream_t st) {
  if (S == 0) return 0;
  hipLaunchKernelGGL(sample_kernel, dim3(S), dim3(1024), 0, st, logits, out, temperature, top_k, top_p, min_p, seeds,
                     V, stride);
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_topk_topp_mask(float* logits, const int* top_k, const float* top_p, const float* min_p, int S,
                                  int V, long stride, hipStream_t st) {
  if (S == 0) return 0;
  hipLaunchKernelGGL(topk_topp_mask_kernel, dim3(S), dim3(1024), 0, st, logits, top_k, top_p, min_p, V, stride);
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_apply_penalties(float* logits, const uint8_t* prompt_mask, const int* out_counts,
                                   const float* rep, const float* freq, const float* pres, int S, int V,
                                   long stride, hipStream_t st) {
  if (S == 0) return 0;
  int gx = (V + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(penalties_kernel, dim3(gx, S), dim3(256), 0, st, logits, prompt_mask, out_counts, rep, freq,
                     pres, V, stride);
  VDA_CHECK_LAUNCH();
  return 0;
}
