// Mixture-of-experts kernels for gfx950.
//
//   vda_moe_topk_softmax   router: softmax over experts + top-k (+ renormalise), one wave per token
//   vda_moe_align          counting sort of the (token, slot) pairs by expert -> sorted ids + offsets
//   vda_moe_gate_up        grouped skinny GEMM  inter[p, :] = act(x[t] . Wg_e^T) * (x[t] . Wu_e^T)  (act: gate_act codes)
//   vda_moe_down           grouped skinny GEMM  out2[p, :] = w[p] * (inter[p] . Wd_e^T)
//   vda_moe_sum            out[t] = sum_j out2[t*k + j]   (entries routed to non-local experts skipped)
//
// p = t * top_k + j is the flat (token, slot) index.  The grouped GEMMs stream each expert's weight
// rows straight from HBM into MFMA A operands (16 rows per workgroup, the 4 waves split K) while up
// to 64 gathered tokens ride on the MFMA column dimension -- the decode regime, where expert weights
// are the traffic.  Work distribution is read from device memory (offsets), so nothing syncs with the
// host and the whole layer is hipGraph-capturable; workgroups beyond an expert's token count exit.
// Counterpart of the reference's csrc/moe/{topk_softmax_kernels,moe_align_sum_kernels}.cu and the
// Triton fused_moe kernel (vllm/model_executor/layers/fused_moe/fused_moe.py).
#include "common.h"

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void moe_topk_softmax_kernel(float* __restrict__ topk_w, int* __restrict__ topk_id,
                                                               const T* __restrict__ logits, int num_tokens, int E,
                                                               int k, int renorm, long stride) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= num_tokens) return;
  constexpr int PER = 8;                 // up to 512 experts
  float v[PER];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = lane + 64 * i;
    v[i] = e < E ? (float)logits[(long)t * stride + e] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    v[i] = (lane + 64 * i) < E ? __expf(v[i] - mx) : -1.f;     // -1 marks "not a candidate"
    sum += fmaxf(v[i], 0.f);
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  float picked = 0.f;
  float my_w = 0.f;
  int my_id = 0;
  for (int j = 0; j < k; ++j) {
    float best = -1.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (v[i] > best) { best = v[i]; bi = lane + 64 * i; }
    }
    // wave arg-max, lowest expert index wins ties
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    // no candidate left (NaN logits, e.g. attention over an empty context in a warm-up run): still hand
    // out a valid, distinct expert id with weight 0, never an out-of-range one
    if (bi == 0x7fffffff) { bi = j; best = 0.f; }
    if ((bi & 63) == lane) v[bi >> 6] = -1.f;
    picked += best;
    if (lane == j) { my_w = best; my_id = bi; }
  }
  if (lane < k) {
    const float scale = renorm ? (picked > 0.f ? 1.f / picked : 0.f) : inv;
    topk_w[(long)t * k + lane] = my_w * scale;
    topk_id[(long)t * k + lane] = my_id;
  }
}

VDA_EXPORT int vda_moe_topk_softmax(float* topk_w, int* topk_id, const void* logits, int num_tokens, int E, int k,
                                    int renorm, long stride, int dtype, hipStream_t st) {
  if (E > 512 || k > 64 || k > E || num_tokens < 0) return -1;
  if (num_tokens == 0) return 0;
  dim3 grid((num_tokens + 3) / 4), block(256);
  if (dtype == VDA_BF16)
    hipLaunchKernelGGL(moe_topk_softmax_kernel<bf16_t>, grid, block, 0, st, topk_w, topk_id, (const bf16_t*)logits,
                       num_tokens, E, k, renorm, stride);
  else if (dtype == VDA_F16)
    hipLaunchKernelGGL(moe_topk_softmax_kernel<f16_t>, grid, block, 0, st, topk_w, topk_id, (const f16_t*)logits,
                       num_tokens, E, k, renorm, stride);
  else if (dtype == VDA_F32)
    hipLaunchKernelGGL(moe_topk_softmax_kernel<float>, grid, block, 0, st, topk_w, topk_id, (const float*)logits,
                       num_tokens, E, k, renorm, stride);
  else return -3;
  VDA_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// ids[p] in [0, E) or negative (expert lives on another rank).  One workgroup; E <= 1024.
__global__ __launch_bounds__(1024) void moe_align_kernel(int* __restrict__ sorted, int* __restrict__ offsets,
                                                         const int* __restrict__ ids, int n, int E) {
  __shared__ int cnt[1024];
  __shared__ int pos[1025];
  const int tid = threadIdx.x;
  if (tid < E) cnt[tid] = 0;
  __syncthreads();
  for (int p = tid; p < n; p += 1024) {
    const int e = ids[p];
    if (e >= 0 && e < E) atomicAdd(&cnt[e], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int e = 0; e < E; ++e) { pos[e] = acc; offsets[e] = acc; acc += cnt[e]; }
    offsets[E] = acc;
  }
  __syncthreads();
  for (int p = tid; p < n; p += 1024) {
    const int e = ids[p];
    if (e >= 0 && e < E) sorted[atomicAdd(&pos[e], 1)] = p;
  }
}

VDA_EXPORT int vda_moe_align(int* sorted, int* offsets, const int* ids, int n, int E, hipStream_t st) {
  if (E > 1024 || n < 0) return -1;
  hipLaunchKernelGGL(moe_align_kernel, dim3(1), dim3(1024), 0, st, sorted, offsets, ids, n, E);
  VDA_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// GATE_UP: rows n0..n0+15 of the gate half AND of the up half of W13[e] ([2I, K]); epilogue silu(g)*u.
// else   : rows n0..n0+15 of W2[e] ([H, K=I]); epilogue * routing weight.
template <typename T, typename WL, bool GATE_UP>
__global__ __launch_bounds__(256) void moe_grouped_kernel(
    T* __restrict__ out, const T* __restrict__ act, const void* __restrict__ w, const float* __restrict__ w_scale,
    const float* __restrict__ topk_w, const int* __restrict__ sorted, const int* __restrict__ offsets,
    int N, int K, int top_k, long act_stride, long out_stride, long w_expert_stride, int act_kind) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  const int e = blockIdx.y;
  const int beg = offsets[e] + blockIdx.z * 64, end = offsets[e + 1];
  if (beg >= end) return;
  const int cnt = min(end - beg, 64);
  const int mt_n = (cnt + 15) >> 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  // K elements of MFMA k-step s for lane group g inside a 128-wide step: permuted so that one load
  // instruction reads 64 contiguous bytes of each weight row (see moe_grouped_rows_kernel)
  // (8-bit weights keep 32 contiguous elements per lane here: measured faster in this form)
  auto koff = [&](int s_) { return WL::kElemBytes == 1 ? 32 * g + 8 * s_ : 32 * s_ + 8 * g; };
  const long wrow0 = (long)e * w_expert_stride + (long)(n0 + r) * K;
  const long wrow1 = wrow0 + (long)N * K;           // the "up" half (GATE_UP only)

  const T* xp[4];
  int pid[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int i = mt * 16 + r;
    pid[mt] = i < cnt ? sorted[beg + i] : -1;
    const int row = pid[mt] < 0 ? 0 : (GATE_UP ? pid[mt] / top_k : pid[mt]);
    xp[mt] = act + (long)row * act_stride;
  }
  f32x4 acc0[4], acc1[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) { acc0[mt] = f32x4{0, 0, 0, 0}; acc1[mt] = f32x4{0, 0, 0, 0}; }

  const int nchunk = K >> 7;
  for (int c = wave; c < nchunk; c += 4) {
    V8 wa[4], wb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      wa[s] = WL::ld(w, wrow0 + (long)c * 128 + koff(s));
      if (GATE_UP) wb[s] = WL::ld(w, wrow1 + (long)c * 128 + koff(s));
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt < mt_n) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const V8 xa = pid[mt] >= 0 ? ld16<V8>(xp[mt] + (long)c * 128 + koff(s)) : V8{};
          acc0[mt] = mfma16(wa[s], xa, acc0[mt]);
          if (GATE_UP) acc1[mt] = mfma16(wb[s], xa, acc1[mt]);
        }
      }
    }
  }

  __shared__ f32x4 red[3][2][4][64];
  if (wave > 0) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      red[wave - 1][0][mt][lane] = acc0[mt];
      if (GATE_UP) red[wave - 1][1][mt][lane] = acc1[mt];
    }
  }
  __syncthreads();
  if (wave != 0) return;
  const float ws = w_scale ? w_scale[e] : 1.f;
  // lane holds result[token pid[mt]][n0 + 4g + i]
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    if (mt < mt_n && pid[mt] >= 0) {
      f32x4 a = acc0[mt] + red[0][0][mt][lane] + red[1][0][mt][lane] + red[2][0][mt][lane];
      V4 o;
      if (GATE_UP) {
        const f32x4 b = acc1[mt] + red[0][1][mt][lane] + red[1][1][mt][lane] + red[2][1][mt][lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gte = a[i] * ws, up = b[i] * ws;
          o[i] = (T)(gate_act(gte, act_kind) * up);
        }
      } else {
        const float tw = topk_w[pid[mt]] * ws;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (T)(a[i] * tw);
      }
      *reinterpret_cast<V4*>(out + (long)pid[mt] * out_stride + n0 + 4 * g) = o;
    }
  }
}

// Row-split form of the grouped skinny GEMM (N % 64 == 0).  Each of the 4 waves owns 16 output rows
// over the whole K, so nothing is reduced across waves; the 64-token x 128-K activation tile of a step is
// copied into LDS once per workgroup (double-buffered) and read by all four waves.  In the K-split form
// above every workgroup streamed the expert's activations from L2 by itself, which at 32 tokens per
// expert was twice the weight traffic; here it is half of it, and the weight stream (the HBM-bound part)
// is prefetched one step ahead.  The K elements a lane group takes for an MFMA k-step are permuted inside a
// 128-wide step so that one load instruction reads 64 contiguous bytes of each of its 16 weight rows
// (see koff below); with four 16-byte pieces 64 bytes apart per instruction the same kernel streamed
// weights at 3 TB/s instead of 4.6-5.3 TB/s.
// MT: 16-token tiles per workgroup pass (4 or 8); an expert with more tokens takes another pass over its
// weights (grid.z), so the caller picks 8 once experts see more than ~32 tokens on average.
template <typename T, typename WL, bool GATE_UP, int MT>
__global__ __launch_bounds__(256) void moe_grouped_rows_kernel(
    T* __restrict__ out, const T* __restrict__ act, const void* __restrict__ w, const float* __restrict__ w_scale,
    const float* __restrict__ topk_w, const int* __restrict__ sorted, const int* __restrict__ offsets,
    int N, int K, int top_k, long act_stride, long out_stride, long w_expert_stride, int act_kind) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  // 256-byte rows of sixteen 16-byte pieces, piece p of row t stored at p ^ (t & 15): the lane groups that
  // ds_read_b128 serves together ({0-3,12-15} of one 16-lane quarter with {4-11} of the next) then hit
  // sixteen different pieces, and the eight contiguous lanes of a ds_write_b128 group eight different ones
  // (with plain padded rows the counters showed one extra LDS cycle per group on every operand read)
  __shared__ __attribute__((aligned(16))) T sx[2][16 * MT][128];
  const int e = blockIdx.y;
  const int beg = offsets[e] + blockIdx.z * (16 * MT), end = offsets[e + 1];
  if (beg >= end) return;                      // uniform over the workgroup
  const int cnt = min(end - beg, 16 * MT);
  const int mt_n = (cnt + 15) >> 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = (blockIdx.x * 4 + wave) * 16;
  // K elements of MFMA k-step s for lane group g inside a 128-wide step, chosen so that one load
  // instruction reads 64 contiguous bytes of each weight row: 16-bit weights 32 s + 8 g (16 bytes per
  // lane), 8-bit weights 64 (s / 2) + 16 g + 8 (s % 2) (the two halves of a lane's 16 bytes are one load).
  auto koff = [&](int s_) { return WL::kElemBytes == 1 ? 64 * (s_ >> 1) + 16 * g + 8 * (s_ & 1) : 32 * s_ + 8 * g; };
  const long wrow0 = (long)e * w_expert_stride + (long)(n0 + r) * K;
  const long wrow1 = wrow0 + (long)N * K;

  // staging role: token (tid >> 4) + 16 i, 16-byte piece tid & 15
  const int piece = tid & 15;
  const T* xrow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int t = (tid >> 4) + 16 * i;
    const int id = t < cnt ? sorted[beg + t] : -1;
    xrow[i] = id < 0 ? nullptr : act + (long)(GATE_UP ? id / top_k : id) * act_stride + piece * 8;
  }
  int pid[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int i = mt * 16 + r;
    pid[mt] = i < cnt ? sorted[beg + i] : -1;
  }
  f32x4 acc0[MT], acc1[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { acc0[mt] = f32x4{0, 0, 0, 0}; acc1[mt] = f32x4{0, 0, 0, 0}; }

  V8 wa[4], wb[4], wan[4], wbn[4], xr[MT];
  auto loadw = [&](int c, V8 (&a)[4], V8 (&b)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      a[s] = WL::ld(w, wrow0 + (long)c * 128 + koff(s));
      if (GATE_UP) b[s] = WL::ld(w, wrow1 + (long)c * 128 + koff(s));
    }
  };
  auto loadx = [&](int c) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
      xr[i] = (i < mt_n && xrow[i]) ? ld16<V8>(xrow[i] + (long)c * 128) : V8{};
  };
  auto storex = [&](int buf) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
      if (i < mt_n) *reinterpret_cast<V8*>(&sx[buf][(tid >> 4) + 16 * i][(piece ^ ((tid >> 4) & 15)) * 8]) = xr[i];
  };

  const int nchunk = K >> 7;
  loadx(0);
  loadw(0, wa, wb);
  storex(0);
  __syncthreads();
  // one K step: request the next step's activations and weights, multiply this step out of LDS buffer
  // c & 1 with the weights in (ca, cb), then park the requested activations in the other buffer.  The two
  // weight register sets swap roles from step to step (no copies).
  auto step = [&](int c, V8 (&ca)[4], V8 (&cb)[4], V8 (&na)[4], V8 (&nb)[4]) {
    const bool more = c + 1 < nchunk;
    if (more) { loadx(c + 1); loadw(c + 1, na, nb); }
    const int buf = c & 1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (mt < mt_n) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const V8 xa = *reinterpret_cast<const V8*>(&sx[buf][mt * 16 + r][((koff(s) >> 3) ^ r) * 8]);
          acc0[mt] = mfma16(ca[s], xa, acc0[mt]);
          if (GATE_UP) acc1[mt] = mfma16(cb[s], xa, acc1[mt]);
        }
      }
    }
    if (more) storex(buf ^ 1);
    __syncthreads();
  };
  for (int c = 0; c < nchunk; c += 2) {
    step(c, wa, wb, wan, wbn);
    if (c + 1 < nchunk) step(c + 1, wan, wbn, wa, wb);
  }

  const float ws = w_scale ? w_scale[e] : 1.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (mt < mt_n && pid[mt] >= 0) {
      V4 o;
      if (GATE_UP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gte = acc0[mt][i] * ws, up = acc1[mt][i] * ws;
          o[i] = (T)(gate_act(gte, act_kind) * up);
        }
      } else {
        const float tw = topk_w[pid[mt]] * ws;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (T)(acc0[mt][i] * tw);
      }
      *reinterpret_cast<V4*>(out + (long)pid[mt] * out_stride + n0 + 4 * g) = o;
    }
  }
}

// w13: [E, 2*I, K] (gate rows then up rows), w2: [E, H, I]; wdtype VDA_FP8 selects e4m3 weights with
// a per-expert fp32 scale.  max_pairs bounds the tokens of one expert (grid.z = ceil(max_pairs/64)).
// rows_form: 0 K-split kernel; 1 / 2 row-split kernel with 64 / 128 tokens per pass (when the row count
// is a multiple of 64); the caller decides from the batch shape (see ops.fused_moe).
VDA_EXPORT int vda_moe_gate_up(void* inter, const void* x, const void* w13, const float* w_scale, const int* sorted,
                               const int* offsets, int E, int I, int K, int top_k, int max_pairs, long x_stride,
                               int dtype, int wdtype, int rows_form, int act_kind, hipStream_t st) {
  if (I % 16 || K % 128 || E < 1 || max_pairs < 1) return -1;
  const bool rows = rows_form && I % 64 == 0;
  const int per = rows && rows_form == 2 ? 128 : 64;
  dim3 grid(rows ? I / 64 : I / 16, E, (max_pairs + per - 1) / per), block(256);
  const long es = (long)2 * I * K;
#define L(TT, WL) do { if (rows && per == 128) hipLaunchKernelGGL((moe_grouped_rows_kernel<TT, WL, true, 8>), grid, block, 0, st, \
        (TT*)inter, (const TT*)x, w13, w_scale, (const float*)nullptr, sorted, offsets, I, K, top_k, x_stride, (long)I, es, act_kind); \
    else if (rows) hipLaunchKernelGGL((moe_grouped_rows_kernel<TT, WL, true, 4>), grid, block, 0, st, (TT*)inter, \
        (const TT*)x, w13, w_scale, (const float*)nullptr, sorted, offsets, I, K, top_k, x_stride, (long)I, es, act_kind); \
    else hipLaunchKernelGGL((moe_grouped_kernel<TT, WL, true>), grid, block, 0, st, (TT*)inter, (const TT*)x, \
        w13, w_scale, (const float*)nullptr, sorted, offsets, I, K, top_k, x_stride, (long)I, es, act_kind); } while (0)
  if (dtype == VDA_BF16) { if (wdtype == VDA_FP8) L(bf16_t, WLoadFp8<bf16_t>); else L(bf16_t, WLoad<bf16_t>); }
  else if (dtype == VDA_F16) { if (wdtype == VDA_FP8) L(f16_t, WLoadFp8<f16_t>); else L(f16_t, WLoad<f16_t>); }
  else return -3;
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_moe_down(void* out2, const void* inter, const void* w2, const float* w_scale, const float* topk_w,
                            const int* sorted, const int* offsets, int E, int H, int I, int max_pairs, int dtype,
                            int wdtype, int rows_form, hipStream_t st) {
  if (H % 16 || I % 128 || E < 1 || max_pairs < 1) return -1;
  const bool rows = rows_form && H % 64 == 0;
  const int per = rows && rows_form == 2 ? 128 : 64;
  dim3 grid(rows ? H / 64 : H / 16, E, (max_pairs + per - 1) / per), block(256);
  const long es = (long)H * I;
#define L(TT, WL) do { if (rows && per == 128) hipLaunchKernelGGL((moe_grouped_rows_kernel<TT, WL, false, 8>), grid, block, 0, st, \
        (TT*)out2, (const TT*)inter, w2, w_scale, topk_w, sorted, offsets, H, I, 1, (long)I, (long)H, es, 0); \
    else if (rows) hipLaunchKernelGGL((moe_grouped_rows_kernel<TT, WL, false, 4>), grid, block, 0, st, (TT*)out2, \
        (const TT*)inter, w2, w_scale, topk_w, sorted, offsets, H, I, 1, (long)I, (long)H, es, 0); \
    else hipLaunchKernelGGL((moe_grouped_kernel<TT, WL, false>), grid, block, 0, st, (TT*)out2, (const TT*)inter, \
        w2, w_scale, topk_w, sorted, offsets, H, I, 1, (long)I, (long)H, es, 0); } while (0)
  if (dtype == VDA_BF16) { if (wdtype == VDA_FP8) L(bf16_t, WLoadFp8<bf16_t>); else L(bf16_t, WLoad<bf16_t>); }
  else if (dtype == VDA_F16) { if (wdtype == VDA_FP8) L(f16_t, WLoadFp8<f16_t>); else L(f16_t, WLoad<f16_t>); }
  else return -3;
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void moe_sum_kernel(T* __restrict__ out, const T* __restrict__ out2,
                                                      const int* __restrict__ ids, int H, int k, long out_stride) {
  using V8 = typename Vec<T>::v8;
  const int t = blockIdx.x;
  for (int h = threadIdx.x * 8; h < H; h += 256 * 8) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < k; ++j) {
      if (ids[(long)t * k + j] < 0) continue;
      const V8 v = ld16<V8>(out2 + ((long)t * k + j) * H + h);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
    }
    V8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (T)acc[i];
    st16<V8>(out + (long)t * out_stride + h, o);
  }
}

VDA_EXPORT int vda_moe_sum(void* out, const void* out2, const int* ids, int num_tokens, int H, int k, long out_stride,
                           int dtype, hipStream_t st) {
  if (H % 8 || num_tokens < 0) return -1;
  if (num_tokens == 0) return 0;
  if (dtype == VDA_BF16)
    hipLaunchKernelGGL(moe_sum_kernel<bf16_t>, dim3(num_tokens), dim3(256), 0, st, (bf16_t*)out, (const bf16_t*)out2,
                       ids, H, k, out_stride);
  else if (dtype == VDA_F16)
    hipLaunchKernelGGL(moe_sum_kernel<f16_t>, dim3(num_tokens), dim3(256), 0, st, (f16_t*)out, (const f16_t*)out2, ids,
                       H, k, out_stride);
  else return -3;
  VDA_CHECK_LAUNCH();
  return 0;
}


// ---------------------------------------------------------------------------------------------
// Expert-parallel dispatch bookkeeping in ONE launch.  For every (token, slot) pair p with global expert id
// ids[p]: the rank that owns the expert (dest = id / experts_per_rank, -1 for ids outside the table: padding, a
// router fed NaNs), a stable counting sort of the pairs by destination (sorted / offsets, the form the row
// exchange packs from), and the pair's position inside its destination's region (slot: where its result comes back
// to).  One workgroup: the sort key has at most 8 values, so a pass is one ballot per value and wave; 64 K pairs
// take 64 trips.  Role of the reference's prepare step in vllm/model_executor/layers/fused_moe/layer.py:1439-1484
// and of the index math of its all2all managers (vllm/distributed/device_communicators/all2all.py:77-264).
__global__ __launch_bounds__(1024) void ep_prep_kernel(const int* __restrict__ ids, int P, int world, int epr,
                                                       int* __restrict__ ids_out, int* __restrict__ dest_out,
                                                       int* __restrict__ sorted, int* __restrict__ offsets,
                                                       int* __restrict__ slot_out) {
  constexpr int NT = 1024, NW = NT / 64, MAXR = 8;
  __shared__ int wave_cnt[NW][MAXR];
  __shared__ int base[MAXR + 1], running[MAXR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto dest_of = [&](int p) {
    if (p >= P) return -1;
    const int e = ids[p];
    return (e >= 0 && e < world * epr) ? e / epr : -1;
  };
  // pass 1: how many pairs go to each rank
  int cnt[MAXR];
#pragma unroll
  for (int d = 0; d < MAXR; ++d) cnt[d] = 0;
  for (int p = tid; p < P; p += NT) {
    const int d = dest_of(p);
#pragma unroll
    for (int q = 0; q < MAXR; ++q) cnt[q] += (d == q);
  }
#pragma unroll
  for (int d = 0; d < MAXR; ++d) {
    int v = cnt[d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) wave_cnt[wave][d] = v;
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int d = 0; d < MAXR; ++d) {
      base[d] = acc;
      if (d < world) offsets[d] = acc;
      int t = 0;
      for (int w = 0; w < NW; ++w) t += wave_cnt[w][d];
      acc += d < world ? t : 0;
      running[d] = 0;
    }
    base[MAXR] = acc;
    offsets[world] = acc;
  }
  __syncthreads();
  // pass 2: positions, chunk by chunk in pair order (stable)
  for (int c0 = 0; c0 < P; c0 += NT) {
    const int p = c0 + tid;
    const int d = dest_of(p);
    int before = 0;
#pragma unroll
    for (int q = 0; q < MAXR; ++q) {
      const unsigned long long m = __ballot(d == q);
      if (lane == 0) wave_cnt[wave][q] = __popcll(m);
      if (d == q) before = __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (p < P) {
      int pos = -1;
      if (d >= 0) {
        int wb = 0;
        for (int w = 0; w < wave; ++w) wb += wave_cnt[w][d];
        const int sl = running[d] + wb + before;
        pos = sl;
        sorted[base[d] + sl] = p;
      }
      slot_out[p] = pos;
      dest_out[p] = d;
      ids_out[p] = d >= 0 ? ids[p] : -1;
    }
    __syncthreads();
    if (tid < MAXR) {
      int t = 0;
      for (int w = 0; w < NW; ++w) t += wave_cnt[w][tid];
      running[tid] += t;
    }
    __syncthreads();
  }
}

VDA_EXPORT int vda_ep_prep(const int* ids, int P, int world, int epr, int* ids_out, int* dest_out, int* sorted,
                           int* offsets, int* slot_out, hipStream_t st) {
  if (P < 0 || world < 1 || world > 8 || epr < 1) return -1;
  hipLaunchKernelGGL(ep_prep_kernel, dim3(1), dim3(1024), 0, st, ids, P, world, epr, ids_out, dest_out, sorted, offsets,
                     slot_out);
  VDA_CHECK_LAUNCH();
  return 0;
}
