// Multi-head latent attention (DeepSeek-V2 / V3) over a latent KV cache.
//
// Per token the cache holds ONE latent row instead of per-head keys and values:
//   kc [num_blocks, BS, DK]   DK = kv_lora_rank + rope_dim: the normalised latent c_kv followed by
//                             the rotated shared key k_pe (what the query side dots against)
//   vt [num_blocks, DV, BS]   DV = kv_lora_rank: the same latent, transposed, so that the P.V MFMA
//                             reads 8 consecutive keys of one latent dimension with one 16-byte load
// With the key up-projection absorbed into the query (q_abs = q_nope . W_UK) and the value
// up-projection applied after attention (o = o_lat . W_UV), every head attends over the same latent
// rows: multi-query attention with head sizes DK / DV.  The cache is (DK + DV) elements per token,
// against 2 * heads * head_dim for expanded heads (57x smaller for DeepSeek-V3).  With an fp8 cache both copies
// hold e4m3 bytes of value / kv_scale (one scale for the latent and the rotated key): half the bytes again; the
// kernel widens them on the way into the MFMA operands and folds the scale into the logits and the output
// (reference: csrc/cache_kernels.cu:309-343 concat_and_cache_mla with kv_cache_dtype fp8).
//
// Kernel: the MFMA scheme of attention.hip (S^T = K.Q^T with keys on the MFMA row, O^T += V^T.P^T),
// 16 (token, head) query columns per workgroup, the 4 waves split the keys and merge through LDS;
// grid.z splits long contexts further (partials + reduce).  One code path serves decode and
// (chunked) prefill; per-column causal limits come from the token index.
// Role of the reference's csrc/cache_kernels.cu::concat_and_cache_mla and its MLA decode backends
// (vllm/v1/attention/backends/mla/*, vllm/attention/ops/triton_decode_attention.py).
#include "common.h"

// 2^x as the bare v_exp_f32 instruction (fexp2() adds a range check and rescale for sub-normal results)
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- cache write -----------------------------------------------------------------------------------
template <typename T, bool F8>
__global__ __launch_bounds__(256) void mla_cache_write_kernel(const T* __restrict__ kv_c, const T* __restrict__ k_pe,
                                                              typename KVElem<T, F8>::type* __restrict__ kc,
                                                              typename KVElem<T, F8>::type* __restrict__ vt,
                                                              const long* __restrict__ slot_mapping, int R, int ROPE,
                                                              int BS, long c_stride, long pe_stride, float inv_scale) {
  using CT = typename KVElem<T, F8>::type;
  const int t = blockIdx.x;
  const long slot = slot_mapping[t];
  if (slot < 0) return;
  const long blk = slot / BS;
  const int off = (int)(slot % BS);
  const int DK = R + ROPE;
  CT* krow = kc + (blk * BS + off) * (long)DK;
  for (int d = threadIdx.x; d < DK; d += 256) {
    if (d < R) {
      const T v = kv_c[(long)t * c_stride + d];
      KVStore<T, F8>::st1(krow + d, v, inv_scale);
      KVStore<T, F8>::st1(vt + (blk * R + d) * (long)BS + off, v, inv_scale);
    } else {
      KVStore<T, F8>::st1(krow + d, k_pe[(long)t * pe_stride + (d - R)], inv_scale);
    }
  }
}

// kv_fp8: the cache holds e4m3 bytes of value / kv_scale
VDA_EXPORT int vda_mla_cache_write(const void* kv_c, const void* k_pe, void* kc, void* vt, const long* slot_mapping,
                                   int T, int R, int ROPE, int BS, long c_stride, long pe_stride, int dtype,
                                   int kv_fp8, float kv_scale, hipStream_t st) {
  if (T <= 0) return 0;
  dim3 grid(T), block(256);
  const float inv = kv_fp8 ? 1.f / kv_scale : 1.f;
#define W(TT, F8)                                                                                                   \
  hipLaunchKernelGGL((mla_cache_write_kernel<TT, F8>), grid, block, 0, st, (const TT*)kv_c, (const TT*)k_pe,          \
                     (typename KVElem<TT, F8>::type*)kc, (typename KVElem<TT, F8>::type*)vt, slot_mapping, R, ROPE, BS, \
                     c_stride, pe_stride, inv)
  if (dtype == VDA_BF16) { if (kv_fp8) W(bf16_t, true); else W(bf16_t, false); }
  else if (dtype == VDA_F16) { if (kv_fp8) W(f16_t, true); else W(f16_t, false); }
  else return -3;
#undef W
  VDA_CHECK_LAUNCH();
  return 0;
}

// ---- attention ---------------------------------------------------------------------------------------
struct MlaParams {
  const void* q;            // [T, H, DK]: absorbed q_nope followed by rotated q_pe
  const void* kc;
  const void* vt;
  void* out;                // [T, H, DV] latent-space output
  float* part_out;          // [T, H, nsplit, DV]
  float* part_lse;          // [T, H, nsplit] (log2 domain)
  const int* block_table;
  const int* seq_lens;
  const int* q_start_loc;
  int H, BS, max_blocks, nsplit;
  long q_stride, out_stride;
  float scale;              // softmax scale, times the cache scale for an fp8 cache
  float v_scale;            // 1, or the cache scale for an fp8 cache
};

template <typename T, int DK, int DV, bool F8>
__global__ __launch_bounds__(256) void mla_attention_kernel(MlaParams p) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  using CT = typename KVElem<T, F8>::type;
  constexpr int KS = DK / 32;        // MFMA k-steps over the query/key width
  constexpr int NT = DV / 16;        // 16-row tiles of the latent output
  constexpr int DQ = DK / 4;         // key dims owned by one lane group
  constexpr int VCH = NT >= 8 ? 8 : NT;   // V tiles in flight at once (register budget)
  static_assert(DK % 32 == 0 && DV % 16 == 0 && DQ % 8 == 0 && NT % VCH == 0, "unsupported latent sizes");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int seq = blockIdx.x;
  const int qs = p.q_start_loc[seq];
  const int q_len = p.q_start_loc[seq + 1] - qs;
  const int seq_len = p.seq_lens[seq];
  const int c0 = blockIdx.y * 16;                    // first (token, head) column of this workgroup
  if (seq_len <= 0 || c0 >= q_len * p.H) return;
  const int ctx = seq_len - q_len;
  const int BS = p.BS;
  const int* bt = p.block_table + (long)seq * p.max_blocks;
  const int nblk_valid = (seq_len + BS - 1) / BS;

  // this lane's query column
  const int c = c0 + r;
  const int ti = c / p.H, head = c % p.H;
  const bool valid = ti < q_len;
  const int limit = valid ? ctx + ti + 1 : 0;
  V8 qf[KS];
  {
    const T* qp = (const T*)p.q + (long)(qs + (valid ? ti : 0)) * p.q_stride + (long)head * DK + g * DQ;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = valid ? ld16<V8>(qp + 8 * s) : V8{};
  }
  int hi = limit;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) hi = max(hi, __shfl_xor(hi, o, 64));
  // key steps of 32: partition over grid.z, then deal to the 4 waves
  int step_hi = (hi + 31) / 32;
  const int per = (step_hi + p.nsplit - 1) / p.nsplit;
  const int a0 = blockIdx.z * per;
  step_hi = min(step_hi, a0 + per);

  float m = -INFINITY, l = 0.f;
  f32x4 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = f32x4{0, 0, 0, 0};
  const CT* kc = (const CT*)p.kc;
  const CT* vt = (const CT*)p.vt;
  const int krow0 = 8 * (r >> 2) + (r & 3);     // tile u, row r <-> key 8*(r>>2) + 4u + (r&3) of the step
  const float sc = p.scale * 1.4426950408889634f;

  for (int step = a0 + wave; step < step_hi; step += 4) {
    const int tb = step * 32 + 8 * g;            // first of this lane's 8 keys
    f32x4 s2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int tok = step * 32 + krow0 + 4 * u;
      const int bi = max(min(tok / BS, nblk_valid - 1), 0);
      const CT* kp = kc + ((long)bt[bi] * BS + (tok % BS)) * DK + g * DQ;
      f32x4 s = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < KS; ++k) s = mfma16(KVLoad<T, F8>::ld(kp + 8 * k), qf[k], s);
      s2[u] = s;
    }
    float sv[8] = {s2[0][0], s2[0][1], s2[0][2], s2[0][3], s2[1][0], s2[1][1], s2[1][2], s2[1][3]};
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sv[j] = (tb + j < limit) ? sv[j] * sc : -INFINITY;
      mx = fmaxf(mx, sv[j]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = fexp2(m - m_use);
    m = m_new;
    float ps = 0.f;
    V8 pb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float e = fexp2(sv[j] - m_use);
      ps += e;
      pb[j] = (T)e;
    }
    l = l * alpha + ps;
    const int bi = max(min(tb / BS, nblk_valid - 1), 0);
    const CT* vp = vt + ((long)bt[bi] * DV + r) * BS + (tb % BS);
#pragma unroll
    for (int n0 = 0; n0 < NT; n0 += VCH) {
      V8 vf[VCH];
#pragma unroll
      for (int n = 0; n < VCH; ++n) vf[n] = KVLoad<T, F8>::ld(vp + (long)(n0 + n) * 16 * BS);
#pragma unroll
      for (int n = 0; n < VCH; ++n) {
        acc[n0 + n] *= alpha;
        acc[n0 + n] = mfma16(vf[n], pb, acc[n0 + n]);
      }
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);

  // ---- merge the 4 waves into wave 0, one sender at a time (one staging buffer keeps LDS at NT KB) ----
  __shared__ f32x4 s_acc[NT][64];
  __shared__ float s_ml[2][64];
  for (int snd = 1; snd < 4; ++snd) {
    if (wave == snd) {
      s_ml[0][lane] = m;
      s_ml[1][lane] = l;
#pragma unroll
      for (int n = 0; n < NT; ++n) s_acc[n][lane] = acc[n];
    }
    __syncthreads();
    if (wave == 0) {
      const float m2 = s_ml[0][lane], l2 = s_ml[1][lane];
      const float M = fmaxf(m, m2);
      const float Mu = (M == -INFINITY) ? 0.f : M;
      const float f1 = fexp2(m - Mu), f2 = fexp2(m2 - Mu);
      l = l * f1 + l2 * f2;
      m = M;
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = acc[n] * f1 + s_acc[n][lane] * f2;
    }
    __syncthreads();
  }
  if (wave != 0 || !valid) return;
  const float inv = l > 0.f ? p.v_scale / l : 0.f;
  const long tok = qs + ti;
  if (p.nsplit == 1) {
    T* op = (T*)p.out + tok * p.out_stride + (long)head * DV + 4 * g;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      V4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (T)(acc[n][i] * inv);
      *reinterpret_cast<V4*>(op + 16 * n) = o;
    }
  } else {
    const long pi = (tok * p.H + head) * p.nsplit + blockIdx.z;
    float* po = p.part_out + pi * DV + 4 * g;
#pragma unroll
    for (int n = 0; n < NT; ++n) *reinterpret_cast<f32x4*>(po + 16 * n) = acc[n] * inv;
    if (g == 0) {
      const float Mu = (m == -INFINITY) ? 0.f : m;
      p.part_lse[pi] = l > 0.f ? Mu + log2f(l) : -INFINITY;
    }
  }
}

// merge split partials: one wave per (token, head)
template <typename T, int DV>
__global__ __launch_bounds__(64) void mla_reduce_kernel(T* out, const float* part_out, const float* part_lse, int H,
                                                        int nsplit, long out_stride) {
  const long th = blockIdx.x;
  const int lane = threadIdx.x;
  // lane s holds partition s (nsplit <= 64): wave reductions for the maximum and the weight sum, weights
  // handed out by shuffles, partitions without keys skipped
  const float mine = lane < nsplit ? part_lse[th * nsplit + lane] : -INFINITY;
  float M = mine;
#pragma unroll
  for (int o_ = 32; o_ > 0; o_ >>= 1) M = fmaxf(M, __shfl_xor(M, o_, 64));
  const float Mu = (M == -INFINITY) ? 0.f : M;
  const float w_mine = (mine == -INFINITY) ? 0.f : fexp2(mine - Mu);
  float W = w_mine;
#pragma unroll
  for (int o_ = 32; o_ > 0; o_ >>= 1) W += __shfl_xor(W, o_, 64);
  constexpr int PER = (DV + 63) / 64;
  float o[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) o[i] = 0.f;
#pragma unroll 2
  for (int s = 0; s < nsplit; ++s) {
    const float w = __shfl(w_mine, s, 64);
    if (w == 0.f) continue;
    const float* po = part_out + (th * nsplit + s) * DV;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int d = lane + 64 * i;
      if (d < DV) o[i] += w * po[d];
    }
  }
  const float inv = W > 0.f ? 1.f / W : 0.f;
  const long tok = th / H, head = th % H;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int d = lane + 64 * i;
    if (d < DV) out[tok * out_stride + head * DV + d] = (T)(o[i] * inv);
  }
}

template <typename T, int DK, int DV>
static int launch_mla(const MlaParams& p, int S, int T_tokens, int max_q, bool f8, hipStream_t st) {
  dim3 block(256), grid(S, (max_q * p.H + 15) / 16, p.nsplit);
  if (f8) hipLaunchKernelGGL((mla_attention_kernel<T, DK, DV, true>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((mla_attention_kernel<T, DK, DV, false>), grid, block, 0, st, p);
  if (p.nsplit > 1)
    hipLaunchKernelGGL((mla_reduce_kernel<T, DV>), dim3((unsigned)((long)T_tokens * p.H)), dim3(64), 0, st,
                       (T*)p.out, p.part_out, p.part_lse, p.H, p.nsplit, p.out_stride);
  return 0;
}

VDA_EXPORT int vda_mla_supported(int DK, int DV) {
  return (DK == 576 && DV == 512) || (DK == 96 && DV == 64) || (DK == 160 && DV == 128);
}

VDA_EXPORT int vda_mla_attention(void* out, const void* q, const void* kc, const void* vt, const int* block_table,
                                 const int* seq_lens, const int* q_start_loc, float* part_out, float* part_lse,
                                 int S, int T_tokens, int H, int DK, int DV, int BS, int max_blocks, int nsplit,
                                 int max_q, long q_stride, long out_stride, float scale, int dtype, int kv_fp8,
                                 float kv_scale, hipStream_t st) {
  if (S <= 0 || T_tokens <= 0) return 0;
  if (!vda_mla_supported(DK, DV) || BS % 8 || nsplit < 1 || nsplit > 64 || (nsplit > 1 && (!part_out || !part_lse)))
    return -1;      // (the reduce kernel keeps one partition per lane)
  MlaParams p;
  p.q = q; p.kc = kc; p.vt = vt; p.out = out; p.part_out = part_out; p.part_lse = part_lse;
  p.block_table = block_table; p.seq_lens = seq_lens; p.q_start_loc = q_start_loc;
  p.H = H; p.BS = BS; p.max_blocks = max_blocks; p.nsplit = nsplit;
  p.q_stride = q_stride; p.out_stride = out_stride;
  p.scale = scale * (kv_fp8 ? kv_scale : 1.f); p.v_scale = kv_fp8 ? kv_scale : 1.f;
  const bool f8 = kv_fp8 != 0;
#define BYD(TT) do { if (DK == 576) launch_mla<TT, 576, 512>(p, S, T_tokens, max_q, f8, st);      \
                     else if (DK == 96) launch_mla<TT, 96, 64>(p, S, T_tokens, max_q, f8, st);    \
                     else launch_mla<TT, 160, 128>(p, S, T_tokens, max_q, f8, st); } while (0)
  if (dtype == VDA_BF16) BYD(bf16_t);
  else if (dtype == VDA_F16) BYD(f16_t);
  else return -3;
#undef BYD
  VDA_CHECK_LAUNCH();
  return 0;
}
