// Wide prefill attention for gfx950, head size 128: 8 waves per workgroup, 32 query columns per wave,
// 64-key tiles, 32x32x16 MFMA.
//
// Same role as the prefill side of the reference's Triton unified attention
// (vllm/attention/ops/triton_unified_attention.py) and its ROCm flash-attention backend
// (vllm/v1/attention/backends/rocm_aiter_fa.py); the structure below is this repo's own.
//
// Why a second form beside paged_prefill_lds_kernel (attention.hip): that kernel gives every wave 32
// columns of 16x16x32 tiles and a 32-key step, which costs ~230 vector-ALU instructions per 32 MFMAs
// (per-lane block-table lookups, address arithmetic for a copy step of only 32 keys, two cross-lane
// maximum exchanges per 16-column tile) -- the VALU, not the matrix pipe, sets its speed.  Here
//   * a column's 32 scores of a 32-key block sit in ONE lane pair (l, l+32) (swapped product
//     S^T = K.Q^T on the 32x32 tile), so the running maximum needs one half-swap per 64 keys;
//   * P never moves between lanes: registers 8t..8t+7 of a score block are, after conversion, the
//     B operand of the P.V product for k-step t; the matching permutation of the key order is applied
//     to V when it is copied into LDS (4-key granules regrouped), which costs nothing;
//   * one barrier and one copy step per 64 keys, block numbers fetched one tile ahead of the loads
//     that need them, loads for tile t+1 in flight during the products of tile t;
//   * workgroup id -> (kv head = id mod Hkv, query tile descending): with 8 kv heads every XCD
//     keeps one head's keys in its L2, and the longest causal tiles start first.
//
// Lane maps (l = lane, c = l & 31, h = l >> 5), 32x32x16:
//   A: row c, k = 8h + j        B: k = 8h + j, column c        C: row (r&3) + 8(r>>2) + 4h, column c
// Scores:  S[kb][r] = key 32kb + (r&3) + 8(r>>2) + 4h  of query column c.
// P.V:     k-step t of a tile (16 keys) takes registers 8(t&1)..+7 of block t>>1; element j of lane
//          half h is key 16t + 8(j>>2) + 4h + (j&3): two 4-key granules, numbers 4t + h and 4t + 2 + h.
// LDS:     K image  [key][16 pieces of 16 B], piece p of key k stored at p ^ (k & 15);
//          V image  [d][8 slots of 16 B]: slot 2t + h holds granules 4t + h and 4t + 2 + h (one
//          ds_read_b128 = one P.V operand), stored at slot ^ ((d >> 1) & 7); the two halves of a
//          16-byte piece of the cache row (granules 2p, 2p + 1) go to slots 2(p>>1) and 2(p>>1) + 1.
#include "attention.h"
#include <type_traits>

namespace {

template <typename V8> struct Mfma32;
template <> struct Mfma32<bf16x8> {
  static __device__ __forceinline__ f32x16 run(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma32<f16x8> {
  static __device__ __forceinline__ f32x16 run(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// max over the lane pair (l, l ^ 32)
__device__ __forceinline__ float pair_max(float x) {
  const unsigned u = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float x) {
  const unsigned u = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#ifndef PW_GLOAD_LATE
#define PW_GLOAD_LATE 1
#endif
constexpr int WD = 128;              // head size
constexpr int KT = 64;               // keys per tile
constexpr int K_BYTES = KT * 256;    // one K image
constexpr int V_BYTES = WD * 128;    // one V image

template <typename T, int BS, bool F8, bool GENERAL>
__global__ __launch_bounds__(512) void paged_prefill_wide_kernel(AttnParams p, int ntiles, int levels) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  using CT = typename KVElem<T, F8>::type;
  constexpr int D = WD;
  constexpr int PPB = BS / 8;          // 16-byte V pieces per d-row of one cache block
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * K_BYTES + 2 * V_BYTES];
  unsigned char* const sKb = smem;
  unsigned char* const sVb = smem + 2 * K_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c32 = lane & 31, hi = lane >> 5;
  const int G = p.Hq / p.Hkv;
  const int kvh = blockIdx.x % p.Hkv;
  int seq, q0;
  if (levels > 0) {
    // Host knows the longest query run: workgroup rank r -> (level j = r / S counted from the END of every
    // sequence, sequence r % S).  The longest causal tiles of all sequences start first, and no search.
    const int r = blockIdx.x / p.Hkv;
    seq = r % p.num_seqs;
    const int n_s = (p.q_start_loc[seq + 1] - p.q_start_loc[seq] + p.q_tile - 1) / p.q_tile;
    const int j = r / p.num_seqs;
    if (j >= n_s) return;                   // uniform over the workgroup
    q0 = (n_s - 1 - j) * p.q_tile;
  } else {
    const int tile = ntiles - 1 - (int)(blockIdx.x / p.Hkv);
    // tile -> sequence: the largest s whose first tile, q_start_loc[s] / q_tile + s (increasing in s), is not
    // behind this one; 64 candidates per load instead of one dependent load per bisection step
    seq = 0;
    for (int base = 0; base < p.num_seqs; base += 64) {
      const int s = base + lane;
      const bool ok = s < p.num_seqs && p.q_start_loc[s] / p.q_tile + s <= tile;
      const unsigned long long mask = __ballot(ok);
      if (mask == 0) break;
      seq = base + 63 - __builtin_clzll(mask);
      if (mask != ~0ull) break;
    }
    seq = __builtin_amdgcn_readfirstlane(seq);
    q0 = (tile - (p.q_start_loc[seq] / p.q_tile + seq)) * p.q_tile;
  }
  const int qs = p.q_start_loc[seq];
  const int q_len = p.q_start_loc[seq + 1] - qs;
  const int seq_len = p.seq_lens[seq];
  if (q0 >= q_len) return;                  // uniform over the workgroup
  const int ctx = seq_len - q_len;
  const int* bt = p.block_table + (long)seq * p.max_blocks;
  const int nblk_valid = (seq_len + BS - 1) / BS;
  const int qt_end = min(q_len, q0 + p.q_tile);

  // this lane's query column
  const int col = wave * 32 + c32;
  const int ti = col / G, hq = col % G;
  const int qt = q0 + ti;
  const bool valid = qt < qt_end;
  const int head = kvh * G + hq;
  const int limit = valid ? (p.causal ? ctx + qt + 1 : seq_len) : 0;
  const int lower = (valid && p.window > 0) ? max(0, limit - p.window) : 0;
  const float slope = (p.alibi && valid) ? p.alibi[head] : 0.f;
  const long col_off = valid ? (long)(qs + qt) : -1;
  V8 qf[8];
  {
    const T* qp = (const T*)p.q + (long)(qs + (valid ? qt : q0)) * p.q_stride + (long)head * D + 8 * hi;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = valid ? ld16<V8>(qp + 16 * s) : V8{};
  }
  int wave_hi = valid ? limit : 0, wave_lo = valid ? lower : 0x7fffffff;
  int full_hi = valid ? limit : 0x7fffffff, full_lo = valid ? lower : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    wave_hi = max(wave_hi, __shfl_xor(wave_hi, o, 64));
    wave_lo = min(wave_lo, __shfl_xor(wave_lo, o, 64));
    full_hi = min(full_hi, __shfl_xor(full_hi, o, 64));
    full_lo = max(full_lo, __shfl_xor(full_lo, o, 64));
  }
  // workgroup-uniform key range (all waves take part in every tile copy)
  const int wg_hi = p.causal ? ctx + qt_end : seq_len;
  const int wg_lo = p.window > 0 ? max(0, ctx + q0 + 1 - p.window) : 0;
  const int kt_lo = wg_lo / KT, kt_hi = (wg_hi + KT - 1) / KT;

  const CT* kc = (const CT*)p.k_cache;
  const CT* vc = (const CT*)p.v_cache;
  const unsigned blk_stride = (unsigned)p.Hkv * BS * D;          // elements per cache block (K and V alike)
  const unsigned k_head = (unsigned)kvh * BS * D, v_head = (unsigned)kvh * D * BS;

  // ---- copy step: 1024 K pieces and 1024 V pieces of 16 bytes per tile, two of each per thread ----
  // K piece pi: key = pi >> 4, piece = pi & 15.   V piece pi: cache block j = pi / (128 * PPB) of the tile,
  // d = (pi / PPB) & 127, piece inside the block row = pi % PPB (consecutive threads read consecutive memory).
  int k_key[2], k_dst[2], v_tok[2], v_dst[2], v_dst2[2];
  unsigned k_src[2], v_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pi = tid + 512 * i;
    const int key = pi >> 4, pc = pi & 15;
    k_key[i] = key;
    k_src[i] = k_head + (unsigned)(key % BS) * D + pc * 8;
    k_dst[i] = key * 256 + 16 * (pc ^ (key & 15));
    const int pin = pi % PPB, d = (pi / PPB) & 127, j = pi / (128 * PPB);
    v_tok[i] = j * BS + pin * 8;
    v_src[i] = v_head + (unsigned)d * BS + pin * 8;
    const int piece = v_tok[i] >> 3;                  // 16-byte piece of the 64-key row
    const int sl = 2 * (piece >> 1), f = (d >> 1) & 7;
    v_dst[i] = d * 128 + 16 * (sl ^ f) + 8 * (piece & 1);
    v_dst2[i] = d * 128 + 16 * ((sl + 1) ^ f) + 8 * (piece & 1);
  }
  int kblk[2], vblk[2];
  auto bload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      kblk[i] = bt[max(min((kt * KT + k_key[i]) / BS, nblk_valid - 1), 0)];
      vblk[i] = bt[max(min((kt * KT + v_tok[i]) / BS, nblk_valid - 1), 0)];
    }
  };
  V8 rk[2], rv[2];
  auto gload = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rk[i] = KVLoad<T, F8>::ld(kc + (size_t)(unsigned)kblk[i] * blk_stride + k_src[i]);
      rv[i] = KVLoad<T, F8>::ld(vc + (size_t)(unsigned)vblk[i] * blk_stride + v_src[i]);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<V8*>(sKb + buf * K_BYTES + k_dst[i]) = rk[i];
      V4 a, b;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = rv[i][e]; b[e] = rv[i][4 + e]; }
      *reinterpret_cast<V4*>(sVb + buf * V_BYTES + v_dst[i]) = a;
      *reinterpret_cast<V4*>(sVb + buf * V_BYTES + v_dst2[i]) = b;
    }
  };

  // ---- operand read addresses (bytes, buffer 0) ----
  int k_rd[8];       // K operand of k-step s: key c32 (+32 for the second block), piece 2s + hi
#pragma unroll
  for (int s = 0; s < 8; ++s) k_rd[s] = c32 * 256 + 16 * ((2 * s + hi) ^ (c32 & 15));
  int v_rd[4];       // V operand of k-step t: row d = c32 (+32 per d block: +4096 bytes), slot 2t + hi
#pragma unroll
  for (int t = 0; t < 4; ++t) v_rd[t] = c32 * 128 + 16 * ((2 * t + hi) ^ ((c32 >> 1) & 7));

  float m = -INFINITY, l = 0.f;       // l: this lane's half of the column's sum
  f32x16 acc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;

  const float sl2 = p.scale * 1.4426950408889634f;

  // A tile is worked in two halves of 32 keys, each with its own online-softmax update, ordered so that the
  // vector ALU and the matrix pipe have independent work inside one wave:
  //     scores(0) . scores(1) || softmax(0) . rescale . P.V(0) || softmax(1) . rescale . P.V(1)
  // (a || b: same basic block, no dependence).  MODE 0: no masking, scale folded into the exponent's fma;
  // 1: causal / window masking only; 2: soft-cap and ALiBi as well (its own kernel instance, GENERAL, so
  // that the common instance does not pay for its registers).
  f32x16 S[2];
  auto qk_half = [&](int buf, int kb) {
    const unsigned char* kb_ = sKb + buf * K_BYTES + kb * 8192;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[kb][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const V8 a = *reinterpret_cast<const V8*>(kb_ + k_rd[s]);
      S[kb] = Mfma32<V8>::run(a, qf[s], S[kb]);
    }
  };
  auto softmax_half = [&](int tb, int kb, V8 (&pb)[2], auto mode) -> float {
    constexpr int MODE = decltype(mode)::value;
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = S[kb][r];
      if constexpr (MODE >= 1) {
        const int tok = tb + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if constexpr (MODE == 2) {
          x *= p.scale;
          if (p.softcap > 0.f) x = p.softcap * tanhf(x / p.softcap);
          x += slope * (float)(tok - (limit - 1));
          x *= 1.4426950408889634f;
        }
        x = (tok < limit && tok >= lower) ? x : -INFINITY;
        S[kb][r] = x;
      }
      mx = fmaxf(mx, x);
    }
    const float mul = MODE == 2 ? 1.f : sl2;
    if constexpr (MODE != 2) mx *= sl2;          // the scale is positive
    mx = pair_max(mx);
    const float m_new = fmaxf(m, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = fexp2(m - m_use);
    m = m_new;
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float e = fexp2(fmaf(S[kb][8 * t + j], mul, -m_use));
        ps += e;
        pb[t][j] = (T)e;
      }
    l = l * alpha + ps;
    return alpha;
  };
  auto rescale = [&](float alpha) {
    if (__any(alpha != 1.f)) {          // the running maximum settles early: most tiles need no rescale
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[db][r] *= alpha;
    }
  };
  auto pv_half = [&](int buf, int kb, const V8 (&pb)[2]) {
    const unsigned char* vb_ = sVb + buf * V_BYTES;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const V8 a = *reinterpret_cast<const V8*>(vb_ + db * 4096 + v_rd[2 * kb + t]);
        acc[db] = Mfma32<V8>::run(a, pb[t], acc[db]);
      }
    }
  };
  // request for tile t+2 (and the block numbers of tile t+3)
  auto issue_next = [&](int kt) {
    if (kt + 2 < kt_hi) gload();
    if (kt + 3 < kt_hi) bload(kt + 3);
  };
  auto tile_compute = [&](int kt, int buf, auto mode) {
    constexpr int MODE = decltype(mode)::value;
    const int tb0 = kt * KT;
    V8 p0[2], p1[2];
    // second half wholly above every column's causal limit (diagonal tiles): nothing to do for it
    const bool second = MODE != 1 || tb0 + 32 < wave_hi;
    // The probabilities are pinned before each rescale branch (otherwise their exponentials sink behind
    // it, out of the block that holds the products they should overlap), and in the unmasked mode the
    // scheduler is told the interleave: one MFMA, then a slice of the softmax.
    auto pin = [](V8 (&pb)[2]) { asm volatile("" : "+v"(pb[0]), "+v"(pb[1])); };
    auto interleave = [] {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);    // VALU
        }
      }
    };
    qk_half(buf, 0);
    if constexpr (MODE == 0) __builtin_amdgcn_sched_barrier(0);
    if (second) qk_half(buf, 1);
#if PW_GLOAD_LATE
    issue_next(kt);
#endif
    const float a0 = softmax_half(tb0, 0, p0, mode);
    pin(p0);
    interleave();
    rescale(a0);
    pv_half(buf, 0, p0);
    if (second) {
      const float a1 = softmax_half(tb0 + 32, 1, p1, mode);
      pin(p1);
      interleave();
      rescale(a1);
      pv_half(buf, 1, p1);
    }
  };

  // Tile t is consumed from LDS buffer t & 1.  Tile t+1 was requested from memory one iteration earlier and
  // sits in registers: it moves into the other buffer at the START of iteration t (nobody reads that
  // buffer: every wave passed the barrier that ended iteration t-1), which frees the registers for the
  // request of tile t+2 -- every load has a whole iteration to land before anything waits for it.
  if (kt_lo < kt_hi) {
    bload(kt_lo);
    gload();
    if (kt_lo + 1 < kt_hi) bload(kt_lo + 1);
    sstore(0);
    if (kt_lo + 1 < kt_hi) gload();
    if (kt_lo + 2 < kt_hi) bload(kt_lo + 2);
  }
  __syncthreads();
  for (int kt = kt_lo, buf = 0; kt < kt_hi; ++kt, buf ^= 1) {
    const int tb0 = kt * KT;
    const bool active = tb0 < wave_hi && tb0 + KT > wave_lo;   // wave-uniform: tile inside this wave's key range
    if (kt + 1 < kt_hi) sstore(buf ^ 1);
#if !PW_GLOAD_LATE
    issue_next(kt);
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (!active) {
#if PW_GLOAD_LATE
      issue_next(kt);
#endif
    } else {
      if constexpr (GENERAL) {
        tile_compute(kt, buf, std::integral_constant<int, 2>{});
      } else {
        if (tb0 + KT <= full_hi && tb0 >= full_lo) tile_compute(kt, buf, std::integral_constant<int, 0>{});
        else tile_compute(kt, buf, std::integral_constant<int, 1>{});
      }
    }
    __syncthreads();
  }

  l = pair_sum(l);
  if (col_off < 0) return;
  const float inv = l > 0.f ? p.v_scale / l : 0.f;
  // Lane l holds d = 8k + 4hi .. +3 of its row for the sixteen groups k = 4db + rr; a half-swap per pair of
  // groups gives every lane 16 contiguous bytes, so the row leaves in 8 stores of 16 bytes, not 16 of 8.
  T* op = (T*)p.out + col_off * p.out_stride + (long)head * D + 8 * hi;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int rr = 0; rr < 4; rr += 2) {
      V4 oa, ob;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        oa[i] = (T)(acc[db][4 * rr + i] * inv);
        ob[i] = (T)(acc[db][4 * rr + 4 + i] * inv);
      }
      u32x2 a = __builtin_bit_cast(u32x2, oa), b = __builtin_bit_cast(u32x2, ob);
      auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
      auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
      u32x4 w;
      w[0] = r0[0]; w[1] = r1[0]; w[2] = r0[1]; w[3] = r1[1];
      *reinterpret_cast<u32x4*>(op + 32 * db + 8 * rr) = w;
    }
  if (p.lse_out && hi == 0)
    p.lse_out[(long)head * p.num_tokens + col_off] =
        l > 0.f ? (m + log2f(l)) * 0.6931471805599453f : -INFINITY;
}

template <typename T, bool F8, bool GENERAL>
int launch_bs(const AttnParams& p, int bs, int ntiles, hipStream_t st) {
  // level order when the host passed the longest query run and the (level, sequence) grid is not much
  // larger than the number of real tiles (a batch of one long and many short prompts keeps the linear order)
  int levels = p.max_q_len > 0 ? (p.max_q_len + p.q_tile - 1) / p.q_tile : 0;
  if ((long)levels * p.num_seqs > 2L * ntiles) levels = 0;
  const dim3 grid((unsigned)(levels > 0 ? levels * p.num_seqs : ntiles) * p.Hkv), block(512);
  switch (bs) {
    case 16: hipLaunchKernelGGL((paged_prefill_wide_kernel<T, 16, F8, GENERAL>), grid, block, 0, st, p, ntiles, levels); break;
    case 32: hipLaunchKernelGGL((paged_prefill_wide_kernel<T, 32, F8, GENERAL>), grid, block, 0, st, p, ntiles, levels); break;
    case 64: hipLaunchKernelGGL((paged_prefill_wide_kernel<T, 64, F8, GENERAL>), grid, block, 0, st, p, ntiles, levels); break;
    default: return -2;
  }
  VDA_CHECK_LAUNCH();
  return 0;
}

template <typename T, bool F8>
int launch_mode(const AttnParams& p, int bs, int ntiles, hipStream_t st) {
  const bool general = p.softcap > 0.f || p.alibi != nullptr;
  return general ? launch_bs<T, F8, true>(p, bs, ntiles, st) : launch_bs<T, F8, false>(p, bs, ntiles, st);
}

}  // namespace

// p.q_tile must be 256 / (Hq / Hkv); the caller has checked D == 128 and that the group size divides 256.
int vda_launch_prefill_wide(const AttnParams& p, int block_size, int dtype, int kv_fp8, hipStream_t st) {
  const int ntiles = p.num_tokens / p.q_tile + p.num_seqs;
  if (dtype == VDA_BF16)
    return kv_fp8 ? launch_mode<bf16_t, true>(p, block_size, ntiles, st) : launch_mode<bf16_t, false>(p, block_size, ntiles, st);
  return kv_fp8 ? launch_mode<f16_t, true>(p, block_size, ntiles, st) : launch_mode<f16_t, false>(p, block_size, ntiles, st);
}
