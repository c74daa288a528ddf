// Paged attention for gfx950: one MFMA kernel body serves decode (split-KV,
// the 4 waves of a workgroup share the query columns and split the keys) and
// varlen prefill / chunked prefill (each wave owns its own query columns).
//
// Covers what the reference spreads over csrc/rocm/attention.cu (MFMA decode),
// csrc/attention/paged_attention_v{1,2}.cu, the Triton unified_attention kernel
// (vllm/attention/ops/triton_unified_attention.py) and
// csrc/attention/merge_attn_states.cu.
//
// Cache layout (chosen for MFMA operands, see sched/kv_cache_interface.py):
//   K: [num_blocks, kv_heads, BS, D]    a token row is D contiguous elements
//   V: [num_blocks, kv_heads, D, BS]    a d-row is BS contiguous tokens
//
// Data flow per 32-key step, per wave (lane l: r = l&15, g = l>>4):
//   S^T = K . Q^T    A = K rows (token on the MFMA row), B = Q^T (query column r)
//                    -> the lane owns scores of column r for keys 8g..8g+7 of the
//                       step (the two 16-key tiles interleave keys so that this
//                       holds), i.e. softmax statistics are per-lane scalars;
//   O^T += V^T . P^T A = V^T rows (d on the MFMA row, 8 contiguous keys per lane
//                       = one 16-byte load from the V layout above), B = P^T which
//                       is exactly the 8 probabilities the lane already holds.
// Nothing goes through LDS except the final cross-wave merge of the decode form.
#include "attention.h"

template <typename T, int D, int BS, int NCT, bool SPLIT, bool F8>
__global__ __launch_bounds__(256) void paged_attn_kernel(AttnParams p) {
  using V8 = typename Vec<T>::v8;
  using CT = typename KVElem<T, F8>::type;
  using V4 = typename Vec<T>::v4;
  constexpr int KS = D / 32;       // MFMA k-steps over the head dimension
  constexpr int NT = D / 16;       // 16-wide d tiles of the output
  constexpr int DQ = D / 4;        // d elements owned by one lane group in QK^T
  constexpr int COLS = 16 * NCT;   // query columns per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int kvh = blockIdx.y;
  const int G = p.Hq / p.Hkv;

  // ---- which sequence / query tokens does this workgroup own ----------------
  int seq, q0;
  if (SPLIT) {
    seq = blockIdx.x;
    q0 = 0;
  } else {
    // tiles of sequence s live in [qsl[s]/QT + s, qsl[s+1]/QT + s + 1): find s
    const int tile = blockIdx.x;
    int lo = 0, hi = p.num_seqs;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (mid < p.num_seqs && p.q_start_loc[mid] / p.q_tile + mid <= tile) lo = mid; else hi = mid - 1;
    }
    seq = lo;
    q0 = (tile - (p.q_start_loc[seq] / p.q_tile + seq)) * p.q_tile;
  }
  const int qs = p.q_start_loc[seq];
  const int q_len = p.q_start_loc[seq + 1] - qs;
  const int seq_len = p.seq_lens[seq];
  if (q0 >= q_len) return;
  const int ctx = seq_len - q_len;        // keys that precede the first query token
  const int* bt = p.block_table + (long)seq * p.max_blocks;
  const int nblk_valid = (seq_len + BS - 1) / BS;

  // ---- per-lane query columns -------------------------------------------------
  // column c of the workgroup -> (query token c / G, head c % G)
  int limit[NCT], lower[NCT];
  long col_off[NCT];           // element offset of (token, head) in q / out
  float slope[NCT];
  V8 qf[NCT][KS];
  int wave_hi = 0, wave_lo = 0x7fffffff;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int c = (SPLIT ? 0 : wave * COLS) + ct * 16 + r;
    const int ti = c / G, h = c % G;
    const int qt = q0 + ti;
    const int qt_end = SPLIT ? q_len : min(q_len, q0 + p.q_tile);
    const bool valid = qt < qt_end && (SPLIT ? c < COLS : true);
    limit[ct] = valid ? (p.causal ? ctx + qt + 1 : seq_len) : 0;
    lower[ct] = (valid && p.window > 0) ? max(0, limit[ct] - p.window) : 0;
    const int head = kvh * G + h;
    slope[ct] = (p.alibi && valid) ? p.alibi[head] : 0.f;
    const T* qp = (const T*)p.q + (long)(qs + (valid ? qt : q0)) * p.q_stride + (long)head * D + g * DQ;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      qf[ct][s] = valid ? ld16<V8>(qp + 8 * s) : V8{};
    }
    col_off[ct] = valid ? (long)(qs + qt) : -1;   // token index; -1 = padding column
    if (valid) { wave_hi = max(wave_hi, limit[ct]); wave_lo = min(wave_lo, lower[ct]); }
  }
  // wave-uniform key range
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    wave_hi = max(wave_hi, __shfl_xor(wave_hi, o, 64));
    wave_lo = min(wave_lo, __shfl_xor(wave_lo, o, 64));
  }
  int step_lo = wave_hi > 0 ? wave_lo / 32 : 0;
  int step_hi = (wave_hi + 31) / 32;
  int step_stride = 1;
  if (SPLIT) {
    // partition the steps over gridDim.z, then deal a partition's steps to the 4 waves
    const int total = step_hi - step_lo;
    const int per = (total + p.nsplit - 1) / p.nsplit;
    const int a = step_lo + blockIdx.z * per;
    step_hi = min(step_hi, a + per);
    step_lo = a + wave;
    step_stride = 4;
  }

  float m[NCT], l[NCT];
  f32x4 acc[NCT][NT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    m[ct] = -INFINITY; l[ct] = 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[ct][n] = f32x4{0, 0, 0, 0};
  }

  const CT* kc = (const CT*)p.k_cache;
  const CT* vc = (const CT*)p.v_cache;
  // K row of this lane inside a step: tile u, row r  <->  key 8*(r>>2) + 4u + (r&3)
  const int krow0 = 8 * (r >> 2) + (r & 3);

  auto load_k = [&](int step, V8 (&kf)[2][KS]) {
    const int tb = step * 32;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int tok = tb + krow0 + 4 * u;
      const int bi = max(min(tok / BS, nblk_valid - 1), 0);
      const long blk = bt[bi];
      const CT* kp = kc + ((blk * p.Hkv + kvh) * BS + (tok % BS)) * (long)D + g * DQ;
#pragma unroll
      for (int s = 0; s < KS; ++s) kf[u][s] = KVLoad<T, F8>::ld(kp + 8 * s);
    }
  };
  auto load_v = [&](int step, V8 (&vf)[NT]) {
    const int tok = step * 32 + 8 * g;
    const int bi = max(min(tok / BS, nblk_valid - 1), 0);
    const long blk = bt[bi];
    const CT* vp = vc + ((blk * p.Hkv + kvh) * (long)D + r) * BS + (tok % BS);
#pragma unroll
    for (int n = 0; n < NT; ++n) vf[n] = KVLoad<T, F8>::ld(vp + (long)n * 16 * BS);
  };

  V8 kf[2][KS], kn[2][KS], vf[NT];
  if (step_lo < step_hi) load_k(step_lo, kf);
  for (int step = step_lo; step < step_hi; step += step_stride) {
    load_v(step, vf);
    const int nxt = step + step_stride;
    if (nxt < step_hi) load_k(nxt, kn);
    const int tb = step * 32 + 8 * g;   // first key of this lane's 8
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        s0 = mfma16(kf[0][s], qf[ct][s], s0);
        s1 = mfma16(kf[1][s], qf[ct][s], s1);
      }
      float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int tok = tb + j;
        float x = sv[j] * p.scale;
        if (p.softcap > 0.f) x = p.softcap * tanhf(x / p.softcap);
        x += slope[ct] * (float)(tok - (limit[ct] - 1));
        x *= 1.4426950408889634f;
        x = (tok < limit[ct] && tok >= lower[ct]) ? x : -INFINITY;
        sv[j] = x;
        mx = fmaxf(mx, x);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m[ct], mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fexp2(m[ct] - m_use);
      m[ct] = m_new;
      float ps = 0.f;
      V8 pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float e = fexp2(sv[j] - m_use);
        ps += e;
        pb[j] = (T)e;
      }
      l[ct] = l[ct] * alpha + ps;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[ct][n] *= alpha;
        acc[ct][n] = mfma16(vf[n], pb, acc[ct][n]);
      }
    }
    if (nxt < step_hi) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < KS; ++s) kf[u][s] = kn[u][s];
    }
  }

  // lane-partial row sums -> full sums
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    l[ct] += __shfl_xor(l[ct], 16, 64);
    l[ct] += __shfl_xor(l[ct], 32, 64);
  }

  if (!SPLIT) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      if (col_off[ct] < 0) continue;
      const int c = wave * COLS + ct * 16 + r;
      const int head = kvh * G + c % G;
      const float inv = l[ct] > 0.f ? p.v_scale / l[ct] : 0.f;
      T* op = (T*)p.out + col_off[ct] * p.out_stride + (long)head * D + 4 * g;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        V4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (T)(acc[ct][n][i] * inv);
        *reinterpret_cast<V4*>(op + 16 * n) = o;
      }
      if (p.lse_out && g == 0)
        p.lse_out[(long)head * p.num_tokens + col_off[ct]] =
            l[ct] > 0.f ? (m[ct] + log2f(l[ct])) * 0.6931471805599453f : -INFINITY;
    }
    return;
  } else {
    // ---- merge the 4 waves: tree over LDS, lane-linear slots ---------------------
    // Partner waves use the same lane -> element map, so a lane reads back exactly
    // the slot positions its partner lane wrote: no layout change, no conflicts.
    __shared__ f32x4 s_acc[2][NCT * NT][64];
    __shared__ float s_ml[2][NCT][2][64];
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      const int first = round == 0 ? 2 : 1;     // sending waves [first, first + cnt)
      const int cnt = round == 0 ? 2 : 1;
      if (wave >= first && wave < first + cnt) {
        const int slot = wave - first;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          s_ml[slot][ct][0][lane] = m[ct];
          s_ml[slot][ct][1][lane] = l[ct];
#pragma unroll
          for (int n = 0; n < NT; ++n) s_acc[slot][ct * NT + n][lane] = acc[ct][n];
        }
      }
      __syncthreads();
      if (wave < cnt) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const float m2 = s_ml[wave][ct][0][lane], l2 = s_ml[wave][ct][1][lane];
          const float M = fmaxf(m[ct], m2);
          const float Mu = (M == -INFINITY) ? 0.f : M;
          const float f1 = fexp2(m[ct] - Mu), f2 = fexp2(m2 - Mu);
          l[ct] = l[ct] * f1 + l2 * f2;
          m[ct] = M;
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[ct][n] = acc[ct][n] * f1 + s_acc[wave][ct * NT + n][lane] * f2;
        }
      }
      __syncthreads();
    }
    if (wave != 0) return;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      if (col_off[ct] < 0) continue;
      const int c = ct * 16 + r;
      const int head = kvh * G + c % G;
      const float L = l[ct];
      const float inv = L > 0.f ? p.v_scale / L : 0.f;
      if (p.nsplit == 1) {
        T* op = (T*)p.out + col_off[ct] * p.out_stride + (long)head * D + 4 * g;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          V4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (T)(acc[ct][n][i] * inv);
          *reinterpret_cast<V4*>(op + 16 * n) = o;
        }
      } else {
        const long pi = (col_off[ct] * p.Hq + head) * p.nsplit + blockIdx.z;
        float* po = p.part_out + pi * D + 4 * g;
#pragma unroll
        for (int n = 0; n < NT; ++n) *reinterpret_cast<f32x4*>(po + 16 * n) = acc[ct][n] * inv;
        if (g == 0) {
          const float Mu = (m[ct] == -INFINITY) ? 0.f : m[ct];
          p.part_lse[pi] = L > 0.f ? Mu + log2f(L) : -INFINITY;
        }
      }
    }
  }
}

// Prefill form with K/V staged through LDS.  The direct-to-register form above makes every wave of a
// workgroup fetch the same keys from L2 (4x the traffic) and is L2-bandwidth bound in prefill; here the
// 256 threads copy each 32-key K tile and V tile once (fully coalesced 16-byte pieces, fp8 caches are
// widened on the way in), double-buffered so that the next tile's global loads fly during the MFMAs, and
// every wave then reads its MFMA operands from LDS.  16-byte pieces are XOR-swizzled so that the eight
// lanes served per LDS cycle hit eight different bank groups for both the row-per-lane K reads and the
// d-row-per-lane V reads.
#ifndef PREFILL_REG_STAGES
#define PREFILL_REG_STAGES 1
#endif
#ifndef PREFILL_WGS_PER_CU
#define PREFILL_WGS_PER_CU 2
#endif
//
// SPARSE: vertical-slash selection for very long prompts.  A workgroup serves 128 query rows of ONE head (the
// selection is per head, so the rows of a GQA group no longer share their keys) and visits only the 32-key steps that
// hold a selected key position or that a selected diagonal crosses inside its rows; it marks them in an LDS bitmap
// built from the two lists when it starts (no index arrays in memory: the reference's block_offset / column_index,
// csrc/attention/vertical_slash_index.cu:26-150, grow with rows x list length) and walks the set bits in order.
// Inside a visited step every key is seen (subject to the causal limit), so the pattern is a superset of the exact
// lines at 32-key granularity.
constexpr int SPARSE_BM_WORDS = 4096;        // 4096 x 32 steps x 32 keys = 4 Mi keys per table
template <typename T, int D, int BS, bool F8, bool SPARSE = false>
__global__ __launch_bounds__(256, PREFILL_WGS_PER_CU) void paged_prefill_lds_kernel(AttnParams p, SparseSel sp) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  using CT = typename KVElem<T, F8>::type;
  constexpr int KS = D / 32, NT = D / 16, DQ = D / 4, NCT = 2, COLS = 32;
  constexpr int KCH = D / 8;          // 16-byte pieces per K row
  constexpr int KP = 32 * KCH;        // K pieces per 32-key step
  constexpr int VP = D * 4;           // V pieces per step (D rows x 4 groups of 8 keys)
  constexpr int KPT = KP / 256, VPT = VP / 256;
  static_assert(KP % 256 == 0 && VP % 256 == 0, "head size must be a multiple of 64");
  __shared__ V8 sK[2][KP];
  __shared__ V8 sV[2][VP];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int G = SPARSE ? 1 : p.Hq / p.Hkv;
  const int kvh = SPARSE ? blockIdx.y / (p.Hq / p.Hkv) : blockIdx.y;
  const int head0 = SPARSE ? blockIdx.y : kvh * G;        // first query head of this workgroup's columns

  const int tile = blockIdx.x;
  int lo = 0, hi = p.num_seqs;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (mid < p.num_seqs && p.q_start_loc[mid] / p.q_tile + mid <= tile) lo = mid; else hi = mid - 1;
  }
  const int seq = lo;
  const int q0 = (tile - (p.q_start_loc[seq] / p.q_tile + seq)) * p.q_tile;
  const int qs = p.q_start_loc[seq];
  const int q_len = p.q_start_loc[seq + 1] - qs;
  const int seq_len = p.seq_lens[seq];
  if (q0 >= q_len) return;                  // uniform over the workgroup
  const int ctx = seq_len - q_len;
  const int* bt = p.block_table + (long)seq * p.max_blocks;
  const int nblk_valid = (seq_len + BS - 1) / BS;
  const int qt_end = min(q_len, q0 + p.q_tile);

  int limit[NCT], lower[NCT];
  long col_off[NCT];
  float slope[NCT];
  V8 qf[NCT][KS];
  int wave_hi = 0, wave_lo = 0x7fffffff;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int c = wave * COLS + ct * 16 + r;
    const int ti = c / G, h = c % G;
    const int qt = q0 + ti;
    const bool valid = qt < qt_end;
    limit[ct] = valid ? (p.causal ? ctx + qt + 1 : seq_len) : 0;
    lower[ct] = (valid && p.window > 0) ? max(0, limit[ct] - p.window) : 0;
    const int head = head0 + h;
    slope[ct] = (p.alibi && valid) ? p.alibi[head] : 0.f;
    const T* qp = (const T*)p.q + (long)(qs + (valid ? qt : q0)) * p.q_stride + (long)head * D + g * DQ;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[ct][s] = valid ? ld16<V8>(qp + 8 * s) : V8{};
    col_off[ct] = valid ? (long)(qs + qt) : -1;
    if (valid) { wave_hi = max(wave_hi, limit[ct]); wave_lo = min(wave_lo, lower[ct]); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    wave_hi = max(wave_hi, __shfl_xor(wave_hi, o, 64));
    wave_lo = min(wave_lo, __shfl_xor(wave_lo, o, 64));
  }
  // workgroup-uniform key range (all waves take part in every tile copy)
  const int wg_hi = p.causal ? ctx + qt_end : seq_len;
  const int wg_lo = p.window > 0 ? max(0, ctx + q0 + 1 - p.window) : 0;
  const int step_lo = wg_lo / 32, step_hi = (wg_hi + 31) / 32;

  __shared__ unsigned bm[SPARSE ? SPARSE_BM_WORDS : 1];
  if constexpr (SPARSE) {
    const int nwords = (step_hi + 31) >> 5;
    for (int i = tid; i < nwords; i += 256) bm[i] = 0u;
    __syncthreads();
    auto mark = [&](int a, int b) {            // steps of the keys a..b of this table (clipped to what is visible)
      a = max(a, 0); b = min(b, wg_hi - 1);
      for (int s = a >> 5; a <= b && s <= (b >> 5); ++s) atomicOr(&bm[s >> 5], 1u << (s & 31));
    };
    const long sh = (long)seq * p.Hq + head0;
    const int kb = sp.key_base[seq];
    const int qa0 = sp.q_base[seq] + q0 - kb, qa1 = sp.q_base[seq] + qt_end - 1 - kb;
    const int nv = min(sp.vcount[sh], sp.NV), ns = min(sp.scount[sh], sp.NS);
    for (int i = tid; i < nv; i += 256) { const int k = sp.vert[sh * sp.NV + i] - kb; mark(k, k); }
    for (int i = tid; i < ns; i += 256) { const int d = sp.slash[sh * sp.NS + i]; if (d >= 0) mark(qa0 - d, qa1 - d); }
    if (p.causal && tid == 0) mark(qa0, qa1);              // a row always sees itself
    __syncthreads();
  }
  // first step >= from that is to be visited (step_hi: none)
  auto next_step = [&](int from) -> int {
    if constexpr (!SPARSE) return from;
    if (from >= step_hi) return step_hi;
    int w = from >> 5;
    const int nwords = (step_hi + 31) >> 5;
    unsigned x = bm[w] & (~0u << (from & 31));
    while (!x) { if (++w >= nwords) return step_hi; x = bm[w]; }
    return min(step_hi, (w << 5) + __ffs(x) - 1);
  };

  float m[NCT], l[NCT];
  f32x4 acc[NCT][NT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    m[ct] = -INFINITY; l[ct] = 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[ct][n] = f32x4{0, 0, 0, 0};
  }
  const CT* kc = (const CT*)p.k_cache;
  const CT* vc = (const CT*)p.v_cache;

  // Two register stages: tile t+2 is requested from memory while tile t is consumed and tile t+1 (requested one
  // iteration earlier) moves from registers into the other LDS buffer -- a full iteration of latency cover.
  V8 rkA[KPT], rvA[VPT];
#if PREFILL_REG_STAGES == 2
  V8 rkB[KPT], rvB[VPT];
#endif
  // Address arithmetic kept in 32 bits except for the one block-number x block-stride product (a single
  // v_mad_u64_u32): offsets inside a block fit 32 bits by construction.
  const unsigned blk_stride = (unsigned)p.Hkv * BS * D;          // elements per cache block (K and V alike)
  const unsigned k_head = (unsigned)kvh * BS * D, v_head = (unsigned)kvh * D * BS;
  auto gload = [&](int step, V8 (&rk)[KPT], V8 (&rv)[VPT]) {
    const int tb = step * 32;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int pi = tid + 256 * i;
      const int key = pi / KCH, ch = pi % KCH;
      const int tok = tb + key;
      const unsigned blk = (unsigned)bt[max(min(tok / BS, nblk_valid - 1), 0)];
      const unsigned off = k_head + (unsigned)(tok % BS) * D + ch * 8;
      rk[i] = KVLoad<T, F8>::ld(kc + (size_t)blk * blk_stride + off);
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int pi = tid + 256 * i;
      const int d = pi >> 2, c = pi & 3;
      const int tok = tb + 8 * c;
      const unsigned blk = (unsigned)bt[max(min(tok / BS, nblk_valid - 1), 0)];
      const unsigned off = v_head + (unsigned)d * BS + (tok % BS);
      rv[i] = KVLoad<T, F8>::ld(vc + (size_t)blk * blk_stride + off);
    }
  };
  // XOR swizzle of the 16-byte pieces of a K row.  ds_read_b128 serves lanes {0-3,12-15,20-27},
  // {4-11,16-19,28-31} (and the same in the upper half) together; with the key a lane reads (krow0 + 4u
  // below) these masks put the sixteen lanes of every such group on sixteen different pieces, where the
  // earlier three-bit mask left them two-way (one extra LDS cycle per group on every K operand read).
  auto kswz = [](int key) { return KCH == 16 ? (key & 15) : (((key & 3) << 1) | ((key >> 3) & 1)); };
  auto sstore = [&](int buf, const V8 (&rk)[KPT], const V8 (&rv)[VPT]) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int pi = tid + 256 * i;
      const int key = pi / KCH, ch = pi % KCH;
      sK[buf][key * KCH + (ch ^ kswz(key))] = rk[i];
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int pi = tid + 256 * i;
      const int d = pi >> 2, c = pi & 3;
      sV[buf][d * 4 + (c ^ ((d >> 1) & 3))] = rv[i];
    }
  };

  const int krow0 = 8 * (r >> 2) + (r & 3);
  // Tiles strictly below every column's causal limit and above every window edge need no masking; with no
  // soft-cap and no ALiBi the score transform is then a single multiply.  Both tests are wave-uniform.
  int full_hi = 0x7fffffff, full_lo = 0;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    if (col_off[ct] >= 0) { full_hi = min(full_hi, limit[ct]); full_lo = max(full_lo, lower[ct]); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    full_hi = min(full_hi, __shfl_xor(full_hi, o, 64));
    full_lo = max(full_lo, __shfl_xor(full_lo, o, 64));
  }
  const bool plain = !(p.softcap > 0.f) && p.alibi == nullptr;
  const float sl2 = p.scale * 1.4426950408889634f;
  auto compute = [&](int step, int buf) {
    const int tb0 = step * 32;
    if (!(tb0 < wave_hi && tb0 + 32 > wave_lo)) return;   // wave-uniform: tile outside this wave's causal range
    const bool fast = plain && tb0 + 32 <= full_hi && tb0 >= full_lo;
    V8 kf[2][KS], vf[NT];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int key = krow0 + 4 * u;
#pragma unroll
      for (int s = 0; s < KS; ++s) kf[u][s] = sK[buf][key * KCH + ((g * KS + s) ^ kswz(key))];
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int d = 16 * n + r;
      vf[n] = sV[buf][d * 4 + (g ^ ((d >> 1) & 3))];
    }
    const int tb = tb0 + 8 * g;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        s0 = mfma16(kf[0][s], qf[ct][s], s0);
        s1 = mfma16(kf[1][s], qf[ct][s], s1);
      }
      float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
      float mx = -INFINITY;
      if (fast) {
        // raw scores: the scale (positive) is applied to the maximum once and folded into the fma that
        // feeds the exponential, so a score costs max + fma + exp + add
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, sv[j]);
        mx *= sl2;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int tok = tb + j;
          float x = sv[j] * p.scale;
          if (p.softcap > 0.f) x = p.softcap * tanhf(x / p.softcap);
          x += slope[ct] * (float)(tok - (limit[ct] - 1));
          x *= 1.4426950408889634f;
          x = (tok < limit[ct] && tok >= lower[ct]) ? x : -INFINITY;
          sv[j] = x;
          mx = fmaxf(mx, x);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m[ct], mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fexp2(m[ct] - m_use);
      m[ct] = m_new;
      float ps = 0.f;
      V8 pb;
      const float mul = fast ? sl2 : 1.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float e = fexp2(fmaf(sv[j], mul, -m_use));
        ps += e;
        pb[j] = (T)e;
      }
      l[ct] = l[ct] * alpha + ps;
      if (__any(alpha != 1.f)) {          // the running maximum settles early: most tiles need no rescale
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[ct][n] *= alpha;
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[ct][n] = mfma16(vf[n], pb, acc[ct][n]);
    }
  };

#if PREFILL_REG_STAGES == 2
  static_assert(!SPARSE, "the sparse form walks its step list with one register stage");
  // prologue: tile 0 -> LDS buffer 0, tile 1 -> register stage A
  if (step_lo < step_hi) {
    gload(step_lo, rkA, rvA);
    sstore(0, rkA, rvA);
    if (step_lo + 1 < step_hi) gload(step_lo + 1, rkA, rvA);
  }
  __syncthreads();
  for (int step = step_lo; step < step_hi; step += 2) {
    // even half: consume buffer 0, stage A holds tile step+1, request tile step+2 into stage B
    if (step + 2 < step_hi) gload(step + 2, rkB, rvB);
    compute(step, 0);
    if (step + 1 < step_hi) sstore(1, rkA, rvA);
    __syncthreads();
    if (step + 1 >= step_hi) break;
    // odd half: consume buffer 1, stage B holds tile step+2, request tile step+3 into stage A
    if (step + 3 < step_hi) gload(step + 3, rkA, rvA);
    compute(step + 1, 1);
    if (step + 2 < step_hi) sstore(0, rkB, rvB);
    __syncthreads();
  }
#else
  // One register stage (16 fewer registers: two workgroups fit a CU, so one wave's global-load and
  // softmax latency is covered by the other wave of its SIMD rather than by a deeper software pipeline).
  // Tile t+1 sits in registers while tile t is consumed, then moves to the other LDS buffer and tile t+2
  // is requested.
  int s0 = next_step(step_lo), s1 = next_step(s0 + 1), s2 = next_step(s1 + 1);
  if (s0 < step_hi) {
    gload(s0, rkA, rvA);
    sstore(0, rkA, rvA);
    if (s1 < step_hi) gload(s1, rkA, rvA);
  }
  __syncthreads();
  for (int buf = 0; s0 < step_hi; buf ^= 1) {
    compute(s0, buf);
    if (s1 < step_hi) sstore(buf ^ 1, rkA, rvA);
    if (s2 < step_hi) gload(s2, rkA, rvA);
    __syncthreads();
    s0 = s1; s1 = s2; s2 = next_step(s2 + 1);
  }
#endif

#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    l[ct] += __shfl_xor(l[ct], 16, 64);
    l[ct] += __shfl_xor(l[ct], 32, 64);
  }
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    if (col_off[ct] < 0) continue;
    const int c = wave * COLS + ct * 16 + r;
    const int head = head0 + c % G;
    const float inv = l[ct] > 0.f ? p.v_scale / l[ct] : 0.f;
    T* op = (T*)p.out + col_off[ct] * p.out_stride + (long)head * D + 4 * g;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      V4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (T)(acc[ct][n][i] * inv);
      *reinterpret_cast<V4*>(op + 16 * n) = o;
    }
    if (p.lse_out && g == 0)
      p.lse_out[(long)head * p.num_tokens + col_off[ct]] =
          l[ct] > 0.f ? (m[ct] + log2f(l[ct])) * 0.6931471805599453f : -INFINITY;
  }
}

// Merge split-KV partials: one wave per (token, head).
template <typename T, int D>
__global__ __launch_bounds__(64) void attn_reduce_kernel(T* out, const float* part_out, const float* part_lse,
                                                         int Hq, int nsplit, long out_stride) {
  const long th = blockIdx.x;          // token * Hq + head
  const int lane = threadIdx.x;
  // lane s holds partition s (nsplit <= 64): maximum and weight sum by wave reductions, weights handed out
  // by shuffles; partitions without keys (short sequences under a split factor fixed at graph capture)
  // carry -inf and are skipped without touching their rows
  const float mine = lane < nsplit ? part_lse[th * nsplit + lane] : -INFINITY;
  float M = mine;
#pragma unroll
  for (int o_ = 32; o_ > 0; o_ >>= 1) M = fmaxf(M, __shfl_xor(M, o_, 64));
  const float Mu = (M == -INFINITY) ? 0.f : M;
  const float w_mine = (mine == -INFINITY) ? 0.f : fexp2(mine - Mu);
  float W = w_mine;
#pragma unroll
  for (int o_ = 32; o_ > 0; o_ >>= 1) W += __shfl_xor(W, o_, 64);
  constexpr int PER = (D + 63) / 64;
  float o[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) o[i] = 0.f;
#pragma unroll 4
  for (int s = 0; s < nsplit; ++s) {
    const float w = __shfl(w_mine, s, 64);
    if (w == 0.f) continue;                    // wave-uniform
    const float* po = part_out + (th * nsplit + s) * D;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int d = lane + 64 * i;
      if (d < D) o[i] += w * po[d];
    }
  }
  const float inv = W > 0.f ? 1.f / W : 0.f;
  const long tok = th / Hq, head = th % Hq;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int d = lane + 64 * i;
    if (d < D) out[tok * out_stride + head * D + d] = (T)(o[i] * inv);
  }
}

// LSE-weighted merge of two partial attention outputs (cascade / prefix+suffix).
// out, prefix, suffix: [T, H, D]; lse: [H, T] natural-log domain.
template <typename T>
__global__ void merge_attn_states_kernel(T* out, float* out_lse, const T* pre, const float* pre_lse,
                                         const T* suf, const float* suf_lse, int Tn, int H, int D) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over T*H*D/8
  const int per = D / 8;
  if (idx >= (long)Tn * H * per) return;
  const int d8 = (idx % per) * 8;
  const long th = idx / per;
  const int h = th % H;
  const long t = th / H;
  float a = pre_lse[(long)h * Tn + t], b = suf_lse[(long)h * Tn + t];
  a = isinf(a) && a > 0 ? -INFINITY : a;
  b = isinf(b) && b > 0 ? -INFINITY : b;
  const float mx = fmaxf(a, b);
  const float mu = (mx == -INFINITY) ? 0.f : mx;
  const float ea = expf(a - mu), eb = expf(b - mu);
  const float den = ea + eb;
  const float wa = den > 0.f ? ea / den : 0.f, wb = den > 0.f ? eb / den : 0.f;
  using V8 = typename Vec<T>::v8;
  const V8 x = ld16<V8>(pre + th * D + d8), y = ld16<V8>(suf + th * D + d8);
  V8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (T)((float)x[i] * wa + (float)y[i] * wb);
  st16<V8>(out + th * D + d8, o);
  if (out_lse && d8 == 0) out_lse[(long)h * Tn + t] = den > 0.f ? mu + logf(den) : -INFINITY;
}

// ---------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------
template <typename T, int D, int BS, bool F8>
static int launch_attn(const AttnParams& p, int form, int grid_x, hipStream_t st) {
  // form 0: decode (split over waves, NCT=1, up to 16 query columns per kv head)
  // form 1: decode with up to 32 columns; form 2: prefill (NCT=2)
  dim3 block(256);
  if (form == 0) {
    dim3 grid(grid_x, p.Hkv, p.nsplit);
    hipLaunchKernelGGL((paged_attn_kernel<T, D, BS, 1, true, F8>), grid, block, 0, st, p);
  } else if (form == 1) {
    dim3 grid(grid_x, p.Hkv, p.nsplit);
    hipLaunchKernelGGL((paged_attn_kernel<T, D, BS, 2, true, F8>), grid, block, 0, st, p);
  } else if (form == 3) {
    dim3 grid(grid_x, p.Hkv, 1);
    if constexpr (D == 64 || D == 128) {          // the LDS tile layout is built for rows of 8 or 16 pieces
      hipLaunchKernelGGL((paged_prefill_lds_kernel<T, D, BS, F8>), grid, block, 0, st, p, SparseSel{});
    } else {
      hipLaunchKernelGGL((paged_attn_kernel<T, D, BS, 2, false, F8>), grid, block, 0, st, p);
    }
  } else {
    dim3 grid(grid_x, p.Hkv, 1);
    hipLaunchKernelGGL((paged_attn_kernel<T, D, BS, 2, false, F8>), grid, block, 0, st, p);
  }
  VDA_CHECK_LAUNCH();
  return 0;
}

template <typename T, int D, bool F8>
static int dispatch_bs(const AttnParams& p, int bs, int form, int grid_x, hipStream_t st) {
  switch (bs) {
    case 16: return launch_attn<T, D, 16, F8>(p, form, grid_x, st);
    case 32: return launch_attn<T, D, 32, F8>(p, form, grid_x, st);
    case 64: return launch_attn<T, D, 64, F8>(p, form, grid_x, st);
  }
  return -2;
}

// Head sizes: any multiple of 32 fits the kernels (a lane group owns D / 4 values in MFMA steps of 8); built are
// 64, 96, 128, 192 and 256, so heads of 80 / 96 run in 96-value rows and 160 / 192 in 192-value rows instead of
// the next power of two (reference head sizes: csrc/attention/paged_attention_v1.cu:150-188).
template <typename T, bool F8>
static int dispatch_d8(const AttnParams& p, int d, int bs, int form, int grid_x, hipStream_t st) {
  switch (d) {
    case 64: return dispatch_bs<T, 64, F8>(p, bs, form, grid_x, st);
    case 128: return dispatch_bs<T, 128, F8>(p, bs, form, grid_x, st);
  }
  if constexpr (!F8) {
    switch (d) {
      case 96: return dispatch_bs<T, 96, F8>(p, bs, form, grid_x, st);
      case 192: return dispatch_bs<T, 192, F8>(p, bs, form, grid_x, st);
      case 256: return dispatch_bs<T, 256, F8>(p, bs, form, grid_x, st);
    }
  }
  return -3;
}

template <typename T>
static int dispatch_d(const AttnParams& p, int d, int bs, int form, int grid_x, hipStream_t st, int kv_fp8) {
  // fp8 caches: head sizes 64 / 128 (the common ones) to bound the number of kernel variants
  if (kv_fp8) return dispatch_d8<T, true>(p, d, bs, form, grid_x, st);
  return dispatch_d8<T, false>(p, d, bs, form, grid_x, st);
}

// Decode / speculative-verify attention.  Every sequence contributes
// q_len = q_start_loc[s+1]-q_start_loc[s] query tokens with q_len * (Hq/Hkv) <= 32.
// num_tokens is only needed to size the split-KV reduce.
VDA_EXPORT int vda_paged_attention_decode(
    void* out, const void* q, const void* k_cache, const void* v_cache, const int* block_table,
    const int* seq_lens, const int* q_start_loc, const float* alibi, float* part_out, float* part_lse,
    int num_seqs, int num_tokens, int Hq, int Hkv, int D, int block_size, int max_blocks, int nsplit,
    int max_q_len, long q_stride, long out_stride, float scale, float softcap, int window, int dtype,
    int kv_fp8, float k_scale, float v_scale, hipStream_t st) {
  if (Hq % Hkv) return -1;
  if (nsplit < 1 || nsplit > 64) return -6;      // the reduce kernel keeps one partition per lane
  const int cols = max_q_len * (Hq / Hkv);
  if (cols > 32 || (cols > 16 && D > 128)) return -4;
  AttnParams p{q, k_cache, v_cache, out, part_out, part_lse, block_table, seq_lens, q_start_loc, alibi,
               num_seqs, Hq, Hkv, max_blocks, nsplit, q_stride, out_stride, scale * (kv_fp8 ? k_scale : 1.f), softcap,
               kv_fp8 ? v_scale : 1.f, window, 0, 1, num_tokens, nullptr};
  const int form = cols <= 16 ? 0 : 1;
  int rc = dtype == VDA_BF16 ? dispatch_d<bf16_t>(p, D, block_size, form, num_seqs, st, kv_fp8)
                             : dispatch_d<f16_t>(p, D, block_size, form, num_seqs, st, kv_fp8);
  if (rc) return rc;
  if (nsplit > 1) {
    const long n = (long)num_tokens * Hq;
#define RED(TT, DD) hipLaunchKernelGGL((attn_reduce_kernel<TT, DD>), dim3(n), dim3(64), 0, st, (TT*)out, \
                                       part_out, part_lse, Hq, nsplit, out_stride)
    if (dtype == VDA_BF16) {
      if (D == 64) RED(bf16_t, 64); else if (D == 96) RED(bf16_t, 96); else if (D == 128) RED(bf16_t, 128);
      else if (D == 192) RED(bf16_t, 192); else RED(bf16_t, 256);
    } else {
      if (D == 64) RED(f16_t, 64); else if (D == 96) RED(f16_t, 96); else if (D == 128) RED(f16_t, 128);
      else if (D == 192) RED(f16_t, 192); else RED(f16_t, 256);
    }
#undef RED
    VDA_CHECK_LAUNCH();
  }
  return 0;
}

// 1 (default): the wide form where it applies, else the 4-wave form with K/V staged through LDS;
// 0: direct-to-register form, 2: always the 4-wave LDS form, 3: the wide form whenever the shape allows
// (0 / 2 / 3 are for A/B measurements and tests)
static int g_prefill_variant = 1;
static long g_prefill_wide_min_wgs = 192;
VDA_EXPORT int vda_set_prefill_variant(int v) { g_prefill_variant = v; return 0; }
VDA_EXPORT int vda_set_prefill_wide_min_wgs(long n) { g_prefill_wide_min_wgs = n; return 0; }

// Varlen attention of new tokens against the paged cache (the new tokens' own K/V must already be in
// the cache): causal, or bidirectional within each sequence (encoder models; the shared-prefix pass of
// cascade attention, where one pseudo-sequence holds every query token of the batch).  lse_out, when
// given, receives the log-sum-exp of every (head, token) for merging with another pass.  q_tile = 128 / (Hq/Hkv) query tokens
// per workgroup; the grid over-allocates one tile per sequence.
VDA_EXPORT int vda_paged_attention_prefill(
    void* out, const void* q, const void* k_cache, const void* v_cache, const int* block_table,
    const int* seq_lens, const int* q_start_loc, const float* alibi, int num_seqs, int num_tokens,
    int Hq, int Hkv, int D, int block_size, int max_blocks, long q_stride, long out_stride, float scale,
    float softcap, int window, int dtype, int kv_fp8, float k_scale, float v_scale, int causal, float* lse_out,
    int max_q_len, hipStream_t st) {
  if (Hq % Hkv) return -1;
  const int G = Hq / Hkv;
  if (G > 128) return -4;
  if (!causal && window > 0) return -5;
  const int q_tile = 128 / G;     // 4 waves x 32 columns
  AttnParams p{q, k_cache, v_cache, out, nullptr, nullptr, block_table, seq_lens, q_start_loc, alibi,
               num_seqs, Hq, Hkv, max_blocks, 1, q_stride, out_stride, scale * (kv_fp8 ? k_scale : 1.f), softcap,
               kv_fp8 ? v_scale : 1.f, window, q_tile, causal, num_tokens, lse_out, max_q_len};
  // Wide form (8 waves, 256 columns, 64-key tiles): head size 128 and enough query tiles to occupy the
  // chip (one workgroup per CU); short chunks keep the 4-wave form, which cuts them into more workgroups.
  if (D == 128 && 256 % G == 0 && (g_prefill_variant == 1 || g_prefill_variant == 3)) {
    const int wide_tile = 256 / G;
    const long wgs = (long)(num_tokens / wide_tile) * Hkv;
    if (g_prefill_variant == 3 || wgs >= g_prefill_wide_min_wgs) {
      p.q_tile = wide_tile;
      return vda_launch_prefill_wide(p, block_size, dtype, kv_fp8, st);
    }
  }
  const int grid_x = num_tokens / q_tile + num_seqs;
  const int form = g_prefill_variant == 0 ? 2 : 3;
  return dtype == VDA_BF16 ? dispatch_d<bf16_t>(p, D, block_size, form, grid_x, st, kv_fp8)
                           : dispatch_d<f16_t>(p, D, block_size, form, grid_x, st, kv_fp8);
}

// Sparse (vertical-slash) prefill: the 4-wave LDS form, one head per workgroup, visiting only the selected 32-key
// steps (see paged_prefill_lds_kernel).  max_keys: the longest table of the batch (sizes the LDS bitmap check).
template <typename T, int D, bool F8>
static int launch_sparse(const AttnParams& p, const SparseSel& sp, int bs, dim3 grid, hipStream_t st) {
  switch (bs) {
    case 16: hipLaunchKernelGGL((paged_prefill_lds_kernel<T, D, 16, F8, true>), grid, dim3(256), 0, st, p, sp); break;
    case 32: hipLaunchKernelGGL((paged_prefill_lds_kernel<T, D, 32, F8, true>), grid, dim3(256), 0, st, p, sp); break;
    case 64: hipLaunchKernelGGL((paged_prefill_lds_kernel<T, D, 64, F8, true>), grid, dim3(256), 0, st, p, sp); break;
    default: return -2;
  }
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_paged_attention_prefill_sparse(
    void* out, const void* q, const void* k_cache, const void* v_cache, const int* block_table,
    const int* seq_lens, const int* q_start_loc, const int* vert, const int* vcount, const int* slash,
    const int* scount, const int* key_base, const int* q_base, int NV, int NS, int num_seqs, int num_tokens,
    int Hq, int Hkv, int D, int block_size, int max_blocks, long q_stride, long out_stride, float scale,
    float softcap, int dtype, int kv_fp8, float k_scale, float v_scale, int causal, float* lse_out, long max_keys,
    hipStream_t st) {
  if (Hq % Hkv) return -1;
  if (D != 64 && D != 128) return -3;
  if (max_keys > (long)SPARSE_BM_WORDS * 32 * 32 || (long)max_blocks * block_size > (long)SPARSE_BM_WORDS * 1024)
    return -7;
  AttnParams p{q, k_cache, v_cache, out, nullptr, nullptr, block_table, seq_lens, q_start_loc, nullptr,
               num_seqs, Hq, Hkv, max_blocks, 1, q_stride, out_stride, scale * (kv_fp8 ? k_scale : 1.f), softcap,
               kv_fp8 ? v_scale : 1.f, 0, 128, causal, num_tokens, lse_out, 0};
  SparseSel sp{vert, vcount, slash, scount, key_base, q_base, NV, NS};
  dim3 grid(num_tokens / 128 + num_seqs, Hq, 1);
#define SPARSE_GO(TT) \
  (D == 64 ? (kv_fp8 ? launch_sparse<TT, 64, true>(p, sp, block_size, grid, st)                \
                     : launch_sparse<TT, 64, false>(p, sp, block_size, grid, st))              \
           : (kv_fp8 ? launch_sparse<TT, 128, true>(p, sp, block_size, grid, st)               \
                     : launch_sparse<TT, 128, false>(p, sp, block_size, grid, st)))
  return dtype == VDA_BF16 ? SPARSE_GO(bf16_t) : SPARSE_GO(f16_t);
#undef SPARSE_GO
}

VDA_EXPORT int vda_merge_attn_states(void* out, float* out_lse, const void* pre, const float* pre_lse,
                                     const void* suf, const float* suf_lse, int T, int H, int D, int dtype,
                                     hipStream_t st) {
  if (D % 8) return -1;
  const long n = (long)T * H * (D / 8);
  const int threads = 256;
  const int blocks = (int)((n + threads - 1) / threads);
  if (dtype == VDA_BF16)
    hipLaunchKernelGGL(merge_attn_states_kernel<bf16_t>, dim3(blocks), dim3(threads), 0, st, (bf16_t*)out,
                       out_lse, (const bf16_t*)pre, pre_lse, (const bf16_t*)suf, suf_lse, T, H, D);
  else
    hipLaunchKernelGGL(merge_attn_states_kernel<f16_t>, dim3(blocks), dim3(threads), 0, st, (f16_t*)out,
                       out_lse, (const f16_t*)pre, pre_lse, (const f16_t*)suf, suf_lse, T, H, D);
  VDA_CHECK_LAUNCH();
  return 0;
}
