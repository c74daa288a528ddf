// 4-bit code-book weights (W4A16 where a code names one of 16 table values instead of an integer level):
// bitsandbytes NF4 / FP4 (groups of 64, first element of a byte in the high nibble) and NVFP4 (E2M1 codes, groups of
// 16, first element in the low nibble).
//
//   codes  uint8 [N, K/2]   two codes per byte, K order
//   scales float [N, K/G]   one fp32 scale per row and group of G input elements (G a multiple of 16)
//   table  float [16]       value of every code;  weight = table[code] * scale
//
// Decode (M <= 64), two forms.  Per-weight form (any G, the one groups of 16 use): 16 RW weight rows per workgroup, a
// lane streams 32 consecutive K elements (one 16-byte load of codes) per 128-wide chunk and row set, looks the 32
// values up in a 16-entry LDS table (lanes that read the same entry are served by one broadcast, different entries
// sit in different banks), scales them and feeds them to MFMA as the A operand; the activation rows ride on the MFMA
// column dimension.  Chunk-layout form (G >= 32, further down): operands straight out of a byte-indexed table, scales
// applied to the accumulators, K splits for narrow layers.  HBM traffic is 4 bits per weight plus 32 / G bits of
// scale.  Larger M: `vda_cb4_dequant` / `vda_cb4_dequant_post` expand the weights to 16 bit once per call and the
// library GEMM does the rest.
// Role of the reference's bitsandbytes matmul_4bit / dequantize_4bit calls
// (vllm/model_executor/layers/quantization/bitsandbytes.py:_apply_4bit_weight) and of the NVFP4 GEMM
// (csrc/quantization/fp4/nvfp4_scaled_mm_kernels.cu; activations stay 16-bit here).
#include "common.h"

template <typename T, bool HI_FIRST>
__device__ __forceinline__ typename Vec<T>::v8 cb4_expand(uint32_t w, const float* tab, float s) {
  typename Vec<T>::v8 o;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const uint32_t byte = (w >> (8 * b)) & 0xffu;
    const uint32_t first = HI_FIRST ? (byte >> 4) : (byte & 15u), second = HI_FIRST ? (byte & 15u) : (byte >> 4);
    o[2 * b] = (T)(tab[first] * s);
    o[2 * b + 1] = (T)(tab[second] * s);
  }
  return o;
}

// RW: row sets of 16 per workgroup -- the activation registers of a chunk serve all of them (see the note on L2
// traffic at cb4_skinny_post_kernel)
template <typename T, int MT, bool HI_FIRST, int RW>
__global__ __launch_bounds__(256) void cb4_skinny_gemm_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                              const uint8_t* __restrict__ codes,
                                                              const float* __restrict__ scales,
                                                              const float* __restrict__ table,
                                                              const T* __restrict__ bias, int M, int N, int K, int G,
                                                              long x_stride, long y_stride) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  __shared__ float tab[16];
  __shared__ f32x4 red[3][MT][64];
  if (threadIdx.x < 16) tab[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16 * RW;
  const int ngroups = K / G;
  const uint8_t* wrow = codes + (long)(n0 + r) * (K / 2) + g * 16;       // 16 bytes = 32 elements per chunk
  const long wset = (long)16 * (K / 2);
  const float* srow = scales + (long)(n0 + r) * ngroups;
  const long sset = (long)16 * ngroups;
  const T* xp[MT];
  bool xv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + r;
    xv[mt] = m < M;
    xp[mt] = x + (long)(xv[mt] ? m : 0) * x_stride + g * 32;
  }
  f32x4 acc[RW][MT];
#pragma unroll
  for (int rw = 0; rw < RW; ++rw)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[rw][mt] = f32x4{0, 0, 0, 0};

  const int nchunk = K >> 7;
  // UN chunks per trip: all their code and scale loads are issued before the first look-up, so a lane keeps
  // UN x RW x 16 bytes of codes in flight
  constexpr int UN = RW >= 4 ? 1 : 4 / RW;
  for (int c = wave; c < nchunk; c += 4 * UN) {
    u32x4 raw[UN][RW];
    float sa[UN][RW], sb[UN][RW];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int cc = c + 4 * u;
      if (cc < nchunk) {
        const int e0 = cc * 128 + g * 32;
#pragma unroll
        for (int rw = 0; rw < RW; ++rw) {
          raw[u][rw] = *reinterpret_cast<const u32x4*>(wrow + rw * wset + (long)cc * 64);
          sa[u][rw] = srow[rw * sset + e0 / G];             // elements 0 .. 15 of the lane's 32
          sb[u][rw] = srow[rw * sset + (e0 + 16) / G];      // elements 16 .. 31 (the same group unless G is 16)
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int cc = c + 4 * u;
      if (cc >= nchunk) break;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        V8 xa[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xa[mt] = xv[mt] ? ld16<V8>(xp[mt] + (long)cc * 128 + 8 * q) : V8{};
#pragma unroll
        for (int rw = 0; rw < RW; ++rw) {
          const V8 wa = cb4_expand<T, HI_FIRST>(raw[u][rw][q], tab, q < 2 ? sa[u][rw] : sb[u][rw]);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[rw][mt] = mfma16(wa, xa[mt], acc[rw][mt]);
        }
      }
    }
  }

#pragma unroll
  for (int rw = 0; rw < RW; ++rw) {
    if (rw) __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) red[wave - 1][mt][lane] = acc[rw][mt];
    }
    __syncthreads();
    if (wave == 0) {
      const int nb = n0 + rw * 16 + 4 * g;
      V4 bv = V4{};
      if (bias) bv = *reinterpret_cast<const V4*>(bias + nb);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 sum = acc[rw][mt] + red[0][mt][lane] + red[1][mt][lane] + red[2][mt][lane];
        if (xv[mt]) {
          V4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (T)(sum[i] + (float)bv[i]);
          *reinterpret_cast<V4*>(y + (long)(mt * 16 + r) * y_stride + nb) = o;
        }
      }
    }
  }
}

VDA_EXPORT int vda_cb4_skinny_gemm(void* y, const void* x, const void* codes, const void* scales, const void* table,
                                   const void* bias, int M, int N, int K, int G, int hi_first, long x_stride,
                                   long y_stride, int dtype, hipStream_t st) {
  if (M < 1 || M > 64 || N < 16 || N % 16 || K < 128 || K % 128 || G < 16 || G % 16 || K % G) return -1;
  if (G > 16 && G % 32) return -1;          // a lane's 16-element halves must not straddle groups
  const int mt = (M + 15) / 16;
  // rows per workgroup: as many as leave one workgroup per CU and the accumulators in registers
  // (below 8 rows x is small beside the codes and more, smaller workgroups hide latency better)
  int rw = 1;
  const int want = M >= 8 ? 256 : 512;
  if (mt <= 2 && N % 64 == 0 && N / 64 >= want) rw = 4;
  else if (N % 32 == 0 && N / 32 >= want) rw = 2;
  dim3 grid(N / (16 * rw)), block(256);
#define L(TT, MM, HF, RR) hipLaunchKernelGGL((cb4_skinny_gemm_kernel<TT, MM, HF, RR>), grid, block, 0, st, (TT*)y,      \
                                             (const TT*)x, (const uint8_t*)codes, (const float*)scales,               \
                                             (const float*)table, (const TT*)bias, M, N, K, G, x_stride, y_stride)
#define BYR(TT, MM, HF) do { if (rw == 4) L(TT, MM, HF, 4); else if (rw == 2) L(TT, MM, HF, 2);                        \
                             else L(TT, MM, HF, 1); } while (0)
#define BYM(TT, HF) do { if (mt == 1) BYR(TT, 1, HF); else if (mt == 2) BYR(TT, 2, HF);                                \
                         else if (rw == 2) { if (mt == 3) L(TT, 3, HF, 2); else L(TT, 4, HF, 2); }                    \
                         else { if (mt == 3) L(TT, 3, HF, 1); else L(TT, 4, HF, 1); } } while (0)
#define BY(TT) do { if (hi_first) BYM(TT, true); else BYM(TT, false); } while (0)
  if (dtype == VDA_BF16) BY(bf16_t);
  else if (dtype == VDA_F16) BY(f16_t);
  else return -3;
#undef BY
#undef BYM
#undef BYR
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

// ---- groups of 32 or more: unscaled operands, scales applied to the accumulators --------------------------------------
// Layout made once after loading (layers/quant.py::CodeBook4LinearMethod.process_weights_after_loading):
//   codes   the four dwords of every 64-byte (128-element) chunk of a row transposed as a 4 x 4 matrix, so that the
//           16 bytes lane group g loads hold, for MFMA q = 0 .. 3, the codes of elements q * 32 + g * 8 .. + 7:
//           one MFMA then sums 32 CONSECUTIVE elements, which lie in one scale group
//   scales  transposed to [K / G, N]: the four output rows a lane owns in the accumulator are one 16-byte load
// The operand registers come straight out of an LDS table indexed by the code BYTE (two table values as packed 16-bit
// numbers: 16 look-ups per lane and chunk, no nibble arithmetic, no multiply, no conversion).  The table is stored
// 32 times, entry e of copy j at word 32 e + j, and lane l reads copy l % 32: lanes l and l + 32 are the only ones
// that can meet in a bank (random bytes would put four or five lanes into the fullest bank of a single copy).  A
// group's partial sum is multiplied by its fp32 scale when the group ends.  Rounding: the table values to 16 bit
// once, products and sums in fp32 -- at least as accurate as rounding every scaled weight to 16 bit.
// A workgroup owns 16 RW rows: the activation registers of a chunk serve RW row sets (every workgroup reads all of
// x from L2 -- with 16 rows that is M / 4 times the bytes of its codes, with 64 rows M / 16 times).
template <typename T, int MT, int GQ, int RW>
__global__ __launch_bounds__(256) void cb4_skinny_post_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                              const uint8_t* __restrict__ codes,
                                                              const float* __restrict__ scales_t,
                                                              const float* __restrict__ table,
                                                              const T* __restrict__ bias, int M, int N, int K, int G,
                                                              int hi_first, long x_stride, long y_stride,
                                                              float* __restrict__ ws) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  __shared__ uint32_t tab2[256 * 32];
  __shared__ uint32_t tab1[256];
  __shared__ f32x4 red[3][MT][64];
  {
    const int b = threadIdx.x;
    const int first = hi_first ? (b >> 4) : (b & 15), second = hi_first ? (b & 15) : (b >> 4);
    union { T h[2]; uint32_t u; } pk;
    pk.h[0] = (T)table[first];
    pk.h[1] = (T)table[second];
    tab1[b] = pk.u;
  }
  __syncthreads();
  // the 32 copies are written in address order (a wave's 64 stores go to 64 banks; entry by entry they would all
  // land in two banks)
#pragma unroll
  for (int i = 0; i < 32; ++i) tab2[i * 256 + threadIdx.x] = tab1[(i * 256 + threadIdx.x) >> 5];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16 * RW;
  const uint32_t* tl = tab2 + (lane & 31);
  const uint8_t* wrow = codes + (long)(n0 + r) * (K / 2) + g * 16;     // + rw * 16 rows
  const long wset = (long)16 * (K / 2);
  const float* srow = scales_t + n0 + 4 * g;                           // + group * N + rw * 16
  const T* xp[MT];
  bool xv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + r;
    xv[mt] = m < M;
    xp[mt] = x + (long)(xv[mt] ? m : 0) * x_stride + g * 8;
  }
  f32x4 acc[RW][MT];
#pragma unroll
  for (int rw = 0; rw < RW; ++rw)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[rw][mt] = f32x4{0, 0, 0, 0};

  // K splits (gridDim.y): workgroup y takes a contiguous run of chunks
  const int per = ((K >> 7) + gridDim.y - 1) / gridDim.y;
  const int nchunk = min(K >> 7, (int)(blockIdx.y + 1) * per);
  constexpr int UN = RW >= 4 ? 1 : 4 / RW;                    // UN x RW 16-byte code loads in flight per lane
  constexpr int NS = 4 / GQ;                                  // scale groups per chunk: 4 (G = 32), 2 (64), 1 (>= 128)
  struct Codes { u32x4 raw[UN][RW]; };
  struct Side { V8 xa[4][MT]; f32x4 sv[RW][NS]; };            // activations and scales of one chunk
  // codes of UN chunks (first chunk c, stride 4) for all row sets: the HBM stream
  auto issue = [&](Codes& s, int c) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int cc = c + 4 * u;
      if (cc < nchunk) {
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
          s.raw[u][rw] = *reinterpret_cast<const u32x4*>(wrow + rw * wset + (long)cc * 64);
      }
    }
  };
  // activations and scales of chunk cc: L2 traffic
  auto load_side = [&](Side& d, int cc) {
    if (cc >= nchunk) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) d.xa[q][mt] = xv[mt] ? ld16<V8>(xp[mt] + (long)cc * 128 + 32 * q) : V8{};
#pragma unroll
    for (int rw = 0; rw < RW; ++rw)
#pragma unroll
      for (int sgi = 0; sgi < NS; ++sgi)
        d.sv[rw][sgi] = *reinterpret_cast<const f32x4*>(srow + (long)((cc * 128 + sgi * GQ * 32) / G) * N + rw * 16);
  };
  auto compute = [&](const u32x4 (&raw)[RW], const Side& d) {
#pragma unroll
    for (int sgi = 0; sgi < NS; ++sgi) {
#pragma unroll
      for (int rw = 0; rw < RW; ++rw) {
        f32x4 part[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) part[mt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int qq = 0; qq < GQ; ++qq) {
          const uint32_t w = raw[rw][sgi * GQ + qq];
          union { uint32_t d[4]; V8 v; } op;
          op.d[0] = tl[(w & 0xffu) << 5];
          op.d[1] = tl[((w >> 8) & 0xffu) << 5];
          op.d[2] = tl[((w >> 16) & 0xffu) << 5];
          op.d[3] = tl[(w >> 24) << 5];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) part[mt] = mfma16(op.v, d.xa[sgi * GQ + qq][mt], part[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[rw][mt][i] = fmaf(part[mt][i], d.sv[rw][sgi][i], acc[rw][mt][i]);
      }
    }
  };
  constexpr int STEP = 4 * UN;
  const int c0 = blockIdx.y * per + wave;
  if constexpr (MT == 1) {
    // up to 16 rows: the codes of the next step are in flight while this one computes.  Loads return in issue order,
    // so a step's activations and scales are issued BEFORE the next step's codes: waiting for them then leaves
    // those codes outstanding.
    Codes ca, cb;
    Side d[UN];
    issue(ca, c0);
    for (int c = c0; c < nchunk; c += 2 * STEP) {
#pragma unroll
      for (int u = 0; u < UN; ++u) load_side(d[u], c + 4 * u);
      issue(cb, c + STEP);
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (c + 4 * u < nchunk) compute(ca.raw[u], d[u]);
#pragma unroll
      for (int u = 0; u < UN; ++u) load_side(d[u], c + STEP + 4 * u);
      issue(ca, c + 2 * STEP);
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (c + STEP + 4 * u < nchunk) compute(cb.raw[u], d[u]);
    }
  } else {
    for (int c = c0; c < nchunk; c += STEP) {
      Codes cs;
      issue(cs, c);
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (c + 4 * u >= nchunk) break;
        Side d;
        load_side(d, c + 4 * u);
        compute(cs.raw[u], d);
      }
    }
  }

#pragma unroll
  for (int rw = 0; rw < RW; ++rw) {
    if (rw) __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) red[wave - 1][mt][lane] = acc[rw][mt];
    }
    __syncthreads();
    if (wave == 0) {
      const int nb = n0 + rw * 16 + 4 * g;
      V4 bv = V4{};
      if (bias) bv = *reinterpret_cast<const V4*>(bias + nb);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 sum = acc[rw][mt] + red[0][mt][lane] + red[1][mt][lane] + red[2][mt][lane];
        if (!xv[mt]) continue;
        if (gridDim.y == 1) {
          V4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (T)(sum[i] + (float)bv[i]);
          *reinterpret_cast<V4*>(y + (long)(mt * 16 + r) * y_stride + nb) = o;
        } else {                                 // K splits: this split's slab of fp32 partial sums
          *reinterpret_cast<f32x4*>(ws + ((long)blockIdx.y * M + mt * 16 + r) * N + nb) = sum;
        }
      }
    }
  }
}

// y[m, n] = sum over the K splits' slabs (+ bias); one thread per four outputs
template <typename T>
__global__ __launch_bounds__(256) void cb4_finish_kernel(T* __restrict__ y, const float* __restrict__ ws,
                                                         const T* __restrict__ bias, int M, int N, int splits,
                                                         long y_stride) {
  using V4 = typename Vec<T>::v4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int nq = N / 4;
  if (i >= (long)M * nq) return;
  const int m = (int)(i / nq), n = (int)(i - (long)m * nq) * 4;
  f32x4 sum = *reinterpret_cast<const f32x4*>(ws + (long)m * N + n);
  for (int k = 1; k < splits; ++k) sum += *reinterpret_cast<const f32x4*>(ws + ((long)k * M + m) * N + n);
  V4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (T)(sum[j] + (bias ? (float)bias[n + j] : 0.0f));
  *reinterpret_cast<V4*>(y + (long)m * y_stride + n) = o;
}

// codes in the transposed-chunk layout, scales_t [K / G, N]
VDA_EXPORT int vda_cb4_skinny_gemm_post(void* y, const void* x, const void* codes, const void* scales_t,
                                        const void* table, const void* bias, int M, int N, int K, int G, int hi_first,
                                        long x_stride, long y_stride, int dtype, void* ws, size_t ws_floats,
                                        int splits, hipStream_t st) {
  if (M < 1 || M > 64 || N < 16 || N % 16 || K < 128 || K % 128 || K % G) return -1;
  if (!(G == 32 || G == 64 || (G >= 128 && G % 128 == 0))) return -1;
  const int mt = (M + 15) / 16;
  const int gq = G == 32 ? 1 : G == 64 ? 2 : 4;
  // rows per workgroup: as many as keep two workgroups per CU busy (512 of them) and the accumulators in registers
  // (from 8 rows on the activations every workgroup re-reads from L2 outweigh its codes unless it owns 64 rows:
  // narrow layers then take 64-row workgroups too and fill the chip with K splits instead)
  int rw = 1;
  // (layers under 16 MiB of codes finish in about ten microseconds either way: the second launch would cost them more
  // than the wider workgroups gain)
  const bool wide_rows = N % 32 == 0 && splits == 0 && ws && (size_t)N * K / 2 >= ((size_t)16 << 20);
  if (N % 64 == 0 && (N / 64 >= 512 || wide_rows)) rw = 4;
  else if (N % 32 == 0 && (N / 32 >= 512 || wide_rows)) rw = 2;
  // K splits (splits = 0: chosen here): enough workgroups for four per CU, at least 16 chunks each; every split writes
  // a slab of fp32 partial sums [M, N] and a second small kernel adds the slabs (no fences, no atomics: an agent-scope
  // fence writes back the whole L2 of an XCD, which cost more than the splits gained)
  const int blocks = N / (16 * rw), nchunk = K >> 7;
  int sk = splits;
  if (sk <= 0) {
    sk = 1;
    if (wide_rows && blocks < 256)
      while (sk < 8 && blocks * sk < 512 && nchunk / (sk * 2) >= 8) sk *= 2;
  }
  if (sk > nchunk) sk = nchunk;
  while (sk > 1 && (!ws || ws_floats < (size_t)sk * M * N)) --sk;
  dim3 grid(blocks, sk), block(256);
#define L(TT, MM, GG, RR) hipLaunchKernelGGL((cb4_skinny_post_kernel<TT, MM, GG, RR>), grid, block, 0, st, (TT*)y,      \
                                             (const TT*)x, (const uint8_t*)codes, (const float*)scales_t,             \
                                             (const float*)table, (const TT*)bias, M, N, K, G, hi_first, x_stride,    \
                                             y_stride, (float*)ws)
#define BYR(TT, MM, GG) do { if (rw == 4) L(TT, MM, GG, 4); else if (rw == 2) L(TT, MM, GG, 2);                       \
                             else L(TT, MM, GG, 1); } while (0)
#define BYM(TT, GG) do { if (mt == 1) BYR(TT, 1, GG); else if (mt == 2) BYR(TT, 2, GG);                                \
                         else if (mt == 3) BYR(TT, 3, GG); else BYR(TT, 4, GG); } while (0)
#define BY(TT) do { if (gq == 1) BYM(TT, 1); else if (gq == 2) BYM(TT, 2); else BYM(TT, 4); } while (0)
  if (dtype == VDA_BF16) BY(bf16_t);
  else if (dtype == VDA_F16) BY(f16_t);
  else return -3;
#undef BY
#undef BYR
#undef BYM
#undef L
  if (sk > 1) {
    dim3 fgrid((unsigned)(((long)M * (N / 4) + 255) / 256));
    if (dtype == VDA_BF16)
      hipLaunchKernelGGL(cb4_finish_kernel<bf16_t>, fgrid, block, 0, st, (bf16_t*)y, (const float*)ws,
                         (const bf16_t*)bias, M, N, sk, y_stride);
    else
      hipLaunchKernelGGL(cb4_finish_kernel<f16_t>, fgrid, block, 0, st, (f16_t*)y, (const float*)ws,
                         (const f16_t*)bias, M, N, sk, y_stride);
  }
  VDA_CHECK_LAUNCH();
  return 0;
}

// expansion of the transposed-chunk layout: stored dword d = 4 g + q of a chunk holds elements 32 q + 8 g .. + 7
template <typename T, bool HI_FIRST>
__global__ __launch_bounds__(256) void cb4_dequant_post_kernel(T* __restrict__ out, const uint32_t* __restrict__ codes,
                                                               const float* __restrict__ scales_t,
                                                               const float* __restrict__ table, long nwords, int N,
                                                               int K, int G) {
  using V8 = typename Vec<T>::v8;
  __shared__ float tab[16];
  if (threadIdx.x < 16) tab[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nwords) return;
  const int wpr = K / 8;
  const long n = i / wpr;
  const int j = (int)(i - n * wpr);
  const int d = j & 15;
  const int e = (j >> 4) * 128 + (d & 3) * 32 + (d >> 2) * 8;
  const float s = scales_t[(long)(e / G) * N + n];
  st16<V8>(out + n * K + e, cb4_expand<T, HI_FIRST>(codes[i], tab, s));
}

VDA_EXPORT int vda_cb4_dequant_post(void* out, const void* codes, const void* scales_t, const void* table, int N,
                                    int K, int G, int hi_first, int dtype, hipStream_t st) {
  if (N < 1 || K < 128 || K % 128 || G < 8 || G % 8 || K % G) return -1;
  const long nwords = (long)N * (K / 8);
  dim3 grid((unsigned)((nwords + 255) / 256)), block(256);
#define L(TT, HF) hipLaunchKernelGGL((cb4_dequant_post_kernel<TT, HF>), grid, block, 0, st, (TT*)out,                    \
                                     (const uint32_t*)codes, (const float*)scales_t, (const float*)table, nwords, N, K, G)
#define BY(TT) do { if (hi_first) L(TT, true); else L(TT, false); } while (0)
  if (dtype == VDA_BF16) BY(bf16_t);
  else if (dtype == VDA_F16) BY(f16_t);
  else return -3;
#undef BY
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

// out[n, 8j .. 8j+7] = table[codes of word j of row n] * scale; one thread per 4 bytes of codes, 16-byte stores
template <typename T, bool HI_FIRST>
__global__ __launch_bounds__(256) void cb4_dequant_kernel(T* __restrict__ out, const uint32_t* __restrict__ codes,
                                                          const float* __restrict__ scales,
                                                          const float* __restrict__ table, long nwords, int K, int G) {
  using V8 = typename Vec<T>::v8;
  __shared__ float tab[16];
  if (threadIdx.x < 16) tab[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nwords) return;
  const int wpr = K / 8;
  const long n = i / wpr;
  const int j = (int)(i - n * wpr);
  const float s = scales[n * (K / G) + (j * 8) / G];
  st16<V8>(out + i * 8, cb4_expand<T, HI_FIRST>(codes[i], tab, s));
}

VDA_EXPORT int vda_cb4_dequant(void* out, const void* codes, const void* scales, const void* table, int N, int K,
                               int G, int hi_first, int dtype, hipStream_t st) {
  if (N < 1 || K < 8 || K % 8 || G < 8 || G % 8 || K % G) return -1;
  const long nwords = (long)N * (K / 8);
  dim3 grid((unsigned)((nwords + 255) / 256)), block(256);
#define L(TT, HF) hipLaunchKernelGGL((cb4_dequant_kernel<TT, HF>), grid, block, 0, st, (TT*)out, (const uint32_t*)codes, \
                                     (const float*)scales, (const float*)table, nwords, K, G)
#define BY(TT) do { if (hi_first) L(TT, true); else L(TT, false); } while (0)
  if (dtype == VDA_BF16) BY(bf16_t);
  else if (dtype == VDA_F16) BY(f16_t);
  else return -3;
#undef BY
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}
