// LDS-tiled MFMA GEMM family for gfx950: the compute-bound regime (many rows per weight matrix).
//
//   vda_moe_tile_map      m-tile table (expert, first sorted position) from the expert offsets
//   vda_moe_align_large   counting sort of (token, slot) pairs by expert over many workgroups
//   vda_tile_gemm         out = epilogue(X . W^T) for one dense weight matrix or for the experts of a
//                         routed MoE layer over expert-sorted (token, slot) pairs
//
// One kernel template serves 16-bit operands (v_mfma_f32_16x16x32_bf16/f16) and e4m3 operands: the
// fp8 form issues v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales, the instruction that runs at
// twice the 16-bit MFMA rate on CDNA4 (the plain 16x16x32 fp8 form runs at the 16-bit rate).
//
// Tile: 256 weight rows x (32 * TN) activation rows x 128 bytes of K per step (128 e4m3 or 64 16-bit
// elements), 8 waves as 4 (weight rows) x 2 (activation rows), each wave 4 x TN accumulator tiles of
// 16 x 16.  The weights are the MFMA A operand and the activations the B operand, so a lane ends up with
// four consecutive output columns of one activation row (an 8-byte store) and the SiLU(gate) * up pair of
// a gated MLP sits in the same lane: a gate/up tile is 128 gate rows plus the 128 up rows of the same
// columns.  Both operands go global -> LDS with 16-byte global_load_lds (no staging registers) into a ring
// of three LDS stages: the fill of step kt + 2 is issued at the start of step kt and is still in flight
// when the step ends (counted s_waitcnt vmcnt + raw s_barrier), so a K step never waits for a load it
// has just issued; one workgroup (two waves per SIMD) per CU.
//
// LDS image: [row][8 chunks of 16 bytes], chunk c of row r stored at c ^ f(r),
//   f(r) = ((r >> 1) & 7) ^ ((((r >> 2) ^ (r >> 3)) & 1) << 1).
// Lane (r = lane & 15, g = lane >> 4) owns bytes [32 g, 32 g + 32) of its row in every K step, for both
// operands alike (any K order works as long as both operands use the same one): two ds_read_b128 of
// chunks 2g and 2g + 1.  ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, ..., i.e. rows
// {0-3, 12-15} at chunk q together with rows {4-11} at chunk q ^ 2; a 128-byte row is half a 256-byte bank
// row (row parity picks the half), and f makes the sixteen 16-byte slots of every group distinct.  The LDS
// side of global_load_lds is lane-linear, so the XOR is applied to the global source chunk instead.
//
// Grouped (MoE) work distribution is read from device memory (tile table), so nothing syncs with the
// host and the layer is hipGraph-capturable; the grid covers the worst-case tile count and surplus
// workgroups exit.  Blocks are numbered so that the blocks resident on one XCD at a time form an
// 8 x 8 patch of tiles (shared weight and activation panels hit in that XCD's L2).
//
// Scales (fp8): SC_TOKEN  out = acc * x_scale[row] * w_scale[expert]            (per-token x per-tensor)
//               SC_BLOCK  acc += partial(K step) * x_scale[row][kb] * w_scale[expert][nb][kb]
//                         (1 x 128 activation groups, 128 x 128 weight blocks: DeepSeek-V3 checkpoints)
// Counterpart of the reference's Triton fused_moe_kernel (vllm/model_executor/layers/fused_moe/
// fused_moe.py:267-476), w8a8_block_fp8_matmul (quantization/utils/fp8_utils.py:570) and
// moe_align_block_size (csrc/moe/moe_align_sum_kernels.cu:17-230).
#include "mfma_tile.h"

enum { EPI_PLAIN = 0, EPI_GATE_UP = 1, EPI_DOWN = 2 };
enum { SC_NONE = 0, SC_TOKEN = 1, SC_BLOCK = 2 };

struct TileGemmArgs {
  void* out;                 // [rows_out, N] 16-bit
  const void* x;             // activations, K contiguous
  const void* w;             // weights [E, W_rows, K], K contiguous
  const float* x_scale;      // SC_TOKEN: [x rows]; SC_BLOCK: [x rows, K / 128]
  const float* w_scale;      // SC_TOKEN: [E]; SC_BLOCK: [E, ceil(W_rows / 128), K / 128]
  const float* topk_w;       // EPI_DOWN: routing weight per pair
  const void* bias;          // EPI_PLAIN: optional [N], output dtype
  const int* sorted;         // grouped: pair ids sorted by expert
  const int* tile_expert;    // grouped: expert of m-tile i
  const int* tile_beg;       // grouped: first sorted position of m-tile i
  const int* tile_end;       // grouped: end of that expert's sorted range
  const int* num_tiles;      // grouped: number of valid m-tiles (device)
  int M;                     // dense: activation rows
  int N;                     // output columns (EPI_GATE_UP: I, the weight has 2 I rows)
  int K;                     // elements
  int top_k;
  int act;                   // EPI_GATE_UP: activation of the gate half (codes of gate_act)
  int m_tiles;               // grid bound on m-tiles
  int n_tiles;
  long x_stride;             // elements
  long out_stride;           // elements
  long w_expert_stride;      // elements
};

// TI: operand element type (bf16_t / f16_t / uint8_t = e4m3); TO: output type.
// 8 waves as 4 (weight rows) x 2 (activation rows); three LDS stages, the fill of step kt + 2 in flight
// across the barrier that ends step kt (counted s_waitcnt, raw s_barrier: a __syncthreads() would drain it).
template <typename TI, typename TO, int EPI, bool GROUPED, int SC, int TN>
__global__ __launch_bounds__(512, 1) void tile_gemm_kernel(const TileGemmArgs a) {
  using V4 = typename Vec<TO>::v4;
  constexpr int EB = sizeof(TI);             // bytes per operand element
  constexpr int XR = 32 * TN;                // activation rows per tile
  constexpr int WBYTES = 256 * 128, XBYTES = XR * 128, STAGE = WBYTES + XBYTES;
  constexpr int XROUNDS = XR / 64;           // 64 rows per fill round (512 threads x 16 bytes)
  constexpr int NG = 4 + XROUNDS;            // global -> LDS pieces per thread and K step
  static_assert(TN == 2 || TN == 4, "activation rows per tile: 64 or 128");
  __shared__ __attribute__((aligned(16))) uint8_t lds[3 * STAGE];

  // ---- which tile ----------------------------------------------------------------------------
  // block ids are dealt round-robin to the 8 XCDs; the 64 consecutive blocks an XCD receives form an
  // 8 (m) x 8 (n) patch of tiles
  const int bid = blockIdx.x;
  const int label = bid & 7, seq = bid >> 3;
  const int patch = (seq >> 6) * 8 + label, within = seq & 63;
  const int pn = (a.n_tiles + 7) >> 3;
  const int mi = (patch / pn) * 8 + (within & 7), ni = (patch % pn) * 8 + (within >> 3);
  if (ni >= a.n_tiles || mi >= a.m_tiles) return;
  int e = 0, beg = 0, cnt;
  if (GROUPED) {
    if (mi >= *a.num_tiles) return;
    e = a.tile_expert[mi];
    beg = a.tile_beg[mi];
    cnt = min(a.tile_end[mi] - beg, XR);
  } else {
    beg = mi * XR;
    cnt = min(a.M - beg, XR);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int n0 = ni * (EPI == EPI_GATE_UP ? 128 : 256);
  const int w_rows = EPI == EPI_GATE_UP ? 2 * a.N : a.N;
  const long kbytes = (long)a.K * EB;
  const uint8_t* wbase = reinterpret_cast<const uint8_t*>(a.w) + (long)e * a.w_expert_stride * EB;
  const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.x);

  // ---- staging roles: thread t of round j fills LDS row 64 j + (t >> 3), slot t & 7 -----------------
  const int srow = tid >> 3, slot = tid & 7;
  const uint8_t* wsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rho = 64 * j + srow;
    int row = EPI == EPI_GATE_UP ? (rho < 128 ? n0 + rho : a.N + n0 + rho - 128) : n0 + rho;
    row = min(row, w_rows - 1);
    wsrc[j] = wbase + (long)row * kbytes + ((slot ^ swz(rho)) << 4);
  }
  const uint8_t* xsrc[XROUNDS];
#pragma unroll
  for (int j = 0; j < XROUNDS; ++j) {
    const int tau = 64 * j + srow;
    const int pos = beg + min(tau, cnt - 1);
    long row = pos;
    if (GROUPED) {
      const int p = a.sorted[pos];
      row = EPI == EPI_GATE_UP ? p / a.top_k : p;
    }
    xsrc[j] = xbase + row * a.x_stride * EB + ((slot ^ swz(tau)) << 4);
  }
  const int nk = (int)(kbytes >> 7);
  auto stage = [&](int kt) {                  // ring slot kt % 3 <- K step min(kt, nk - 1) (a re-fill of the
    uint8_t* lw = lds + (kt % 3) * STAGE + wave * 1024;      // last step keeps the operation count uniform)
    uint8_t* lx = lw + WBYTES;
    const long ko = (long)min(kt, nk - 1) * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      glds16(wsrc[j] + ko, lw + j * 8192);
#pragma unroll
    for (int j = 0; j < XROUNDS; ++j)
      glds16(xsrc[j] + ko, lx + j * 8192);
  };

  // ---- compute roles -----------------------------------------------------------------------------------
  // weight m-tile mt (0..3) of this wave starts at LDS row: gate/up tiles hold 128 gate rows, then the 128 up
  // rows of the same columns, and a wave takes two gate tiles and the up tiles of the same 32 columns
  auto wrow = [&](int mt) { return EPI == EPI_GATE_UP ? (mt < 2 ? 32 * wr + 16 * mt : 128 + 32 * wr + 16 * (mt - 2))
                                                      : 64 * wr + 16 * mt; };
  const int f = swz(r);                        // tile rows start at multiples of 16: f depends on r only
  const int c0 = ((2 * g) ^ f) << 4, c1 = ((2 * g + 1) ^ f) << 4;
  int woff[4], xoff[TN];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) woff[mt] = (wrow(mt) + r) * 128;
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) xoff[nt] = WBYTES + (16 * TN * wc + 16 * nt + r) * 128;

  f32x4 acc[4][TN];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};

  // SC_BLOCK: activation-group scale rows of this lane's tokens, weight-block scale rows of its m-tiles
  const float* xs_row[TN];
  const float* ws_row[4];
  const int kblocks = a.K >> 7;
  if (SC == SC_BLOCK) {
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      const int tau = 16 * TN * wc + 16 * nt + r;
      const int pos = beg + min(tau, cnt - 1);
      long row = pos;
      if (GROUPED) {
        const int p = a.sorted[pos];
        row = EPI == EPI_GATE_UP ? p / a.top_k : p;
      }
      xs_row[nt] = a.x_scale + row * kblocks;
    }
    const int nblocks = (w_rows + 127) >> 7;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int rho = wrow(mt);
      int row = EPI == EPI_GATE_UP ? (rho < 128 ? n0 + rho : a.N + n0 + rho - 128) : n0 + rho;
      row = min(row, w_rows - 1);
      ws_row[mt] = a.w_scale + ((long)e * nblocks + (row >> 7)) * kblocks;
    }
  }

  stage(0);
  stage(1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG) : "memory");          // step 0 has landed, step 1 is in flight
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    stage(kt + 2);                             // into the slot read during step kt - 1 (all waves are past it)
    float sx[TN], sw[4];
    if (SC == SC_BLOCK) {
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) sx[nt] = xs_row[nt][kt];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) sw[mt] = ws_row[mt][kt];
    }
    const uint8_t* base = lds + (kt % 3) * STAGE;
    i32x4 wa[4][2], xb[TN][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      wa[mt][0] = *reinterpret_cast<const i32x4*>(base + woff[mt] + c0);
      wa[mt][1] = *reinterpret_cast<const i32x4*>(base + woff[mt] + c1);
    }
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      xb[nt][0] = *reinterpret_cast<const i32x4*>(base + xoff[nt] + c0);
      xb[nt][1] = *reinterpret_cast<const i32x4*>(base + xoff[nt] + c1);
    }
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if (SC == SC_BLOCK) {
          const f32x4 p = mma(wa[mt][0], wa[mt][1], xb[nt][0], xb[nt][1], f32x4{0, 0, 0, 0}, TI{});
          const float s = sw[mt] * sx[nt];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[mt][nt][i] += p[i] * s;
        } else {
          acc[mt][nt] = mma(wa[mt][0], wa[mt][1], xb[nt][0], xb[nt][1], acc[mt][nt], TI{});
        }
      }
    // step kt + 1 must have landed before anyone reads it; the fill of step kt + 2 stays in flight.
    // (SC_BLOCK reads its scales with ordinary loads: they are older than that fill and done by now.)
    // This step's LDS reads are retired too (lgkmcnt): the next step's fill overwrites the slot they read.
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NG) : "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: lane holds out[token tau(nt)][n(mt) + 4 g + i] -----------------------------------------
  const float wsc = SC == SC_TOKEN ? a.w_scale[e] : 1.f;
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int tau = 16 * TN * wc + 16 * nt + r;
    if (tau >= cnt) continue;
    long orow = beg + tau;
    float s = wsc;
    float tw = 1.f;
    if (GROUPED) {
      const int p = a.sorted[beg + tau];
      orow = p;
      if (SC == SC_TOKEN) s *= a.x_scale[EPI == EPI_GATE_UP ? p / a.top_k : p];
      if (EPI == EPI_DOWN) tw = a.topk_w[p];
    } else if (SC == SC_TOKEN) {
      s *= a.x_scale[orow];
    }
    TO* op = reinterpret_cast<TO*>(a.out) + orow * a.out_stride;
    if (EPI == EPI_GATE_UP) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int n = n0 + 32 * wr + 16 * mt + 4 * g;
        V4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gte = acc[mt][nt][i] * s, up = acc[mt + 2][nt][i] * s;
          o[i] = (TO)(gate_act(gte, a.act) * up);
        }
        if (n + 3 < a.N) *reinterpret_cast<V4*>(op + n) = o;
        else
          for (int i = 0; i < 4; ++i) if (n + i < a.N) op[n + i] = o[i];
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int n = n0 + 64 * wr + 16 * mt + 4 * g;
        V4 o;
        V4 bv = V4{};
        if (EPI == EPI_PLAIN && a.bias && n + 3 < a.N) bv = *reinterpret_cast<const V4*>(reinterpret_cast<const TO*>(a.bias) + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[mt][nt][i] * s * tw;
          if (EPI == EPI_PLAIN && a.bias) v += n + 3 < a.N ? (float)bv[i] : (n + i < a.N ? (float)reinterpret_cast<const TO*>(a.bias)[n + i] : 0.f);
          o[i] = (TO)v;
        }
        if (n + 3 < a.N) *reinterpret_cast<V4*>(op + n) = o;
        else
          for (int i = 0; i < 4; ++i) if (n + i < a.N) op[n + i] = o[i];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// m-tile table: for expert e with c tokens, ceil(c / rows) tiles.  One workgroup, E <= 1024.
__global__ __launch_bounds__(1024) void moe_tile_map_kernel(int* __restrict__ tile_expert, int* __restrict__ tile_beg,
                                                            int* __restrict__ tile_end, int* __restrict__ num_tiles,
                                                            const int* __restrict__ offsets, int E, int rows,
                                                            int max_tiles) {
  __shared__ int pre[1025];
  const int tid = threadIdx.x;
  int nt = 0, b = 0, en = 0;
  if (tid < E) { b = offsets[tid]; en = offsets[tid + 1]; nt = (en - b + rows - 1) / rows; }
  pre[tid + 1] = nt;
  if (tid == 0) pre[0] = 0;
  __syncthreads();
  // inclusive scan over 1024 entries (Hillis-Steele, 10 rounds)
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = tid + 1 > d ? pre[tid + 1 - d] : 0;
    __syncthreads();
    pre[tid + 1] += v;
    __syncthreads();
  }
  if (tid < E) {
    const int first = pre[tid];
    for (int i = 0; i < nt; ++i) {
      const int t = first + i;
      if (t < max_tiles) { tile_expert[t] = tid; tile_beg[t] = b + i * rows; tile_end[t] = en; }
    }
  }
  if (tid == 0) *num_tiles = min(pre[E], max_tiles);
}

VDA_EXPORT int vda_moe_tile_map(int* tile_expert, int* tile_beg, int* tile_end, int* num_tiles, const int* offsets,
                                int E, int rows, int max_tiles, hipStream_t st) {
  if (E < 1 || E > 1024 || rows < 1 || max_tiles < 1) return -1;
  hipLaunchKernelGGL(moe_tile_map_kernel, dim3(1), dim3(1024), 0, st, tile_expert, tile_beg, tile_end, num_tiles,
                     offsets, E, rows, max_tiles);
  VDA_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Counting sort over many workgroups.  counts[E] must be zero on entry (the caller's memset node);
// pass 1 histograms, pass 2 (one workgroup) turns counts into offsets and cursors, pass 3 reserves one
// range per (workgroup, expert) with a single global atomic and scatters.  Order inside an expert is
// not fixed between runs; every output row is computed independently of its position, so results are.
constexpr int ALIGN_CHUNK = 2048;       // pairs per workgroup

__global__ __launch_bounds__(256) void moe_hist_kernel(int* __restrict__ counts, const int* __restrict__ ids, int n, int E) {
  __shared__ int h[1024];
  for (int i = threadIdx.x; i < E; i += 256) h[i] = 0;
  __syncthreads();
  const int base = blockIdx.x * ALIGN_CHUNK;
  for (int p = base + threadIdx.x; p < min(base + ALIGN_CHUNK, n); p += 256) {
    const int e = ids[p];
    if (e >= 0 && e < E) atomicAdd(&h[e], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < E; i += 256) if (h[i]) atomicAdd(&counts[i], h[i]);
}

__global__ __launch_bounds__(1024) void moe_scan_kernel(int* __restrict__ offsets, int* __restrict__ cursor,
                                                        const int* __restrict__ counts, int E) {
  __shared__ int pre[1025];
  const int tid = threadIdx.x;
  pre[tid + 1] = tid < E ? counts[tid] : 0;
  if (tid == 0) pre[0] = 0;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = tid + 1 > d ? pre[tid + 1 - d] : 0;
    __syncthreads();
    pre[tid + 1] += v;
    __syncthreads();
  }
  if (tid < E) { offsets[tid] = pre[tid]; cursor[tid] = pre[tid]; }
  if (tid == 0) offsets[E] = pre[E];
}

__global__ __launch_bounds__(256) void moe_scatter_kernel(int* __restrict__ sorted, int* __restrict__ cursor,
                                                          const int* __restrict__ ids, int n, int E) {
  __shared__ int h[1024];
  __shared__ int start[1024];
  for (int i = threadIdx.x; i < E; i += 256) h[i] = 0;
  __syncthreads();
  const int base = blockIdx.x * ALIGN_CHUNK, lim = min(base + ALIGN_CHUNK, n);
  for (int p = base + threadIdx.x; p < lim; p += 256) {
    const int e = ids[p];
    if (e >= 0 && e < E) atomicAdd(&h[e], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < E; i += 256) { start[i] = h[i] ? atomicAdd(&cursor[i], h[i]) : 0; h[i] = 0; }
  __syncthreads();
  for (int p = base + threadIdx.x; p < lim; p += 256) {
    const int e = ids[p];
    if (e >= 0 && e < E) sorted[start[e] + atomicAdd(&h[e], 1)] = p;
  }
}

VDA_EXPORT int vda_moe_align_large(int* sorted, int* offsets, int* counts, int* cursor, const int* ids, int n, int E,
                                   hipStream_t st) {
  if (E < 1 || E > 1024 || n < 0) return -1;
  hipError_t err = hipMemsetAsync(counts, 0, sizeof(int) * E, st);
  if (err != hipSuccess) return (int)err;
  const int blocks = (n + ALIGN_CHUNK - 1) / ALIGN_CHUNK;
  if (blocks > 0) hipLaunchKernelGGL(moe_hist_kernel, dim3(blocks), dim3(256), 0, st, counts, ids, n, E);
  hipLaunchKernelGGL(moe_scan_kernel, dim3(1), dim3(1024), 0, st, offsets, cursor, counts, E);
  if (blocks > 0) hipLaunchKernelGGL(moe_scatter_kernel, dim3(blocks), dim3(256), 0, st, sorted, cursor, ids, n, E);
  VDA_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// in_dtype: operand type (VDA_BF16 / VDA_F16 / VDA_FP8); out_dtype VDA_BF16 / VDA_F16.
// rows_per_tile: 128 or 64 activation rows per tile.  Grouped when sorted != nullptr.
template <typename TI, typename TO, int SC>
static int launch_tile_gemm(const TileGemmArgs& a, int epi, bool grouped, int rows_per_tile, hipStream_t st) {
  const int pm = (a.m_tiles + 7) / 8, pn = (a.n_tiles + 7) / 8;
  const dim3 grid((unsigned)((pm * pn + 7) / 8 * 8 * 64)), block(512);
#define LAUNCH(EPI, GR, TN) hipLaunchKernelGGL((tile_gemm_kernel<TI, TO, EPI, GR, SC, TN>), grid, block, 0, st, a)
#define BY_TN(EPI, GR) do { if (rows_per_tile == 128) LAUNCH(EPI, GR, 4); else LAUNCH(EPI, GR, 2); } while (0)
  if (grouped) {
    if (epi == EPI_GATE_UP) BY_TN(EPI_GATE_UP, true);
    else if (epi == EPI_DOWN) BY_TN(EPI_DOWN, true);
    else return -4;
  } else {
    if (epi != EPI_PLAIN) return -4;
    BY_TN(EPI_PLAIN, false);
  }
#undef BY_TN
#undef LAUNCH
  return 0;
}

VDA_EXPORT int vda_tile_gemm(void* out, const void* x, const void* w, const float* x_scale, const float* w_scale,
                             const float* topk_w, const void* bias, const int* sorted, const int* tile_expert,
                             const int* tile_beg, const int* tile_end, const int* num_tiles, int M, int N, int K,
                             int top_k, int max_m_tiles, long x_stride, long out_stride, long w_expert_stride,
                             int epi, int scale_mode, int rows_per_tile, int in_dtype, int out_dtype, int act,
                             hipStream_t st) {
  const bool grouped = sorted != nullptr;
  const int eb = in_dtype == VDA_FP8 ? 1 : 2;
  if (rows_per_tile != 128 && rows_per_tile != 64) return -1;
  if (N < 1 || K < 1 || ((long)K * eb) % 128) return -1;
  if ((x_stride * eb) % 16 || (w_expert_stride * eb) % 16) return -1;       // 16-byte global_load_lds
  if (epi == EPI_GATE_UP && N % 64) return -1;
  if (N % 4 && out_stride % 4) return -1;
  if (scale_mode != SC_NONE && in_dtype != VDA_FP8) return -1;
  if (scale_mode != SC_NONE && (!x_scale || !w_scale)) return -1;
  if (scale_mode == SC_BLOCK && K % 128) return -1;
  if (grouped && (!tile_expert || !tile_beg || !tile_end || !num_tiles || max_m_tiles < 1)) return -1;
  if (!grouped && M < 1) return M == 0 ? 0 : -1;
  TileGemmArgs a;
  a.out = out; a.x = x; a.w = w; a.x_scale = x_scale; a.w_scale = w_scale; a.topk_w = topk_w; a.bias = bias;
  a.sorted = sorted; a.tile_expert = tile_expert; a.tile_beg = tile_beg; a.tile_end = tile_end; a.num_tiles = num_tiles;
  a.M = M; a.N = N; a.K = K; a.top_k = top_k > 0 ? top_k : 1; a.act = act;
  a.m_tiles = grouped ? max_m_tiles : (M + rows_per_tile - 1) / rows_per_tile;
  a.n_tiles = epi == EPI_GATE_UP ? (N + 127) / 128 : (N + 255) / 256;
  a.x_stride = x_stride; a.out_stride = out_stride; a.w_expert_stride = w_expert_stride;
  int rc;
#define BY_SC(TI, TO) (scale_mode == SC_NONE ? launch_tile_gemm<TI, TO, SC_NONE>(a, epi, grouped, rows_per_tile, st) \
    : scale_mode == SC_TOKEN ? launch_tile_gemm<TI, TO, SC_TOKEN>(a, epi, grouped, rows_per_tile, st) \
    : launch_tile_gemm<TI, TO, SC_BLOCK>(a, epi, grouped, rows_per_tile, st))
  if (in_dtype == VDA_FP8) {
    if (scale_mode == SC_NONE) return -1;
    if (out_dtype == VDA_BF16) rc = BY_SC(uint8_t, bf16_t);
    else if (out_dtype == VDA_F16) rc = BY_SC(uint8_t, f16_t);
    else return -3;
  } else if (in_dtype == VDA_BF16 && out_dtype == VDA_BF16) {
    rc = launch_tile_gemm<bf16_t, bf16_t, SC_NONE>(a, epi, grouped, rows_per_tile, st);
  } else if (in_dtype == VDA_F16 && out_dtype == VDA_F16) {
    rc = launch_tile_gemm<f16_t, f16_t, SC_NONE>(a, epi, grouped, rows_per_tile, st);
  } else return -3;
#undef BY_SC
  if (rc) return rc;
  VDA_CHECK_LAUNCH();
  return 0;
}
