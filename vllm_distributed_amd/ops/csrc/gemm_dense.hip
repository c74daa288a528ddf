// Dense 16-bit GEMM for gfx950, every M: out[M, N] = epilogue(X[M, K] . W[N, K]^T).
//
//   vda_dense_gemm          one launch (+ a reduction launch when K is split)
//
// Structure: a workgroup of 8 waves owns a tile of 64 WT weight rows x 32 XT activation rows and walks K in
// units of 32 elements (64 bytes per row).  The waves form two groups of four (waves 0-3 / 4-7: the two
// waves that share a SIMD are in different groups); a group owns one half of the tile's activation rows and
// all of its weight rows, wave wn of a group the weight rows [16 WT wn, 16 WT (wn + 1)).  The groups run the
// same loop half a step apart:
//
//        barrier #2u      R(u)                 barrier #2u+1     C(u)            barrier #2u+2
//   group 0:   issue LDS-DMA of unit u+D-1, read unit u's fragments  |  4 WT XT / 4 MFMAs from registers |
//   group 1:   ... MFMAs of unit u-1 ...                              |  R(u)                              | C(u)
//
// so on every SIMD one wave issues MFMAs back to back from registers while its partner issues the
// global -> LDS copies of a later unit and the ds_read_b128 of its next fragments: the matrix pipe never
// waits for LDS or for address arithmetic.  Units live in a ring of D LDS slots filled by 16-byte
// global_load_lds, D - 1 units ahead; a counted s_waitcnt vmcnt leaves D - 2 units in flight across every
// barrier (raw s_barrier: a __syncthreads() would drain the queue).
//   read-after-write: a thread waits for its own copies of unit u+1 at the end of R(u), before a barrier
//     that every reader of unit u+1 passes before its R(u+1);
//   write-after-read: the copy of unit u+D-1 lands in the slot of unit u-1, whose last readers (group 1 in
//     its R(u-1)) drained their LDS reads (lgkmcnt(0)) before barrier #2u, which precedes the issue.
//
// LDS image of a unit: [row][4 chunks of 16 bytes], chunk c of row r stored at position c ^ f(r & 15),
// f(r) = (-(r >> 2)) & 3: the 16 lanes of a ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) then touch 16
// different 16-byte slots of the 256-byte bank row.  The LDS side of global_load_lds is lane-linear, so the
// XOR is applied to the global source chunk.
//
// The weights are the MFMA A operand and the activations the B operand (v_mfma_f32_16x16x32): a lane ends
// with four consecutive output columns of one activation row, and with GATE_UP the SiLU(gate) * up pair of a
// gated MLP sits in the same lane (a tile holds 32 WT gate rows and the 32 WT up rows of the same columns).
// Epilogues: bias, residual add, SiLU(gate) * up; with K split over workgroups the tile goes to an fp32
// workspace and the reduction kernel applies the epilogue.
//
// Takes the place of the BLAS library call behind the reference's UnquantizedLinearMethod
// (vllm/model_executor/layers/linear.py:184-226, dispatch vllm/model_executor/layers/utils.py:66-116) and
// of its skinny GEMMs above four rows (csrc/rocm/skinny_gemms.cu:294-1290).
#include "gemm_dense.h"

// Diagnostic instantiations (STAMP) sum, per wave, the shader cycles spent in each part of a loop step into a
// buffer of their own (never an output): [0] issue of copies and fragment reads until the reads are back,
// [1] wait for the copies, [2] barrier before the MFMAs, [3] MFMAs, [4] barrier after them, [5] steps.
__device__ __forceinline__ unsigned long long dg_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

template <typename T, int EPI, int WT, int XT, bool STAMP = false>
__global__ __launch_bounds__(512, 1) void dense_gemm_kernel(const DenseGemmArgs a) {
  unsigned long long rt0 = 0, rt1 = 0, rt2 = 0;
  if (STAMP) rt0 = __builtin_amdgcn_s_memrealtime();
  using V4 = typename Vec<T>::v4;
  using V8 = typename Vec<T>::v8;
  constexpr int WR = 64 * WT;                      // weight rows per tile
  constexpr int XR = 32 * XT;                      // activation rows per tile
  constexpr int WROUNDS = WR / 128 > 0 ? WR / 128 : 1, XROUNDS = XR / 128 > 0 ? XR / 128 : 1;
  constexpr int WB = WROUNDS * 8192, XB = XROUNDS * 8192, UNIT = WB + XB;
  constexpr int NG = WROUNDS + XROUNDS;            // global -> LDS pieces per thread and unit
  constexpr int D = UNIT <= 16384 ? 8 : UNIT <= 24576 ? 6 : 5;       // ring slots (<= 160 KiB, all of the LDS)
  constexpr int AHEAD = NG * (D - 2);              // copies left in flight across a barrier
  static_assert(WT == 2 || WT == 4, "128 or 256 weight rows per tile");
  static_assert(XT == 1 || XT == 2 || XT == 4 || XT == 8, "32 .. 256 activation rows per tile");
  static_assert(AHEAD < 64, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) uint8_t lds[D * UNIT];

  // ---- which tile, which K range --------------------------------------------------------------------
  int mi, ni, sp;
  if (a.m_tiles == 1) {                            // decode: consecutive blocks stream different weight tiles
    mi = 0; ni = blockIdx.x % a.n_tiles; sp = blockIdx.x / a.n_tiles;
  } else {                                         // the 64 consecutive blocks an XCD receives form an 8 x 8 patch
    const int bid = blockIdx.x, label = bid & 7, seq = bid >> 3;
    const int patch = (seq >> 6) * 8 + label, within = seq & 63;
    const int pn = (a.n_tiles + 7) >> 3;
    mi = (patch / pn) * 8 + (within & 7); ni = (patch % pn) * 8 + (within >> 3); sp = blockIdx.y;
  }
  if (ni >= a.n_tiles || mi >= a.m_tiles || sp >= a.splits) return;
  const int nu_all = a.K >> 5;
  const int u0 = sp * a.units_per_split;
  const int nun = min(a.units_per_split, nu_all - u0);
  if (nun <= 0) return;
  const int beg = mi * XR, cnt = min(a.M - beg, XR);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int grp = wave >> 2, wn = wave & 3;
  constexpr int NCOL = EPI == DG_GATE_UP ? WR / 2 : WR;         // output columns per tile
  const int n0 = ni * NCOL;
  const int w_rows = EPI == DG_GATE_UP ? 2 * a.N : a.N;
  const unsigned kbytes = (unsigned)a.K * 2u;
  const uint8_t* wbase = reinterpret_cast<const uint8_t*>(a.w) + (long)u0 * 64;
  const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.x) + (long)u0 * 64;

  // ---- staging roles: in round j, lane L of wave w fills LDS row 128 j + 16 w + (L >> 2), position L & 3 --
  unsigned voff_w[WROUNDS], voff_x[XROUNDS];       // byte offsets from wbase / xbase (host: both < 2^32)
  {
    const int lrow = 16 * wave + (lane >> 2), pos = lane & 3;
#pragma unroll
    for (int j = 0; j < WROUNDS; ++j) {
      const int rho = min(128 * j + lrow, WR - 1);
      int row = EPI == DG_GATE_UP ? (rho < WR / 2 ? n0 + rho : a.N + n0 + rho - WR / 2) : n0 + rho;
      row = min(row, w_rows - 1);
      voff_w[j] = (unsigned)row * kbytes + ((pos ^ swz64(rho & 15)) << 4);
    }
#pragma unroll
    for (int j = 0; j < XROUNDS; ++j) {
      const int tau = 128 * j + lrow;
      const int row = beg + min(tau, cnt - 1);
      voff_x[j] = (unsigned)row * (unsigned)(a.x_stride * 2) + ((pos ^ swz64(tau & 15)) << 4);
    }
  }
  auto issue = [&](int u, int slot) {              // ring slot <- unit min(u, nun - 1): a re-fill of the last
    const long ko = (long)min(u, nun - 1) * 64;    // unit keeps the operation count uniform
    const uint8_t* wb = wbase + ko;
    const uint8_t* xb = xbase + ko;
    uint8_t* lw = lds + slot * UNIT + wave * 1024;
#pragma unroll
    for (int j = 0; j < WROUNDS; ++j) glds16(wb + voff_w[j], lw + j * 8192);
#pragma unroll
    for (int j = 0; j < XROUNDS; ++j) glds16(xb + voff_x[j], lw + WB + j * 8192);
  };

  // ---- compute roles ------------------------------------------------------------------------------------
  // weight fragment mt of this wave starts at LDS row wrow(mt); with GATE_UP the first WT / 2 fragments are
  // gate rows and the others the up rows of the same columns
  constexpr int H = WT / 2;
  const int cpos = (g ^ swz64(r)) << 4;
  const int wl = ((EPI == DG_GATE_UP ? 16 * H * wn : 16 * WT * wn) + r) * 64 + cpos;
  const int xl = WB + (16 * XT * grp + r) * 64 + cpos;
  auto wofs = [&](int mt) { return EPI == DG_GATE_UP ? (mt < H ? mt * 1024 : (WR / 2) * 64 + (mt - H) * 1024) : mt * 1024; };

  f32x4 acc[WT][XT];
#pragma unroll
  for (int mt = 0; mt < WT; ++mt)
#pragma unroll
    for (int nt = 0; nt < XT; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};

  // ---- pipeline ---------------------------------------------------------------------------------------
#pragma unroll
  for (int u = 0; u < D - 1; ++u) issue(u, u);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD) : "memory");        // unit 0 has landed
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();                         // group 1 runs half a step behind
  int rs = 0, wsl = D - 1;
  unsigned long long st[5] = {0, 0, 0, 0, 0}, t0 = 0, t1;
  if (STAMP) { t0 = dg_stamp(); rt1 = __builtin_amdgcn_s_memrealtime(); }
  for (int u = 0; u < nun; ++u) {
    const uint8_t* base = lds + rs * UNIT;
    V8 wf[WT], xf[XT];
#pragma unroll
    for (int nt = 0; nt < XT; ++nt) xf[nt] = *reinterpret_cast<const V8*>(base + xl + nt * 1024);
#pragma unroll
    for (int mt = 0; mt < WT; ++mt) wf[mt] = *reinterpret_cast<const V8*>(base + wl + wofs(mt));
    __builtin_amdgcn_sched_barrier(0);             // fragment reads first: their latency runs under the copy issue
    issue(u + D - 1, wsl);
    if (STAMP) { t1 = dg_stamp(); st[0] += t1 - t0; t0 = t1; }
    // own copies of unit u + 1 have landed (D - 2 units stay in flight); this unit's LDS reads are done
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AHEAD) : "memory");
    if (STAMP) { t1 = dg_stamp(); st[1] += t1 - t0; t0 = t1; }
    __builtin_amdgcn_s_barrier();
    if (STAMP) { t1 = dg_stamp(); st[2] += t1 - t0; t0 = t1; }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nt = 0; nt < XT; ++nt)
#pragma unroll
      for (int mt = 0; mt < WT; ++mt) acc[mt][nt] = mfma16(wf[mt], xf[nt], acc[mt][nt]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    if (STAMP) { t1 = dg_stamp(); st[3] += t1 - t0; t0 = t1; }
    __builtin_amdgcn_s_barrier();
    if (STAMP) { t1 = dg_stamp(); st[4] += t1 - t0; t0 = t1; }
    rs = rs + 1 == D ? 0 : rs + 1;
    wsl = wsl + 1 == D ? 0 : wsl + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  if (STAMP) {
#pragma unroll
    for (int mt = 0; mt < WT; ++mt)                 // the diagnostic build writes no output: keep the MFMAs alive
#pragma unroll
      for (int nt = 0; nt < XT; ++nt) asm volatile("" ::"v"(acc[mt][nt]));
    rt2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && blockIdx.x < 64) {
      unsigned long long* d = reinterpret_cast<unsigned long long*>(a.ws) + (blockIdx.x * 8 + wave) * 6;
      for (int i = 0; i < 5; ++i) d[i] = st[i];
      d[5] = (unsigned long long)nun;
    }
    if (tid == 0) {            // per workgroup: start, loop start, loop end (100 MHz ticks), XCC id
      unsigned long long* d = reinterpret_cast<unsigned long long*>(a.ws) + 64 * 8 * 6 + (long)blockIdx.x * 4;
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      d[0] = rt0; d[1] = rt1; d[2] = rt2; d[3] = xcc & 15;
    }
    return;
  }

  dense_epilogue<T, EPI, WT, XT>(a, acc, sp, beg, cnt, n0, r, g, grp, wn);
}

// ---------------------------------------------------------------------------------------------------------
// Compute-bound form (many activation rows): 256 x 256 tile, K walked in units of 64 elements so that every
// LDS-DMA instruction copies 8 rows of one full 128-byte line each (a 64-byte row segment costs the address
// path as much as a whole line).  Two LDS slots of 64 KiB; a unit is consumed in two halves of K = 32:
//
//   R0(u): issue the copies of unit u + 1 (8 per thread), read the K-half-0 fragments of unit u
//   C0(u): 32 MFMAs           R1(u): read the K-half-1 fragments, wait for the copies of unit u + 1
//   C1(u): 32 MFMAs
//
// with the two wave groups one slot apart as above.  Write-after-read: unit u + 1 overwrites the slot of unit
// u - 1, last read by group 1 in its R1(u - 1), which drains its LDS reads before the barrier that precedes
// group 0's R0(u).  Read-after-write: every thread waits for its own copies (vmcnt(0)) at the end of its
// R1(u), before a barrier that every reader of unit u + 1 passes first.
// LDS image: [row][8 chunks of 16 bytes], chunk c of row r at position c ^ ((r >> 1) & 7); lane (r, g) reads
// chunk 4 h + g of its row in K half h: conflict-free for the ds_read_b128 lane groups.
template <typename T, int EPI, bool STAMP = false>
__global__ __launch_bounds__(512, 1) void dense_gemm_k64_kernel(const DenseGemmArgs a) {
  unsigned long long rt0 = 0, rt1 = 0, rt2 = 0;
  if (STAMP) rt0 = __builtin_amdgcn_s_memrealtime();
  using V8 = typename Vec<T>::v8;
  constexpr int WT = 4, XT = 8, WR = 256, XR = 256, H = 2;
  constexpr int WB = WR * 128, UNIT = WB + XR * 128;
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * UNIT];

  int mi, ni, sp;
  if (a.m_tiles == 1) {
    mi = 0; ni = blockIdx.x % a.n_tiles; sp = blockIdx.x / a.n_tiles;
  } else {
    const int bid = blockIdx.x, label = bid & 7, seq = bid >> 3;
    const int patch = (seq >> 6) * 8 + label, within = seq & 63;
    const int pn = (a.n_tiles + 7) >> 3;
    mi = (patch / pn) * 8 + (within & 7); ni = (patch % pn) * 8 + (within >> 3); sp = blockIdx.y;
  }
  if (ni >= a.n_tiles || mi >= a.m_tiles || sp >= a.splits) return;
  const int nu_all = a.K >> 6;                     // units of 64 elements (host: K % 64 == 0)
  const int ups = a.units_per_split >> 1;          // the host counts units of 32
  const int u0 = sp * ups;
  const int nun = min(ups, nu_all - u0);
  if (nun <= 0) return;
  const int beg = mi * XR, cnt = min(a.M - beg, XR);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int grp = wave >> 2, wn = wave & 3;
  constexpr int NCOL = EPI == DG_GATE_UP ? WR / 2 : WR;
  const int n0 = ni * NCOL;
  const int w_rows = EPI == DG_GATE_UP ? 2 * a.N : a.N;
  const unsigned kbytes = (unsigned)a.K * 2u;
  const uint8_t* wbase = reinterpret_cast<const uint8_t*>(a.w) + (long)u0 * 128;
  const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.x) + (long)u0 * 128;

  // staging roles: in round j, lane L of wave w fills LDS row 64 j + 8 w + (L >> 3), position L & 7
  unsigned voff_w[4], voff_x[4];
  {
    const int lrow = 8 * wave + (lane >> 3), pos = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rho = 64 * j + lrow;
      int row = EPI == DG_GATE_UP ? (rho < WR / 2 ? n0 + rho : a.N + n0 + rho - WR / 2) : n0 + rho;
      row = min(row, w_rows - 1);
      voff_w[j] = (unsigned)row * kbytes + ((pos ^ ((rho >> 1) & 7)) << 4);
      const int xrow = beg + min(rho, cnt - 1);
      voff_x[j] = (unsigned)xrow * (unsigned)(a.x_stride * 2) + ((pos ^ ((rho >> 1) & 7)) << 4);
    }
  }
  auto issue = [&](int u, int slot) {
    const long ko = (long)min(u, nun - 1) * 128;
    const uint8_t* wb = wbase + ko;
    const uint8_t* xb = xbase + ko;
    uint8_t* lw = lds + slot * UNIT + wave * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(wb + voff_w[j], lw + j * 8192);
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(xb + voff_x[j], lw + WB + j * 8192);
  };

  const int fr = (r >> 1) & 7;
  const int cpos = (g ^ fr) << 4;                  // K half 0; half 1 is cpos ^ 64
  const int wl = ((EPI == DG_GATE_UP ? 32 * wn : 64 * wn) + r) * 128;
  const int xl = WB + (128 * grp + r) * 128;
  auto wofs = [&](int mt) { return EPI == DG_GATE_UP ? (mt < H ? mt * 2048 : (WR / 2) * 128 + (mt - H) * 2048) : mt * 2048; };

  f32x4 acc[WT][XT];
#pragma unroll
  for (int mt = 0; mt < WT; ++mt)
#pragma unroll
    for (int nt = 0; nt < XT; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();
  unsigned long long st[5] = {0, 0, 0, 0, 0}, t0 = 0, t1;
  if (STAMP) { t0 = dg_stamp(); rt1 = __builtin_amdgcn_s_memrealtime(); }
  for (int u = 0; u < nun; ++u) {
    const uint8_t* base = lds + (u & 1) * UNIT;
    V8 wf[WT], xf[XT];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cp = h ? cpos ^ 64 : cpos;
#pragma unroll
      for (int nt = 0; nt < XT; ++nt) xf[nt] = *reinterpret_cast<const V8*>(base + xl + cp + nt * 2048);
#pragma unroll
      for (int mt = 0; mt < WT; ++mt) wf[mt] = *reinterpret_cast<const V8*>(base + wl + cp + wofs(mt));
      __builtin_amdgcn_sched_barrier(0);           // fragment reads first: their latency runs under the copy issue
      if (h == 0) issue(u + 1, (u + 1) & 1);
      if (STAMP) { t1 = dg_stamp(); st[0] += t1 - t0; t0 = t1; }
      if (h == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (STAMP) { t1 = dg_stamp(); st[1] += t1 - t0; t0 = t1; }
      __builtin_amdgcn_s_barrier();
      if (STAMP) { t1 = dg_stamp(); st[2] += t1 - t0; t0 = t1; }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int nt = 0; nt < XT; ++nt)
#pragma unroll
        for (int mt = 0; mt < WT; ++mt) acc[mt][nt] = mfma16(wf[mt], xf[nt], acc[mt][nt]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      if (STAMP) { t1 = dg_stamp(); st[3] += t1 - t0; t0 = t1; }
      __builtin_amdgcn_s_barrier();
      if (STAMP) { t1 = dg_stamp(); st[4] += t1 - t0; t0 = t1; }
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  if (STAMP) {
#pragma unroll
    for (int mt = 0; mt < WT; ++mt)                 // the diagnostic build writes no output: keep the MFMAs alive
#pragma unroll
      for (int nt = 0; nt < XT; ++nt) asm volatile("" ::"v"(acc[mt][nt]));
    rt2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && blockIdx.x < 64) {
      unsigned long long* d = reinterpret_cast<unsigned long long*>(a.ws) + (blockIdx.x * 8 + wave) * 6;
      for (int i = 0; i < 5; ++i) d[i] = st[i];
      d[5] = (unsigned long long)(2 * nun);
    }
    if (tid == 0) {            // per workgroup: start, loop start, loop end (100 MHz ticks), XCC id
      unsigned long long* d = reinterpret_cast<unsigned long long*>(a.ws) + 64 * 8 * 6 + (long)blockIdx.x * 4;
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      d[0] = rt0; d[1] = rt1; d[2] = rt2; d[3] = xcc & 15;
    }
    return;
  }
  dense_epilogue<T, EPI, WT, XT>(a, acc, sp, beg, cnt, n0, r, g, grp, wn);
}

template <typename T, int EPI>
static int launch_dense_gemm(const DenseGemmArgs& a, int wt, int xt, bool k64, hipStream_t st) {
  dim3 grid;
  if (a.m_tiles == 1) grid = dim3((unsigned)(a.n_tiles * a.splits));
  else {
    const int pm = (a.m_tiles + 7) / 8, pn = (a.n_tiles + 7) / 8;
    grid = dim3((unsigned)((pm * pn + 7) / 8 * 8 * 64), (unsigned)a.splits);
  }
#define LAUNCH(WT, XT) hipLaunchKernelGGL((dense_gemm_kernel<T, EPI, WT, XT>), grid, dim3(512), 0, st, a)
#define BY_XT(WT) do { if (xt == 8) LAUNCH(WT, 8); else if (xt == 4) LAUNCH(WT, 4); else if (xt == 2) LAUNCH(WT, 2); \
                       else LAUNCH(WT, 1); } while (0)
  if (k64) hipLaunchKernelGGL((dense_gemm_k64_kernel<T, EPI>), grid, dim3(512), 0, st, a);
  else if (wt == 4) BY_XT(4);
  else BY_XT(2);
#undef BY_XT
#undef LAUNCH
  if (a.splits > 1) {
    const long quads = (long)a.M * (a.N >> 2);
    hipLaunchKernelGGL((dense_gemm_reduce_kernel<T, EPI>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, a);
  }
  return 0;
}

// x [M, K] (row stride x_stride elements), w [N or 2 N, K] contiguous, out [M, N] (row stride out_stride), optional
// bias [N] and residual [M, N] (PLAIN only).  K % 32 == 0; strides multiples of 8 elements; the byte size of x and
// of w below 4 GiB (32-bit offsets in the copy addresses).  wt / splits: 0 = chosen here (wt 2 or 4: 128 or 256
// weight rows per tile).  ws: fp32 scratch of ws_bytes for split K (may be null: K is then not split).
// Returns 0, or a negative code for shapes it does not take.
VDA_EXPORT int vda_dense_gemm(void* out, const void* x, const void* w, const void* bias, const void* residual,
                              float* ws, size_t ws_bytes, int M, int N, int K, long x_stride, long out_stride,
                              long res_stride, int epi, int dtype, int wt, int splits, int variant, hipStream_t st) {
  if (M < 1) return M == 0 ? 0 : -1;
  if (N < 1 || K < 32 || K % 32) return -1;
  if (x_stride % 8 || (N % 4 && out_stride % 4)) return -1;
  if (epi != DG_PLAIN && epi != DG_GATE_UP) return -4;
  if (epi == DG_GATE_UP && (bias || residual || N % 4)) return -4;
  if (dtype != VDA_BF16 && dtype != VDA_F16) return -3;
  const long w_rows = epi == DG_GATE_UP ? 2L * N : N;
  if (w_rows * K * 2 >= (1LL << 32) || (long)M * x_stride * 2 >= (1LL << 32)) return -2;
  int xt = M > 128 ? 8 : M > 64 ? 4 : M > 32 ? 2 : 1;
  const int nu = K >> 5;
  const int m_tiles = (M + 32 * xt - 1) / (32 * xt);
  const bool can_split = !(N % 4 || out_stride % 4 || !ws);
  // Tile height and K split, when the caller leaves them open: the pair with the smaller estimated time.  A
  // workgroup streams (64 wt weight rows + 32 xt activation rows) x 64 bytes per unit; weight rows come from HBM at
  // about 24 GB/s per CU, activation rows from L2 (a sixteenth of that per row); a split adds the reduction
  // launch and its round trip over the fp32 partial sums.  Splits never create a second round of workgroups.
  auto plan = [&](int t, int forced_splits, double* cost) {
    const int ncol = epi == DG_GATE_UP ? 32 * t : 64 * t;
    const long tiles = (long)m_tiles * ((N + ncol - 1) / ncol);
    int sp = forced_splits;
    if (sp < 1) {
      sp = 1;
      if (tiles * 4 < 256 * 3) sp = (int)(256 / tiles);
      sp = min(sp, max(nu / 16, 1));
      sp = min(sp, 16);
    }
    if (!can_split) sp = 1;
    while (sp > 1 && (size_t)sp * M * w_rows * sizeof(float) > ws_bytes) --sp;
    const long rounds = (tiles * sp + 255) / 256;
    const double unit_us = 0.17 * t + 0.011 * xt;
    *cost = rounds * ((nu + sp - 1) / sp) * unit_us + (sp > 1 ? 4.0 + (double)M * w_rows * sp * 8 / 3e6 : 0.0);
    return sp;
  };
  if (wt != 2 && wt != 4) {
    double c4, c2;
    plan(4, splits, &c4);
    plan(2, splits, &c2);
    wt = (m_tiles > 1 || c4 <= c2) ? 4 : 2;
  }
  double unused;
  splits = plan(wt, splits, &unused);
  const int ncol = epi == DG_GATE_UP ? 32 * wt : 64 * wt;
  DenseGemmArgs a;
  a.out = out; a.x = x; a.w = w; a.bias = bias; a.residual = residual; a.ws = ws;
  a.M = M; a.N = N; a.K = K; a.x_stride = x_stride; a.out_stride = out_stride; a.res_stride = res_stride;
  a.m_tiles = m_tiles;
  a.n_tiles = (N + ncol - 1) / ncol;
  // variant: 0 = chosen here, 1 = ring of 32-element units, 2 = 64-element units (256 x 256 tiles, K % 64 == 0)
  const bool k64 = variant == 2 || (variant == 0 && a.m_tiles > 1);
  const bool use_k64 = k64 && xt == 8 && wt == 4 && K % 64 == 0;
  a.units_per_split = (nu + splits - 1) / splits;
  if (use_k64) a.units_per_split = (a.units_per_split + 1) & ~1;       // whole 64-element units per split
  a.splits = (nu + a.units_per_split - 1) / a.units_per_split;
  int rc;
  if (dtype == VDA_BF16) rc = epi == DG_GATE_UP ? launch_dense_gemm<bf16_t, DG_GATE_UP>(a, wt, xt, use_k64, st)
                                                : launch_dense_gemm<bf16_t, DG_PLAIN>(a, wt, xt, use_k64, st);
  else rc = epi == DG_GATE_UP ? launch_dense_gemm<f16_t, DG_GATE_UP>(a, wt, xt, use_k64, st)
                              : launch_dense_gemm<f16_t, DG_PLAIN>(a, wt, xt, use_k64, st);
  if (rc) return rc;
  VDA_CHECK_LAUNCH();
  return 0;
}

// Diagnostic: one launch of the stamped 256 x 256 kernels (variant 1 or 2, bf16, no epilogue: `out` is not
// written); dbg receives 64 blocks x 8 waves x 6 counters (see dg_stamp).
VDA_EXPORT int vda_dense_gemm_stamp(void* dbg, const void* x, const void* w, int M, int N, int K, int variant,
                                    hipStream_t st) {
  if (M < 257 || N < 256 || K % 64 || (long)N * K * 2 >= (1LL << 32) || (long)M * K * 2 >= (1LL << 32)) return -1;
  DenseGemmArgs a;
  a.out = nullptr; a.x = x; a.w = w; a.bias = nullptr; a.residual = nullptr; a.ws = reinterpret_cast<float*>(dbg);
  a.M = M; a.N = N; a.K = K; a.x_stride = K; a.out_stride = N; a.res_stride = 0;
  a.m_tiles = (M + 255) / 256; a.n_tiles = (N + 255) / 256; a.splits = 1; a.units_per_split = K >> 5;
  const int pm = (a.m_tiles + 7) / 8, pn = (a.n_tiles + 7) / 8;
  const dim3 grid((unsigned)((pm * pn + 7) / 8 * 8 * 64));
  if (variant == 2) hipLaunchKernelGGL((dense_gemm_k64_kernel<bf16_t, DG_PLAIN, true>), grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL((dense_gemm_kernel<bf16_t, DG_PLAIN, 4, 8, true>), grid, dim3(512), 0, st, a);
  VDA_CHECK_LAUNCH();
  return 0;
}
