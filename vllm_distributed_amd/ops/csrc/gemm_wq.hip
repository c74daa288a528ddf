// Weight-only quantised GEMM for every M: out[M, N] = X[M, K] . dequant(Wq)[N, K]^T (+ bias), 4- or 8-bit weights
// with one (scale, zero) pair per output row and group of G input elements (value = (q - zero) * scale), 16-bit
// activations: GPTQ / AWQ W4A16, 8-bit GPTQ, GGUF Q4_0 / Q8_0 after repacking.  The weights stay packed in HBM and
// in LDS; they are widened to the activation type only on their way into the MFMA operand registers.
//
// Two kernels.  Up to 128 activation rows (decode): the weight stream is the whole cost, so wq_stream_kernel keeps
// it out of LDS altogether: packed words go global -> registers eight (4 bit) or four (8 bit) K steps ahead, are
// widened in registers and fed to the MFMAs as the A operand, while the activations of a K step are shared by the
// waves through an LDS ring (the scheme of gemm_decode.hip with a quarter / half of its HBM bytes).  More rows:
// wq_gemm_kernel, the two-group schedule of gemm_dense.hip (one wave group issues 32 MFMAs from registers while the other issues
// the LDS-DMA copies of a later K unit, reads its next fragments AND widens them: the nibble extraction and the
// scale / zero arithmetic run on the vector ALU of the group that is not using the matrix pipe).  A K unit is 32
// elements: per tile 256 weight rows x 16 bytes (4 bit) or 32 bytes (8 bit), 256 (scale, zero) pairs and the
// activation rows x 64 bytes.
//
// Layouts (made by ops.wq_pack_* at load time):
//   qweight 4 bit  uint32 [N, K / 8]: word j of row n holds K elements 8 j .. 8 j + 7, element e in nibble
//                  e / 2 + 4 * (e % 2): (w >> 4 i) & 0x000F000F is the pair (2 i, 2 i + 1) as two 16-bit fields
//   qweight 8 bit  uint8 [N, K]: unsigned bytes (signed formats are stored + 128 with zero = 128)
//   sz             T [K / G, N, 2]: (scale, zero) of row n in group k / G; G a multiple of 32
// Takes the place of the reference's csrc/quantization/gptq/q_gemm.cu:1815 (exllama GEMM), awq/gemm_kernels.cu,
// gptq_marlin/ (W4A16 / W8A16 for any batch) and gguf/gguf_kernel.cu:91,207 for the formats above.
#include <type_traits>

#include "gemm_dense.h"

struct WqGemmArgs {
  DenseGemmArgs d;        // out, x, bias, residual, ws, M, N, K, tiles, splits, strides (d.w: packed weights)
  const void* sz;         // [K / G, N, 2]
  int group;              // G
  int gshift;             // log2(G / 32) when G is a power of two (30 when one group spans K), else -1
  int per_element;        // streaming kernel: scale every weight element even where the unscaled form applies
};

template <typename T, int BITS>
__device__ __forceinline__ typename Vec<T>::v8 wq_expand(const uint32_t* q, float s, float nzs) {
  typename Vec<T>::v8 o;
  if (BITS == 4) {
    const uint32_t lo = q[0] & 0x0F0F0F0Fu, hi = (q[0] >> 4) & 0x0F0F0F0Fu;
    o[0] = (T)fmaf((float)(lo & 0xff), s, nzs);
    o[4] = (T)fmaf((float)((lo >> 8) & 0xff), s, nzs);
    o[1] = (T)fmaf((float)((lo >> 16) & 0xff), s, nzs);
    o[5] = (T)fmaf((float)(lo >> 24), s, nzs);
    o[2] = (T)fmaf((float)(hi & 0xff), s, nzs);
    o[6] = (T)fmaf((float)((hi >> 8) & 0xff), s, nzs);
    o[3] = (T)fmaf((float)((hi >> 16) & 0xff), s, nzs);
    o[7] = (T)fmaf((float)(hi >> 24), s, nzs);
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      o[4 * h + 0] = (T)fmaf((float)(q[h] & 0xff), s, nzs);
      o[4 * h + 1] = (T)fmaf((float)((q[h] >> 8) & 0xff), s, nzs);
      o[4 * h + 2] = (T)fmaf((float)((q[h] >> 16) & 0xff), s, nzs);
      o[4 * h + 3] = (T)fmaf((float)(q[h] >> 24), s, nzs);
    }
  }
  return o;
}

__device__ __forceinline__ void glds4(const uint8_t* src, uint8_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

template <typename T, int BITS, int XT>
__global__ __launch_bounds__(512, 1) void wq_gemm_kernel(const WqGemmArgs wa) {
  using V8 = typename Vec<T>::v8;
  const DenseGemmArgs& a = wa.d;
  constexpr int WT = 4, WR = 256, XR = 32 * XT;
  constexpr int ROWB = 4 * BITS;                   // packed bytes of one weight row in a unit
  constexpr int WB = WR * ROWB;                    // 4 KiB or 8 KiB
  constexpr int SB = WR * 4;                       // (scale, zero) pairs
  constexpr int XROUNDS = XR / 128 > 0 ? XR / 128 : 1, XB = XROUNDS * 8192;
  constexpr int UNIT = WB + SB + XB;
  constexpr int NG = 2 + XROUNDS;
  constexpr int D = UNIT <= 16384 ? 8 : 6;         // ring slots (XT = 8 with 8-bit weights: 25 KiB x 6)
  constexpr int AHEAD = NG * (D - 2);
  static_assert(AHEAD < 64, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) uint8_t lds[D * UNIT];

  int mi, ni, sp;
  if (a.m_tiles == 1) {
    mi = 0; ni = blockIdx.x % a.n_tiles; sp = blockIdx.x / a.n_tiles;
  } else {
    // workgroup -> tile: each XCD (blockIdx.x % 8) takes a contiguous run of the tile list, and the list walks the
    // output in columns of eight weight tiles, row by row: the 32 workgroups an XCD runs at a time cover about
    // 4 x 8 tiles, so its L2 serves every weight tile four times and every activation tile eight times
    const int label = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int tiles = a.m_tiles * a.n_tiles;
    const int t = label * ((tiles + 7) >> 3) + seq;
    if (seq >= ((tiles + 7) >> 3) || t >= tiles) return;
    const int col = t / (8 * a.m_tiles), rem = t - col * 8 * a.m_tiles;
    const int width = min(8, a.n_tiles - 8 * col);
    mi = rem / width; ni = 8 * col + rem % width; sp = blockIdx.y;
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
  const int n0 = ni * WR;
  const uint8_t* wbase = reinterpret_cast<const uint8_t*>(a.w) + (long)u0 * ROWB;
  const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.x) + (long)u0 * 64;
  const uint8_t* szbase = reinterpret_cast<const uint8_t*>(wa.sz);
  const int units_per_group = wa.group >> 5;

  // ---- staging roles ------------------------------------------------------------------------------------
  // weights: 4 bit: lane L of wave w copies the 16 bytes of row 64 (w & 3) + L (waves 4-7 write the same bytes
  // again, which keeps the copy count per wave uniform); 8 bit: thread t copies half t & 1 of row t >> 1.
  // pairs: lane L of wave w copies the 4 bytes of row 64 (w & 3) + L.
  unsigned voff_w, voff_s, voff_x[XROUNDS];
  uint8_t* lds_w;
  {
    const int srow = 64 * wn + lane;
    if (BITS == 4) {
      voff_w = (unsigned)min(n0 + srow, a.N - 1) * (unsigned)(a.K >> 1);
      lds_w = lds + wn * 1024;
    } else {
      voff_w = (unsigned)min(n0 + (tid >> 1), a.N - 1) * (unsigned)a.K + ((tid & 1) << 4);
      lds_w = lds + wave * 1024;
    }
    voff_s = (unsigned)min(n0 + srow, a.N - 1) * 4u;
    const int lrow = 16 * wave + (lane >> 2), pos = lane & 3;
#pragma unroll
    for (int j = 0; j < XROUNDS; ++j) {
      const int tau = 128 * j + lrow;
      voff_x[j] = (unsigned)(beg + min(tau, cnt - 1)) * (unsigned)(a.x_stride * 2) + ((pos ^ swz64(tau & 15)) << 4);
    }
  }
  auto issue = [&](int u, int slot) {
    const int uu = min(u, nun - 1);
    uint8_t* base = lds + slot * UNIT;
    glds16(wbase + (long)uu * ROWB + voff_w, lds_w + slot * UNIT);
    const long gi = (u0 + uu) / units_per_group;
    glds4(szbase + gi * a.N * 4 + voff_s, base + WB + wn * 256);
    const uint8_t* xb = xbase + (long)uu * 64;
#pragma unroll
    for (int j = 0; j < XROUNDS; ++j) glds16(xb + voff_x[j], base + WB + SB + wave * 1024 + j * 8192);
  };

  // ---- compute roles: wave (grp, wn) owns weight rows [64 wn, 64 wn + 64) and activation rows of half grp ----
  const int wl = (64 * wn + r) * ROWB + g * (ROWB / 4);
  const int sl = WB + (64 * wn + r) * 4;
  const int xl = WB + SB + (16 * XT * grp + r) * 64 + ((g ^ swz64(r)) << 4);

  f32x4 acc[WT][XT];
#pragma unroll
  for (int mt = 0; mt < WT; ++mt)
#pragma unroll
    for (int nt = 0; nt < XT; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};

#pragma unroll
  for (int u = 0; u < D - 1; ++u) issue(u, u);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD) : "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();
  int rs = 0, wsl = D - 1;
  for (int u = 0; u < nun; ++u) {
    issue(u + D - 1, wsl);
    const uint8_t* base = lds + rs * UNIT;
    V8 wf[WT], xf[XT];
#pragma unroll
    for (int nt = 0; nt < XT; ++nt) xf[nt] = *reinterpret_cast<const V8*>(base + xl + nt * 1024);
    uint32_t q[WT][2], pr[WT];
#pragma unroll
    for (int mt = 0; mt < WT; ++mt) {
      const uint8_t* wp = base + wl + mt * 16 * ROWB;
      q[mt][0] = *reinterpret_cast<const uint32_t*>(wp);
      q[mt][1] = BITS == 8 ? *reinterpret_cast<const uint32_t*>(wp + 4) : 0u;
      pr[mt] = *reinterpret_cast<const uint32_t*>(base + sl + mt * 64);
    }
#pragma unroll
    for (int mt = 0; mt < WT; ++mt) {
      const typename Vec<T>::v4 pv = __builtin_bit_cast(typename Vec<T>::v4, u32x2{pr[mt], 0u});
      const float s = (float)pv[0];
      wf[mt] = wq_expand<T, BITS>(q[mt], s, -(float)pv[1] * s);
    }
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AHEAD) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nt = 0; nt < XT; ++nt)
#pragma unroll
      for (int mt = 0; mt < WT; ++mt) acc[mt][nt] = mfma16(wf[mt], xf[nt], acc[mt][nt]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rs = rs + 1 == D ? 0 : rs + 1;
    wsl = wsl + 1 == D ? 0 : wsl + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  dense_epilogue<T, DG_PLAIN, WT, XT>(a, acc, sp, beg, cnt, n0, r, g, grp, wn);
}


// ---- decode: weights global -> registers, activations through an LDS ring -----------------------------------
// A K step is 128 elements.  Lane (r, g) of a wave holds, for each of the wave's two 16-row weight fragments, the
// 32 consecutive elements [128 kt + 32 g, + 32) of row r (16 bytes of nibbles / 32 bytes) and that row's
// (scale, zero) pair; MFMA j of the step multiplies elements 32 g + 8 j .. + 7, so the activation operand of lane
// (c, g) is the 16-byte chunk 4 g + j of activation row c's 256 bytes.  The ring stores chunk i of row c at
// position i ^ (c & 15): the sixteen lanes of one g read sixteen different positions (all 64 banks, once each).
// Vector-memory operations complete in order: every iteration issues one ring fill and then one weight step and
// waits with a counted s_waitcnt for the fill the next iteration reads; the weight steps are issued from inline
// assembly so the compiler cannot drain the queue at the loop head.
//
// POST (4-bit weights whose groups are whole K steps): the weights enter the MFMAs unscaled.  Or-ing a float's
// exponent over a nibble field makes the 16-bit float BIAS + q (BIAS = 128 for bf16, 1024 for fp16, both exact), so
// a packed word becomes an operand register pair by pair with one shift and one and-or; a third MFMA per step
// multiplies the activations by ones, and at the end of a group
//     total += scale * (sum_k (BIAS + q_k) x_k  -  (BIAS + zero) * sum_k x_k)
// in fp32 on the accumulators: 2 x 4 fused multiply-adds per fragment instead of a convert, a multiply-add and a
// pack per weight element, which is what bounds the per-element form below the HBM rate.
template <typename T> struct WqMagic;
template <> struct WqMagic<bf16_t> { static constexpr uint32_t OR = 0x43004300u; static constexpr float BIAS = 128.f; };
template <> struct WqMagic<f16_t> { static constexpr uint32_t OR = 0x64006400u; static constexpr float BIAS = 1024.f; };

template <typename T, int BITS, int MT16, int WM, bool POST>
__global__ __launch_bounds__(256 * WM, WM == 1 ? 2 : 1) void wq_stream_kernel(const WqGemmArgs wa) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  const DenseGemmArgs& a = wa.d;
  constexpr int WAVES = 4 * WM;
  constexpr int SX = 4;                                // activation ring depth (filled three steps ahead)
  constexpr int DW = BITS == 4 && !(POST && MT16 >= 4) ? 8 : 4;   // weight steps in registers
  static_assert(!POST || BITS == 4, "unscaled operands exist for 4-bit weights only");
  constexpr int NQ = BITS / 4;                         // 16-byte loads per fragment and step
  constexpr int LW = 2 * NQ + 2;                       // vector-memory operations of one weight step
  constexpr int STEPB = 16 * BITS;                     // packed bytes of one row in a step
  constexpr int XB = MT16 * WM * 16 * 256;             // bytes of one ring slot
  constexpr int P = MT16;                              // 1 KiB pieces per wave per ring slot
  static_assert(P * WAVES * 1024 == XB, "rows must divide over the waves");
  // issued after the ring fill that the next iteration reads: three weight steps and two fills
  constexpr int YOUNGER = 3 * LW + 2 * P;
  static_assert(YOUNGER < 64, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) uint8_t lds[SX * XB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int wr = wave & 3, wm = wave >> 2;
  const int n0 = (blockIdx.x * 4 + wr) * 32;
  const int nk_all = a.K >> 7;
  const int kt0 = blockIdx.y * a.units_per_split;
  const int nk = min(a.units_per_split, nk_all - kt0);           // >= 1 by the host's choice of splits

  const uint8_t* wp[2];
  const uint8_t* sp[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = min(n0 + 16 * mt + r, a.N - 1);
    wp[mt] = reinterpret_cast<const uint8_t*>(a.w) + (long)row * ((long)a.K * BITS / 8) + (long)kt0 * STEPB + g * (STEPB / 4);
    // per-element form: the pair of the lane's own row; POST: the pairs of the four output rows 4 g .. 4 g + 3
    sp[mt] = reinterpret_cast<const uint8_t*>(wa.sz) + (POST ? (long)min(n0 + 16 * mt + 4 * g, a.N - 4) * 4 : (long)row * 4);
  }
  const uint8_t* xp[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int tau = 4 * WAVES * j + (tid >> 4);
    const long row = min(tau, a.M - 1);
    xp[j] = reinterpret_cast<const uint8_t*>(a.x) + row * a.x_stride * 2 + (long)kt0 * 256 + (((tid & 15) ^ (tau & 15)) << 4);
  }
  auto fill = [&](int kt) {          // ring slot kt % 4 <- K step min(kt, nk - 1)
    const long ko = (long)min(kt, nk - 1) * 256;
    uint8_t* dst = lds + (kt & 3) * XB + wave * 1024;
#pragma unroll
    for (int j = 0; j < P; ++j) glds16(xp[j] + ko, dst + j * WAVES * 1024);
  };
  i32x4 wreg[DW][2][NQ];
  typename std::conditional<POST, i32x4, uint32_t>::type sreg[DW][2];
  const long plane = (long)a.N * 4;
  auto wload = [&](int u, int kt) {
    const int kk = min(kt, nk - 1);
    const long so = (long)((((kt0 + kk) << 2) + g) >> wa.gshift) * plane;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const uint8_t* p = wp[mt] + (long)kk * STEPB;
      if constexpr (NQ == 1)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(wreg[u][mt][0]) : "v"(p) : "memory");
      else
        asm volatile("global_load_dwordx4 %0, %2, off nt\n\tglobal_load_dwordx4 %1, %2, off offset:16 nt"
                     : "=&v"(wreg[u][mt][0]), "=&v"(wreg[u][mt][NQ - 1]) : "v"(p) : "memory");
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const uint8_t* p = sp[mt] + so;
      if constexpr (POST) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(sreg[u][mt]) : "v"(p) : "memory");
      else asm volatile("global_load_dword %0, %1, off" : "=&v"(sreg[u][mt]) : "v"(p) : "memory");
    }
  };

  f32x4 acc[2][MT16];
  f32x4 part[POST ? 2 : 1][POST ? MT16 : 1], xsum[POST ? MT16 : 1];      // POST: this group's unscaled sums
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < MT16; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};
  if constexpr (POST) {
#pragma unroll
    for (int nt = 0; nt < MT16; ++nt) { part[0][nt] = part[1][nt] = xsum[nt] = f32x4{0, 0, 0, 0}; }
  }
  V8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (T)1.f;

  // prologue in the loop's own order (W .. W | X0 W X1 W X2 W), so the same counted wait holds from the start
#pragma unroll
  for (int u = 0; u < DW - 3; ++u) wload(u, u);
#pragma unroll
  for (int u = 0; u < 3; ++u) { fill(u); wload(DW - 3 + u, DW - 3 + u); }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");      // X0 and every weight step before it
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  for (int kb = 0; kb < nk; kb += DW) {
#pragma unroll
    for (int u = 0; u < DW; ++u) {
      const int kt = kb + u;
      fill(kt + 3);                                                  // slot (kt - 1) % 4: read last iteration
      if (kt < nk) {
        const uint8_t* base = lds + (u & 3) * XB + (wm * MT16 * 16 + r) * 256;
        if constexpr (POST) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            V8 wf[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const uint32_t w = (uint32_t)wreg[u][mt][0][j];
              const u32x4 d = {(w & 0x000F000Fu) | WqMagic<T>::OR, ((w >> 4) & 0x000F000Fu) | WqMagic<T>::OR,
                               ((w >> 8) & 0x000F000Fu) | WqMagic<T>::OR, ((w >> 12) & 0x000F000Fu) | WqMagic<T>::OR};
              wf[mt] = __builtin_bit_cast(V8, d);
            }
            const int ch = ((4 * g + j) ^ r) << 4;
#pragma unroll
            for (int nt = 0; nt < MT16; ++nt) {
              const V8 xf = *reinterpret_cast<const V8*>(base + nt * 4096 + ch);
              part[0][nt] = mfma16(wf[0], xf, part[0][nt]);
              part[1][nt] = mfma16(wf[1], xf, part[1][nt]);
              xsum[nt] = mfma16(ones, xf, xsum[nt]);
            }
          }
          const int here = (kt0 + kt) << 2;
          if ((here >> wa.gshift) != ((here + 4) >> wa.gshift) || kt + 1 == nk) {   // the group ends with this step
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const V8 pv = __builtin_bit_cast(V8, sreg[u][mt]);               // (scale, zero) of rows 4 g .. 4 g + 3
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float s = (float)pv[2 * i], zb = (float)pv[2 * i + 1] + WqMagic<T>::BIAS;
#pragma unroll
                for (int nt = 0; nt < MT16; ++nt) {
                  acc[mt][nt][i] = fmaf(s, fmaf(-zb, xsum[nt][i], part[mt][nt][i]), acc[mt][nt][i]);
                  part[mt][nt][i] = 0.f;
                }
              }
            }
#pragma unroll
            for (int nt = 0; nt < MT16; ++nt) xsum[nt] = f32x4{0, 0, 0, 0};
          }
        } else {
        float sc[2], nz[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const V4 pv = __builtin_bit_cast(V4, u32x2{sreg[u][mt], 0u});
          sc[mt] = (float)pv[0];
          nz[mt] = -(float)pv[1] * sc[mt];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          V8 wf[2];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            uint32_t q[2];
            if (BITS == 4) { q[0] = (uint32_t)wreg[u][mt][0][j]; q[1] = 0u; }
            else { q[0] = (uint32_t)wreg[u][mt][(2 * j) >> 2][(2 * j) & 3]; q[1] = (uint32_t)wreg[u][mt][(2 * j + 1) >> 2][(2 * j + 1) & 3]; }
            wf[mt] = wq_expand<T, BITS>(q, sc[mt], nz[mt]);
          }
          const int ch = ((4 * g + j) ^ r) << 4;
#pragma unroll
          for (int nt = 0; nt < MT16; ++nt) {
            const V8 xf = *reinterpret_cast<const V8*>(base + nt * 4096 + ch);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = mfma16(wf[mt], xf, acc[mt][nt]);
          }
        }
        }
      }
      wload(u, kt + DW);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(YOUNGER) : "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // nothing in flight into registers or LDS

  // ---- epilogue: lane holds out[token 16 (wm MT16 + nt) + r][n0 + 16 mt + 4 g + i] ------------------------
  float* wsp = a.splits > 1 ? a.ws + (long)blockIdx.y * a.M * a.N : nullptr;
#pragma unroll
  for (int nt = 0; nt < MT16; ++nt) {
    const int tok = 16 * (wm * MT16 + nt) + r;
    if (tok >= a.M) continue;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int n = n0 + 16 * mt + 4 * g;
      if (n >= a.N) continue;
      if (wsp) {
        float* op = wsp + (long)tok * a.N + n;
        if (n + 3 < a.N) *reinterpret_cast<f32x4*>(op) = acc[mt][nt];
        else
          for (int i = 0; i < 4; ++i) if (n + i < a.N) op[i] = acc[mt][nt][i];
      } else {
        T* op = reinterpret_cast<T*>(a.out) + (long)tok * a.out_stride;
        V4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[mt][nt][i];
          if (a.bias && n + i < a.N) v += (float)reinterpret_cast<const T*>(a.bias)[n + i];
          o[i] = (T)v;
        }
        if (n + 3 < a.N && a.out_stride % 4 == 0) *reinterpret_cast<V4*>(op + n) = o;
        else
          for (int i = 0; i < 4; ++i) if (n + i < a.N) op[n + i] = o[i];
      }
    }
  }
}

template <typename T, int BITS>
static int launch_wq_stream(const WqGemmArgs& wa, hipStream_t st) {
  const DenseGemmArgs& a = wa.d;
  const dim3 grid((unsigned)((a.N + 127) / 128), (unsigned)a.splits);
  // unscaled operands need 4-bit weights, groups made of whole K steps and four whole output rows per lane
  const bool post = BITS == 4 && !wa.per_element && (wa.group % 128 == 0) && a.N % 4 == 0;
#define LAUNCH(MT16, WM)                                                                                              \
  do {                                                                                                                \
    if constexpr (BITS == 4) {                                                                                        \
      if (post) { hipLaunchKernelGGL((wq_stream_kernel<T, 4, MT16, WM, true>), grid, dim3(256 * WM), 0, st, wa); break; } \
    }                                                                                                                 \
    hipLaunchKernelGGL((wq_stream_kernel<T, BITS, MT16, WM, false>), grid, dim3(256 * WM), 0, st, wa);                \
  } while (0)
  if (a.M <= 16) LAUNCH(1, 1);
  else if (a.M <= 32) LAUNCH(2, 1);
  else if (a.M <= 64) LAUNCH(4, 1);
  else LAUNCH(4, 2);
#undef LAUNCH
  if (a.splits > 1) {
    const long quads = (long)a.M * (a.N >> 2);
    hipLaunchKernelGGL((dense_gemm_reduce_kernel<T, DG_PLAIN>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, a);
  }
  return 0;
}

template <typename T, int BITS>
static int launch_wq_gemm(const WqGemmArgs& wa, int xt, hipStream_t st) {
  const DenseGemmArgs& a = wa.d;
  dim3 grid;
  if (a.m_tiles == 1) grid = dim3((unsigned)(a.n_tiles * a.splits));
  else grid = dim3((unsigned)((a.m_tiles * a.n_tiles + 7) / 8 * 8), (unsigned)a.splits);
#define LAUNCH(XT) hipLaunchKernelGGL((wq_gemm_kernel<T, BITS, XT>), grid, dim3(512), 0, st, wa)
  if (xt == 8) LAUNCH(8); else if (xt == 4) LAUNCH(4); else if (xt == 2) LAUNCH(2); else LAUNCH(1);
#undef LAUNCH
  if (a.splits > 1) {
    const long quads = (long)a.M * (a.N >> 2);
    hipLaunchKernelGGL((dense_gemm_reduce_kernel<T, DG_PLAIN>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, a);
  }
  return 0;
}

// x [M, K] 16-bit (row stride x_stride elements), qw packed [N, K] (bits 4 or 8), sz [K / group, N, 2] of x's type,
// out [M, N].  K % 32 == 0, group % 32 == 0 and K % group == 0, N % 4 == 0 for split K.  variant: 0 picks by shape,
// 1 the tiled kernel, 2 the streaming kernel (M <= 128, K % 128 == 0, group a power of two or K), 3 the streaming
// kernel in its per-element form.  Returns 0 or a negative code for shapes it does not take.
VDA_EXPORT int vda_wq_gemm(void* out, const void* x, const void* qw, const void* sz, const void* bias, float* ws,
                           size_t ws_bytes, int M, int N, int K, int group, int bits, long x_stride, long out_stride,
                           int dtype, int splits, int variant, hipStream_t st) {
  if (M < 1) return M == 0 ? 0 : -1;
  if (N < 1 || K < 32 || K % 32 || group < 32 || group % 32 || K % group) return -1;
  if (bits != 4 && bits != 8) return -1;
  if (x_stride % 8 || (N % 4 && out_stride % 4)) return -1;
  if (dtype != VDA_BF16 && dtype != VDA_F16) return -3;
  if ((long)N * K * bits / 8 >= (1LL << 32) || (long)M * x_stride * 2 >= (1LL << 32)) return -2;
  WqGemmArgs wa;
  DenseGemmArgs& a = wa.d;
  a.out = out; a.x = x; a.w = qw; a.bias = bias; a.residual = nullptr; a.ws = ws;
  a.M = M; a.N = N; a.K = K; a.x_stride = x_stride; a.out_stride = out_stride; a.res_stride = 0;
  wa.sz = sz; wa.group = group;
  wa.gshift = -1;
  if (group == K) wa.gshift = 30;
  else
    for (int sh = 0; sh < 16; ++sh) if (group == (32 << sh)) wa.gshift = sh;
  const bool can_stream = M <= 128 && K % 128 == 0 && wa.gshift >= 0;
  wa.per_element = variant == 3;
  if (variant == 3) variant = 2;
  if (variant == 2 && !can_stream) return -1;
  const bool no_split = N % 4 || out_stride % 4 || !ws;
  int rc;
  if (variant == 2 || (variant == 0 && can_stream && (M <= 64 || bits == 4))) {   // 8 bit, 65..128 rows: tiled is faster
    const int nk = K >> 7;
    const int groups = (N + 127) / 128;
    const int slots = M <= 64 ? 512 : 256;
    if (splits < 1) {
      splits = 1;
      if (groups * 4 < slots * 3) splits = (slots + groups - 1) / groups;
      splits = min(splits, max(nk / 8, 1));
      splits = min(splits, 32);
    }
    splits = min(splits, nk);
    if (no_split) splits = 1;
    while (splits > 1 && (size_t)splits * M * N * sizeof(float) > ws_bytes) --splits;
    a.units_per_split = (nk + splits - 1) / splits;
    a.splits = (nk + a.units_per_split - 1) / a.units_per_split;
    a.m_tiles = 1; a.n_tiles = groups;
    if (dtype == VDA_BF16) rc = bits == 4 ? launch_wq_stream<bf16_t, 4>(wa, st) : launch_wq_stream<bf16_t, 8>(wa, st);
    else rc = bits == 4 ? launch_wq_stream<f16_t, 4>(wa, st) : launch_wq_stream<f16_t, 8>(wa, st);
  } else {
    const int xt = M > 128 ? 8 : M > 64 ? 4 : M > 32 ? 2 : 1;
    a.m_tiles = (M + 32 * xt - 1) / (32 * xt);
    a.n_tiles = (N + 255) / 256;
    const int nu = K >> 5;
    if (splits < 1) {
      splits = 1;
      const long tiles = (long)a.m_tiles * a.n_tiles;
      if (tiles * 4 < 256 * 3) splits = (int)(256 / tiles);
      splits = min(splits, max(nu / 16, 1));
      splits = min(splits, 16);
    }
    splits = min(splits, nu);
    if (no_split) splits = 1;
    while (splits > 1 && (size_t)splits * M * N * sizeof(float) > ws_bytes) --splits;
    a.units_per_split = (nu + splits - 1) / splits;
    a.splits = (nu + a.units_per_split - 1) / a.units_per_split;
    if (dtype == VDA_BF16) rc = bits == 4 ? launch_wq_gemm<bf16_t, 4>(wa, xt, st) : launch_wq_gemm<bf16_t, 8>(wa, xt, st);
    else rc = bits == 4 ? launch_wq_gemm<f16_t, 4>(wa, xt, st) : launch_wq_gemm<f16_t, 8>(wa, xt, st);
  }
  if (rc) return rc;
  VDA_CHECK_LAUNCH();
  return 0;
}
