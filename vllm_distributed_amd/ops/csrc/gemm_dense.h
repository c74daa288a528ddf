// Shared by the ping-pong MFMA GEMM kernels (gemm_dense.hip: 16-bit weights; gemm_wq.hip: packed 4- / 8-bit
// weights widened on the way into the MFMAs): launch arguments, the epilogue and the split-K reduction.
#pragma once
#include "mfma_tile.h"

enum { DG_PLAIN = 0, DG_GATE_UP = 1 };

struct DenseGemmArgs {
  void* out;              // [M, N] 16-bit, row stride out_stride
  const void* x;          // [M, K], row stride x_stride
  const void* w;          // [W rows, K] contiguous; W rows = N (PLAIN) or 2 N (GATE_UP: gate rows, then up rows)
  const void* bias;       // optional [N]
  const void* residual;   // optional [M, N], row stride res_stride
  float* ws;              // [splits, M, W rows] fp32 when splits > 1
  int M, N, K;
  int m_tiles, n_tiles, splits, units_per_split;
  long x_stride, out_stride, res_stride;   // elements
};

__device__ __forceinline__ int swz64(int r) { return (-(r >> 2)) & 3; }

// Epilogue shared by the GEMM kernels: lane (r = lane & 15, g = lane >> 4) of wave (grp, wn) holds
// out[token 16 (XT grp + nt) + r][column of weight fragment mt + 4 g + i] in acc[mt][nt][i].
// Vector-memory operations retire in order and stores count: a load between two stores makes the wave wait
// for every earlier store (a round trip to L2 each).  So a tile that lies wholly inside the output takes a
// straight-line path: every bias / residual load first, then nothing but arithmetic and stores.
template <typename T, int EPI, int WT, int XT>
__device__ __forceinline__ void dense_epilogue(const DenseGemmArgs& a, const f32x4 (&acc)[WT][XT], int sp, int beg,
                                               int cnt, int n0, int r, int g, int grp, int wn) {
  using V4 = typename Vec<T>::v4;
  constexpr int WR = 64 * WT, H = WT / 2, XR = 32 * XT;
  constexpr int NCOL = EPI == DG_GATE_UP ? WR / 2 : WR;
  const int w_rows = EPI == DG_GATE_UP ? 2 * a.N : a.N;
  const bool interior = cnt == XR && n0 + NCOL <= a.N;        // the same for every lane of the workgroup
  // weight row (as stored) of fragment mt, element 0 of this lane
  auto wrow_of = [&](int mt) {
    const int rho = (EPI == DG_GATE_UP ? (mt < H ? 16 * H * wn + 16 * mt : WR / 2 + 16 * H * wn + 16 * (mt - H))
                                       : 16 * WT * wn + 16 * mt) + 4 * g;
    return EPI == DG_GATE_UP ? (rho < WR / 2 ? n0 + rho : a.N + n0 + rho - WR / 2) : n0 + rho;
  };
  if (a.splits > 1) {                              // fp32 partial tile, weight-row indexed (gate and up rows as stored)
    float* wsp = a.ws + (long)sp * a.M * w_rows;
    if (interior) {
#pragma unroll
      for (int nt = 0; nt < XT; ++nt) {
        float* op = wsp + (long)(beg + 16 * (XT * grp + nt) + r) * w_rows;
#pragma unroll
        for (int mt = 0; mt < WT; ++mt) *reinterpret_cast<f32x4*>(op + wrow_of(mt)) = acc[mt][nt];
      }
      return;
    }
#pragma unroll
    for (int nt = 0; nt < XT; ++nt) {
      const int tau = 16 * (XT * grp + nt) + r;
      if (tau >= cnt) continue;
      float* op = wsp + (long)(beg + tau) * w_rows;
#pragma unroll
      for (int mt = 0; mt < WT; ++mt) {
        const int row = wrow_of(mt);
        const int lim = EPI == DG_GATE_UP ? (mt < H ? a.N : 2 * a.N) : a.N;
        if (row + 3 < lim) *reinterpret_cast<f32x4*>(op + row) = acc[mt][nt];
        else
          for (int i = 0; i < 4; ++i) if (row + i < lim) op[row + i] = acc[mt][nt][i];
      }
    }
    return;
  }
  if (interior) {
    T* ob = reinterpret_cast<T*>(a.out) + (long)(beg + 16 * XT * grp + r) * a.out_stride;
    if (EPI == DG_GATE_UP) {
#pragma unroll
      for (int nt = 0; nt < XT; ++nt)
#pragma unroll
        for (int mt = 0; mt < H; ++mt) {
          V4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float gte = acc[mt][nt][i], up = acc[mt + H][nt][i];
            o[i] = (T)(gte / (1.f + __expf(-gte)) * up);
          }
          *reinterpret_cast<V4*>(ob + (long)nt * 16 * a.out_stride + n0 + 16 * H * wn + 16 * mt + 4 * g) = o;
        }
      return;
    }
    const int nb = n0 + 16 * WT * wn + 4 * g;
    if (a.bias || a.residual) {                    // uniform: loads first, all of them, then stores only
      V4 bv[WT], rv[WT][XT];
#pragma unroll
      for (int mt = 0; mt < WT; ++mt) {
        bv[mt] = a.bias ? *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(a.bias) + nb + 16 * mt) : V4{};
#pragma unroll
        for (int nt = 0; nt < XT; ++nt)
          rv[mt][nt] = a.residual ? *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(a.residual) +
                                                                 (long)(beg + 16 * (XT * grp + nt) + r) * a.res_stride +
                                                                 nb + 16 * mt)
                                  : V4{};
      }
#pragma unroll
      for (int nt = 0; nt < XT; ++nt)
#pragma unroll
        for (int mt = 0; mt < WT; ++mt) {
          V4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (T)(acc[mt][nt][i] + (float)bv[mt][i] + (float)rv[mt][nt][i]);
          *reinterpret_cast<V4*>(ob + (long)nt * 16 * a.out_stride + nb + 16 * mt) = o;
        }
      return;
    }
#pragma unroll
    for (int nt = 0; nt < XT; ++nt)
#pragma unroll
      for (int mt = 0; mt < WT; ++mt) {
        V4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (T)acc[mt][nt][i];
        *reinterpret_cast<V4*>(ob + (long)nt * 16 * a.out_stride + nb + 16 * mt) = o;
      }
    return;
  }
  // ---- edge tiles: per-element bounds ------------------------------------------------------------------
#pragma unroll
  for (int nt = 0; nt < XT; ++nt) {
    const int tau = 16 * (XT * grp + nt) + r;
    if (tau >= cnt) continue;
    T* op = reinterpret_cast<T*>(a.out) + (long)(beg + tau) * a.out_stride;
    const T* rp = a.residual ? reinterpret_cast<const T*>(a.residual) + (long)(beg + tau) * a.res_stride : nullptr;
    if (EPI == DG_GATE_UP) {
#pragma unroll
      for (int mt = 0; mt < H; ++mt) {
        const int n = n0 + 16 * H * wn + 16 * mt + 4 * g;
        V4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gte = acc[mt][nt][i], up = acc[mt + H][nt][i];
          o[i] = (T)(gte / (1.f + __expf(-gte)) * up);
        }
        if (n + 3 < a.N) *reinterpret_cast<V4*>(op + n) = o;
        else
          for (int i = 0; i < 4; ++i) if (n + i < a.N) op[n + i] = o[i];
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < WT; ++mt) {
        const int n = n0 + 16 * WT * wn + 16 * mt + 4 * g;
        if (n >= a.N) continue;
        const bool full = n + 3 < a.N;
        V4 o, bv = V4{}, rv = V4{};
        if (full) {
          if (a.bias) bv = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(a.bias) + n);
          if (rp) rv = *reinterpret_cast<const V4*>(rp + n);
        } else {
          for (int i = 0; i < 4; ++i) if (n + i < a.N) {
            if (a.bias) bv[i] = reinterpret_cast<const T*>(a.bias)[n + i];
            if (rp) rv[i] = rp[n + i];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (T)(acc[mt][nt][i] + (float)bv[i] + (float)rv[i]);
        if (full) *reinterpret_cast<V4*>(op + n) = o;
        else
          for (int i = 0; i < 4; ++i) if (n + i < a.N) op[n + i] = o[i];
      }
    }
  }
}

// out[m][n] = epilogue(sum over splits of ws[s][m][row(n)]); four columns per thread (N % 4 == 0)
template <typename T, int EPI>
__global__ __launch_bounds__(256) void dense_gemm_reduce_kernel(const DenseGemmArgs a) {
  using V4 = typename Vec<T>::v4;
  const int nq = a.N >> 2;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= (long)a.M * nq) return;
  const int m = (int)(q / nq), n = (int)(q % nq) * 4;
  const int w_rows = EPI == DG_GATE_UP ? 2 * a.N : a.N;
  const long plane = (long)a.M * w_rows;
  const float* p = a.ws + (long)m * w_rows + n;
  f32x4 s = *reinterpret_cast<const f32x4*>(p), t = f32x4{0, 0, 0, 0};
  if (EPI == DG_GATE_UP) t = *reinterpret_cast<const f32x4*>(p + a.N);
  for (int k = 1; k < a.splits; ++k) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + k * plane);
    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    if (EPI == DG_GATE_UP) {
      const f32x4 z = *reinterpret_cast<const f32x4*>(p + k * plane + a.N);
      t[0] += z[0]; t[1] += z[1]; t[2] += z[2]; t[3] += z[3];
    }
  }
  V4 o;
  if (EPI == DG_GATE_UP) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (T)(s[i] / (1.f + __expf(-s[i])) * t[i]);
  } else {
    V4 bv = V4{}, rv = V4{};
    if (a.bias) bv = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(a.bias) + n);
    if (a.residual) rv = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(a.residual) + (long)m * a.res_stride + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (T)(s[i] + (float)bv[i] + (float)rv[i]);
  }
  *reinterpret_cast<V4*>(reinterpret_cast<T*>(a.out) + (long)m * a.out_stride + n) = o;
}

