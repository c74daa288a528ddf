// Weight-streaming MFMA GEMM for decode batches: out[M, N] = X[M, K] . W[N, K]^T (+ bias), 16 <= M <= 256.
//
// In this regime every weight byte is used M times and the layer is bound by how fast the weights leave
// HBM, so the kernel is built around the weight stream:
//   * a wave owns 32 weight rows and ALL M activation rows; its weight fragments go global -> registers
//     (the MFMA A operand: lane (r, g) holds bytes [32 g, 32 g + 32) of row r of a 128-byte K step, one
//     full cache line per row), non-temporal, four K steps ahead, and never touch LDS;
//   * the activations of a K step (M x 128 bytes) are shared by the waves of a workgroup through a
//     four-deep LDS ring filled with 16-byte global_load_lds (same swizzled image as gemm_tile.hip);
//   * vector-memory operations complete in order, so each iteration issues the ring fill first and the
//     weight loads second and then waits with a COUNTED s_waitcnt for the fill the next iteration reads:
//     the three youngest weight steps stay in flight across the barrier (>= 96 KiB per CU outstanding).
//     The weight loads are issued from inline assembly: the compiler does not track them, so it cannot
//     put its own (conservative, queue-draining) wait at the loop head.  A weight step is issued before
//     the ring fill of the same K step, hence it has landed whenever that fill has;
//   * K is split over workgroups (blockIdx.y) so that even an 8192-row matrix puts a workgroup on every
//     CU; partial sums go to an fp32 workspace and a second kernel adds them (+ bias) into the output.
// A workgroup covers 128 weight rows with 4 waves (two workgroups per CU).  For 128 < M <= 256 it has
// 8 waves, 4 (weight rows) x 2 (halves of the activation rows), one workgroup per CU with a 128 KiB ring:
// the two waves of a row group load the same weight lines (the second one hits in the cache), which
// keeps the accumulators of a wave at 32 x 128 and leaves registers for the weight ring.
//
// Takes the place of the reference's skinny GEMMs at larger batch (csrc/rocm/skinny_gemms.cu:
// wvSplitK :293-1210 serves M <= 4 there, everything above falls to the BLAS library).
#include "mfma_tile.h"

struct DecodeGemmArgs {
  const void* x;        // [M, K], row stride x_stride elements
  const void* w;        // [N, K] contiguous
  void* out;            // [M, N] 16-bit (splits == 1)
  float* ws;            // [splits, M, N] fp32 (splits > 1)
  const void* bias;     // optional [N] (splits == 1; otherwise added by the reduction)
  int M, N, K;
  long x_stride;        // elements
  long out_stride;      // elements
  int splits;
  int steps_per_split;  // K steps of 128 bytes per split
};

// MT16: 16-row activation tiles per wave; WM: waves along the activation rows (1 or 2)
template <typename T, int MT16, int WM>
__global__ __launch_bounds__(256 * WM, WM == 1 ? 2 : 1) void decode_gemm_kernel(const DecodeGemmArgs a) {
  using V4 = typename Vec<T>::v4;
  constexpr int WAVES = 4 * WM;
  constexpr int SX = 4;                          // activation ring depth (filled three steps ahead)
  constexpr int DW = 4;                          // weight register ring (loaded four steps ahead)
  constexpr int XB = MT16 * WM * 16 * 128;       // bytes of one ring slot
  constexpr int P = MT16 / 2;                    // 1 KiB pieces per wave per ring slot
  static_assert(P >= 1 && P * WAVES * 8 == MT16 * WM * 16, "rows must divide over the waves");
  // operations issued after the ring fill that the next iteration reads: three weight steps (4 loads
  // each) and two fills (P pieces each)
  constexpr int YOUNGER = 3 * 4 + 2 * P;
  __shared__ __attribute__((aligned(16))) uint8_t lds[SX * XB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int wr = wave & 3, wm = wave >> 2;
  const int n0 = (blockIdx.x * 4 + wr) * 32;
  const int nk_all = a.K >> 6;                                   // 64 elements = 128 bytes
  const int kt0 = blockIdx.y * a.steps_per_split;
  const int nk = min(a.steps_per_split, nk_all - kt0);          // >= 1 by the host's choice of splits
  const long kbytes = (long)a.K * 2;

  // ---- weight stream: rows n0 + 16 mt + r, bytes 128 kt + 32 g ... + 32 ------------------------------
  const uint8_t* wp[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = min(n0 + 16 * mt + r, a.N - 1);
    wp[mt] = reinterpret_cast<const uint8_t*>(a.w) + (long)row * kbytes + (long)kt0 * 128 + 32 * g;
  }
  // ---- activation ring: thread t of piece j fills row 8 WAVES j + (t >> 3), slot t & 7 ----------------
  const uint8_t* xp[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int tau = 8 * WAVES * j + (tid >> 3);
    const long row = min(tau, a.M - 1);
    xp[j] = reinterpret_cast<const uint8_t*>(a.x) + row * a.x_stride * 2 + (long)kt0 * 128 +
            (((tid & 7) ^ swz(tau)) << 4);
  }
  auto fill = [&](int kt) {        // ring slot kt % 4 <- K step min(kt, nk - 1)
    const long ko = (long)min(kt, nk - 1) * 128;
    uint8_t* dst = lds + (kt & 3) * XB + wave * 1024;
#pragma unroll
    for (int j = 0; j < P; ++j) glds16(xp[j] + ko, dst + j * WAVES * 1024);
  };
  i32x4 wreg[DW][2][2];
  auto wload = [&](int u, int kt) {
    const long ko = (long)min(kt, nk - 1) * 128;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const uint8_t* p = wp[mt] + ko;
      asm volatile("global_load_dwordx4 %0, %2, off nt\n\tglobal_load_dwordx4 %1, %2, off offset:16 nt"
                   : "=&v"(wreg[u][mt][0]), "=&v"(wreg[u][mt][1]) : "v"(p) : "memory");
    }
  };

  const int f = swz(r);
  const int c0 = ((2 * g) ^ f) << 4, c1 = ((2 * g + 1) ^ f) << 4;
  f32x4 acc[2][MT16];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < MT16; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};

  // ---- prologue: weight steps 0 .. 3, activation steps 0 .. 2; every weight step goes out before the
  // ring fill of the same K step, here and in the loop (W0 X0 W1 X1 W2 X2 W3 | X3 W4 | X4 W5 | ...) --------
#pragma unroll
  for (int u = 0; u < 3; ++u) { wload(u, u); fill(u); }
  wload(3, 3);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");      // W0 and X0 have landed
  __builtin_amdgcn_s_barrier();

  for (int kb = 0; kb < nk; kb += 4) {                             // nk % 4 == 0 (host)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kt = kb + u;
      fill(kt + 3);                                                // slot (kt - 1) % 4: read last iteration
      const uint8_t* base = lds + u * XB + (wm * MT16 * 16 + r) * 128;
#pragma unroll
      for (int nt = 0; nt < MT16; ++nt) {
        const i32x4 x0 = *reinterpret_cast<const i32x4*>(base + nt * 2048 + c0);
        const i32x4 x1 = *reinterpret_cast<const i32x4*>(base + nt * 2048 + c1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt][nt] = mma(wreg[u][mt][0], wreg[u][mt][1], x0, x1, acc[mt][nt], T{});
      }
      wload(u, kt + 4);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

  // ---- epilogue: lane holds out[token 16 nt + r][n0 + 16 mt + 4 g + i] ---------------------------------
  if (a.splits == 1) {
#pragma unroll
    for (int nt = 0; nt < MT16; ++nt) {
      const int tok = 16 * (wm * MT16 + nt) + r;
      if (tok >= a.M) continue;
      T* op = reinterpret_cast<T*>(a.out) + (long)tok * a.out_stride;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int n = n0 + 16 * mt + 4 * g;
        if (n >= a.N) continue;
        V4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[mt][nt][i];
          if (a.bias && n + i < a.N) v += (float)reinterpret_cast<const T*>(a.bias)[n + i];
          o[i] = (T)v;
        }
        if (n + 3 < a.N) *reinterpret_cast<V4*>(op + n) = o;
        else
          for (int i = 0; i < 4; ++i) if (n + i < a.N) op[n + i] = o[i];
      }
    }
  } else {
    float* wsp = a.ws + (long)blockIdx.y * a.M * a.N;
#pragma unroll
    for (int nt = 0; nt < MT16; ++nt) {
      const int tok = 16 * (wm * MT16 + nt) + r;
      if (tok >= a.M) continue;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int n = n0 + 16 * mt + 4 * g;
        if (n >= a.N) continue;
        float* op = wsp + (long)tok * a.N + n;
        if (n + 3 < a.N) *reinterpret_cast<f32x4*>(op) = acc[mt][nt];
        else
          for (int i = 0; i < 4; ++i) if (n + i < a.N) op[i] = acc[mt][nt][i];
      }
    }
  }
}

// out[m][n] = sum over splits of ws[s][m][n] (+ bias[n]); four columns per thread (N % 4 == 0)
template <typename T>
__global__ __launch_bounds__(256) void decode_gemm_reduce_kernel(T* __restrict__ out, const float* __restrict__ ws,
                                                                 const T* __restrict__ bias, int M, int N,
                                                                 long out_stride, int splits) {
  using V4 = typename Vec<T>::v4;
  const long quads = (long)M * (N >> 2);
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= quads) return;
  const int m = (int)(q / (N >> 2)), n = (int)(q % (N >> 2)) * 4;
  const long plane = (long)M * N;
  const float* p = ws + (long)m * N + n;
  f32x4 s = *reinterpret_cast<const f32x4*>(p);
  for (int k = 1; k < splits; ++k) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + k * plane);
    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
  }
  V4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (T)(s[i] + (bias ? (float)bias[n + i] : 0.f));
  *reinterpret_cast<V4*>(out + (long)m * out_stride + n) = o;
}

// How K is cut: enough workgroups to cover every CU's slots, at least 8 K steps per split (the pipeline
// prologue is 3), at most what the workspace holds.  Returns the number of splits.
static int decode_gemm_splits(int M, int N, int K, int wm, size_t ws_bytes, int* steps_per_split) {
  const int nk = K >> 6;                         // a multiple of 4 (K % 256 == 0)
  const int groups = (N + 127) / 128;
  const int slots = wm == 1 ? 512 : 256;
  int s = 1;
  if (groups * 4 < slots * 3) s = (slots + groups - 1) / groups;
  s = min(s, max(nk / 8, 1));
  s = min(s, 32);
  if (N % 4) s = 1;
  while (s > 1 && (size_t)s * M * N * sizeof(float) > ws_bytes) --s;
  int steps = ((nk + s - 1) / s + 3) / 4 * 4;    // the kernel's loop runs four steps per trip
  s = (nk + steps - 1) / steps;
  *steps_per_split = steps;
  return s;
}

template <typename T>
static int launch_decode_gemm(const DecodeGemmArgs& a, int wm, hipStream_t st) {
  const dim3 grid((unsigned)((a.N + 127) / 128), (unsigned)a.splits);
#define LAUNCH(MT16, WM) hipLaunchKernelGGL((decode_gemm_kernel<T, MT16, WM>), grid, dim3(256 * WM), 0, st, a)
  if (wm == 2) LAUNCH(8, 2);
  else if (a.M <= 32) LAUNCH(2, 1);
  else if (a.M <= 64) LAUNCH(4, 1);
  else LAUNCH(8, 1);
#undef LAUNCH
  if (a.splits > 1) {
    const long quads = (long)a.M * (a.N >> 2);
    hipLaunchKernelGGL((decode_gemm_reduce_kernel<T>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<T*>(a.out), a.ws, reinterpret_cast<const T*>(a.bias), a.M, a.N, a.out_stride,
                       a.splits);
  }
  return 0;
}

// x [M, K] (row stride x_stride elements), w [N, K] contiguous, out [M, N] (row stride out_stride);
// 1 <= M <= 256, K % 256 == 0, strides multiples of 8 elements.  ws: fp32 scratch of ws_bytes (may be null:
// K is then not split).  dtype VDA_BF16 / VDA_F16.  Returns 0, or a negative code for shapes it does not take.
VDA_EXPORT int vda_decode_gemm(void* out, const void* x, const void* w, const void* bias, float* ws, size_t ws_bytes,
                               int M, int N, int K, long x_stride, long out_stride, int dtype, hipStream_t st) {
  if (M < 1 || M > 256 || N < 1 || K < 256 || K % 256) return -1;
  if (x_stride % 8 || (N % 4 && out_stride % 4)) return -1;
  if (dtype != VDA_BF16 && dtype != VDA_F16) return -3;
  const int wm = M > 128 ? 2 : 1;
  DecodeGemmArgs a;
  a.x = x; a.w = w; a.out = out; a.ws = ws; a.bias = bias; a.M = M; a.N = N; a.K = K;
  a.x_stride = x_stride; a.out_stride = out_stride;
  a.splits = decode_gemm_splits(M, N, K, wm, ws ? ws_bytes : 0, &a.steps_per_split);
  if (a.splits > 1 && out_stride % 4) { a.splits = 1; a.steps_per_split = K >> 6; }
  const int rc = dtype == VDA_BF16 ? launch_decode_gemm<bf16_t>(a, wm, st) : launch_decode_gemm<f16_t>(a, wm, st);
  if (rc) return rc;
  VDA_CHECK_LAUNCH();
  return 0;
}
