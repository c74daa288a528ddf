// State-space (Mamba / Mamba-2) kernels over a ragged batch with per-request state slots.
//
// Both kernels take the same batch description as the attention kernels:
//   qsl[S+1]       first token row of every sequence in the flat [T, ...] inputs
//   seq_lens[S]    tokens of the sequence after this step (0 = padded graph row: skipped)
//   block_table    row s, column 0 holds the state slot of sequence s
// A sequence carries state in from earlier steps when seq_len - new_tokens > 0, so chunked
// prefill, decode and mixed batches are one code path.
// Role of the reference's csrc/mamba/mamba_ssm/selective_scan_fwd.cu and
// csrc/mamba/causal_conv1d/causal_conv1d.cu (varlen forward + single-token update).
#include "common.h"

__device__ __forceinline__ float silu_f(float v) { return v / (1.f + __expf(-v)); }
__device__ __forceinline__ float softplus_f(float v) { return v > 20.f ? v : log1pf(__expf(v)); }

// ---- depthwise causal convolution ----------------------------------------------------------------
// One thread per channel and token chunk: channels are the fast (coalesced) dimension of x and out,
// the W-1 inputs before the chunk come from x itself or, at the start of the step, from conv_state.
// mode 0: convolve and roll the state (only valid when one chunk covers every sequence)
// mode 1: convolve only          mode 2: roll the state only (launched after a mode-1 pass)
template <typename T, int W>
__global__ __launch_bounds__(256) void causal_conv1d_kernel(
    T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ weight, const T* __restrict__ bias,
    T* __restrict__ conv_state, const int* __restrict__ qsl, const int* __restrict__ seq_lens,
    const int* __restrict__ block_table, long bt_stride, int C, long x_stride, long o_stride, int act, int chunk,
    int mode) {
  const int s = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int sl = seq_lens[s];
  if (sl <= 0) return;
  const int t0 = qsl[s], L = qsl[s + 1] - t0;
  const bool has_init = sl - L > 0;
  const int cs = mode == 2 ? L : blockIdx.z * chunk;
  if (cs > L || (mode != 2 && cs >= L)) return;
  const int ce = min(L, cs + chunk);
  T* st = conv_state + ((long)block_table[s * bt_stride] * C + c) * (W - 1);
  float w[W], win[W - 1];
#pragma unroll
  for (int j = 0; j < W; ++j) w[j] = (float)weight[(long)c * W + j];
  const float b = bias ? (float)bias[c] : 0.f;
#pragma unroll
  for (int j = 0; j < W - 1; ++j) {
    const int p = cs - (W - 1) + j;
    win[j] = p >= 0 ? (float)x[(long)(t0 + p) * x_stride + c] : (has_init ? (float)st[W - 1 + p] : 0.f);
  }
  if (mode != 2) {
    for (int t = cs; t < ce; ++t) {
      const float cur = (float)x[(long)(t0 + t) * x_stride + c];
      float acc = b + w[W - 1] * cur;
#pragma unroll
      for (int j = 0; j < W - 1; ++j) acc += w[j] * win[j];
#pragma unroll
      for (int j = 0; j < W - 2; ++j) win[j] = win[j + 1];
      win[W - 2] = cur;
      out[(long)(t0 + t) * o_stride + c] = (T)(act ? silu_f(acc) : acc);
    }
  }
  if (mode == 2 || (mode == 0 && ce == L)) {
#pragma unroll
    for (int j = 0; j < W - 1; ++j) st[j] = (T)win[j];
  }
}

VDA_EXPORT int vda_causal_conv1d(void* out, const void* x, const void* weight, const void* bias, void* conv_state,
                                 const int* qsl, const int* seq_lens, const int* block_table, long bt_stride, int S,
                                 int C, int W, int max_q, long x_stride, long o_stride, int act, int dtype,
                                 hipStream_t st) {
  if (S <= 0 || C <= 0) return 0;
  if (W < 2 || W > 4) return -1;
  const int chunk = 64;
  const int nz = max(1, (max_q + chunk - 1) / chunk);
  dim3 block(256), grid((C + 255) / 256, S, nz);
#define L(TT, WW, MODE, G) hipLaunchKernelGGL((causal_conv1d_kernel<TT, WW>), G, block, 0, st, (TT*)out, (const TT*)x, \
    (const TT*)weight, (const TT*)bias, (TT*)conv_state, qsl, seq_lens, block_table, bt_stride, C, x_stride, o_stride,  \
    act, chunk, MODE)
#define BYW(TT, MODE, G) do { if (W == 4) L(TT, 4, MODE, G); else if (W == 3) L(TT, 3, MODE, G); else L(TT, 2, MODE, G); } while (0)
#define RUN(TT) do { if (nz == 1) BYW(TT, 0, grid); else { BYW(TT, 1, grid); dim3 g2((C + 255) / 256, S, 1); BYW(TT, 2, g2); } } while (0)
  if (dtype == VDA_BF16) RUN(bf16_t);
  else if (dtype == VDA_F16) RUN(f16_t);
  else if (dtype == VDA_F32) RUN(float);
  else return -3;
#undef RUN
#undef BYW
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

// ---- selective scan ----------------------------------------------------------------------------
// h[c, n] <- exp(dt * A[c, n]) * h[c, n] + dt * B[t, n] * u[t, c];  y[t, c] = sum_n C[t, n] h[c, n] + D[c] u[t, c]
// NL = N / 16 lanes share one channel (16 states each, y summed with shuffles), so Mamba (N = 16) uses one
// lane per channel and Mamba-2 (N = 128) eight.
//   dt_div      channels per dt / A / D entry (1: per channel; head_dim for Mamba-2 heads)
//   a_scalar    A holds one value per dt entry instead of [C, N]
//   group_ch    channels per B / C group (B, C are [T, groups * N])
// MODE 0  whole sequences in one pass (decode, short chunks): tokens walked in order, state read and written.
// Long prefills are cut into chunks of `chunk` tokens (grid.z) and run in three passes, since the
// recurrence is linear in h:  h_end = P * h_start + H_loc  with P = prod exp(dt A), H_loc = chunk result from 0.
// MODE 1  per chunk: P and H_loc into the workspace (no output)
//         (then scan_combine_kernel walks the chunks of a sequence: start state of every chunk, final state)
// MODE 2  per chunk: outputs, starting from the chunk's start state in the workspace
template <typename T, int N, int MODE>
__global__ __launch_bounds__(256) void selective_scan_kernel(
    T* __restrict__ out, const T* __restrict__ u, const T* __restrict__ delta, const float* __restrict__ A,
    const T* __restrict__ Bm, const T* __restrict__ Cm, const float* __restrict__ D, const T* __restrict__ z,
    const float* __restrict__ delta_bias, float* __restrict__ ssm_state, const int* __restrict__ qsl,
    const int* __restrict__ seq_lens, const int* __restrict__ block_table, long bt_stride, int C, long u_stride,
    long d_stride, long b_stride, long c_stride, long z_stride, long o_stride, int dt_div, int a_scalar,
    int group_ch, int softplus, int chunk, float* __restrict__ ws_p, float* __restrict__ ws_h) {
  constexpr int NL = N / 16;
  const int s = blockIdx.y;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int c = gid / NL, part = gid % NL;
  if (c >= C) return;       // NL divides 64, so the lanes of one channel leave together
  const int sl = seq_lens[s];
  if (sl <= 0) return;
  const int t0 = qsl[s], L = qsl[s + 1] - t0;
  const bool has_init = sl - L > 0;
  int ta = t0, tb = t0 + L;
  long ws_off = 0;
  if (MODE != 0) {
    ta = t0 + blockIdx.z * chunk;
    if (ta >= tb) return;
    tb = min(tb, ta + chunk);
    ws_off = ((((long)s * gridDim.z + blockIdx.z) * C + c) * N) + part * 16;
  }
  float* hp = ssm_state + ((long)block_table[s * bt_stride] * C + c) * N + part * 16;
  const int e = c / dt_div;
  float a[16], h[16], pr[16];
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    a[n] = a_scalar ? A[e] : A[(long)c * N + part * 16 + n];
    if (MODE == 0) h[n] = has_init ? hp[n] : 0.f;
    else if (MODE == 1) h[n] = 0.f;
    else h[n] = ws_p[ws_off + n];            // start state left there by the combine pass
    pr[n] = 1.f;
  }
  const float dv = D ? D[e] : 0.f;
  const float db = delta_bias ? delta_bias[e] : 0.f;
  const long bc_off = (long)(c / group_ch) * N + part * 16;
  for (int t = ta; t < tb; ++t) {
    float dt = (float)delta[(long)t * d_stride + e] + db;
    if (softplus) dt = softplus_f(dt);
    const float x = (float)u[(long)t * u_stride + c];
    const float dx = dt * x;
    const T* bp = Bm + (long)t * b_stride + bc_off;
    const T* cp = Cm + (long)t * c_stride + bc_off;
    float y = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      const float dec = __expf(dt * a[n]);
      h[n] = dec * h[n] + dx * (float)bp[n];
      if (MODE == 1) pr[n] *= dec;
      else y += (float)cp[n] * h[n];
    }
    if (MODE != 1) {
#pragma unroll
      for (int o = NL / 2; o > 0; o >>= 1) y += __shfl_xor(y, o, 64);
      if (part == 0) {
        y += dv * x;
        if (z) y *= silu_f((float)z[(long)t * z_stride + c]);
        out[(long)t * o_stride + c] = (T)y;
      }
    }
  }
  if (MODE == 0) {
#pragma unroll
    for (int n = 0; n < 16; ++n) hp[n] = h[n];
  } else if (MODE == 1) {
#pragma unroll
    for (int n = 0; n < 16; ++n) { ws_p[ws_off + n] = pr[n]; ws_h[ws_off + n] = h[n]; }
  }
}

// One thread per (sequence, channel, state): walk the sequence's chunks, replace every chunk's P by the
// state the chunk starts from, and leave the final state in the request's slot.
__global__ __launch_bounds__(256) void scan_combine_kernel(float* __restrict__ ws_p, const float* __restrict__ ws_h,
                                                           float* __restrict__ ssm_state,
                                                           const int* __restrict__ qsl,
                                                           const int* __restrict__ seq_lens,
                                                           const int* __restrict__ block_table, long bt_stride,
                                                           long CN, int chunk, int nchunks) {
  const int s = blockIdx.y;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= CN) return;
  const int sl = seq_lens[s];
  if (sl <= 0) return;
  const int L = qsl[s + 1] - qsl[s];
  float* hp = ssm_state + (long)block_table[s * bt_stride] * CN + i;
  float h = (sl - L > 0) ? *hp : 0.f;
  const int nc = (L + chunk - 1) / chunk;
  for (int k = 0; k < nc; ++k) {
    const long off = ((long)s * nchunks + k) * CN + i;
    const float p = ws_p[off];
    ws_p[off] = h;
    h = p * h + ws_h[off];
  }
  *hp = h;
}

VDA_EXPORT int vda_selective_scan(void* out, const void* u, const void* delta, const float* A, const void* Bm,
                                  const void* Cm, const float* D, const void* z, const float* delta_bias,
                                  float* ssm_state, const int* qsl, const int* seq_lens, const int* block_table,
                                  long bt_stride, int S, int C, int N, long u_stride, long d_stride, long b_stride,
                                  long c_stride, long z_stride, long o_stride, int dt_div, int a_scalar, int group_ch,
                                  int softplus, int chunk, int nchunks, float* ws_p, float* ws_h, int dtype,
                                  hipStream_t st) {
  if (S <= 0 || C <= 0) return 0;
  if (N != 16 && N != 32 && N != 64 && N != 128 && N != 256) return -1;
  if (dt_div < 1 || group_ch < 1) return -2;
  const bool chunked = nchunks > 1;
  if (chunked && (!ws_p || !ws_h || chunk < 1)) return -4;
  const long threads = (long)C * (N / 16);
  dim3 block(256), grid((unsigned)((threads + 255) / 256), S, chunked ? nchunks : 1);
#define L(TT, NN, MODE) hipLaunchKernelGGL((selective_scan_kernel<TT, NN, MODE>), grid, block, 0, st, (TT*)out,             (const TT*)u, (const TT*)delta, A, (const TT*)Bm, (const TT*)Cm, D, (const TT*)z, delta_bias, ssm_state, qsl,           seq_lens, block_table, bt_stride, C, u_stride, d_stride, b_stride, c_stride, z_stride, o_stride, dt_div, a_scalar,      group_ch, softplus, chunk, ws_p, ws_h)
#define BYN(TT, MODE) do { if (N == 16) L(TT, 16, MODE); else if (N == 32) L(TT, 32, MODE);                                    else if (N == 64) L(TT, 64, MODE); else if (N == 128) L(TT, 128, MODE);                             else L(TT, 256, MODE); } while (0)
#define RUN(TT) do {                                                                                                   \
    if (!chunked) { BYN(TT, 0); }                                                                                      \
    else {                                                                                                             \
      BYN(TT, 1);                                                                                                      \
      const long CN = (long)C * N;                                                                                     \
      hipLaunchKernelGGL(scan_combine_kernel, dim3((unsigned)((CN + 255) / 256), S), block, 0, st, ws_p, ws_h,         \
                         ssm_state, qsl, seq_lens, block_table, bt_stride, CN, chunk, nchunks);                        \
      BYN(TT, 2);                                                                                                      \
    } } while (0)
  if (dtype == VDA_BF16) RUN(bf16_t);
  else if (dtype == VDA_F16) RUN(f16_t);
  else if (dtype == VDA_F32) RUN(float);
  else return -3;
#undef RUN
#undef BYN
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}
