// Lightning (linear) attention, one-token step: per sequence and head
//     S <- exp(-slope) * S + k^T v        (S: [D, D] fp32 in the sequence's state slot)
//     out = q S
// in ONE pass over S (read, update, write back, reduce) -- the state is the only large operand (64 KB per head at
// D = 128), so the step runs at the HBM rate of 2 x |S|.
// One workgroup per (sequence, head): 256 threads as (256 / (D/4)) row groups x D/4 column quads; a row of S is read
// as consecutive float4 (512 B at D = 128), every thread keeps a partial q.S sum for its four columns, the row
// groups are summed through LDS.
// Role of the reference's vllm/model_executor/layers/lightning_attn.py::linear_decode_forward_triton.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void lightning_decode_kernel(
    float* __restrict__ out, const T* __restrict__ qkv, float* __restrict__ state, const float* __restrict__ slope,
    const int* __restrict__ seq_lens, const int* __restrict__ block_table, long bt_stride, int H, int D,
    long qkv_stride) {
  __shared__ float sq[256], sk[256], sv[256];
  __shared__ float red[1024];
  const int s = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int sl = seq_lens[s];
  if (sl <= 0) return;                               // padded row of a captured graph
  const T* tok = qkv + (long)s * qkv_stride + (long)h * 3 * D;
  if (tid < D) {
    sq[tid] = (float)tok[tid];
    sk[tid] = (float)tok[D + tid];
    sv[tid] = (float)tok[2 * D + tid];
  }
  __syncthreads();
  const int quads = D >> 2, groups = 256 / quads;
  const int cq = tid % quads, rg = tid / quads;
  const float decay = sl <= 1 ? 0.f : __expf(-slope[h]);      // the first token of a sequence starts from zero
  float* S = state + (((long)block_table[s * bt_stride] * H + h) * D) * D;
  const float4 v4 = *reinterpret_cast<const float4*>(&sv[cq * 4]);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int d = rg; d < D; d += groups) {
    float4* p = reinterpret_cast<float4*>(S + (long)d * D + cq * 4);
    float4 x = decay == 0.f ? make_float4(0.f, 0.f, 0.f, 0.f) : *p;
    const float kd = sk[d], qd = sq[d];
    x.x = x.x * decay + kd * v4.x;
    x.y = x.y * decay + kd * v4.y;
    x.z = x.z * decay + kd * v4.z;
    x.w = x.w * decay + kd * v4.w;
    *p = x;
    acc.x += qd * x.x;
    acc.y += qd * x.y;
    acc.z += qd * x.z;
    acc.w += qd * x.w;
  }
  *reinterpret_cast<float4*>(&red[rg * D + cq * 4]) = acc;
  __syncthreads();
  if (tid < D) {
    float t = 0.f;
    for (int g = 0; g < groups; ++g) t += red[g * D + tid];
    out[((long)s * H + h) * D + tid] = t;
  }
}

VDA_EXPORT int vda_lightning_decode(float* out, const void* qkv, float* state, const float* slope, const int* seq_lens,
                                    const int* block_table, long bt_stride, int S, int H, int D, long qkv_stride,
                                    int dtype, hipStream_t st) {
  if (S <= 0) return 0;
  if (D < 16 || D > 256 || (D & 3) || 256 % (D >> 2)) return -1;
#define RUN(T)                                                                                              \
  hipLaunchKernelGGL(lightning_decode_kernel<T>, dim3(S, H), dim3(256), 0, st, out, (const T*)qkv, state, \
                     slope, seq_lens, block_table, bt_stride, H, D, qkv_stride)
  if (dtype == VDA_BF16) RUN(bf16_t);
  else if (dtype == VDA_F16) RUN(f16_t);
  else if (dtype == VDA_F32) RUN(float);
  else return -3;
#undef RUN
  VDA_CHECK_LAUNCH();
  return 0;
}
