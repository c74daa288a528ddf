// Rotary embedding, KV-cache writes, and the fused rope + cache-write used on
// the hot path (one launch instead of three per layer).
// Reference ops: csrc/pos_encoding_kernels.cu (rotary_embedding),
// csrc/cache_kernels.cu (reshape_and_cache_flash, copy_blocks, gather_cache).
//
// cos_sin cache: fp32 [max_pos, rot_dim] = [cos(rot_dim/2) | sin(rot_dim/2)].
// K cache [nblk, Hkv, BS, D]; V cache [nblk, Hkv, D, BS] (see attention.hip).
#include "common.h"

// Rotate 8 pairs of one head. NEOX: pairs are (x[i], x[i + rot/2]); else (x[2i], x[2i+1]).
template <typename T, bool NEOX>
__device__ __forceinline__ void rotate8(T* head_ptr, const float* cs, int rot_half, int i8) {
  using V8 = typename Vec<T>::v8;
  const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + i8), c1 = *reinterpret_cast<const f32x4*>(cs + i8 + 4);
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(cs + rot_half + i8),
              s1 = *reinterpret_cast<const f32x4*>(cs + rot_half + i8 + 4);
  const float c[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
  const float s[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
  if (NEOX) {
    V8 a = ld16<V8>(head_ptr + i8), b = ld16<V8>(head_ptr + rot_half + i8);
    V8 oa, ob;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (float)a[j], y = (float)b[j];
      oa[j] = (T)(x * c[j] - y * s[j]);
      ob[j] = (T)(y * c[j] + x * s[j]);
    }
    st16<V8>(head_ptr + i8, oa);
    st16<V8>(head_ptr + rot_half + i8, ob);
  } else {
    V8 a = ld16<V8>(head_ptr + 2 * i8), b = ld16<V8>(head_ptr + 2 * i8 + 8);
    V8 oa, ob;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = (float)a[2 * j], y0 = (float)a[2 * j + 1];
      oa[2 * j] = (T)(x0 * c[j] - y0 * s[j]);
      oa[2 * j + 1] = (T)(y0 * c[j] + x0 * s[j]);
      const float x1 = (float)b[2 * j], y1 = (float)b[2 * j + 1];
      ob[2 * j] = (T)(x1 * c[j + 4] - y1 * s[j + 4]);
      ob[2 * j + 1] = (T)(y1 * c[j + 4] + x1 * s[j + 4]);
    }
    st16<V8>(head_ptr + 2 * i8, oa);
    st16<V8>(head_ptr + 2 * i8 + 8, ob);
  }
}

// In-place RoPE on q [T, Hq, D] and k [T, Hkv, D] (k may be null).
template <typename T, bool NEOX>
__global__ __launch_bounds__(256) void rotary_kernel(const long* __restrict__ positions, T* __restrict__ q,
                                                     T* __restrict__ k, const float* __restrict__ cos_sin,
                                                     int rot_dim, int D, int Hq, int Hkv, long q_stride,
                                                     long k_stride) {
  const long t = blockIdx.x;
  const float* cs = cos_sin + positions[t] * rot_dim;
  const int rot_half = rot_dim >> 1;
  const int per_head = rot_half >> 3;       // 8-pair chunks per head
  const int nq = Hq * per_head, nk = (k ? Hkv : 0) * per_head;
  for (int i = threadIdx.x; i < nq + nk; i += blockDim.x) {
    if (i < nq) {
      rotate8<T, NEOX>(q + t * q_stride + (long)(i / per_head) * D, cs, rot_half, (i % per_head) * 8);
    } else {
      const int j = i - nq;
      rotate8<T, NEOX>(k + t * k_stride + (long)(j / per_head) * D, cs, rot_half, (j % per_head) * 8);
    }
  }
}

// Write k [T, Hkv, D], v [T, Hkv, D] into the paged cache by slot_mapping (-1 = skip).
template <typename T, bool F8>
__global__ __launch_bounds__(256) void reshape_and_cache_kernel(const T* __restrict__ k, const T* __restrict__ v,
                                                                typename KVElem<T, F8>::type* __restrict__ kc,
                                                                typename KVElem<T, F8>::type* __restrict__ vc,
                                                                const long* __restrict__ slot_mapping, int Hkv,
                                                                int D, int BS, long k_stride, long v_stride,
                                                                float k_inv, float v_inv) {
  using V8 = typename Vec<T>::v8;
  using CT = typename KVElem<T, F8>::type;
  const long t = blockIdx.x;
  const long slot = slot_mapping[t];
  if (slot < 0) return;
  const long blk = slot / BS;
  const int off = (int)(slot % BS);
  const int nv = Hkv * (D >> 3);
  for (int i = threadIdx.x; i < nv; i += blockDim.x) {
    const int h = i / (D >> 3), d8 = (i % (D >> 3)) * 8;
    const V8 kk = ld16<V8>(k + t * k_stride + (long)h * D + d8);
    KVStore<T, F8>::st8(kc + ((blk * Hkv + h) * BS + off) * (long)D + d8, kk, k_inv);
    const V8 vv = ld16<V8>(v + t * v_stride + (long)h * D + d8);
    CT* vp = vc + ((blk * Hkv + h) * (long)D + d8) * BS + off;
#pragma unroll
    for (int j = 0; j < 8; ++j) KVStore<T, F8>::st1(vp + (long)j * BS, vv[j], v_inv);
  }
}

// Fused: rotate q in place, rotate k and write it to the K cache, write v to the
// V cache.  qkv is one row-major buffer [T, (Hq + 2 Hkv) * D].
template <typename T, bool NEOX, bool F8>
__global__ __launch_bounds__(256) void rope_and_cache_kernel(
    const long* __restrict__ positions, T* __restrict__ qkv, const float* __restrict__ cos_sin,
    typename KVElem<T, F8>::type* __restrict__ kc, typename KVElem<T, F8>::type* __restrict__ vc,
    const long* __restrict__ slot_mapping, int rot_dim, int D, int Hq, int Hkv, int BS, long row_stride,
    float k_inv, float v_inv) {
  using V8 = typename Vec<T>::v8;
  using CT = typename KVElem<T, F8>::type;
  const long t = blockIdx.x;
  T* row = qkv + t * row_stride;
  T* q = row;
  T* k = row + (long)Hq * D;
  const T* v = k + (long)Hkv * D;
  const float* cs = cos_sin + positions[t] * rot_dim;
  const int rot_half = rot_dim >> 1;
  const int per_head = rot_half >> 3;
  const int nrot = (Hq + Hkv) * per_head;
  for (int i = threadIdx.x; i < nrot; i += blockDim.x) {
    const int h = i / per_head;     // q heads first, then k heads (contiguous in the row)
    rotate8<T, NEOX>(q + (long)h * D, cs, rot_half, (i % per_head) * 8);
  }
  const long slot = slot_mapping[t];
  if (slot < 0) return;
  __syncthreads();                  // k of this row is rotated (same block wrote it)
  const long blk = slot / BS;
  const int off = (int)(slot % BS);
  const int nv = Hkv * (D >> 3);
  for (int i = threadIdx.x; i < nv; i += blockDim.x) {
    const int h = i / (D >> 3), d8 = (i % (D >> 3)) * 8;
    const V8 kk = ld16<V8>(k + (long)h * D + d8);
    KVStore<T, F8>::st8(kc + ((blk * Hkv + h) * BS + off) * (long)D + d8, kk, k_inv);
    const V8 vv = ld16<V8>(v + (long)h * D + d8);
    CT* vp = vc + ((blk * Hkv + h) * (long)D + d8) * BS + off;
#pragma unroll
    for (int j = 0; j < 8; ++j) KVStore<T, F8>::st1(vp + (long)j * BS, vv[j], v_inv);
  }
}

VDA_EXPORT int vda_rotary_embedding(const long* positions, void* q, void* k, const float* cos_sin, int T_,
                                    int rot_dim, int D, int Hq, int Hkv, long q_stride, long k_stride,
                                    int is_neox, int dtype, hipStream_t st) {
  if (rot_dim % 16 || D % 8 || rot_dim > D) return -1;
  if (T_ == 0) return 0;
  dim3 grid(T_), block(256);
#define L(TT, N) hipLaunchKernelGGL((rotary_kernel<TT, N>), grid, block, 0, st, positions, (TT*)q, (TT*)k, cos_sin, \
                                    rot_dim, D, Hq, Hkv, q_stride, k_stride)
  if (dtype == VDA_BF16) { if (is_neox) L(bf16_t, true); else L(bf16_t, false); }
  else if (dtype == VDA_F16) { if (is_neox) L(f16_t, true); else L(f16_t, false); }
  else return -3;
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

// kv_fp8 != 0: the caches hold e4m3 bytes, stored value = x / scale.
VDA_EXPORT int vda_reshape_and_cache(const void* k, const void* v, void* kc, void* vc, const long* slot_mapping,
                                     int T_, int Hkv, int D, int BS, long k_stride, long v_stride, int dtype,
                                     int kv_fp8, float k_scale, float v_scale, hipStream_t st) {
  if (D % 8) return -1;
  if (T_ == 0) return 0;
  dim3 grid(T_), block(256);
  const float ki = 1.f / k_scale, vi = 1.f / v_scale;
#define L(TT, F) hipLaunchKernelGGL((reshape_and_cache_kernel<TT, F>), grid, block, 0, st, (const TT*)k, (const TT*)v, \
                                    (typename KVElem<TT, F>::type*)kc, (typename KVElem<TT, F>::type*)vc, slot_mapping, \
                                    Hkv, D, BS, k_stride, v_stride, ki, vi)
  if (dtype == VDA_BF16) { if (kv_fp8) L(bf16_t, true); else L(bf16_t, false); }
  else if (dtype == VDA_F16) { if (kv_fp8) L(f16_t, true); else L(f16_t, false); }
  else return -3;
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_rope_and_cache(const long* positions, void* qkv, const float* cos_sin, void* kc, void* vc,
                                  const long* slot_mapping, int T_, int rot_dim, int D, int Hq, int Hkv, int BS,
                                  long row_stride, int is_neox, int dtype, int kv_fp8, float k_scale,
                                  float v_scale, hipStream_t st) {
  if (rot_dim % 16 || D % 8 || rot_dim > D) return -1;
  if (T_ == 0) return 0;
  dim3 grid(T_), block(256);
  const float ki = 1.f / k_scale, vi = 1.f / v_scale;
#define L(TT, N, F) hipLaunchKernelGGL((rope_and_cache_kernel<TT, N, F>), grid, block, 0, st, positions, (TT*)qkv, \
                                       cos_sin, (typename KVElem<TT, F>::type*)kc, (typename KVElem<TT, F>::type*)vc, \
                                       slot_mapping, rot_dim, D, Hq, Hkv, BS, row_stride, ki, vi)
#define LN(TT, F) do { if (is_neox) L(TT, true, F); else L(TT, false, F); } while (0)
  if (dtype == VDA_BF16) { if (kv_fp8) LN(bf16_t, true); else LN(bf16_t, false); }
  else if (dtype == VDA_F16) { if (kv_fp8) LN(f16_t, true); else LN(f16_t, false); }
  else return -3;
#undef LN
#undef L
  VDA_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------
// block-level cache maintenance
// ---------------------------------------------------------------------------
// Copy whole blocks inside one cache tensor: pairs[i] = (src_block, dst_block).
__global__ __launch_bounds__(256) void copy_blocks_kernel(u32x4* __restrict__ cache, const long* __restrict__ pairs,
                                                          long block_vec16) {
  const long src = pairs[2 * blockIdx.x], dst = pairs[2 * blockIdx.x + 1];
  const u32x4* s = cache + src * block_vec16;
  u32x4* d = cache + dst * block_vec16;
  for (long i = threadIdx.x; i < block_vec16; i += blockDim.x) d[i] = s[i];
}

VDA_EXPORT int vda_copy_blocks(void* cache, const long* pairs, int num_pairs, long block_bytes, hipStream_t st) {
  if (block_bytes % 16) return -1;
  if (num_pairs == 0) return 0;
  hipLaunchKernelGGL(copy_blocks_kernel, dim3(num_pairs), dim3(256), 0, st, (u32x4*)cache, pairs, block_bytes / 16);
  VDA_CHECK_LAUNCH();
  return 0;
}

// Copy blocks between two cache tensors (possibly on different devices mapped
// through peer access, or pinned host memory): `copy` 16-byte units starting at unit `dst_off` of block
// dst_ids[i] of dst come from unit `src_off` on of block src_ids[i] of src.  The two sides may have
// different block pitches: a decode instance with another tensor-parallel degree than the prefill
// instance copies only its slice of the KV heads of every block (heads are the outer dimension of a block).
__global__ __launch_bounds__(256) void transfer_blocks_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src,
                                                              const long* __restrict__ dst_ids,
                                                              const long* __restrict__ src_ids, long copy,
                                                              long dst_pitch, long src_pitch, long dst_off,
                                                              long src_off) {
  const u32x4* s = src + src_ids[blockIdx.x] * src_pitch + src_off;
  u32x4* d = dst + dst_ids[blockIdx.x] * dst_pitch + dst_off;
  for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < copy; i += (long)gridDim.y * blockDim.x)
    d[i] = s[i];
}

VDA_EXPORT int vda_transfer_blocks_ex(void* dst, const void* src, const long* dst_ids, const long* src_ids, int n,
                                      long copy_bytes, long dst_pitch_bytes, long src_pitch_bytes, long dst_off_bytes,
                                      long src_off_bytes, hipStream_t st) {
  if (copy_bytes % 16 || dst_pitch_bytes % 16 || src_pitch_bytes % 16 || dst_off_bytes % 16 || src_off_bytes % 16)
    return -1;
  if (copy_bytes + dst_off_bytes > dst_pitch_bytes || copy_bytes + src_off_bytes > src_pitch_bytes) return -1;
  if (n == 0 || copy_bytes == 0) return 0;
  const long v = copy_bytes / 16;
  int gy = (int)((v + 255) / 256);
  if (gy > 16) gy = 16;
  hipLaunchKernelGGL(transfer_blocks_kernel, dim3(n, gy), dim3(256), 0, st, (u32x4*)dst, (const u32x4*)src, dst_ids,
                     src_ids, v, dst_pitch_bytes / 16, src_pitch_bytes / 16, dst_off_bytes / 16, src_off_bytes / 16);
  VDA_CHECK_LAUNCH();
  return 0;
}

VDA_EXPORT int vda_transfer_blocks(void* dst, const void* src, const long* dst_ids, const long* src_ids, int n,
                                   long block_bytes, hipStream_t st) {
  return vda_transfer_blocks_ex(dst, src, dst_ids, src_ids, n, block_bytes, block_bytes, block_bytes, 0, 0, st);
}

// Gather the K (or V) rows of one sequence back into token-major [T, Hkv, D]
// (debug, KV export for the storage connector, chunked-context paths).
template <typename T>
__global__ __launch_bounds__(256) void gather_cache_kernel(T* __restrict__ k_out, T* __restrict__ v_out,
                                                           const T* __restrict__ kc, const T* __restrict__ vc,
                                                           const long* __restrict__ slots, int Hkv, int D, int BS) {
  using V8 = typename Vec<T>::v8;
  const long t = blockIdx.x;
  const long slot = slots[t];
  const long blk = slot / BS;
  const int off = (int)(slot % BS);
  const int nv = Hkv * (D >> 3);
  for (int i = threadIdx.x; i < nv; i += blockDim.x) {
    const int h = i / (D >> 3), d8 = (i % (D >> 3)) * 8;
    st16<V8>(k_out + (t * Hkv + h) * (long)D + d8, ld16<V8>(kc + ((blk * Hkv + h) * BS + off) * (long)D + d8));
    const T* vp = vc + ((blk * Hkv + h) * (long)D + d8) * BS + off;
    V8 vv;
#pragma unroll
    for (int j = 0; j < 8; ++j) vv[j] = vp[(long)j * BS];
    st16<V8>(v_out + (t * Hkv + h) * (long)D + d8, vv);
  }
}

VDA_EXPORT int vda_gather_cache(void* k_out, void* v_out, const void* kc, const void* vc, const long* slots, int T_,
                                int Hkv, int D, int BS, int dtype, hipStream_t st) {
  if (D % 8) return -1;
  if (T_ == 0) return 0;
  if (dtype == VDA_BF16)
    hipLaunchKernelGGL(gather_cache_kernel<bf16_t>, dim3(T_), dim3(256), 0, st, (bf16_t*)k_out, (bf16_t*)v_out,
                       (const bf16_t*)kc, (const bf16_t*)vc, slots, Hkv, D, BS);
  else if (dtype == VDA_F16)
    hipLaunchKernelGGL(gather_cache_kernel<f16_t>, dim3(T_), dim3(256), 0, st, (f16_t*)k_out, (f16_t*)v_out,
                       (const f16_t*)kc, (const f16_t*)vc, slots, Hkv, D, BS);
  else return -3;
  VDA_CHECK_LAUNCH();
  return 0;
}
