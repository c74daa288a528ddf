// Peer-memory all-reduce over xGMI for one node (2..8 GPUs, fully connected).
//
// Every rank owns an IPC-exported arena: two staging halves (selected by the
// parity of a per-block call counter, which removes the end-of-call barrier)
// and a small uncached signal block.  A call first copies the input into the
// rank's own staging half, so ANY device tensor can be reduced and nothing has
// to be registered when a hipGraph is captured.
//
//   one-shot  (small, latency bound): barrier; every rank reads the whole buffer
//             of all peers over all links at once and sums in fp32.
//   two-shot  (large, bandwidth bound): barrier; rank r reduces slice r of all
//             peers into its result area (direct reduce-scatter); barrier; every
//             rank gathers slice s from peer s (direct all-gather).  Per link
//             traffic is M/N per phase, all N-1 links busy in parallel; a ring would
//             be bound by one link.
//
// Cross-device ordering: data stores -> system-scope fence -> release store of a
// monotone arrival counter into the peer's signal block; the waiter acquires at
// system scope before reading peer data.  Every spin is bounded: on time-out an
// error word is set and the kernel leaves, so a lost peer can never hang the GPU.
// Role of the reference's csrc/custom_all_reduce.cuh (1-stage :296-312, 2-stage
// :319-364, ROCm scoped atomics :243-282) and csrc/quickreduce.
#include "peer_sync.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

struct CarCtx {
  int rank, world;
  size_t half_bytes;             // bytes of one staging half (input area + result area)
  size_t in_bytes;               // input area bytes inside a half
  char* data[MAX_RANKS];         // peer arenas (own entry is the local pointer)
  Signal* sig[MAX_RANKS];
  unsigned long long spin_limit;
  uint32_t* host_err;            // pinned host word: non-zero after a time-out
};

struct CarArgs {
  char* data[MAX_RANKS];
  Signal* sig[MAX_RANKS];
  int rank, world;
  size_t half_bytes, in_bytes;
  unsigned long long spin_limit;
  uint32_t* host_err;
};

template <typename T> struct Pack { T v[16 / sizeof(T)]; };

template <typename T> __device__ __forceinline__ void acc_add(float* a, const u32x4& raw) {
  constexpr int N = 16 / sizeof(T);
  const T* p = reinterpret_cast<const T*>(&raw);
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] += (float)p[i];
}
template <typename T> __device__ __forceinline__ u32x4 acc_pack(const float* a) {
  constexpr int N = 16 / sizeof(T);
  u32x4 raw;
  T* p = reinterpret_cast<T*>(&raw);
#pragma unroll
  for (int i = 0; i < N; ++i) p[i] = (T)a[i];
  return raw;
}

__device__ __forceinline__ bool cross_barrier(const CarArgs& a, uint32_t target) {
  return peer_barrier(a.sig, a.rank, a.world, a.spin_limit, target, a.host_err);
}

// Element ownership is FIXED: 16-byte element i belongs to thread (i mod GRID_THREADS) of a
// MAX_BLOCKS x 512 grid, for every call and every message size.  A call launches only the blocks
// that own elements of its message (two for a 16 KiB decode message, all 64 from 512 KiB up): the
// barrier arrival words, pass counters and call parities are per block, and the number of blocks is
// a function of the message size alone, so every rank leaves the same blocks out and a block that
// sits out a call simply keeps its counters.  Each
// block only synchronises with the same block of the peers, and it only ever touches
// elements it owns, so "my block passed barrier k+1" implies "the peer block finished
// reading my call-k data": that is what makes the two staging halves (call parity) safe
// without an end-of-call barrier.  In the two-shot form sweep j (elements j*GRID_THREADS..)
// is reduced by rank j mod world, which keeps the ownership rule and spreads slices evenly.
template <typename T, bool TWO_SHOT>
__global__ __launch_bounds__(512) void car_kernel(CarArgs a, const u32x4* __restrict__ in, u32x4* __restrict__ out,
                                                  size_t n16) {
  Signal* me = a.sig[a.rank];
  __shared__ uint32_t s_cnt, s_call;
  if (threadIdx.x == 0) { s_cnt = me->counter[blockIdx.x]; s_call = me->calls[blockIdx.x]; }
  __syncthreads();
  const uint32_t cnt = s_cnt;                       // barriers this block has passed so far
  const size_t half_off = (size_t)(s_call & 1) * a.half_bytes;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)MAX_BLOCKS * blockDim.x;      // of the full grid, whatever was launched
  constexpr int N = 16 / sizeof(T);

  // 0) my input -> my staging half (peers read it from there)
  u32x4* my_in = reinterpret_cast<u32x4*>(a.data[a.rank] + half_off);
  for (size_t i = tid; i < n16; i += stride) my_in[i] = in[i];
  bool ok = cross_barrier(a, cnt + 1);

  if (!TWO_SHOT) {
    if (ok) {
      for (size_t i = tid; i < n16; i += stride) {
        u32x4 v[MAX_RANKS];
#pragma unroll
        for (int r = 0; r < MAX_RANKS; ++r)
          if (r < a.world) v[r] = reinterpret_cast<const u32x4*>(a.data[r] + half_off)[i];
        float acc[N];
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = 0.f;
#pragma unroll
        for (int r = 0; r < MAX_RANKS; ++r)
          if (r < a.world) acc_add<T>(acc, v[r]);
        out[i] = acc_pack<T>(acc);
      }
    }
    if (threadIdx.x == 0) { me->counter[blockIdx.x] = cnt + 1; me->calls[blockIdx.x] = s_call + 1; }
    return;
  }

  u32x4* my_res = reinterpret_cast<u32x4*>(a.data[a.rank] + half_off + a.in_bytes);
  if (ok) {
    // sweeps j = rank, rank + world, ... are mine
    size_t k = 0;
    for (size_t i = tid + (size_t)a.rank * stride; i < n16; i += stride * a.world, ++k) {
      u32x4 v[MAX_RANKS];
#pragma unroll
      for (int r = 0; r < MAX_RANKS; ++r)
        if (r < a.world) v[r] = reinterpret_cast<const u32x4*>(a.data[(a.rank + r) % a.world] + half_off)[i];
      float acc[N];
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] = 0.f;
#pragma unroll
      for (int r = 0; r < MAX_RANKS; ++r)
        if (r < a.world) acc_add<T>(acc, v[r]);
      my_res[tid + k * stride] = acc_pack<T>(acc);
    }
  }
  ok = cross_barrier(a, cnt + 2) && ok;
  if (ok) {
    size_t j = 0;
    for (size_t i = tid; i < n16; i += stride, ++j) {
      const int src = (int)(j % a.world);
      const u32x4* res = reinterpret_cast<const u32x4*>(a.data[src] + half_off + a.in_bytes);
      out[i] = res[tid + (j / a.world) * stride];
    }
  }
  if (threadIdx.x == 0) { me->counter[blockIdx.x] = cnt + 2; me->calls[blockIdx.x] = s_call + 1; }
}

// Reduce-scatter and all-gather over dimension 0 with the same arena, counters and ownership rule as car_kernel
// (element = its index in the FULL message, so a block meets the same elements in every kind of call and the
// parity-selected halves stay safe when the three kinds alternate):
//   GATHER = false: in is the full message [n16]; rank r gets out = sum over ranks of elements
//                   [r * n16 / world, (r + 1) * n16 / world) -- the first half of a two-shot all-reduce with
//                   contiguous slices (what sequence parallelism needs: whole token rows per rank);
//   GATHER = true:  in is this rank's slice; out is the full message -- the second half.
// Role of the reference's reduce_scatter / all_gather in vllm/distributed/device_communicators/base_device_communicator.py
// as used by vllm/compilation/sequence_parallelism.py:103-115.
template <typename T, bool GATHER>
__global__ __launch_bounds__(512) void car_slice_kernel(CarArgs a, const u32x4* __restrict__ in, u32x4* __restrict__ out,
                                                        size_t n16) {
  Signal* me = a.sig[a.rank];
  __shared__ uint32_t s_cnt, s_call;
  if (threadIdx.x == 0) { s_cnt = me->counter[blockIdx.x]; s_call = me->calls[blockIdx.x]; }
  __syncthreads();
  const uint32_t cnt = s_cnt;
  const size_t half_off = (size_t)(s_call & 1) * a.half_bytes;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)MAX_BLOCKS * blockDim.x;
  const size_t slice = n16 / a.world, lo = slice * a.rank, hi = lo + slice;
  constexpr int N = 16 / sizeof(T);
  u32x4* my_in = reinterpret_cast<u32x4*>(a.data[a.rank] + half_off);
  if (GATHER) {
    for (size_t i = tid; i < n16; i += stride)
      if (i >= lo && i < hi) my_in[i] = in[i - lo];
  } else {
    for (size_t i = tid; i < n16; i += stride)
      if (i < lo || i >= hi) my_in[i] = in[i];              // what the peers read; my own slice stays in `in`
  }
  const bool ok = cross_barrier(a, cnt + 1);
  if (ok) {
    if (GATHER) {
      for (size_t i = tid; i < n16; i += stride) {
        const int src = (int)(i / slice);
        out[i] = src == a.rank ? in[i - lo] : reinterpret_cast<const u32x4*>(a.data[src] + half_off)[i];
      }
    } else {
      for (size_t i = tid; i < n16; i += stride) {
        if (i < lo || i >= hi) continue;
        u32x4 v[MAX_RANKS];
#pragma unroll
        for (int r = 0; r < MAX_RANKS; ++r)
          if (r < a.world) v[r] = r == a.rank ? in[i] : reinterpret_cast<const u32x4*>(a.data[r] + half_off)[i];
        float acc[N];
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = 0.f;
#pragma unroll
        for (int r = 0; r < MAX_RANKS; ++r)
          if (r < a.world) acc_add<T>(acc, v[r]);
        out[i - lo] = acc_pack<T>(acc);
      }
    }
  }
  if (threadIdx.x == 0) { me->counter[blockIdx.x] = cnt + 1; me->calls[blockIdx.x] = s_call + 1; }
}

// ---------------------------------------------------------------------------
// host API
// ---------------------------------------------------------------------------
// Allocate the arena (2 staging halves) and the uncached signal block; return IPC handles.
VDA_EXPORT int vda_car_alloc(size_t arena_bytes, void** data, void** sig, char* data_handle /*64*/,
                             char* sig_handle /*64*/) {
  HIPCHK(hipMalloc(data, arena_bytes));
  HIPCHK(hipMemset(*data, 0, arena_bytes));
  HIPCHK(hipExtMallocWithFlags(sig, sizeof(Signal), hipDeviceMallocUncached));
  HIPCHK(hipMemset(*sig, 0, sizeof(Signal)));
  HIPCHK(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  HIPCHK(hipIpcGetMemHandle(&h, *data));
  memcpy(data_handle, &h, sizeof(h));
  HIPCHK(hipIpcGetMemHandle(&h, *sig));
  memcpy(sig_handle, &h, sizeof(h));
  return 0;
}

VDA_EXPORT int vda_car_open(const char* handle, void** ptr) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  HIPCHK(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
  return 0;
}

VDA_EXPORT int vda_car_close(void* ptr) { return (int)hipIpcCloseMemHandle(ptr); }
VDA_EXPORT int vda_car_free(void* data, void* sig) { hipFree(data); hipFree(sig); return 0; }
VDA_EXPORT int vda_car_handle_size() { return (int)sizeof(hipIpcMemHandle_t); }

VDA_EXPORT void* vda_car_create(int rank, int world, void** peer_data, void** peer_sig, size_t half_bytes,
                                unsigned long long spin_limit) {
  if (world < 2 || world > MAX_RANKS) return nullptr;
  CarCtx* c = new CarCtx();
  c->rank = rank; c->world = world; c->half_bytes = half_bytes;
  // input area : result area = world : 1 (a result slice is 1/world of the input)
  // the result area holds ceil(sweeps / world) whole sweeps of MAX_BLOCKS*512 elements
  const size_t sweep = (size_t)MAX_BLOCKS * 512 * 16;
  size_t in_b = half_bytes / (world + 1) * world;
  in_b -= in_b % sweep;
  while (in_b > 0 && in_b + ((in_b / sweep + world - 1) / world) * sweep > half_bytes) in_b -= sweep;
  if (in_b == 0) { delete c; return nullptr; }
  c->in_bytes = in_b;
  for (int i = 0; i < world; ++i) { c->data[i] = (char*)peer_data[i]; c->sig[i] = (Signal*)peer_sig[i]; }
  c->spin_limit = spin_limit;
  c->host_err = alloc_host_err();
  return c;
}

VDA_EXPORT void vda_car_destroy(void* ctx) {
  CarCtx* c = (CarCtx*)ctx;
  if (c->host_err) hipHostFree(c->host_err);
  delete c;
}
VDA_EXPORT size_t vda_car_max_bytes(void* ctx) { return ((CarCtx*)ctx)->in_bytes; }

// Non-zero (1 + the peer that did not arrive) after a barrier time-out.  Reads the pinned host word the
// kernels write, so polling it costs nothing; valid once the stream that ran the call has been synchronised.
VDA_EXPORT int vda_car_error(void* ctx) {
  CarCtx* c = (CarCtx*)ctx;
  if (c->host_err) return (int)*(volatile uint32_t*)c->host_err;
  uint32_t e = 0;
  if (hipMemcpy(&e, &c->sig[c->rank]->error, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)e;
}

// dtype: 0 bf16, 1 f16, 2 f32.  nbytes must be a multiple of 16 and <= vda_car_max_bytes.
// algo: 0 auto, 1 one-shot, 2 two-shot.  one-shot and two-shot calls share the call counters,
// so the SAME algorithm must be chosen on every rank for a given call (it is a pure function
// of nbytes and world size here).
VDA_EXPORT int vda_car_allreduce(void* ctx, const void* in, void* out, size_t nbytes, int dtype, int algo,
                                 size_t oneshot_max, hipStream_t st) {
  CarCtx* c = (CarCtx*)ctx;
  if (nbytes % 16 || nbytes > c->in_bytes || nbytes == 0) return -1;
  const size_t n16 = nbytes / 16;
  if (algo == 0) algo = (c->world == 2 || nbytes <= oneshot_max) ? 1 : 2;
  CarArgs a;
  for (int i = 0; i < MAX_RANKS; ++i) { a.data[i] = c->data[i < c->world ? i : 0]; a.sig[i] = c->sig[i < c->world ? i : 0]; }
  a.rank = c->rank; a.world = c->world; a.half_bytes = c->half_bytes; a.in_bytes = c->in_bytes;
  a.spin_limit = c->spin_limit; a.host_err = c->host_err;
  const int threads = 512;
  const int blocks = (int)(n16 >= (size_t)MAX_BLOCKS * threads ? MAX_BLOCKS : (n16 + threads - 1) / threads);
#define L(T, TS) hipLaunchKernelGGL((car_kernel<T, TS>), dim3(blocks), dim3(threads), 0, st, a, (const u32x4*)in, \
                                    (u32x4*)out, n16)
  if (algo == 1) {
    if (dtype == 0) L(bf16_t, false); else if (dtype == 1) L(f16_t, false); else L(float, false);
  } else {
    if (dtype == 0) L(bf16_t, true); else if (dtype == 1) L(f16_t, true); else L(float, true);
  }
#undef L
  return (int)hipGetLastError();
}

// Reduce-scatter (gather = 0: in full [nbytes], out [nbytes / world]) or all-gather (gather = 1: in
// [nbytes / world], out full [nbytes]) over dimension 0.  nbytes: the FULL message, a multiple of 16 * world and
// <= vda_car_max_bytes.  Shares counters with vda_car_allreduce: the same sequence of calls on every rank.
VDA_EXPORT int vda_car_slice(void* ctx, const void* in, void* out, size_t nbytes, int dtype, int gather,
                             hipStream_t st) {
  CarCtx* c = (CarCtx*)ctx;
  if (nbytes == 0 || nbytes % ((size_t)16 * c->world) || nbytes > c->in_bytes) return -1;
  const size_t n16 = nbytes / 16;
  CarArgs a;
  for (int i = 0; i < MAX_RANKS; ++i) { a.data[i] = c->data[i < c->world ? i : 0]; a.sig[i] = c->sig[i < c->world ? i : 0]; }
  a.rank = c->rank; a.world = c->world; a.half_bytes = c->half_bytes; a.in_bytes = c->in_bytes;
  a.spin_limit = c->spin_limit; a.host_err = c->host_err;
  const int threads = 512;
  const int blocks = (int)(n16 >= (size_t)MAX_BLOCKS * threads ? MAX_BLOCKS : (n16 + threads - 1) / threads);
#define L(T, G) hipLaunchKernelGGL((car_slice_kernel<T, G>), dim3(blocks), dim3(threads), 0, st, a, (const u32x4*)in, \
                                   (u32x4*)out, n16)
  if (gather) {
    L(float, true);            // a copy: the element type does not matter
  } else {
    if (dtype == 0) L(bf16_t, false); else if (dtype == 1) L(f16_t, false); else L(float, false);
  }
#undef L
  return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------
// Fused all-reduce + residual add + RMSNorm for decode-sized batches:
//     s = sum over ranks of in;  residual += s;  out = rmsnorm(residual) * (w + wbias)
// in ONE kernel, where the unfused path runs the one-shot all-reduce, writes the sum, and the norm kernel
// reads it back together with the residual (two launches, one extra round trip of [M, H] through memory).
// Role of the reference's all-reduce + RMSNorm fusion pass (vllm/compilation/collective_fusion.py,
// AllReduceRMSNormPattern / AllReduceFusedAddRMSNormPattern, which lowers to a flashinfer kernel there).
//
// Whole ROWS are owned by workgroups: row r belongs to workgroup r mod MAX_BLOCKS on every rank, and its
// staging copy lives in that workgroup's own region of the staging half (region b = bytes
// [b * region, (b + 1) * region), row r at slot r / MAX_BLOCKS), whatever M and H a call has.  A
// workgroup thus only ever touches its own region, locally and on the peers, which is the ownership
// rule that makes the parity-selected halves safe without an end-of-call barrier (see car_kernel); the
// element-strided ownership of car_kernel is a different partition, so this kernel has a context
// (arena, counters) of its own.  Rounding follows the unfused path exactly: the sum is rounded to T,
// added to the residual in fp32 and rounded again, statistics are taken from the rounded values.
template <typename T, int EMAX>
__global__ __launch_bounds__(512) void far_kernel(CarArgs a, const u32x4* __restrict__ in, u32x4* __restrict__ residual,
                                                  const u32x4* __restrict__ weight, u32x4* __restrict__ out, int M,
                                                  int row16, int H, float eps, float wbias, size_t region) {
  Signal* me = a.sig[a.rank];
  __shared__ uint32_t s_cnt, s_call;
  __shared__ float red[8];
  if (threadIdx.x == 0) { s_cnt = me->counter[blockIdx.x]; s_call = me->calls[blockIdx.x]; }
  __syncthreads();
  const uint32_t cnt = s_cnt;
  const size_t off = (size_t)(s_call & 1) * a.half_bytes + (size_t)blockIdx.x * region;
  constexpr int N = 16 / sizeof(T);

  // 0) my partial rows -> my region of my staging half
  u32x4* mine = reinterpret_cast<u32x4*>(a.data[a.rank] + off);
  int slot = 0;
  for (int r = blockIdx.x; r < M; r += MAX_BLOCKS, ++slot)
    for (int e = threadIdx.x; e < row16; e += 512) mine[(size_t)slot * row16 + e] = in[(size_t)r * row16 + e];
  const bool ok = cross_barrier(a, cnt + 1);

  if (ok) {
    slot = 0;
    for (int r = blockIdx.x; r < M; r += MAX_BLOCKS, ++slot) {
      float x[EMAX][N];
      float ss = 0.f;
#pragma unroll
      for (int v = 0; v < EMAX; ++v) {
        const int e = threadIdx.x + v * 512;
        if (e < row16) {
          u32x4 pv[MAX_RANKS];
#pragma unroll
          for (int q = 0; q < MAX_RANKS; ++q)
            if (q < a.world) pv[q] = reinterpret_cast<const u32x4*>(a.data[q] + off)[(size_t)slot * row16 + e];
          float acc[N];
#pragma unroll
          for (int k = 0; k < N; ++k) acc[k] = 0.f;
#pragma unroll
          for (int q = 0; q < MAX_RANKS; ++q)
            if (q < a.world) acc_add<T>(acc, pv[q]);
          u32x4 rraw = residual[(size_t)r * row16 + e];
          T* rp = reinterpret_cast<T*>(&rraw);
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const T s = (T)acc[k];                          // what the unfused all-reduce would have stored
            const T t = (T)((float)s + (float)rp[k]);
            rp[k] = t;
            x[v][k] = (float)t;
            ss += x[v][k] * x[v][k];
          }
          residual[(size_t)r * row16 + e] = rraw;
        }
      }
      // block-wide sum of squares
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
      __syncthreads();                       // red[] of the previous row has been read by everyone
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) tot += red[w];
      const float rstd = rsqrtf(tot / (float)H + eps);
#pragma unroll
      for (int v = 0; v < EMAX; ++v) {
        const int e = threadIdx.x + v * 512;
        if (e < row16) {
          const u32x4 wraw = weight[e];
          const T* wp = reinterpret_cast<const T*>(&wraw);
          u32x4 oraw;
          T* op = reinterpret_cast<T*>(&oraw);
#pragma unroll
          for (int k = 0; k < N; ++k) op[k] = (T)((float)(T)(x[v][k] * rstd) * ((float)wp[k] + wbias));
          out[(size_t)r * row16 + e] = oraw;
        }
      }
    }
  }
  if (threadIdx.x == 0) { me->counter[blockIdx.x] = cnt + 1; me->calls[blockIdx.x] = s_call + 1; }
}

// A context of this kernel is a CarCtx whose in_bytes field holds the bytes of one workgroup region.
VDA_EXPORT void* vda_far_create(int rank, int world, void** peer_data, void** peer_sig, size_t half_bytes,
                                unsigned long long spin_limit) {
  if (world < 2 || world > MAX_RANKS) return nullptr;
  CarCtx* c = new CarCtx();
  c->rank = rank; c->world = world; c->half_bytes = half_bytes;
  c->in_bytes = half_bytes / MAX_BLOCKS / 16 * 16;
  if (c->in_bytes == 0) { delete c; return nullptr; }
  for (int i = 0; i < world; ++i) { c->data[i] = (char*)peer_data[i]; c->sig[i] = (Signal*)peer_sig[i]; }
  c->spin_limit = spin_limit;
  c->host_err = alloc_host_err();
  return c;
}

// in: [M, H] partial sums; residual: [M, H] updated in place; weight: [H]; out: [M, H]; all contiguous and of
// `dtype` (0 bf16, 1 f16).  Returns -1 when the shape does not fit (the caller takes the unfused path; the
// answer depends on M, H and the context only, so every rank decides alike).
VDA_EXPORT int vda_far_run(void* ctx, const void* in, void* residual, const void* weight, void* out, int M, int H,
                           float eps, float wbias, int dtype, hipStream_t st) {
  CarCtx* c = (CarCtx*)ctx;
  if (dtype != 0 && dtype != 1) return -1;
  const size_t row_bytes = (size_t)H * 2;
  if (M <= 0 || H <= 0 || row_bytes % 16) return -1;
  const int row16 = (int)(row_bytes / 16);
  const size_t slots = ((size_t)M + MAX_BLOCKS - 1) / MAX_BLOCKS;
  if (row16 > 4 * 512 || slots * row_bytes > c->in_bytes) return -1;
  CarArgs a;
  for (int i = 0; i < MAX_RANKS; ++i) { a.data[i] = c->data[i < c->world ? i : 0]; a.sig[i] = c->sig[i < c->world ? i : 0]; }
  a.rank = c->rank; a.world = c->world; a.half_bytes = c->half_bytes; a.in_bytes = c->in_bytes;
  a.spin_limit = c->spin_limit; a.host_err = c->host_err;
  // rows r = b, b + MAX_BLOCKS, ... belong to workgroup b: with fewer rows than workgroups the rest own nothing
  const int blocks = M < MAX_BLOCKS ? M : MAX_BLOCKS;
#define L(T, E) hipLaunchKernelGGL((far_kernel<T, E>), dim3(blocks), dim3(512), 0, st, a, (const u32x4*)in, \
                                   (u32x4*)residual, (const u32x4*)weight, (u32x4*)out, M, row16, H, eps, wbias, c->in_bytes)
  const int emax = (row16 + 511) / 512;
  if (dtype == 0) { if (emax <= 1) L(bf16_t, 1); else if (emax <= 2) L(bf16_t, 2); else L(bf16_t, 4); }
  else { if (emax <= 1) L(f16_t, 1); else if (emax <= 2) L(f16_t, 2); else L(f16_t, 4); }
#undef L
  return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------
// Quantised two-shot all-reduce ("quick reduce"): the bytes that cross xGMI are int8, int6 or int4
// with one fp32 scale per 32 values, so a bandwidth-bound all-reduce moves 1.8x / 2.3x / 3.2x less; the
// fourth codec ("FP") sends the 16-bit values themselves through the same two-shot schedule.
// Both shots carry quantised data: inputs are quantised into the staging half, the reducer
// dequantises all peers' slices, sums in fp32 and re-quantises its result, the gather phase
// dequantises into the output.  The error is two rounding steps of absmax/127 (or /7) per group.
//
// Unit = 32 values (64 input bytes); unit u belongs to thread (u mod GRID_THREADS) for every call,
// the reducer of unit (sweep j, block b) is rank (j + b) mod world, and a result lives at the
// unit's own index in the reducer's result area.  A block therefore only ever touches units of
// its own block index, on every rank and in every phase, which is the ownership rule that makes
// the parity-selected staging halves safe (see car_kernel).  The arena of this path is separate
// from the unquantised one, because the two kernels own elements differently.
// Role of the reference's csrc/quickreduce (quick_reduce_impl.cuh:543-696, codecs :60-540).
// ---------------------------------------------------------------------------
template <int BITS> struct QCodec;
template <> struct QCodec<8> {
  static constexpr int DATA_BYTES = 32;
  static constexpr float QMAX = 127.f;
  struct Packed { u32x4 d[2]; };
  __device__ static __forceinline__ Packed encode(const float* v, float inv) {
    Packed p;
    uint32_t* w = reinterpret_cast<uint32_t*>(p.d);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint32_t x = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int q = __float2int_rn(v[i * 4 + k] * inv);
        q = q < -127 ? -127 : (q > 127 ? 127 : q);
        x |= (uint32_t)(q & 0xFF) << (8 * k);
      }
      w[i] = x;
    }
    return p;
  }
  __device__ static __forceinline__ void accumulate(float* acc, const Packed& p, float scale) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p.d);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[i * 4 + k] += (float)(int8_t)(w[i] >> (8 * k)) * scale;
  }
  __device__ static __forceinline__ Packed load(const char* base, size_t unit) {
    const u32x4* p = reinterpret_cast<const u32x4*>(base + unit * DATA_BYTES);
    Packed r; r.d[0] = p[0]; r.d[1] = p[1]; return r;
  }
  __device__ static __forceinline__ void store(char* base, size_t unit, const Packed& v) {
    u32x4* p = reinterpret_cast<u32x4*>(base + unit * DATA_BYTES);
    p[0] = v.d[0]; p[1] = v.d[1];
  }
};
template <> struct QCodec<4> {
  static constexpr int DATA_BYTES = 16;
  static constexpr float QMAX = 7.f;
  struct Packed { u32x4 d; };
  __device__ static __forceinline__ Packed encode(const float* v, float inv) {
    Packed p;
    uint32_t* w = reinterpret_cast<uint32_t*>(&p.d);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t x = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int q = __float2int_rn(v[i * 8 + k] * inv);
        q = q < -7 ? -7 : (q > 7 ? 7 : q);
        x |= (uint32_t)(q & 0xF) << (4 * k);
      }
      w[i] = x;
    }
    return p;
  }
  __device__ static __forceinline__ void accumulate(float* acc, const Packed& p, float scale) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&p.d);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int q = ((int)(w[i] << (28 - 4 * k))) >> 28;     // sign-extend the nibble
        acc[i * 8 + k] += (float)q * scale;
      }
  }
  __device__ static __forceinline__ Packed load(const char* base, size_t unit) {
    Packed r; r.d = reinterpret_cast<const u32x4*>(base + unit * DATA_BYTES)[0]; return r;
  }
  __device__ static __forceinline__ void store(char* base, size_t unit, const Packed& v) {
    reinterpret_cast<u32x4*>(base + unit * DATA_BYTES)[0] = v.d;
  }
};

// 6 bits per value: 32 values are a 192-bit stream (value i at bit 6 i), six 32-bit words, moved as three 8-byte
// pieces (a unit's 24 bytes are 8-byte aligned).
template <> struct QCodec<6> {
  static constexpr int DATA_BYTES = 24;
  static constexpr float QMAX = 31.f;
  struct Packed { uint32_t w[6]; };
  __device__ static __forceinline__ Packed encode(const float* v, float inv) {
    Packed p;
#pragma unroll
    for (int i = 0; i < 6; ++i) p.w[i] = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      int q = __float2int_rn(v[i] * inv);
      q = q < -31 ? -31 : (q > 31 ? 31 : q);
      const uint32_t u = (uint32_t)q & 63u;
      const int bit = 6 * i, word = bit >> 5, off = bit & 31;
      p.w[word] |= u << off;
      if (off > 26) p.w[word + 1] |= u >> (32 - off);
    }
    return p;
  }
  __device__ static __forceinline__ void accumulate(float* acc, const Packed& p, float scale) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int bit = 6 * i, word = bit >> 5, off = bit & 31;
      uint32_t u = p.w[word] >> off;
      if (off > 26) u |= p.w[word + 1] << (32 - off);
      const int q = (int)((u & 63u) ^ 32u) - 32;             // sign-extend six bits
      acc[i] += (float)q * scale;
    }
  }
  __device__ static __forceinline__ Packed load(const char* base, size_t unit) {
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t* s = reinterpret_cast<const u32x2_t*>(base + unit * DATA_BYTES);
    Packed r;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const u32x2_t t = s[i]; r.w[2 * i] = t[0]; r.w[2 * i + 1] = t[1]; }
    return r;
  }
  __device__ static __forceinline__ void store(char* base, size_t unit, const Packed& v) {
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t* d = reinterpret_cast<u32x2_t*>(base + unit * DATA_BYTES);
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = u32x2_t{v.w[2 * i], v.w[2 * i + 1]};
  }
};
// "FP": no quantisation, bf16 on the wire (exact for bf16 inputs; fp16 inputs round to bf16's 8 bits once per
// shot).  The scale plane carries ones.
template <> struct QCodec<16> {
  static constexpr int DATA_BYTES = 64;
  static constexpr float QMAX = 1.f;
  struct Packed { u32x4 d[4]; };
  __device__ static __forceinline__ Packed encode(const float* v, float) {
    Packed p;
#pragma unroll
    for (int q = 0; q < 4; ++q) p.d[q] = acc_pack<bf16_t>(v + q * 8);
    return p;
  }
  __device__ static __forceinline__ void accumulate(float* acc, const Packed& p, float) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc_add<bf16_t>(acc + q * 8, p.d[q]);
  }
  __device__ static __forceinline__ Packed load(const char* base, size_t unit) {
    const u32x4* s = reinterpret_cast<const u32x4*>(base + unit * DATA_BYTES);
    Packed r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r.d[q] = s[q];
    return r;
  }
  __device__ static __forceinline__ void store(char* base, size_t unit, const Packed& v) {
    u32x4* d = reinterpret_cast<u32x4*>(base + unit * DATA_BYTES);
#pragma unroll
    for (int q = 0; q < 4; ++q) d[q] = v.d[q];
  }
};

// One staging half of the quantised path: [in data][in scales][result data][result scales], each
// sized for `cap` units (CarArgs::in_bytes carries cap).
template <int BITS> struct QLayout {
  using C = QCodec<BITS>;
  __host__ __device__ static size_t half_bytes(size_t cap) { return 2 * cap * (C::DATA_BYTES + 4); }
  __device__ static char* in_data(char* half, size_t) { return half; }
  __device__ static float* in_scale(char* half, size_t cap) { return reinterpret_cast<float*>(half + cap * C::DATA_BYTES); }
  __device__ static char* res_data(char* half, size_t cap) { return half + cap * (C::DATA_BYTES + 4); }
  __device__ static float* res_scale(char* half, size_t cap) {
    return reinterpret_cast<float*>(half + cap * (C::DATA_BYTES + 4) + cap * C::DATA_BYTES);
  }
};

template <typename T> __device__ __forceinline__ void unit_to_float(const u32x4* src, float* v) {
  // 32 values of a 2-byte type = 4 x 16 bytes
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32x4 raw = src[q];
    const T* p = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[q * 8 + i] = (float)p[i];
  }
}

template <int BITS> __device__ __forceinline__ float group_scale(const float* v, float& inv) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) m = fmaxf(m, fabsf(v[i]));
  if (BITS == 16) { inv = 1.f; return 1.f; }
  const float scale = m > 0.f ? m / QCodec<BITS>::QMAX : 1.f;
  inv = 1.f / scale;
  return scale;
}

template <typename T, int BITS>
__global__ __launch_bounds__(512) void qr_kernel(CarArgs a, const u32x4* __restrict__ in, u32x4* __restrict__ out,
                                                 size_t nunits) {
  using C = QCodec<BITS>;
  using L = QLayout<BITS>;
  Signal* me = a.sig[a.rank];
  __shared__ uint32_t s_cnt, s_call;
  if (threadIdx.x == 0) { s_cnt = me->counter[blockIdx.x]; s_call = me->calls[blockIdx.x]; }
  __syncthreads();
  const uint32_t cnt = s_cnt;
  const size_t cap = a.in_bytes;                    // capacity in units
  const size_t half_off = (size_t)(s_call & 1) * a.half_bytes;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  char* my_half = a.data[a.rank] + half_off;

  // 0) quantise my input into my staging half
  for (size_t u = tid; u < nunits; u += stride) {
    float v[32];
    unit_to_float<T>(in + u * 4, v);
    float inv;
    const float sc = group_scale<BITS>(v, inv);
    C::store(L::in_data(my_half, cap), u, C::encode(v, inv));
    L::in_scale(my_half, cap)[u] = sc;
  }
  bool ok = cross_barrier(a, cnt + 1);

  // 1) reduce the units this rank is responsible for, keep the result quantised
  if (ok) {
    size_t j = 0;
    for (size_t u = tid; u < nunits; u += stride, ++j) {
      if ((int)((j + blockIdx.x) % a.world) != a.rank) continue;
      float acc[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = 0.f;
#pragma unroll
      for (int r = 0; r < MAX_RANKS; ++r) {
        if (r < a.world) {
          char* half = a.data[(a.rank + r) % a.world] + half_off;
          const typename C::Packed p = C::load(L::in_data(half, cap), u);
          const float sc = L::in_scale(half, cap)[u];
          C::accumulate(acc, p, sc);
        }
      }
      float inv;
      const float sc = group_scale<BITS>(acc, inv);
      C::store(L::res_data(my_half, cap), u, C::encode(acc, inv));
      L::res_scale(my_half, cap)[u] = sc;
    }
  }
  ok = cross_barrier(a, cnt + 2) && ok;

  // 2) gather every unit from its reducer
  if (ok) {
    size_t j = 0;
    for (size_t u = tid; u < nunits; u += stride, ++j) {
      const int src = (int)((j + blockIdx.x) % a.world);
      char* half = a.data[src] + half_off;
      const typename C::Packed p = C::load(L::res_data(half, cap), u);
      const float sc = L::res_scale(half, cap)[u];
      float v[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = 0.f;
      C::accumulate(v, p, sc);
#pragma unroll
      for (int q = 0; q < 4; ++q) out[u * 4 + q] = acc_pack<T>(v + q * 8);
    }
  }
  if (threadIdx.x == 0) { me->counter[blockIdx.x] = cnt + 2; me->calls[blockIdx.x] = s_call + 1; }
}

// Bytes of one staging half for `max_bytes` of 2-byte input at the given codec width.
static size_t qr_capacity(size_t max_bytes) {
  // units, rounded up to a multiple of 4 so that every plane of the layout starts 16-byte aligned
  return ((max_bytes + 63) / 64 + 3) / 4 * 4;
}

VDA_EXPORT size_t vda_qr_half_bytes(size_t max_bytes, int bits) {
  const size_t cap = qr_capacity(max_bytes);
  return bits == 16 ? QLayout<16>::half_bytes(cap) : bits == 8 ? QLayout<8>::half_bytes(cap)
       : bits == 6 ? QLayout<6>::half_bytes(cap) : QLayout<4>::half_bytes(cap);
}

// The context is a CarCtx whose in_bytes field holds the capacity in units; arenas and signal
// blocks come from vda_car_alloc (a separate pair from the unquantised path).
VDA_EXPORT void* vda_qr_create(int rank, int world, void** peer_data, void** peer_sig, size_t max_bytes, int bits,
                               unsigned long long spin_limit) {
  if (world < 2 || world > MAX_RANKS || (bits != 16 && bits != 8 && bits != 6 && bits != 4)) return nullptr;
  CarCtx* c = new CarCtx();
  c->rank = rank; c->world = world;
  c->in_bytes = qr_capacity(max_bytes);
  c->half_bytes = vda_qr_half_bytes(max_bytes, bits);
  for (int i = 0; i < world; ++i) { c->data[i] = (char*)peer_data[i]; c->sig[i] = (Signal*)peer_sig[i]; }
  c->spin_limit = spin_limit;
  c->host_err = alloc_host_err();
  return c;
}

// dtype: 0 bf16, 1 f16.  nbytes: multiple of 64, at most the capacity given at creation.
VDA_EXPORT int vda_qr_allreduce(void* ctx, const void* in, void* out, size_t nbytes, int dtype, int bits,
                                hipStream_t st) {
  CarCtx* c = (CarCtx*)ctx;
  if (nbytes % 64 || nbytes == 0 || nbytes / 64 > c->in_bytes || dtype < 0 || dtype > 1) return -1;
  const size_t nunits = nbytes / 64;
  CarArgs a;
  for (int i = 0; i < MAX_RANKS; ++i) { a.data[i] = c->data[i < c->world ? i : 0]; a.sig[i] = c->sig[i < c->world ? i : 0]; }
  a.rank = c->rank; a.world = c->world; a.half_bytes = c->half_bytes; a.in_bytes = c->in_bytes;
  a.spin_limit = c->spin_limit; a.host_err = c->host_err;
#define L(T, B) hipLaunchKernelGGL((qr_kernel<T, B>), dim3(MAX_BLOCKS), dim3(512), 0, st, a, (const u32x4*)in, \
                                   (u32x4*)out, nunits)
  if (bits == 16)     { if (dtype == 0) L(bf16_t, 16); else L(f16_t, 16); }
  else if (bits == 8) { if (dtype == 0) L(bf16_t, 8); else L(f16_t, 8); }
  else if (bits == 6) { if (dtype == 0) L(bf16_t, 6); else L(f16_t, 6); }
  else                { if (dtype == 0) L(bf16_t, 4); else L(f16_t, 4); }
#undef L
  return (int)hipGetLastError();
}
