// Peer-memory all-to-all-v of rows over xGMI for one node (2..8 GPUs, fully connected).
//
// Every rank owns an IPC-exported arena of two halves (call parity, as in the all-reduce).  A half holds
//   header [MAX_BLOCKS][MAX_RANKS]  row counts per destination (one copy per workgroup)
//   meta   [world][cap]             one int32 per row (e.g. the expert a token goes to)
//   rows   [world][cap][row_bytes]  rows grouped by DESTINATION rank
// A call packs the rows this rank sends into its own half (local stores), meets the peers at a
// cross-device barrier, and then PULLS region [me] of every peer's half over all links at once into
// ordinary local memory: recv[src * cap + j] is the j-th row rank `src` sent here, recv_counts[src]
// how many.  Capacity is fixed (`cap` rows per (source, destination) pair), counts travel with the
// data and live on the device, so a call has a fixed launch shape, never syncs with the host and
// can be captured in a hipGraph; rows past a count are simply not moved.
//
// Ownership rule that makes per-workgroup barriers and the parity halves sufficient: row j of any
// region belongs to workgroup j mod gridDim.x, on every rank, when packing and when pulling.  So a
// workgroup only reads what the same-numbered workgroup of a peer packed before the barrier they share,
// and "my workgroup passed barrier k+1" implies "the peers' workgroup finished pulling my call-k rows".
//
// Two packing forms:
//   gather   rows are x[sorted[offsets[d] + j] / k_div] for destination d (rows sorted by destination:
//            expert-parallel dispatch, token-parallel scatter)
//   regions  x is already laid out [world][cap][row_bytes] with counts[d] valid rows (the way back:
//            results of rows received by an earlier call return to their senders in the same slots)
// Role of the reference's all2all managers (vllm/distributed/device_communicators/all2all.py:22-264,
// which lean on NVSHMEM / DeepEP) and of the all_to_all_single calls of its token-parallel layers
// (vllm/model_executor/layers/token_parallel_linear.py:147-236, 328-374).
#include "peer_sync.h"

struct A2aCtx {
  uint32_t* gsync;          // local grid barrier of the combine kernel: [0] arrivals, [1 + b] barriers workgroup b passed
  int rank, world, cap, row_bytes;
  size_t half_bytes, meta_off, rows_off;
  char* data[MAX_RANKS];
  Signal* sig[MAX_RANKS];
  unsigned long long spin_limit;
  uint32_t* host_err;
};

struct A2aArgs {
  char* data[MAX_RANKS];
  Signal* sig[MAX_RANKS];
  int rank, world, cap, row_bytes;
  size_t half_bytes, meta_off, rows_off;
  unsigned long long spin_limit;
  uint32_t* host_err;
};

constexpr int A2A_THREADS = 512;

// copy one row of `units` 16-byte pieces with the whole workgroup
__device__ __forceinline__ void copy_row(u32x4* __restrict__ dst, const u32x4* __restrict__ src, int units) {
  int i = threadIdx.x;
  for (; i + 3 * A2A_THREADS < units; i += 4 * A2A_THREADS) {
    const u32x4 a = src[i], b = src[i + A2A_THREADS], c = src[i + 2 * A2A_THREADS], d = src[i + 3 * A2A_THREADS];
    dst[i] = a; dst[i + A2A_THREADS] = b; dst[i + 2 * A2A_THREADS] = c; dst[i + 3 * A2A_THREADS] = d;
  }
  for (; i < units; i += A2A_THREADS) dst[i] = src[i];
}

template <bool GATHER>
__global__ __launch_bounds__(A2A_THREADS) void a2a_kernel(A2aArgs a, const char* __restrict__ x, long x_row_stride,
                                                           const int* __restrict__ sorted,
                                                           const int* __restrict__ offsets_or_counts, int k_div,
                                                           const int* __restrict__ meta_in, char* __restrict__ recv,
                                                           long recv_row_stride, int copy_bytes,
                                                           int* __restrict__ recv_meta, int* __restrict__ recv_counts,
                                                           int meta_sub, int fill_meta) {
  Signal* me = a.sig[a.rank];
  __shared__ uint32_t s_cnt, s_call;
  __shared__ int s_n[MAX_RANKS];
  if (threadIdx.x == 0) { s_cnt = me->counter[blockIdx.x]; s_call = me->calls[blockIdx.x]; }
  __syncthreads();
  const uint32_t cnt = s_cnt;
  const size_t half_off = (size_t)(s_call & 1) * a.half_bytes;
  const int b = blockIdx.x, B = gridDim.x;
  const int units = copy_bytes >> 4;            // bytes moved per row (<= the arena's row pitch)
  char* mine = a.data[a.rank] + half_off;

  // ---- pack: my rows for destination d -> my half, region d -----------------------------------------
  for (int d = 0; d < a.world; ++d) {
    int n, first = 0;
    if (GATHER) { first = offsets_or_counts[d]; n = offsets_or_counts[d + 1] - first; }
    else n = offsets_or_counts[d];
    n = min(max(n, 0), a.cap);
    if (threadIdx.x == 0) reinterpret_cast<int*>(mine)[b * MAX_RANKS + d] = n;
    for (int j = b; j < n; j += B) {
      const long src_idx = GATHER ? sorted[first + j] : (long)d * a.cap + j;
      const long src_row = GATHER ? src_idx / k_div : src_idx;
      copy_row(reinterpret_cast<u32x4*>(mine + a.rows_off + ((size_t)d * a.cap + j) * a.row_bytes),
               reinterpret_cast<const u32x4*>(x + src_row * x_row_stride), units);
      if (meta_in && threadIdx.x == 0)
        reinterpret_cast<int*>(mine + a.meta_off)[(size_t)d * a.cap + j] = meta_in[src_idx];
    }
  }
  const bool ok = peer_barrier(a.sig, a.rank, a.world, a.spin_limit, cnt + 1, a.host_err);

  // ---- pull: region [me] of every peer -> recv[src] -------------------------------------------------
  if (ok) {
    if (threadIdx.x < a.world) {
      const int s = threadIdx.x;
      const int n = reinterpret_cast<const int*>(a.data[s] + half_off)[b * MAX_RANKS + a.rank];
      s_n[s] = n;
      if (b == 0) recv_counts[s] = n;
    }
    __syncthreads();
    for (int i = 0; i < a.world; ++i) {
      const int s = (a.rank + i) % a.world;          // start with myself, then walk the peers: spreads link load
      const char* peer = a.data[s] + half_off;
      const int n = s_n[s];
      for (int j = b; j < n; j += B) {
        copy_row(reinterpret_cast<u32x4*>(recv + ((size_t)s * a.cap + j) * recv_row_stride),
                 reinterpret_cast<const u32x4*>(peer + a.rows_off + ((size_t)a.rank * a.cap + j) * a.row_bytes), units);
        if (recv_meta && threadIdx.x == 0)
          recv_meta[(size_t)s * a.cap + j] =
              reinterpret_cast<const int*>(peer + a.meta_off)[(size_t)a.rank * a.cap + j] - meta_sub;
      }
    }
    if (recv_meta && fill_meta) {             // slots past a source's count: -1 ("no row here")
      const int total = a.world * a.cap;
      for (int idx = b * A2A_THREADS + threadIdx.x; idx < total; idx += B * A2A_THREADS)
        if (idx % a.cap >= s_n[idx / a.cap]) recv_meta[idx] = -1;
    }
  }
  if (threadIdx.x == 0) { me->counter[blockIdx.x] = cnt + 1; me->calls[blockIdx.x] = s_call + 1; }
}

// The way back of an expert-parallel dispatch with the weighting fused in: y [world][cap] rows (results of the rows
// received, grouped by the rank they came from) return to their senders exactly like a regions-form exchange, and
// then -- after a barrier among this GPU's own workgroups, all of which are resident -- every token's k rows are
// read back from the scratch they landed in (still in L2 for decode sizes), weighted and summed in fp32:
//     out[t] = sum_j w[t, j] * ret[dest[t k + j] * cap + slot[t k + j]]
// One launch instead of an exchange, a gather, a broadcast multiply and a reduction.
template <typename T>
__global__ __launch_bounds__(A2A_THREADS) void a2a_combine_kernel(A2aArgs a, uint32_t* __restrict__ gsync,
                                                                   const char* __restrict__ y, long y_row_stride,
                                                                   const int* __restrict__ counts,
                                                                   char* __restrict__ ret, long ret_row_stride,
                                                                   int copy_bytes, const int* __restrict__ dest,
                                                                   const int* __restrict__ slot,
                                                                   const float* __restrict__ topk_w, T* __restrict__ out,
                                                                   long out_row_stride, int Tn, int k) {
  Signal* me = a.sig[a.rank];
  __shared__ uint32_t s_cnt, s_call, s_seen;
  __shared__ int s_n[MAX_RANKS];
  __shared__ int s_ok;
  if (threadIdx.x == 0) { s_cnt = me->counter[blockIdx.x]; s_call = me->calls[blockIdx.x]; s_seen = gsync[1 + blockIdx.x]; }
  __syncthreads();
  const uint32_t cnt = s_cnt;
  const size_t half_off = (size_t)(s_call & 1) * a.half_bytes;
  const int b = blockIdx.x, B = gridDim.x;
  const int units = copy_bytes >> 4;
  char* mine = a.data[a.rank] + half_off;
  for (int d = 0; d < a.world; ++d) {
    const int n = min(max(counts[d], 0), a.cap);
    if (threadIdx.x == 0) reinterpret_cast<int*>(mine)[b * MAX_RANKS + d] = n;
    for (int j = b; j < n; j += B)
      copy_row(reinterpret_cast<u32x4*>(mine + a.rows_off + ((size_t)d * a.cap + j) * a.row_bytes),
               reinterpret_cast<const u32x4*>(y + ((long)d * a.cap + j) * y_row_stride), units);
  }
  bool ok = peer_barrier(a.sig, a.rank, a.world, a.spin_limit, cnt + 1, a.host_err);
  if (ok) {
    if (threadIdx.x < a.world)
      s_n[threadIdx.x] = reinterpret_cast<const int*>(a.data[threadIdx.x] + half_off)[b * MAX_RANKS + a.rank];
    __syncthreads();
    for (int i = 0; i < a.world; ++i) {
      const int s = (a.rank + i) % a.world;
      const char* peer = a.data[s] + half_off;
      const int n = s_n[s];
      for (int j = b; j < n; j += B)
        copy_row(reinterpret_cast<u32x4*>(ret + ((size_t)s * a.cap + j) * ret_row_stride),
                 reinterpret_cast<const u32x4*>(peer + a.rows_off + ((size_t)a.rank * a.cap + j) * a.row_bytes), units);
    }
  }
  if (threadIdx.x == 0) { me->counter[blockIdx.x] = cnt + 1; me->calls[blockIdx.x] = s_call + 1; }
  // ---- every workgroup of THIS GPU has its rows in ret ----------------------------------------------
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t target = (s_seen + 1) * (uint32_t)B;
    __hip_atomic_fetch_add(&gsync[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long spins = 0;
    int good = 1;
    while ((int32_t)(__hip_atomic_load(&gsync[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > a.spin_limit) { good = 0; break; }
    }
    gsync[1 + b] = s_seen + 1;
    s_ok = good;
  }
  __syncthreads();
  __threadfence();
  if (!ok || !s_ok) return;
  constexpr int N = 16 / sizeof(T);
  for (int t = b; t < Tn; t += B) {
    for (int e = threadIdx.x; e < units; e += A2A_THREADS) {
      float acc[N];
#pragma unroll
      for (int i = 0; i < N; ++i) acc[i] = 0.f;
      for (int j = 0; j < k; ++j) {
        const int p = t * k + j;
        const int d = dest[p];
        if (d < 0) continue;
        const float w = topk_w[p];
        const u32x4 raw = reinterpret_cast<const u32x4*>(ret + ((size_t)d * a.cap + slot[p]) * ret_row_stride)[e];
        const T* v = reinterpret_cast<const T*>(&raw);
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] += w * (float)v[i];
      }
      u32x4 o;
      T* op = reinterpret_cast<T*>(&o);
#pragma unroll
      for (int i = 0; i < N; ++i) op[i] = (T)acc[i];
      reinterpret_cast<u32x4*>(reinterpret_cast<char*>(out) + (long)t * out_row_stride)[e] = o;
    }
  }
}

// ---------------------------------------------------------------------------
// host API (arena allocation / IPC exchange is shared with the all-reduce: vda_car_alloc / vda_car_open)
// ---------------------------------------------------------------------------
static size_t a2a_meta_off() { return (size_t)MAX_BLOCKS * MAX_RANKS * sizeof(int); }
static size_t a2a_rows_off(int world, int cap) {
  const size_t m = a2a_meta_off() + (size_t)world * cap * sizeof(int);
  return (m + 255) / 256 * 256;
}

VDA_EXPORT size_t vda_a2a_half_bytes(int world, int cap, int row_bytes) {
  const size_t h = a2a_rows_off(world, cap) + (size_t)world * cap * row_bytes;
  return (h + 4095) / 4096 * 4096;
}

VDA_EXPORT void* vda_a2a_create(int rank, int world, void** peer_data, void** peer_sig, int cap, int row_bytes,
                                unsigned long long spin_limit) {
  if (world < 2 || world > MAX_RANKS || cap < 1 || row_bytes < 16 || row_bytes % 16) return nullptr;
  A2aCtx* c = new A2aCtx();
  c->rank = rank; c->world = world; c->cap = cap; c->row_bytes = row_bytes;
  c->half_bytes = vda_a2a_half_bytes(world, cap, row_bytes);
  c->meta_off = a2a_meta_off();
  c->rows_off = a2a_rows_off(world, cap);
  for (int i = 0; i < world; ++i) { c->data[i] = (char*)peer_data[i]; c->sig[i] = (Signal*)peer_sig[i]; }
  c->spin_limit = spin_limit;
  c->host_err = alloc_host_err();
  c->gsync = nullptr;
  if (hipMalloc((void**)&c->gsync, (1 + MAX_BLOCKS) * sizeof(uint32_t)) != hipSuccess ||
      hipMemset(c->gsync, 0, (1 + MAX_BLOCKS) * sizeof(uint32_t)) != hipSuccess) {
    delete c;
    return nullptr;
  }
  return c;
}

VDA_EXPORT void vda_a2a_destroy(void* ctx) {
  A2aCtx* c = (A2aCtx*)ctx;
  if (c->host_err) hipHostFree(c->host_err);
  if (c->gsync) hipFree(c->gsync);
  delete c;
}

VDA_EXPORT int vda_a2a_error(void* ctx) {
  A2aCtx* c = (A2aCtx*)ctx;
  if (c->host_err) return (int)*(volatile uint32_t*)c->host_err;
  uint32_t e = 0;
  if (hipMemcpy(&e, &c->sig[c->rank]->error, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)e;
}

// gather != 0: x rows are picked through sorted / offsets[world + 1] (row = sorted[i] / k_div, meta_in
// indexed by sorted[i]); gather == 0: x is [world][cap][row_bytes] and offsets_or_counts holds counts[world]
// (meta_in indexed like x).  copy_bytes per row (a multiple of 16, at most the context's row pitch);
// x_row_stride / recv_row_stride in bytes, multiples of 16.  recv holds world * cap rows, recv_meta
// [world * cap] (optional), recv_counts [world].  cap: rows per (source, destination) pair of THIS call
// (0 = the context's capacity, never more; every rank must pass the same value): small batches then
// use small regions and the consumers of recv see few padded rows.
VDA_EXPORT int vda_a2a_exchange(void* ctx, const void* x, long x_row_stride, const int* sorted,
                                const int* offsets_or_counts, int k_div, const int* meta_in, void* recv,
                                long recv_row_stride, int copy_bytes, int* recv_meta, int* recv_counts, int gather,
                                int cap, int meta_sub, int fill_meta, hipStream_t st) {
  A2aCtx* c = (A2aCtx*)ctx;
  if (cap <= 0) cap = c->cap;
  if (cap > c->cap) return -2;
  if (!x || !offsets_or_counts || !recv || !recv_counts || x_row_stride % 16 || recv_row_stride % 16 ||
      copy_bytes < 16 || copy_bytes % 16 || copy_bytes > c->row_bytes || recv_row_stride < copy_bytes ||
      (gather && !sorted) || k_div < 1)
    return -1;
  A2aArgs a;
  for (int i = 0; i < MAX_RANKS; ++i) { a.data[i] = c->data[i < c->world ? i : 0]; a.sig[i] = c->sig[i < c->world ? i : 0]; }
  a.rank = c->rank; a.world = c->world; a.cap = cap; a.row_bytes = c->row_bytes;
  a.half_bytes = c->half_bytes; a.meta_off = c->meta_off; a.rows_off = c->rows_off; a.spin_limit = c->spin_limit;
  a.host_err = c->host_err;
  if (gather)
    hipLaunchKernelGGL(a2a_kernel<true>, dim3(MAX_BLOCKS), dim3(A2A_THREADS), 0, st, a, (const char*)x, x_row_stride,
                       sorted, offsets_or_counts, k_div, meta_in, (char*)recv, recv_row_stride, copy_bytes, recv_meta,
                       recv_counts, meta_sub, fill_meta);
  else
    hipLaunchKernelGGL(a2a_kernel<false>, dim3(MAX_BLOCKS), dim3(A2A_THREADS), 0, st, a, (const char*)x, x_row_stride,
                       sorted, offsets_or_counts, k_div, meta_in, (char*)recv, recv_row_stride, copy_bytes, recv_meta,
                       recv_counts, meta_sub, fill_meta);
  return (int)hipGetLastError();
}

// y: [world][cap] rows grouped by the rank they go back to, counts[world] valid rows each (the recv_counts of the
// dispatch); ret: scratch of world * cap rows; dest / slot: per (token, slot) pair of THIS rank the owner rank (-1:
// none) and the position inside that owner's region; topk_w fp32 [T * k]; out [T] rows of dtype (0 bf16, 1 f16).
VDA_EXPORT int vda_a2a_combine(void* ctx, const void* y, long y_row_stride, const int* counts, void* ret,
                               long ret_row_stride, int copy_bytes, const int* dest, const int* slot,
                               const float* topk_w, void* out, long out_row_stride, int T, int k, int dtype, int cap,
                               hipStream_t st) {
  A2aCtx* c = (A2aCtx*)ctx;
  if (cap <= 0) cap = c->cap;
  if (cap > c->cap) return -2;
  if (!y || !counts || !ret || !dest || !slot || !topk_w || !out || y_row_stride % 16 || ret_row_stride % 16 ||
      out_row_stride % 16 || copy_bytes < 16 || copy_bytes % 16 || copy_bytes > c->row_bytes ||
      ret_row_stride < copy_bytes || T < 0 || k < 1 || (dtype != 0 && dtype != 1))
    return -1;
  A2aArgs a;
  for (int i = 0; i < MAX_RANKS; ++i) { a.data[i] = c->data[i < c->world ? i : 0]; a.sig[i] = c->sig[i < c->world ? i : 0]; }
  a.rank = c->rank; a.world = c->world; a.cap = cap; a.row_bytes = c->row_bytes;
  a.half_bytes = c->half_bytes; a.meta_off = c->meta_off; a.rows_off = c->rows_off; a.spin_limit = c->spin_limit;
  a.host_err = c->host_err;
  if (dtype == 0)
    hipLaunchKernelGGL(a2a_combine_kernel<__bf16>, dim3(MAX_BLOCKS), dim3(A2A_THREADS), 0, st, a, c->gsync, (const char*)y,
                       y_row_stride, counts, (char*)ret, ret_row_stride, copy_bytes, dest, slot, topk_w, (__bf16*)out,
                       out_row_stride, T, k);
  else
    hipLaunchKernelGGL(a2a_combine_kernel<_Float16>, dim3(MAX_BLOCKS), dim3(A2A_THREADS), 0, st, a, c->gsync,
                       (const char*)y, y_row_stride, counts, (char*)ret, ret_row_stride, copy_bytes, dest, slot, topk_w,
                       (_Float16*)out, out_row_stride, T, k);
  return (int)hipGetLastError();
}
