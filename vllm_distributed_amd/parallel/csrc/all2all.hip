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
                                                           int* __restrict__ recv_meta, int* __restrict__ recv_counts) {
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
          recv_meta[(size_t)s * a.cap + j] = reinterpret_cast<const int*>(peer + a.meta_off)[(size_t)a.rank * a.cap + j];
      }
    }
  }
  if (threadIdx.x == 0) { me->counter[blockIdx.x] = cnt + 1; me->calls[blockIdx.x] = s_call + 1; }
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
  return c;
}

VDA_EXPORT void vda_a2a_destroy(void* ctx) {
  A2aCtx* c = (A2aCtx*)ctx;
  if (c->host_err) hipHostFree(c->host_err);
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
                                int cap, hipStream_t st) {
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
                       recv_counts);
  else
    hipLaunchKernelGGL(a2a_kernel<false>, dim3(MAX_BLOCKS), dim3(A2A_THREADS), 0, st, a, (const char*)x, x_row_stride,
                       sorted, offsets_or_counts, k_div, meta_in, (char*)recv, recv_row_stride, copy_bytes, recv_meta,
                       recv_counts);
  return (int)hipGetLastError();
}
