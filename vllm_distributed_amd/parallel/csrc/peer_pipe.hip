// Peer-memory pipe between neighbouring pipeline stages on one node: the activations of a step go from stage s to
// stage s + 1 as a peer WRITE into a two-slot ring in the receiver's IPC-exported arena, followed by a flag; the
// receiver's kernel waits for the flag, copies the slot out and acknowledges it, which frees the slot for the send
// after next.  No host takes part in a transfer: sequence counters live in device memory, the launch shape depends
// on the message size only, so a hop is two kernel launches (one per side) and can sit in a captured graph.
//
// Ownership (the rule of custom_allreduce.hip): 16-byte element i belongs to workgroup (i mod (MAX_BLOCKS * 512)) / 512
// on both sides, a call launches the workgroups that own elements of its message, and every workgroup keeps its own
// send / receive counters and flag words.  A workgroup thus only ever meets the same-numbered workgroup of its
// neighbour, and a slot is rewritten only after that workgroup acknowledged the message sent two sends earlier.
// Every spin is bounded; a time-out sets the error word and the kernel leaves without touching the payload.
// Takes the place of the RCCL send / recv plus pickled metadata of the reference's pipeline hop
// (vllm/distributed/parallel_state.py:629-749 send_tensor_dict / recv_tensor_dict).
#include "peer_sync.h"

struct PipeSignal {
  alignas(128) uint32_t ready[MAX_BLOCKS];    // written by the upstream stage: messages landed in my ring, per workgroup
  alignas(128) uint32_t ack[MAX_BLOCKS];      // written by the downstream stage: messages of mine it has copied out
  alignas(128) uint32_t sent[MAX_BLOCKS];     // owner only
  alignas(128) uint32_t recvd[MAX_BLOCKS];    // owner only
  alignas(128) uint32_t error;
};
static_assert(sizeof(PipeSignal) <= sizeof(Signal), "the pipe reuses the signal allocation of the collectives");

struct PipeCtx {
  int rank, world;
  size_t slot_bytes;
  char* data[MAX_RANKS];
  PipeSignal* sig[MAX_RANKS];
  unsigned long long spin_limit;
  uint32_t* host_err;
};

struct PipeArgs {
  char* ring;                 // send: the receiver's arena; receive: mine
  PipeSignal* mine;
  PipeSignal* peer;           // send: the receiver's signals; receive: the sender's
  size_t slot_bytes;
  unsigned long long spin_limit;
  uint32_t* host_err;
  int peer_rank;
};

// thread 0 waits until *word >= target (wrap-safe); false after spin_limit polls
__device__ __forceinline__ bool pipe_wait(const PipeArgs& a, uint32_t* word, uint32_t target) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    ok = 1;
    unsigned long long spins = 0;
    while ((int32_t)(__scoped_atomic_load_n(word, __ATOMIC_ACQUIRE, __MEMORY_SCOPE_SYSTEM) - target) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > a.spin_limit) {
        __scoped_atomic_store_n(&a.mine->error, 1u + (uint32_t)a.peer_rank, __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM);
        if (a.host_err) __scoped_atomic_store_n(a.host_err, 1u + (uint32_t)a.peer_rank, __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM);
        ok = 0;
        break;
      }
    }
  }
  __syncthreads();
  __threadfence_system();
  return ok != 0;
}

__global__ __launch_bounds__(512) void pipe_send_kernel(PipeArgs a, const u32x4* __restrict__ in, size_t n16) {
  const int b = blockIdx.x;
  __shared__ uint32_t s_c;
  if (threadIdx.x == 0) s_c = a.mine->sent[b];
  __syncthreads();
  const uint32_t c = s_c;
  // the slot was last used by send c - 2: wait until the receiver has copied that one out
  const bool ok = c < 2 ? true : pipe_wait(a, &a.mine->ack[b], c - 1);
  if (ok) {
    u32x4* dst = reinterpret_cast<u32x4*>(a.ring + (size_t)(c & 1) * a.slot_bytes);
    const size_t stride = (size_t)MAX_BLOCKS * blockDim.x;
    for (size_t i = (size_t)b * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = in[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
      __scoped_atomic_store_n(&a.peer->ready[b], c + 1, __ATOMIC_RELEASE, __MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 0) a.mine->sent[b] = c + 1;
}

__global__ __launch_bounds__(512) void pipe_recv_kernel(PipeArgs a, u32x4* __restrict__ out, size_t n16) {
  const int b = blockIdx.x;
  __shared__ uint32_t s_c;
  if (threadIdx.x == 0) s_c = a.mine->recvd[b];
  __syncthreads();
  const uint32_t c = s_c;
  const bool ok = pipe_wait(a, &a.mine->ready[b], c + 1);
  if (ok) {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.ring + (size_t)(c & 1) * a.slot_bytes);
    const size_t stride = (size_t)MAX_BLOCKS * blockDim.x;
    for (size_t i = (size_t)b * blockDim.x + threadIdx.x; i < n16; i += stride)
      out[i] = __builtin_nontemporal_load(src + i);
    __syncthreads();                    // every read of the slot has been issued and returned
  }
  if (threadIdx.x == 0) {
    __scoped_atomic_store_n(&a.peer->ack[b], c + 1, __ATOMIC_RELEASE, __MEMORY_SCOPE_SYSTEM);
    a.mine->recvd[b] = c + 1;
  }
}

VDA_EXPORT void* vda_pipe_create(int rank, int world, void** peer_data, void** peer_sig, size_t slot_bytes,
                                 unsigned long long spin_limit) {
  if (world < 2 || world > MAX_RANKS || slot_bytes == 0 || slot_bytes % 16) return nullptr;
  PipeCtx* c = new PipeCtx();
  c->rank = rank; c->world = world; c->slot_bytes = slot_bytes;
  for (int i = 0; i < world; ++i) { c->data[i] = (char*)peer_data[i]; c->sig[i] = (PipeSignal*)peer_sig[i]; }
  c->spin_limit = spin_limit;
  c->host_err = alloc_host_err();
  return c;
}

VDA_EXPORT void vda_pipe_destroy(void* ctx) {
  PipeCtx* c = (PipeCtx*)ctx;
  if (c->host_err) hipHostFree(c->host_err);
  delete c;
}

VDA_EXPORT int vda_pipe_error(void* ctx) {
  PipeCtx* c = (PipeCtx*)ctx;
  return c->host_err ? (int)*(volatile uint32_t*)c->host_err : 0;
}

static int pipe_blocks(size_t n16) {
  return (int)(n16 >= (size_t)MAX_BLOCKS * 512 ? MAX_BLOCKS : (n16 + 511) / 512);
}

// Send nbytes (a multiple of 16, at most the slot size) to the next rank of the ring / receive from the previous one.
// Sends and receives of a pair must come in the same order with the same sizes.
VDA_EXPORT int vda_pipe_send(void* ctx, const void* in, size_t nbytes, hipStream_t st) {
  PipeCtx* c = (PipeCtx*)ctx;
  if (nbytes == 0 || nbytes % 16 || nbytes > c->slot_bytes) return -1;
  const int nxt = (c->rank + 1) % c->world;
  PipeArgs a{c->data[nxt], c->sig[c->rank], c->sig[nxt], c->slot_bytes, c->spin_limit, c->host_err, nxt};
  hipLaunchKernelGGL(pipe_send_kernel, dim3(pipe_blocks(nbytes / 16)), dim3(512), 0, st, a, (const u32x4*)in, nbytes / 16);
  return (int)hipGetLastError();
}

VDA_EXPORT int vda_pipe_recv(void* ctx, void* out, size_t nbytes, hipStream_t st) {
  PipeCtx* c = (PipeCtx*)ctx;
  if (nbytes == 0 || nbytes % 16 || nbytes > c->slot_bytes) return -1;
  const int prv = (c->rank + c->world - 1) % c->world;
  PipeArgs a{c->data[c->rank], c->sig[c->rank], c->sig[prv], c->slot_bytes, c->spin_limit, c->host_err, prv};
  hipLaunchKernelGGL(pipe_recv_kernel, dim3(pipe_blocks(nbytes / 16)), dim3(512), 0, st, a, (u32x4*)out, nbytes / 16);
  return (int)hipGetLastError();
}
