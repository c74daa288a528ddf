// Shared pieces of the peer-memory collectives (all-reduce, all-to-all): the uncached signal block
// every rank exports, and the bounded cross-device barrier between same-numbered workgroups.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#define VDA_EXPORT extern "C" __attribute__((visibility("default")))
#define MAX_RANKS 8
#define MAX_BLOCKS 64

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Signal {
  // arrival[block][src_rank]: written by src_rank's block `block`, read by the owner
  alignas(128) uint32_t arrival[MAX_BLOCKS][MAX_RANKS];
  alignas(128) uint32_t counter[MAX_BLOCKS];   // owner only: barriers passed by this block
  alignas(128) uint32_t calls[MAX_BLOCKS];     // owner only: calls made (parity selects the staging half)
  alignas(128) uint32_t error;                 // non-zero after a time-out
};

// All ranks' block `blockIdx.x` meet: stores made before the call are visible to the peers' block
// after it.  Every spin is bounded; on time-out the error word of this rank is set to 1 + the peer
// that did not arrive and false is returned (the caller skips its reads and leaves).
__device__ __forceinline__ bool peer_barrier(Signal* const* sig, int rank, int world, unsigned long long spin_limit,
                                             uint32_t target, uint32_t* host_err) {
  __threadfence_system();
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  if (threadIdx.x < world) {
    const int peer = threadIdx.x;
    __scoped_atomic_store_n(&sig[peer]->arrival[blockIdx.x][rank], target, __ATOMIC_RELEASE, __MEMORY_SCOPE_SYSTEM);
    unsigned long long spins = 0;
    while ((int32_t)(__scoped_atomic_load_n(&sig[rank]->arrival[blockIdx.x][peer], __ATOMIC_ACQUIRE,
                                            __MEMORY_SCOPE_SYSTEM) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > spin_limit) {
        __scoped_atomic_store_n(&sig[rank]->error, 1u + (uint32_t)peer, __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM);
        // the same word in host-mapped memory: the worker polls it after every step without a device copy
        if (host_err) __scoped_atomic_store_n(host_err, 1u + (uint32_t)peer, __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM);
        ok = 0;
        break;
      }
    }
  }
  __syncthreads();
  __threadfence_system();
  return ok != 0;
}

// One pinned, device-visible word per context for the time-out report.
static inline uint32_t* alloc_host_err() {
  uint32_t* p = nullptr;
  if (hipHostMalloc((void**)&p, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess) return nullptr;
  *p = 0;
  return p;
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
