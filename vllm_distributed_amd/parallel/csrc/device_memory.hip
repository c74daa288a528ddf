// Device memory that other processes can map: one hipMalloc per region with its IPC handle.
// The KV-cache arena of an instance that serves KV to peers (disaggregated prefill) is allocated
// here so that a decode instance on another GPU of the node maps it once and then pulls blocks
// over xGMI with an ordinary copy kernel -- no staging, no per-transfer handshake.
#include <hip/hip_runtime.h>
#include <string.h>

#define VDA_EXPORT extern "C" __attribute__((visibility("default")))

VDA_EXPORT int vda_mem_alloc(size_t bytes, int device, void** ptr, char* handle /*64*/) {
  int prev = 0;
  hipError_t e = hipGetDevice(&prev);
  if (e != hipSuccess) return (int)e;
  if ((e = hipSetDevice(device)) != hipSuccess) return (int)e;
  e = hipMalloc(ptr, bytes);
  if (e == hipSuccess) e = hipMemset(*ptr, 0, bytes);
  if (e == hipSuccess) {
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, *ptr);
    if (e == hipSuccess) memcpy(handle, &h, sizeof(h));
  }
  hipSetDevice(prev);
  return (int)e;
}

VDA_EXPORT int vda_mem_free(void* ptr) { return (int)hipFree(ptr); }

VDA_EXPORT int vda_mem_open(const char* handle, int device, void** ptr) {
  int prev = 0;
  hipError_t e = hipGetDevice(&prev);
  if (e != hipSuccess) return (int)e;
  if ((e = hipSetDevice(device)) != hipSuccess) return (int)e;
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
  hipSetDevice(prev);
  return (int)e;
}

VDA_EXPORT int vda_mem_close(void* ptr) { return (int)hipIpcCloseMemHandle(ptr); }

// ---------------------------------------------------------------------------
// Sleep-mode allocator: device memory whose virtual addresses survive while the physical pages
// go away.  Tensors of a tagged pool ("weights", "kv_cache") are backed by hipMemCreate pages
// mapped at reserved addresses; `sleep` copies a pool to pinned host memory (or discards it),
// unmaps and releases the pages, `wake` creates new pages at the SAME addresses and copies the
// contents back.  Captured hipGraphs, KV-cache views and raw pointers held by kernels therefore
// stay valid across a sleep, which a plain free / re-allocate cannot promise.
// vda_vmm_malloc / vda_vmm_free have the signature of PyTorch's pluggable allocator, so a
// torch MemPool can route every allocation made inside a `with` block here; bookkeeping lives in
// this file (no Python callbacks on the allocation path).
// Role of the reference's csrc/cumem_allocator.cpp + vllm/device_allocator/cumem.py.
// ---------------------------------------------------------------------------
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
struct VmmBlock {
  void* va = nullptr;
  size_t size = 0;            // rounded to the allocation granularity
  hipMemGenericAllocationHandle_t handle{};
  int device = 0;
  int tag = 0;
  bool mapped = false;
  void* host_copy = nullptr;
};
// Leaked on purpose: PyTorch frees pool segments from static destructors at process exit, after
// this library's own statics would have been destroyed.
std::mutex& g_vmm_mu = *new std::mutex();
std::unordered_map<void*, VmmBlock>& g_vmm_blocks = *new std::unordered_map<void*, VmmBlock>();
int g_vmm_tag = 0;
int g_vmm_last_error = 0;
bool g_vmm_exiting = false;     // set by an atexit hook: the HIP runtime may already be gone
bool g_vmm_hooked = false;

hipMemAllocationProp vmm_prop(int device) {
  hipMemAllocationProp p{};
  p.type = hipMemAllocationTypePinned;
  p.location.type = hipMemLocationTypeDevice;
  p.location.id = device;
  return p;
}

hipError_t vmm_back(VmmBlock& b) {
  hipMemAllocationProp p = vmm_prop(b.device);
  hipError_t e = hipMemCreate(&b.handle, b.size, &p, 0);
  if (e != hipSuccess) return e;
  e = hipMemMap(b.va, b.size, 0, b.handle, 0);
  if (e != hipSuccess) { hipMemRelease(b.handle); return e; }
  hipMemAccessDesc d{};
  d.location.type = hipMemLocationTypeDevice;
  d.location.id = b.device;
  d.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(b.va, b.size, &d, 1);
  if (e != hipSuccess) { hipMemUnmap(b.va, b.size); hipMemRelease(b.handle); return e; }
  b.mapped = true;
  return hipSuccess;
}

hipError_t vmm_unback(VmmBlock& b) {
  hipError_t e = hipMemUnmap(b.va, b.size);
  if (e != hipSuccess) return e;
  e = hipMemRelease(b.handle);
  b.mapped = false;
  return e;
}
}  // namespace

VDA_EXPORT void vda_vmm_set_tag(int tag) { std::lock_guard<std::mutex> l(g_vmm_mu); g_vmm_tag = tag; }
VDA_EXPORT int vda_vmm_last_error() { return g_vmm_last_error; }

VDA_EXPORT void* vda_vmm_malloc(ssize_t size, int device, hipStream_t) {
  if (size <= 0) return nullptr;
  if (!g_vmm_hooked) {
    // registered after PyTorch's statics, so it runs before their destructors release the pools
    g_vmm_hooked = true;
    atexit([] { g_vmm_exiting = true; });
  }
  int prev = 0;
  hipGetDevice(&prev);
  hipSetDevice(device);
  hipMemAllocationProp p = vmm_prop(device);
  size_t gran = 0;
  hipError_t e = hipMemGetAllocationGranularity(&gran, &p, hipMemAllocationGranularityRecommended);
  if (e != hipSuccess || gran == 0) gran = 2u << 20;
  VmmBlock b;
  b.size = ((size_t)size + gran - 1) / gran * gran;
  b.device = device;
  e = hipMemAddressReserve(&b.va, b.size, gran, nullptr, 0);
  if (e == hipSuccess) {
    e = vmm_back(b);
    if (e != hipSuccess) hipMemAddressFree(b.va, b.size);
  }
  hipSetDevice(prev);
  if (e != hipSuccess) { g_vmm_last_error = (int)e; return nullptr; }
  std::lock_guard<std::mutex> l(g_vmm_mu);
  b.tag = g_vmm_tag;
  g_vmm_blocks[b.va] = b;
  return b.va;
}

VDA_EXPORT void vda_vmm_free(void* ptr, ssize_t, int, hipStream_t) {
  if (g_vmm_exiting) return;      // the process is going away; the driver reclaims everything
  VmmBlock b;
  {
    std::lock_guard<std::mutex> l(g_vmm_mu);
    auto it = g_vmm_blocks.find(ptr);
    if (it == g_vmm_blocks.end()) return;
    b = it->second;
    g_vmm_blocks.erase(it);
  }
  hipDeviceSynchronize();
  if (b.mapped) vmm_unback(b);
  if (b.host_copy) hipHostFree(b.host_copy);
  hipMemAddressFree(b.va, b.size);
}

// Release the physical memory of every mapped block with this tag; with `offload` the contents
// are first copied to pinned host memory.  Returns the number of bytes released, or -1.
VDA_EXPORT long long vda_vmm_sleep(int tag, int offload) {
  std::lock_guard<std::mutex> l(g_vmm_mu);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  long long freed = 0;
  for (auto& kv : g_vmm_blocks) {
    VmmBlock& b = kv.second;
    if (b.tag != tag || !b.mapped) continue;
    if (offload) {
      hipError_t e = hipHostMalloc(&b.host_copy, b.size, hipHostMallocDefault);
      if (e == hipSuccess) e = hipMemcpy(b.host_copy, b.va, b.size, hipMemcpyDeviceToHost);
      if (e != hipSuccess) { g_vmm_last_error = (int)e; return -1; }
    }
    hipError_t e = vmm_unback(b);
    if (e != hipSuccess) { g_vmm_last_error = (int)e; return -1; }
    freed += (long long)b.size;
  }
  return freed;
}

// Give every sleeping block with this tag new pages at its old address (contents restored when a
// host copy exists, undefined otherwise).  Returns bytes mapped, or -1.
VDA_EXPORT long long vda_vmm_wake(int tag) {
  std::lock_guard<std::mutex> l(g_vmm_mu);
  long long mapped = 0;
  for (auto& kv : g_vmm_blocks) {
    VmmBlock& b = kv.second;
    if (b.tag != tag || b.mapped) continue;
    int prev = 0;
    hipGetDevice(&prev);
    hipSetDevice(b.device);
    hipError_t e = vmm_back(b);
    if (e == hipSuccess && b.host_copy) {
      e = hipMemcpy(b.va, b.host_copy, b.size, hipMemcpyHostToDevice);
      hipHostFree(b.host_copy);
      b.host_copy = nullptr;
    } else if (e == hipSuccess) {
      // dropped, not saved: the fresh pages hold whatever their last owner left.  Cleared like the
      // first allocation was, because kernels multiply masked-out cache slots by zero weights and a
      // NaN bit pattern there would survive the product.
      e = hipMemset(b.va, 0, b.size);
      if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    hipSetDevice(prev);
    if (e != hipSuccess) { g_vmm_last_error = (int)e; return -1; }
    mapped += (long long)b.size;
  }
  return mapped;
}

// Bytes held by a tag: which = 0 all blocks, 1 mapped only, 2 asleep only.
VDA_EXPORT long long vda_vmm_bytes(int tag, int which) {
  std::lock_guard<std::mutex> l(g_vmm_mu);
  long long n = 0;
  for (auto& kv : g_vmm_blocks) {
    const VmmBlock& b = kv.second;
    if (b.tag != tag) continue;
    if (which == 0 || (which == 1 && b.mapped) || (which == 2 && !b.mapped)) n += (long long)b.size;
  }
  return n;
}
