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
