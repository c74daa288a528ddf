// Free list of KV-cache blocks for the scheduler's block pool: a doubly linked list over two int arrays with
// O(1) pop-front, append and removal from the middle, in eviction order ("least recently freed first, and within
// one request tail blocks first").  The scheduler frees and allocates blocks for every running request on every
// step; the batched entry points below move one request's (or one step's) whole list of blocks per call.
// Behaviour follows the reference's vllm/v1/core/kv_cache_utils.py:126-280 (FreeKVCacheBlockQueue).
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace {
struct FreeList {
    std::vector<int32_t> prev, next;
    std::vector<uint8_t> in;
    int32_t head = -1, tail = -1;
    int64_t count = 0;
};

inline void unlink(FreeList* f, int32_t b) {
    const int32_t p = f->prev[b], n = f->next[b];
    if (p >= 0) f->next[p] = n; else f->head = n;
    if (n >= 0) f->prev[n] = p; else f->tail = p;
    f->prev[b] = f->next[b] = -1;
    f->in[b] = 0;
    --f->count;
}

inline void push_back(FreeList* f, int32_t b) {
    f->prev[b] = f->tail;
    f->next[b] = -1;
    if (f->tail >= 0) f->next[f->tail] = b; else f->head = b;
    f->tail = b;
    f->in[b] = 1;
    ++f->count;
}
}  // namespace

#define VDA_EXPORT extern "C" __attribute__((visibility("default")))

// Blocks 0 .. num_blocks - 1 in id order, minus those listed in skip.
VDA_EXPORT void* vda_fbq_create(int64_t num_blocks, const int64_t* skip, int64_t n_skip) {
    if (num_blocks < 0 || num_blocks > (int64_t(1) << 31) - 2) return nullptr;
    FreeList* f = new FreeList();
    f->prev.assign(num_blocks, -1);
    f->next.assign(num_blocks, -1);
    f->in.assign(num_blocks, 0);
    std::vector<uint8_t> out(num_blocks, 0);
    for (int64_t i = 0; i < n_skip; ++i)
        if (skip[i] >= 0 && skip[i] < num_blocks) out[skip[i]] = 1;
    for (int64_t b = 0; b < num_blocks; ++b)
        if (!out[b]) push_back(f, int32_t(b));
    return f;
}

VDA_EXPORT void vda_fbq_destroy(void* h) { delete static_cast<FreeList*>(h); }
VDA_EXPORT int64_t vda_fbq_count(void* h) { return static_cast<FreeList*>(h)->count; }
VDA_EXPORT int vda_fbq_contains(void* h, int64_t b) {
    FreeList* f = static_cast<FreeList*>(h);
    return b >= 0 && b < int64_t(f->in.size()) && f->in[b];
}

// Pop the n least recently freed blocks into out; -1 when fewer are free (nothing is popped then).
VDA_EXPORT int64_t vda_fbq_popleft_n(void* h, int64_t n, int64_t* out) {
    FreeList* f = static_cast<FreeList*>(h);
    if (n < 0 || n > f->count) return -1;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t b = f->head;
        unlink(f, b);
        out[i] = b;
    }
    return n;
}

// Take block b out of the middle (a cached block is reused); -1 when it is not free.
VDA_EXPORT int vda_fbq_remove(void* h, int64_t b) {
    FreeList* f = static_cast<FreeList*>(h);
    if (b < 0 || b >= int64_t(f->in.size()) || !f->in[b]) return -1;
    unlink(f, int32_t(b));
    return 0;
}

// Append ids in the given order; -1 - i when ids[i] is already free or out of range (the ones before it stay appended).
VDA_EXPORT int64_t vda_fbq_append_n(void* h, const int64_t* ids, int64_t n) {
    FreeList* f = static_cast<FreeList*>(h);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t b = ids[i];
        if (b < 0 || b >= int64_t(f->in.size()) || f->in[b]) return -1 - i;
        push_back(f, int32_t(b));
    }
    return n;
}

// The free blocks in eviction order; out holds count entries.
VDA_EXPORT int64_t vda_fbq_list(void* h, int64_t* out) {
    FreeList* f = static_cast<FreeList*>(h);
    int64_t i = 0;
    for (int32_t b = f->head; b >= 0; b = f->next[b]) out[i++] = b;
    return i;
}
