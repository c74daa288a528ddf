// Single-writer / multi-reader broadcast ring in POSIX shared memory.
//
// The executor process publishes one RPC message and every worker process on the node
// reads it without a socket hop or a per-reader copy on the writer side.  Layout:
//
//   [Header][Slot 0][Slot 1]...[Slot n-1]
//   Slot = { seq (u32 futex word), len, flags, acks } + payload[slot_bytes]
//
// Message k lives in slot k % n_slots.  The writer may reuse a slot once all readers
// acknowledged the message that was in it (acks == n_readers); a reader may read a slot
// once seq == (k + 1) (low 32 bits).  Waiting spins briefly, then sleeps on the futex word,
// so an idle engine burns no CPU while wake-up latency under load stays at the spin cost.
// Messages larger than a slot are split over consecutive slots (flag MORE).
//
// Behavioural counterpart of the reference's vllm/distributed/device_communicators/shm_broadcast.py
// (ShmRingBuffer/MessageQueue), built around futexes instead of polling + a zmq side channel.
#include <atomic>
#include <cerrno>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <linux/futex.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace {

constexpr uint32_t MAGIC = 0x56444152;   // "VDAR"
constexpr uint32_t FLAG_MORE = 1;

struct alignas(64) Header {
    uint32_t magic;
    uint32_t n_slots;
    uint32_t n_readers;
    uint32_t pad;
    uint64_t slot_bytes;
    std::atomic<uint32_t> closed;
    std::atomic<uint32_t> readers_attached;
};

struct alignas(64) Slot {
    std::atomic<uint32_t> seq;    // futex word: low 32 bits of (message index + 1)
    std::atomic<uint32_t> acks;   // futex word: readers done with the current content
    uint32_t flags;
    uint32_t pad;
    uint64_t len;     // bytes in this slot
    uint64_t total;   // bytes of the whole message (same in each of its slots)
};

struct Ring {
    Header* hdr;
    uint8_t* base;
    size_t map_bytes;
    uint64_t cursor;     // next message index for this endpoint
    int is_writer;
    char name[128];
};

inline size_t slot_stride(uint64_t slot_bytes) { return sizeof(Slot) + ((slot_bytes + 63) & ~uint64_t(63)); }
inline Slot* slot_at(Ring* r, uint64_t i) {
    return reinterpret_cast<Slot*>(r->base + sizeof(Header) + (i % r->hdr->n_slots) * slot_stride(r->hdr->slot_bytes));
}
inline uint8_t* payload(Slot* s) { return reinterpret_cast<uint8_t*>(s) + sizeof(Slot); }

inline long futex(std::atomic<uint32_t>* addr, int op, uint32_t val, const timespec* ts) {
    return syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), op, val, ts, nullptr, 0);
}

inline double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

// Wait until *addr == want.  Returns 0, -1 on timeout, -2 if the ring was closed.
int wait_equal(Ring* r, std::atomic<uint32_t>* addr, uint32_t want, double timeout_s) {
    for (int i = 0; i < 2000; ++i) {
        if (addr->load(std::memory_order_acquire) == want) return 0;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    const double deadline = timeout_s < 0 ? -1.0 : now_s() + timeout_s;
    for (;;) {
        uint32_t cur = addr->load(std::memory_order_acquire);
        if (cur == want) return 0;
        if (r->hdr->closed.load(std::memory_order_acquire)) return -2;
        timespec ts{0, 50 * 1000 * 1000};   // wake at least every 50 ms to notice close/timeouts
        if (deadline >= 0) {
            double left = deadline - now_s();
            if (left <= 0) return -1;
            if (left < 0.05) { ts.tv_sec = 0; ts.tv_nsec = long(left * 1e9); }
        }
        futex(addr, FUTEX_WAIT, cur, &ts);
    }
}

}  // namespace

extern "C" {

#define VDA_API __attribute__((visibility("default")))

VDA_API void* vda_ring_create(const char* name, uint32_t n_slots, uint64_t slot_bytes, uint32_t n_readers) {
    if (n_slots == 0 || slot_bytes == 0) return nullptr;
    size_t bytes = sizeof(Header) + size_t(n_slots) * slot_stride(slot_bytes);
    shm_unlink(name);
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return nullptr;
    if (ftruncate(fd, off_t(bytes)) != 0) { close(fd); shm_unlink(name); return nullptr; }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(name); return nullptr; }
    std::memset(p, 0, sizeof(Header));
    auto* r = new Ring();
    r->hdr = static_cast<Header*>(p);
    r->base = static_cast<uint8_t*>(p);
    r->map_bytes = bytes;
    r->cursor = 0;
    r->is_writer = 1;
    std::snprintf(r->name, sizeof(r->name), "%s", name);
    r->hdr->n_slots = n_slots;
    r->hdr->n_readers = n_readers;
    r->hdr->slot_bytes = slot_bytes;
    for (uint32_t i = 0; i < n_slots; ++i) {
        Slot* s = slot_at(r, i);
        s->seq.store(0, std::memory_order_relaxed);
        s->acks.store(n_readers, std::memory_order_relaxed);   // "free"
        s->flags = 0;
        s->len = 0;
    }
    std::atomic_thread_fence(std::memory_order_release);
    r->hdr->magic = MAGIC;
    return r;
}

VDA_API void* vda_ring_attach(const char* name) {
    int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0 || size_t(st.st_size) < sizeof(Header)) { close(fd); return nullptr; }
    void* p = mmap(nullptr, size_t(st.st_size), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return nullptr;
    auto* h = static_cast<Header*>(p);
    if (h->magic != MAGIC) { munmap(p, size_t(st.st_size)); return nullptr; }
    auto* r = new Ring();
    r->hdr = h;
    r->base = static_cast<uint8_t*>(p);
    r->map_bytes = size_t(st.st_size);
    r->cursor = 0;
    r->is_writer = 0;
    std::snprintf(r->name, sizeof(r->name), "%s", name);
    h->readers_attached.fetch_add(1, std::memory_order_acq_rel);
    return r;
}

VDA_API uint64_t vda_ring_slot_bytes(void* h) { return static_cast<Ring*>(h)->hdr->slot_bytes; }
VDA_API uint32_t vda_ring_readers_attached(void* h) {
    return static_cast<Ring*>(h)->hdr->readers_attached.load(std::memory_order_acquire);
}

// Writer: publish one message (split over slots when it exceeds a slot).  0 ok, -1 timeout, -2 closed.
VDA_API int vda_ring_write(void* h, const void* data, uint64_t len, double timeout_s) {
    Ring* r = static_cast<Ring*>(h);
    const uint8_t* src = static_cast<const uint8_t*>(data);
    const uint64_t cap = r->hdr->slot_bytes;
    uint64_t off = 0;
    do {
        Slot* s = slot_at(r, r->cursor);
        int rc = wait_equal(r, &s->acks, r->hdr->n_readers, timeout_s);
        if (rc != 0) return rc;
        uint64_t n = len - off < cap ? len - off : cap;
        std::memcpy(payload(s), src + off, n);
        off += n;
        s->len = n;
        s->total = len;
        s->flags = off < len ? FLAG_MORE : 0;
        s->acks.store(0, std::memory_order_relaxed);
        s->seq.store(uint32_t(r->cursor + 1), std::memory_order_release);
        futex(&s->seq, FUTEX_WAKE, INT_MAX, nullptr);
        ++r->cursor;
    } while (off < len);
    return 0;
}

// Reader: size of the next message, or <0 (-1 timeout, -2 closed).  Blocks until the first
// slot of the message is published.
VDA_API int64_t vda_ring_next_size(void* h, double timeout_s) {
    Ring* r = static_cast<Ring*>(h);
    Slot* s = slot_at(r, r->cursor);
    int rc = wait_equal(r, &s->seq, uint32_t(r->cursor + 1), timeout_s);
    if (rc != 0) return rc;
    return int64_t(s->total);
}

// Reader: copy the next message into dst (capacity cap).  Returns bytes or <0
// (-1 timeout, -2 closed, -4 dst too small -- the message is consumed up to the failure).
VDA_API int64_t vda_ring_read(void* h, void* dst, uint64_t cap, double timeout_s) {
    Ring* r = static_cast<Ring*>(h);
    uint8_t* out = static_cast<uint8_t*>(dst);
    uint64_t off = 0;
    for (;;) {
        Slot* s = slot_at(r, r->cursor);
        int rc = wait_equal(r, &s->seq, uint32_t(r->cursor + 1), timeout_s);
        if (rc != 0) return rc;
        uint64_t n = s->len;
        bool more = s->flags & FLAG_MORE;
        bool fits = off + n <= cap;
        if (fits) std::memcpy(out + off, payload(s), n);
        off += n;
        uint32_t done = s->acks.fetch_add(1, std::memory_order_acq_rel) + 1;
        if (done == r->hdr->n_readers) futex(&s->acks, FUTEX_WAKE, 1, nullptr);
        ++r->cursor;
        if (!fits) return -4;
        if (!more) return int64_t(off);
    }
}

VDA_API void vda_ring_close(void* h, int unlink_name) {
    Ring* r = static_cast<Ring*>(h);
    if (r->is_writer) {
        r->hdr->closed.store(1, std::memory_order_release);
        for (uint32_t i = 0; i < r->hdr->n_slots; ++i) futex(&slot_at(r, i)->seq, FUTEX_WAKE, INT_MAX, nullptr);
    }
    if (unlink_name) shm_unlink(r->name);
    munmap(r->base, r->map_bytes);
    delete r;
}

}  // extern "C"
