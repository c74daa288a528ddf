// Parallel positional reads of byte ranges of one file into caller-provided (pinned) host memory.
// The checkpoint loader hands over every tensor of a window at once; the ranges are cut into pieces of at most
// PIECE bytes and a pool of threads pulls pieces off a shared counter, so one large tensor is read by many threads
// and many small ones do not serialise on a single reader.  No page-cache mapping is created (pread, not mmap): a
// 140 GB checkpoint streams through a fixed window instead of faulting its way through the address space.
#include <atomic>
#include <cerrno>
#include <cstdint>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {
constexpr uint64_t PIECE = 32ull << 20;
struct Piece { uint64_t off; uint64_t len; char* dst; };
}

extern "C" __attribute__((visibility("default"))) int vda_read_ranges(const char* path, int64_t n, const uint64_t* offsets, const uint64_t* sizes,
                               void* const* dsts, int threads) {
    int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return -errno;
#ifdef POSIX_FADV_SEQUENTIAL
    posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
#endif
    std::vector<Piece> pieces;
    for (int64_t i = 0; i < n; ++i)
        for (uint64_t done = 0; done < sizes[i]; done += PIECE)
            pieces.push_back({offsets[i] + done, sizes[i] - done < PIECE ? sizes[i] - done : PIECE,
                              static_cast<char*>(dsts[i]) + done});
    std::atomic<size_t> next{0};
    std::atomic<int> err{0};
    auto work = [&]() {
        for (;;) {
            size_t k = next.fetch_add(1, std::memory_order_relaxed);
            if (k >= pieces.size() || err.load(std::memory_order_relaxed)) return;
            const Piece& p = pieces[k];
            uint64_t got = 0;
            while (got < p.len) {
                ssize_t r = pread(fd, p.dst + got, p.len - got, static_cast<off_t>(p.off + got));
                if (r < 0) { if (errno == EINTR) continue; err.store(errno); return; }
                if (r == 0) { err.store(EIO); return; }          // range beyond the end of the file
                got += static_cast<uint64_t>(r);
            }
        }
    };
    int t = threads < 1 ? 1 : threads;
    if (static_cast<size_t>(t) > pieces.size()) t = pieces.empty() ? 1 : static_cast<int>(pieces.size());
    std::vector<std::thread> pool;
    for (int i = 1; i < t; ++i) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    close(fd);
    return -err.load();
}
