// rhj_host.cpp — the host-only concurrency of librhj.so, kept free of HIP so that it builds and runs under
// -fsanitize=address,undefined / thread on a machine without a GPU (make asan / make tsan, tests/host_san_driver.cpp):
//   * the library's one API lock (every entry point runs under it: rhj_internal.h);
//   * the mover that takes a join's pairs from a ring of pinned staging blocks into the caller-visible result nodes
//     with several host threads (the D2H half of RadixHashJoin() with host pointers, rhj_device.hip: rhj_host_join).
#include "rhj_internal.h"

#include <atomic>
#include <mutex>
#include <string.h>
#include <thread>
#include <vector>

static std::recursive_mutex g_api_mutex;

extern "C" void rhj_api_lock(void) { g_api_mutex.lock(); }
extern "C" void rhj_api_unlock(void) { g_api_mutex.unlock(); }

// Moves `total` elements of `elem` bytes that arrive in blocks of `blk` elements through `ring` staging buffers (block b
// lands in staging[b % ring]) into the nodes (node i holds elements [i * node_elems, (i + 1) * node_elems)).
//   issue(ctx, b)  starts the copy of block b into its staging buffer (called from the calling thread, in block order, at
//                  most ring - 1 blocks ahead of the block being moved: a buffer is reused only after every mover is
//                  through with the block it held);
//   wait(ctx, b)   returns once block b has landed.
// The nodes are plain malloc memory (FreeResult = free(buff); free(node), results.c:144-153): freshly mapped pages, so
// whoever writes them first pays the page faults — one thread filling them measured 5.8 GB/s.  Mover thread t of
// `threads` therefore moves the t-th slice of every block as soon as the block is announced, each faulting in its own pages.
extern "C" int rhj_move_blocks(uint64_t total, uint64_t elem, uint64_t node_elems, char *const *nodes, uint64_t blk, int ring,
                               char *const *staging, unsigned threads, int (*issue)(void *ctx, uint64_t b),
                               int (*wait)(void *ctx, uint64_t b), void *ctx)
{
    if (total == 0) return 0;
    if (ring < 2 || blk == 0 || node_elems == 0 || elem == 0) return -1;
    const uint64_t nblk = (total + blk - 1) / blk;
    if (threads < 1) threads = 1;
    auto block_elems = [&](uint64_t b) { return total - b * blk < blk ? total - b * blk : blk; };
    // elements [first, first + cnt) of the result, at `src`, into the nodes
    auto move = [&](const char *src, uint64_t first, uint64_t cnt) {
        while (cnt) {
            const uint64_t node = first / node_elems, at = first % node_elems;
            const uint64_t take = cnt < node_elems - at ? cnt : node_elems - at;
            memcpy(nodes[(size_t)node] + at * elem, src, take * elem);
            src += take * elem; first += take; cnt -= take;
        }
    };
    std::vector<std::atomic<int>> ready((size_t)nblk), done((size_t)nblk);
    for (uint64_t b = 0; b < nblk; ++b) { ready[(size_t)b].store(0); done[(size_t)b].store(0); }
    std::atomic<int> abort_flag{0};
    auto worker = [&](unsigned t) {
        for (uint64_t b = 0; b < nblk; ++b) {
            while (!ready[(size_t)b].load(std::memory_order_acquire)) {
                if (abort_flag.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            const uint64_t cnt = block_elems(b);
            const uint64_t per = (cnt + threads - 1) / threads, o = (uint64_t)t * per;
            if (o < cnt) move(staging[b % (uint64_t)ring] + o * elem, b * blk + o, cnt - o < per ? cnt - o : per);
            done[(size_t)b].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    if (threads > 1)
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back(worker, t);
    int rc = 0;
    const uint64_t ahead = (uint64_t)ring - 1;
    for (uint64_t b = 0; b < nblk && b < ahead && !rc; ++b) rc = issue(ctx, b);
    for (uint64_t b = 0; b < nblk && !rc; ++b) {
        if (b + ahead < nblk) {
            // the staging buffer of block b + ring - 1 held block b - 1: wait until every mover is through with it
            if (b > 0 && threads > 1)
                while (done[(size_t)(b - 1)].load(std::memory_order_acquire) < (int)threads) std::this_thread::yield();
            rc = issue(ctx, b + ahead);
            if (rc) break;
        }
        rc = wait(ctx, b);
        if (rc) break;
        if (threads == 1) move(staging[b % (uint64_t)ring], b * blk, block_elems(b));
        else ready[(size_t)b].store(1, std::memory_order_release);
    }
    if (rc) abort_flag.store(1);
    for (auto &th : pool) th.join();
    return rc ? -1 : 0;
}
