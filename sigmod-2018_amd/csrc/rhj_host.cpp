// rhj_host.cpp — the host-only concurrency of librhj.so, kept free of HIP so that it builds and runs under
// -fsanitize=address,undefined / thread on a machine without a GPU (make asan / make tsan, tests/host_san_driver.cpp):
//   * the library's one API lock (every entry point runs under it: rhj_internal.h);
//   * the mover that takes a join's pairs from a ring of pinned staging blocks into the caller-visible result nodes
//     with several host threads (the D2H half of RadixHashJoin() with host pointers, rhj_device.hip: rhj_host_join).
#include "rhj_internal.h"

#include <atomic>
#include <chrono>
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
// A waiting thread yields a few hundred times (a block is due within microseconds while the copies stream) and then sleeps in
// short naps: the movers and the issuing thread share the host with the caller's own threads (the reference engine's pool).
static inline void backoff(unsigned &spins)
{
    if (++spins < 256) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(20));
}

static int move_blocks(uint64_t base, uint64_t total, uint64_t elem, uint64_t node_elems, char *const *nodes, uint64_t blk, int ring,
                       char *const *staging, unsigned threads, int (*issue)(void *ctx, uint64_t b),
                       int (*wait)(void *ctx, uint64_t b), void *ctx)
{
    const uint64_t nblk = (total + blk - 1) / blk;
    auto block_elems = [&](uint64_t b) { return total - b * blk < blk ? total - b * blk : blk; };
    // elements [first, first + cnt) of the result, at `src`, into the nodes
    auto move = [&](const char *src, uint64_t first, uint64_t cnt) {
        first += base;                                   // (the list's element in front of which this call's elements begin)
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
    auto worker = [&](unsigned t, unsigned nthreads) {
        for (uint64_t b = 0; b < nblk; ++b) {
            unsigned spins = 0;
            while (!ready[(size_t)b].load(std::memory_order_acquire)) {
                if (abort_flag.load(std::memory_order_relaxed)) return;
                backoff(spins);
            }
            const uint64_t cnt = block_elems(b);
            const uint64_t per = (cnt + nthreads - 1) / nthreads, o = (uint64_t)t * per;
            if (o < cnt) move(staging[b % (uint64_t)ring] + o * elem, b * blk + o, cnt - o < per ? cnt - o : per);
            done[(size_t)b].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    if (threads > 1) {
        // a host that refuses the threads (thread or memory limits: std::system_error, bad_alloc) gets the move done inline
        try {
            pool.reserve(threads);
            for (unsigned t = 0; t < threads; ++t) pool.emplace_back(worker, t, threads);
        } catch (...) {
            abort_flag.store(1);
            for (auto &th : pool) th.join();
            pool.clear();
            abort_flag.store(0);
            threads = 1;
        }
    }
    int rc = 0;
    const uint64_t ahead = (uint64_t)ring - 1;
    for (uint64_t b = 0; b < nblk && b < ahead && !rc; ++b) rc = issue(ctx, b);
    for (uint64_t b = 0; b < nblk && !rc; ++b) {
        if (b + ahead < nblk) {
            // the staging buffer of block b + ring - 1 held block b - 1: wait until every mover is through with it
            if (b > 0 && threads > 1) {
                unsigned spins = 0;
                while (done[(size_t)(b - 1)].load(std::memory_order_acquire) < (int)threads) backoff(spins);
            }
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

// extern "C": no exception may leave (the caller is rhj_abi.c, in the middle of RadixHashJoin): what the vectors or the
// thread library throw becomes -1.  `base`: the elements are elements [base, base + total) of the list the nodes hold (one
// device's share of a join sharded over several devices; 0 for a whole list).
extern "C" int rhj_move_blocks_at(uint64_t base, uint64_t total, uint64_t elem, uint64_t node_elems, char *const *nodes, uint64_t blk,
                                  int ring, char *const *staging, unsigned threads, int (*issue)(void *ctx, uint64_t b),
                                  int (*wait)(void *ctx, uint64_t b), void *ctx)
{
    if (total == 0) return 0;
    if (ring < 2 || blk == 0 || node_elems == 0 || elem == 0) return -1;
    if (threads < 1) threads = 1;
    try {
        return move_blocks(base, total, elem, node_elems, nodes, blk, ring, staging, threads, issue, wait, ctx);
    } catch (...) {
        return -1;
    }
}

extern "C" int rhj_move_blocks(uint64_t total, uint64_t elem, uint64_t node_elems, char *const *nodes, uint64_t blk, int ring,
                               char *const *staging, unsigned threads, int (*issue)(void *ctx, uint64_t b),
                               int (*wait)(void *ctx, uint64_t b), void *ctx)
{
    return rhj_move_blocks_at(0, total, elem, node_elems, nodes, blk, ring, staging, threads, issue, wait, ctx);
}
