// Cost of a grid-wide barrier on MI355X (256 workgroups x 1024 threads, one per CU), by the fences it carries.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/gridbar.hip -o tools/micro/gridbar
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct GridBar { uint32_t count, epoch, abort, done; };

// MODE bit 0: release fence (thread 0), bit 1: acquire fence (thread 0), bit 2: acquire fence on every wave
template <int MODE>
__device__ __forceinline__ void grid_sync(GridBar *gb, uint32_t G)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE & 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t ep = __hip_atomic_load(&gb->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&gb->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1u) {
            __hip_atomic_store(&gb->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&gb->epoch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(&gb->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ep) {
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) break;
            }
        }
        if (MODE & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (MODE & 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_bar(GridBar *gb, uint4 *buf, uint32_t per_wg_bytes, int rounds, uint64_t *clk)
{
    const uint32_t G = gridDim.x;
    uint64_t t0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        // dirty some lines: every workgroup writes per_wg_bytes of its own region
        uint4 *mine = buf + (size_t)blockIdx.x * (per_wg_bytes / 16);
        for (uint32_t i = threadIdx.x; i < per_wg_bytes / 16; i += 1024) mine[i] = make_uint4(r, i, 0, 0);
        grid_sync<MODE>(gb, G);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = __builtin_amdgcn_s_memrealtime() - t0;
}

template <int MODE> void run(const char *name, GridBar *gb, uint4 *buf, uint64_t *clk, uint32_t bytes)
{
    const int rounds = 20;
    hipMemset(gb, 0, sizeof(GridBar));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_bar<MODE>, dim3(256), dim3(1024), 0, 0, gb, buf, bytes, rounds, clk);
        hipDeviceSynchronize();
    }
    uint64_t h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6u KB/wg dirtied: %.2f us per round\n", name, bytes >> 10, h * 0.01 / rounds);
}

int main()
{
    GridBar *gb; uint4 *buf; uint64_t *clk;
    hipMalloc(&gb, 256); hipMalloc(&buf, (size_t)256 << 20); hipMalloc(&clk, 64);
    for (uint32_t bytes : {0u, 16u << 10, 128u << 10}) {
        run<0>("no fences", gb, buf, clk, bytes);
        run<1>("release (thread 0)", gb, buf, clk, bytes);
        run<2>("acquire (thread 0)", gb, buf, clk, bytes);
        run<3>("release + acquire (thread 0)", gb, buf, clk, bytes);
        run<7>("release + acquire (thread 0) + acquire/wave", gb, buf, clk, bytes);
    }
    return 0;
}
