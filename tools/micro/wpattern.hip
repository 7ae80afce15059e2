// Store-pattern microbenchmark (MI355X): how fast do N waves write when each writes 1 KB per instruction into a region of
// its own, against workgroup-wide contiguous bursts?  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/wpattern.hip -o tools/micro/wpattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

// mode 0: every WAVE owns a contiguous region of `per_wave` bytes and writes it 1 KB per instruction
// mode 1: every WORKGROUP owns a region; its 4 waves write 4 KB per step contiguously
// gap: dependent global loads between two stores (0 = none) — a gather-like wait
__global__ __launch_bounds__(256) void k_store(uint4 *out, const uint32_t *src, size_t per_wave, int mode, int gap, uint32_t mask)
{
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t gw = (size_t)blockIdx.x * 4 + w;
    const size_t iters = per_wave / 1024;
    uint32_t x = (uint32_t)gw * 2654435761u + lane;
    for (size_t i = 0; i < iters; ++i) {
        for (int g = 0; g < gap; ++g) x = src[(x * 1664525u + 1013904223u) & mask] + (uint32_t)i;
        size_t at;
        if (mode == 0) at = gw * (per_wave / 16) + i * 64 + lane;
        else           at = (size_t)blockIdx.x * (per_wave / 16) * 4 + (i * 4 + w) * 64 + lane;
        out[at] = make_uint4(x, (uint32_t)i, lane, 0u);
    }
}

int main(int argc, char **argv)
{
    const size_t total = (size_t)1600 << 20;                       // 1.6 GB
    uint4 *out; uint32_t *src;
    hipMalloc(&out, total); hipMalloc(&src, 64 << 20); hipMemset(src, 1, 64 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int gap = 0; gap <= 1; ++gap)
        for (int mode = 0; mode <= 1; ++mode)
            for (size_t per_wave : {(size_t)16 << 10, (size_t)64 << 10, (size_t)400 << 10, (size_t)1600 << 10}) {
                const size_t waves = total / per_wave;
                const unsigned grid = (unsigned)(waves / 4);
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEventRecord(a);
                    hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, 0, out, src, per_wave, mode, gap, (uint32_t)((16u << 20) - 1));
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
                }
                printf("gap %d mode %s region %5zu KB per wave, %7zu waves: %.3f ms = %.2f TB/s\n", gap, mode ? "wg  " : "wave", per_wave >> 10, waves, best,
                       total / (best * 1e-3) / 1e12);
            }
    return 0;
}
