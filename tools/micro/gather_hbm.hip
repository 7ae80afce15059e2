// What the chip delivers for k_gather_tables' access pattern (rhj_inter.hip: dst[i] = src[idx[i]], 8-byte values, idx a random
// match list): N random 8-byte reads out of a table far larger than the 256 MB Infinity Cache, index read + coalesced write
// included, against the same kernel on a sorted index (a stream) — the ceiling for `inter_res.c:95-101`'s gathers.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/gather_hbm.hip -o tools/micro/gather_hbm ; run: tools/micro/gather_hbm
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void fill_idx(uint64_t *idx, uint64_t n, uint64_t rows, int sorted)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        uint64_t x = i * 0x9e3779b97f4a7c15ull; x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
        idx[i] = sorted ? (uint64_t)((__uint128_t)i * rows / n) : x % rows;
    }
}
// one row a thread (k_gather_tables as it is)
__global__ __launch_bounds__(256) void gather1(uint64_t *dst, const uint64_t *src, const uint64_t *idx, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
// V rows a thread, all index reads, then all gathers, in flight together
template <int V>
__global__ __launch_bounds__(256) void gatherv(uint64_t *dst, const uint64_t *src, const uint64_t *idx, uint64_t n)
{
    const uint64_t i0 = (uint64_t)blockIdx.x * 256 * V + threadIdx.x;
    uint64_t p[V], v[V];
#pragma unroll
    for (int k = 0; k < V; ++k) p[k] = i0 + k * 256 < n ? idx[i0 + k * 256] : 0;
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = src[p[k]];
#pragma unroll
    for (int k = 0; k < V; ++k) if (i0 + k * 256 < n) dst[i0 + k * 256] = v[k];
}
int main()
{
    const uint64_t n = 100000000;
    uint64_t *idx, *dst, *src;
    CHECK(hipMalloc((void **)&idx, n * 8)); CHECK(hipMalloc((void **)&dst, n * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (uint64_t rows : {(uint64_t)4000000, (uint64_t)25000000, (uint64_t)100000000, (uint64_t)400000000}) {
        CHECK(hipMalloc((void **)&src, rows * 8)); CHECK(hipMemset(src, 1, rows * 8));
        for (int sorted = 0; sorted < 2; ++sorted) {
            hipLaunchKernelGGL(fill_idx, dim3(4096), dim3(256), 0, 0, idx, n, rows, sorted);
            for (int var = 0; var < 3; ++var) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    CHECK(hipEventRecord(e0));
                    if (var == 0) hipLaunchKernelGGL(gather1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dst, src, idx, n);
                    else if (var == 1) hipLaunchKernelGGL(gatherv<4>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, dst, src, idx, n);
                    else hipLaunchKernelGGL(gatherv<8>, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, dst, src, idx, n);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
                }
                printf("table %4llu MB  %s index  %d row(s) a thread: %.3f ms for 100 M rows = %.1f G rows/s\n", (unsigned long long)(rows * 8 >> 20),
                       sorted ? "sorted" : "random", var == 0 ? 1 : var == 1 ? 4 : 8, best, n / best / 1e6);
            }
        }
        CHECK(hipFree(src));
    }
    return 0;
}
