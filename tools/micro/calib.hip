// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, in the access patterns of this
// library's kernels (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Every kernel is launched three times over 1.2 .. 1.6 GB (well past the 256 MB Infinity Cache); the counters
// of the last dispatch divided by the bytes the pattern moves by construction are the factors tools/pmc_summary.py applies.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/calib.hip -o tools/micro/calib
// Run (one counter a pass, the program itself behind --):
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calib_fetch -- tools/micro/calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/calib_write -- tools/micro/calib
//   python3 tools/calib_summary.py  ->  gpurun_out/calibration.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct __attribute__((aligned(4))) T12 { uint32_t a, b, c; };

// 16 B a lane, coalesced (the partition's reads of the caller's tuples, the pair stores)
__global__ __launch_bounds__(256) void calib_stream_read16(const uint4 *in, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = in[i]; acc ^= v.x ^ v.w; }
    if (acc == 0x12345u) sink[threadIdx.x] = acc;
}
// 12 B a lane (global_load_dwordx3: the 12-byte partitioned tuples, pass 2 and the join's probe side)
__global__ __launch_bounds__(256) void calib_stream_read12(const T12 *in, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const T12 v = in[i]; acc ^= v.a ^ v.c; }
    if (acc == 0x12345u) sink[threadIdx.x] = acc;
}
// the 8-byte key of every 12-byte tuple (the join's build passes)
__global__ __launch_bounds__(256) void calib_stream_read8of12(const T12 *in, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint2 v = *reinterpret_cast<const uint2 *>(&in[i]); acc ^= v.x ^ v.y; }
    if (acc == 0x12345u) sink[threadIdx.x] = acc;
}
// 8 B a lane (a filter's column)
__global__ __launch_bounds__(256) void calib_stream_read8(const uint2 *in, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint2 v = in[i]; acc ^= v.x ^ v.y; }
    if (acc == 0x12345u) sink[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void calib_stream_write16(uint4 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ __launch_bounds__(256) void calib_stream_write12(T12 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = T12{(uint32_t)i, 1u, 2u};
}
// The fused join's gathers: every workgroup (one a CU, 1024 threads, four loads in flight a lane) takes random records from ITS
// region through a buffer descriptor with sc1 — BYTES = 12: dwordx3 from a region of `elems` 12-byte tuples (a bucket's build
// side: 24 414 = C3 at 12 radix bits, 293 KB), BYTES = 4: one dword from a region of `elems` row ids (98 KB)
template <int BYTES>
__global__ __launch_bounds__(1024) void calib_gather(const uint32_t *base, uint32_t elems, uint32_t rounds, uint32_t *sink)
{
    const uint64_t addr = (uint64_t)(base + (size_t)blockIdx.x * elems * (BYTES / 4));
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)addr, 0, (int)(elems * BYTES), 0x00020000);
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u, acc = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x = x * 1664525u + 1013904223u;
            const uint32_t e = __umulhi(x, elems);
            if (BYTES == 12) {
                typedef uint32_t v3 __attribute__((ext_vector_type(3)));
                const v3 t = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(e * 12u), 0, 16 /* sc1 */);
                v[k] = t.x ^ t.z;
            } else
                v[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(e * 4u), 0, 16 /* sc1 */);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc ^= v[k];
    }
    if (acc == 0x12345u) sink[threadIdx.x] = acc;
}

int main()
{
    const size_t bytes = (size_t)1600 << 20;                         // 1.6 GB: 100 M 16-byte tuples
    void *a, *b;
    uint32_t *sink;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc((void **)&sink, 4096));
    CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
    const size_t n16 = bytes / 16, n12 = 100000000, n8 = bytes / 8;
    const unsigned grid = 256 * 16;
    printf("pattern bytes_read bytes_written\n");
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_stream_read16, dim3(grid), dim3(256), 0, 0, (const uint4 *)a, n16, sink);
        hipLaunchKernelGGL(calib_stream_read12, dim3(grid), dim3(256), 0, 0, (const T12 *)a, n12, sink);
        hipLaunchKernelGGL(calib_stream_read8of12, dim3(grid), dim3(256), 0, 0, (const T12 *)a, n12, sink);
        hipLaunchKernelGGL(calib_stream_read8, dim3(grid), dim3(256), 0, 0, (const uint2 *)a, n8, sink);
        hipLaunchKernelGGL(calib_stream_write16, dim3(grid), dim3(256), 0, 0, (uint4 *)b, n16);
        hipLaunchKernelGGL(calib_stream_write12, dim3(grid), dim3(256), 0, 0, (T12 *)b, n12);
        // 256 workgroups x 96 rounds x 1024 threads x 4 = 100.7 M gathers, as C3's probe phase
        hipLaunchKernelGGL(calib_gather<12>, dim3(256), dim3(1024), 0, 0, (const uint32_t *)a, 24414u, 96u, sink);
        hipLaunchKernelGGL(calib_gather<4>, dim3(256), dim3(1024), 0, 0, (const uint32_t *)a, 24414u, 96u, sink);
        CHECK(hipDeviceSynchronize());
    }
    printf("calib_stream_read16 %zu 0\ncalib_stream_read12 %zu 0\ncalib_stream_read8of12 %zu 0\ncalib_stream_read8 %zu 0\n", n16 * 16, n12 * 12, n12 * 8, n8 * 8);
    printf("calib_stream_write16 0 %zu\ncalib_stream_write12 0 %zu\n", n16 * 16, n12 * 12);
    printf("calib_gather<12> %zu 0\ncalib_gather<4> %zu 0\n", (size_t)256 * 96 * 1024 * 4, (size_t)256 * 96 * 1024 * 4);   // (gathers, not bytes: the factor is bytes a gather)
    return 0;
}
