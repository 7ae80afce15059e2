// rhj_diag.hip.h — diagnostics build only (-DRHJ_INSTRUMENT): gather-rate microbenchmark of the fused kernel's access pattern
// (part of the device code of librhj.so; rhj_kernels.hip.h includes all of it)
#pragma once
#include "rhj_join_fused.hip.h"

namespace rhj {

#ifdef RHJ_INSTRUMENT
// Diagnostics build only (tools/gather_bench.py): the access pattern of phase 1 of the fused join without the
// join — every workgroup gathers random 16-byte tuples from its own contiguous region (a bucket's build side)
// through the same sc1 buffer loads, FJ_V in flight per lane; optionally it also streams `stream_elems` tuples per
// round like the probe side.  Gives the chip's rate for this pattern: the ceiling phase 1 can be compared with.
__global__ __launch_bounds__(FJ_BLOCK) void k_gather_bench(const rhj_tuple *base, uint32_t region_elems, uint32_t rounds,
                                                           const uint4 *stream, uint32_t stream_per_round, uint4 *sink)
{
    FjGather<false> G;
    G.init(base, (uint64_t)blockIdx.x * region_elems, region_elems);
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint4 *sp = stream + ((size_t)blockIdx.x * rounds) * stream_per_round * FJ_BLOCK + threadIdx.x;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint4 v[FJ_V];
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            x = x * 1664525u + 1013904223u;
            v[k] = G.load(__umulhi(x, region_elems));
        }
        for (uint32_t q = 0; q < stream_per_round; ++q) { const uint4 t = sp[((size_t)r * stream_per_round + q) * FJ_BLOCK]; acc.y ^= t.x; }
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) acc.x ^= v[k].x ^ v[k].z;
    }
    if (acc.x == 0x12345u && acc.y == 0x54321u) sink[threadIdx.x] = acc;
}
#endif

}  // namespace rhj
