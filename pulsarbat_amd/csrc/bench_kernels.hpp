// bench_kernels.hpp -- the streaming yardstick of bench.py's path_roofline and of the allocation-class probes (pbhip.hip).
#pragma once
#include "fft_core.hpp"

namespace PBH_NS {
#ifndef PBH_F64
// ---- streaming copy: the achievable-HBM yardstick --------------------------------------------------------
// One contiguous 16-KiB chunk per workgroup, four 16-byte loads in flight per lane: the fastest plain copy on MI355X
// (tools/micro/membench.hip, profiles/r01_membench.txt: 5.93 TB/s; a grid-stride loop reaches 4.7-5.0, hipMemcpyDtoD 5.3).
// Round 4: the round-3 form guarded every element (`if (base + u*256 < n) v[u] = ...`), which made hipcc keep the
// conditionally initialised float4 v[4] in LDS and wait for each load before issuing the next (global_load_dwordx4 ->
// s_waitcnt vmcnt(0) -> ds_write_b128, four times: SQ_INSTS_LDS 4.2e6 in a copy kernel).  Full chunks now take a
// guard-free path with the four loads in registers; only the last, partial chunk of a launch is guarded.
// RMW = true: the same chunks read, modified and written back IN PLACE -- the ceiling of the three middle passes, which
// update the planar work buffer where it stands.
template <bool RMW>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ in, float4* __restrict__ out, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (base - threadIdx.x + 1024 <= n) {   // block-uniform: a full chunk
        const float4 a = in[base], b = in[base + 256], c = in[base + 512], d = in[base + 768];
        if constexpr (RMW) {
            out[base] = make_float4(a.x + 1.0f, a.y, a.z, a.w);
            out[base + 256] = make_float4(b.x + 1.0f, b.y, b.z, b.w);
            out[base + 512] = make_float4(c.x + 1.0f, c.y, c.z, c.w);
            out[base + 768] = make_float4(d.x + 1.0f, d.y, d.z, d.w);
        } else {
            out[base] = a;
            out[base + 256] = b;
            out[base + 512] = c;
            out[base + 768] = d;
        }
    } else {
        for (int u = 0; u < 4; ++u)
            if (base + u * 256 < n) {
                float4 a = in[base + u * 256];
                if constexpr (RMW) a.x += 1.0f;
                out[base + u * 256] = a;
            }
    }
}

#endif  // !PBH_F64
}  // namespace PBH_NS
