// rt_stream_kernels.hip — the instantiations of k_stream, in a translation unit of their own because they are compiled with
// another instruction-scheduling strategy than the rest of the library (`-mllvm -amdgpu-sched-strategy=max-ilp`, __graft_entry__.py):
// measured +1.6 % on both triangle workloads for k_stream and -0.9 % for k_trace on the sphere workload, so only k_stream gets it.
#include "rt_stream.hpp"

namespace rtk {

// the host-side handle (hipLaunchKernel / occupancy queries) of k_stream<counting, philox, compact nodes, triangles>; a scene without
// triangles takes the instantiation without traversal code (the node format plays no part in it)
const void* stream_kernel(bool counting, bool philox, bool compact, bool triangles)
{
    if (!triangles) {
        switch ((counting ? 2 : 0) | (philox ? 1 : 0)) {
        case 0: return (const void*)k_stream<false, false, false, false>;
        case 1: return (const void*)k_stream<false, true, false, false>;
        case 2: return (const void*)k_stream<true, false, false, false>;
        default: return (const void*)k_stream<true, true, false, false>;
        }
    }
    const int v = (counting ? 4 : 0) | (philox ? 2 : 0) | (compact ? 1 : 0);
    switch (v) {
    case 0: return (const void*)k_stream<false, false, false>;
    case 1: return (const void*)k_stream<false, false, true>;
    case 2: return (const void*)k_stream<false, true, false>;
    case 3: return (const void*)k_stream<false, true, true>;
    case 4: return (const void*)k_stream<true, false, false>;
    case 5: return (const void*)k_stream<true, false, true>;
    case 6: return (const void*)k_stream<true, true, false>;
    default: return (const void*)k_stream<true, true, true>;
    }
}

} // namespace rtk
