// det_reduce.h -- run-to-run reproducible cross-block reduction of per-channel sums (gfx950).
//
// The BatchNorm statistics decide ReLU masks downstream, so a last-ulp wobble from atomicAdd ordering turns into O(1)
// gradient differences between two runs of the same step.  Every block stores its 32 partials (2 moments x 16 channels of
// one channel block) to scratch; the last block to arrive (ticket counter) adds them in chunk order, which does not depend
// on which block happened to be last.
//   scratch layout: float part[DRC_BN_MAX_CHUNKS][CB][32], then unsigned tickets[CB] (zero on entry, zero again on exit).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/disprcnn_hip.h"

namespace drc_det {

// called by all (>= 256) threads of the block; `v`, c_index and moment are those of thread threadIdx.x < 32 (ignored elsewhere)
__device__ __forceinline__ void finish(float v, int cb, int CB, int c_index, int moment, float* __restrict__ sums, float* scratch) {
    __shared__ unsigned s_last;
    float* part = scratch;
    unsigned* tickets = (unsigned*)(scratch + (size_t)DRC_BN_MAX_CHUNKS * CB * 32);
    if (threadIdx.x < 32) part[((size_t)blockIdx.x * CB + cb) * 32 + threadIdx.x] = v;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(tickets + cb, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // fixed-order sum of the gridDim.x partials: 8 groups of 32 threads take the chunks i = grp, grp+8, ... (independent
    // coherent loads, so they pipeline), then the 8 group sums are added in group order
    __shared__ float s_grp[8][32];
    const int grp = threadIdx.x >> 5, k = threadIdx.x & 31;
    if (grp < 8) {
        float t = 0.f;
        for (unsigned i = grp; i < gridDim.x; i += 8)
            t += __hip_atomic_load(part + ((size_t)i * CB + cb) * 32 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_grp[grp][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += s_grp[i][threadIdx.x];
        sums[moment * CB * 16 + c_index] = t;
    }
    if (threadIdx.x == 0) tickets[cb] = 0u;
}

}  // namespace drc_det
