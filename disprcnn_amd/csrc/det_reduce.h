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

// called by all threads of the block; `v` is meaningful for threadIdx.x < 32: moment k>>2... see callers (index = threadIdx.x)
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
    if (threadIdx.x < 32) {
        const volatile float* pv = part;
        float t = 0.f;
        for (unsigned i = 0; i < gridDim.x; ++i) t += pv[((size_t)i * CB + cb) * 32 + threadIdx.x];
        sums[moment * CB * 16 + c_index] = t;
    }
    if (threadIdx.x == 0) tickets[cb] = 0u;
}

}  // namespace drc_det
