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

// ---- (count, mean, M2) statistics, combined with Chan's pairwise update: one pass over the data gives a cancellation-free variance.
struct Stat { float n, mean, m2; };
__device__ __forceinline__ Stat merge(const Stat a, const Stat b) {
    Stat r;
    r.n = a.n + b.n;
    const float inv = r.n > 0.f ? 1.f / r.n : 0.f;
    const float d = b.mean - a.mean;
    r.mean = a.mean + d * (b.n * inv);
    r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n * inv);
    return r;
}

// called by all (>= 256) threads of the block; `st` is the block's statistic of channel cb*16 + threadIdx.x for threadIdx.x < 16.
// scratch layout: float part[DRC_BN_MAX_CHUNKS][CB][32] (mean in [0,16), M2 in [16,32)) -- the count of chunk i is recomputed
// from n_of(i) -- then unsigned tickets[CB].  out_mean / out_m2: [CB*16].
template <class NOf>
__device__ __forceinline__ void finish_stat(Stat st, int cb, int CB, float* __restrict__ out_mean, float* __restrict__ out_m2, float* scratch,
                                            NOf n_of) {
    __shared__ unsigned s_last2;
    float* part = scratch;
    unsigned* tickets = (unsigned*)(scratch + (size_t)DRC_BN_MAX_CHUNKS * CB * 32);
    if (threadIdx.x < 16) {
        part[((size_t)blockIdx.x * CB + cb) * 32 + threadIdx.x] = st.mean;
        part[((size_t)blockIdx.x * CB + cb) * 32 + 16 + threadIdx.x] = st.m2;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last2 = (atomicAdd(tickets + cb, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last2) return;
    __threadfence();
    // fixed-order combine: 16 groups of 16 threads merge the chunks i = grp, grp+16, ... in order, then the 16 group results
    // are merged in group order -- the result does not depend on which block happened to be last
    __shared__ float s_g[16][16][3];
    const int grp = threadIdx.x >> 4, c = threadIdx.x & 15;
    if (grp < 16) {
        Stat acc = {0.f, 0.f, 0.f};
        for (unsigned i = grp; i < gridDim.x; i += 16) {
            Stat b;
            b.n = n_of(i);
            b.mean = __hip_atomic_load(part + ((size_t)i * CB + cb) * 32 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b.m2 = __hip_atomic_load(part + ((size_t)i * CB + cb) * 32 + 16 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc = merge(acc, b);
        }
        s_g[grp][c][0] = acc.n; s_g[grp][c][1] = acc.mean; s_g[grp][c][2] = acc.m2;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        Stat acc = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = merge(acc, Stat{s_g[i][threadIdx.x][0], s_g[i][threadIdx.x][1], s_g[i][threadIdx.x][2]});
        out_mean[cb * 16 + threadIdx.x] = acc.mean;
        out_m2[cb * 16 + threadIdx.x] = acc.m2;
    }
    if (threadIdx.x == 0) tickets[cb] = 0u;
}

}  // namespace drc_det
