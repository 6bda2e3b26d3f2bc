// det_reduce.h -- run-to-run reproducible cross-block reduction of per-channel sums (gfx950).
//
// The BatchNorm statistics decide ReLU masks downstream, so a last-ulp wobble from atomicAdd ordering turns into O(1)
// gradient differences between two runs of the same step.  Every block stores its 32 partials (2 moments x 16 channels of
// one channel block) to scratch; the last block to arrive (ticket counter) adds them in chunk order, which does not depend
// on which block happened to be last.
//   scratch layout: float part[DRC_BN_MAX_CHUNKS][CB][32], then the ticket words (zero on entry, zero again on exit):
//   unsigned t1[CB][32 groups][32] and unsigned t2[CB][32], one counter per 128-byte line.  Same-line device-scope atomics
//   serialise at ~90 ns each on MI355X (measured: 1,024 blocks on one line = 90 us of a 93 us launch), so arrival is counted
//   in two levels: 16 blocks share a first-level line, the last of each group bumps the channel block's second-level counter.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/disprcnn_hip.h"

namespace drc_det {

// Called by all threads of the block after its partials were stored BY WAVE 0 (threads < 64).  True (block-uniformly) in the one
// block per channel block that arrived last; that block then sees every other block's partials with plain loads.
// Only wave 0 executes the agent-scope release: the fence is an L2 write-back request per WAVE, and with every wave of every
// block issuing one (4,096 per full-resolution launch) the requests queued at the eight L2s were most of the kernel's time.
__device__ __forceinline__ bool arrive_last(int cb, int CB, float* scratch) {
    __shared__ unsigned s_last_arrival;
    unsigned* t1 = (unsigned*)(scratch + (size_t)DRC_BN_MAX_CHUNKS * CB * 32);
    unsigned* t2 = t1 + (size_t)CB * 32 * 32;
    if (threadIdx.x < 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (threadIdx.x == 0) {
            const unsigned grp = blockIdx.x >> 4, ngrp = (gridDim.x + 15u) >> 4;
            const unsigned rest = gridDim.x - grp * 16u;
            const unsigned in_grp = rest < 16u ? rest : 16u;
            unsigned last = 0u;
            if (atomicAdd(t1 + ((size_t)cb * 32 + grp) * 32, 1u) == in_grp - 1u) {
                __threadfence();
                last = atomicAdd(t2 + (size_t)cb * 32, 1u) == ngrp - 1u ? 1u : 0u;
            }
            s_last_arrival = last;
        }
    }
    __syncthreads();
    if (!s_last_arrival) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}
// the last block re-arms the counters of its channel block (any thread count >= 32)
__device__ __forceinline__ void rearm(int cb, int CB, float* scratch) {
    unsigned* t1 = (unsigned*)(scratch + (size_t)DRC_BN_MAX_CHUNKS * CB * 32);
    unsigned* t2 = t1 + (size_t)CB * 32 * 32;
    if (threadIdx.x < 32) t1[((size_t)cb * 32 + threadIdx.x) * 32] = 0u;
    if (threadIdx.x == 0) t2[(size_t)cb * 32] = 0u;
}

// called by all (>= 256) threads of the block; `v`, c_index and moment are those of thread threadIdx.x < 32 (ignored elsewhere)
__device__ __forceinline__ void finish(float v, int cb, int CB, int c_index, int moment, float* __restrict__ sums, float* scratch) {
    float* part = scratch;
    if (threadIdx.x < 32) part[((size_t)blockIdx.x * CB + cb) * 32 + threadIdx.x] = v;
    if (!arrive_last(cb, CB, scratch)) return;
    // fixed-order sum of the gridDim.x partials: 8 groups of 32 threads take the chunks i = grp, grp+8, ... (independent
    // coherent loads, so they pipeline), then the 8 group sums are added in group order
    __shared__ float s_grp[8][32];
    const int grp = threadIdx.x >> 5, k = threadIdx.x & 31;
    if (grp < 8) {
        // the agent-scope fence above invalidated this CU's L1 (and the partials were released to memory by their writers), so
        // plain loads see them -- and, unlike atomic loads, all of a thread's loads are in flight together
        float vals[DRC_BN_MAX_CHUNKS / 8];
#pragma unroll
        for (int it = 0; it < DRC_BN_MAX_CHUNKS / 8; ++it) {
            const unsigned i = grp + it * 8;
            vals[it] = i < gridDim.x ? part[((size_t)i * CB + cb) * 32 + k] : 0.f;
        }
        float t = 0.f;
#pragma unroll
        for (int it = 0; it < DRC_BN_MAX_CHUNKS / 8; ++it) t += vals[it];
        s_grp[grp][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += s_grp[i][threadIdx.x];
        sums[moment * CB * 16 + c_index] = t;
    }
    rearm(cb, CB, scratch);
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
// from n_of(i) -- then the ticket words.  out_mean / out_m2: [CB*16].
template <class NOf>
__device__ __forceinline__ void finish_stat(Stat st, int cb, int CB, float* __restrict__ out_mean, float* __restrict__ out_m2, float* scratch,
                                            NOf n_of) {
    float* part = scratch;
    if (threadIdx.x < 16) {
        part[((size_t)blockIdx.x * CB + cb) * 32 + threadIdx.x] = st.mean;
        part[((size_t)blockIdx.x * CB + cb) * 32 + 16 + threadIdx.x] = st.m2;
    }
    if (!arrive_last(cb, CB, scratch)) return;
    // fixed-order combine: 16 groups of 16 threads merge the chunks i = grp, grp+16, ... in order, then the 16 group results
    // are merged in group order -- the result does not depend on which block happened to be last
    __shared__ float s_g[16][16][3];
    const int grp = threadIdx.x >> 4, c = threadIdx.x & 15;
    if (grp < 16) {
        float vm[DRC_BN_MAX_CHUNKS / 16], v2[DRC_BN_MAX_CHUNKS / 16];      // plain loads after the fence, all in flight (see finish)
#pragma unroll
        for (int it = 0; it < DRC_BN_MAX_CHUNKS / 16; ++it) {
            const unsigned i = grp + it * 16;
            vm[it] = i < gridDim.x ? part[((size_t)i * CB + cb) * 32 + c] : 0.f;
            v2[it] = i < gridDim.x ? part[((size_t)i * CB + cb) * 32 + 16 + c] : 0.f;
        }
        Stat acc = {0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < DRC_BN_MAX_CHUNKS / 16; ++it) {
            const unsigned i = grp + it * 16;
            if (i < gridDim.x) acc = merge(acc, Stat{n_of(i), vm[it], v2[it]});
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
    rearm(cb, CB, scratch);
}

}  // namespace drc_det
