// blocked_walk.h -- geometry of channel-blocked tensors and a row-wise walker for the elementwise / reduction kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace drc_blk {

struct BlkGeom { int N, CB, D, H, W, pd, ph, pw, cb_total, cb_off; };   // CB blocks [cb_off, cb_off+CB) of cb_total

__device__ __forceinline__ long blk_off(const BlkGeom& g, int n, int cb, int d, int y, int x) {
    const long Wp = g.W + 2 * g.pw, Hp = g.H + 2 * g.ph, Dp = g.D + 2 * g.pd;
    return ((((long)n * g.cb_total + g.cb_off + cb) * Dp + (d + g.pd)) * Hp + (y + g.ph)) * Wp * 16 + (long)(x + g.pw) * 16;
}

// Every block of the launch (gridDim.x blocks per channel block) takes one contiguous run of interior rows (n, d, y) and
// calls f(n, d, y, x, q) for each float4 quad q of each voxel x of its rows.  Thread t owns quad (t & 3) and one or more
// fixed x positions, so the row index advances by increments (no 64-bit division per element) and a warp reads whole rows:
// W * 64 contiguous bytes.  The assignment depends only on the geometry and the grid, never on timing.
template <int THREADS, class F>
__device__ __forceinline__ void walk_rows(const BlkGeom& g, F f) {
    const int tpr = g.W * 4;                                   // threads per row
    const int rpb = tpr >= THREADS ? 1 : THREADS / tpr;        // rows in flight per block
    const int rows = g.N * g.D * g.H;
    int chunk = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
    chunk = (chunk + rpb - 1) / rpb * rpb;
    const int r0 = (int)blockIdx.x * chunk;
    const int r1 = r0 + chunk < rows ? r0 + chunk : rows;
    const int rsub = (int)threadIdx.x / tpr;
    const int xq0 = (int)threadIdx.x - rsub * tpr;
    if (rsub >= rpb) return;
    int row = r0 + rsub;
    if (row >= r1) return;
    int y = row % g.H, t = row / g.H;
    int d = t % g.D, n = t / g.D;
    auto next = [&]() {
        y += rpb;
        while (y >= g.H) { y -= g.H; if (++d == g.D) { d = 0; ++n; } }
    };
    // four rows per trip: the calls sit back to back in straight-line code, so their loads are in flight together (one load per
    // trip leaves a streaming reduction latency-bound).  The call order is FIXED but not plain row order: a thread that owns several
    // xq of a row (tpr > THREADS) visits rows 0..3 of the trip for its first xq, then rows 0..3 for the next one -- the streaming
    // reductions built on this walker (bn_stats, bn_bwd_reduce) are deterministic and their goldens were recorded with this order
    for (; row + 3 * rpb < r1; row += 4 * rpb) {
        const int n0 = n, d0 = d, y0 = y; next();
        const int n1 = n, d1 = d, y1 = y; next();
        const int n2 = n, d2 = d, y2 = y; next();
        const int n3 = n, d3 = d, y3 = y; next();
        for (int xq = xq0; xq < tpr; xq += THREADS) {
            f(n0, d0, y0, xq >> 2, xq & 3);
            f(n1, d1, y1, xq >> 2, xq & 3);
            f(n2, d2, y2, xq >> 2, xq & 3);
            f(n3, d3, y3, xq >> 2, xq & 3);
        }
    }
    for (; row < r1; row += rpb) {
        for (int xq = xq0; xq < tpr; xq += THREADS) f(n, d, y, xq >> 2, xq & 3);
        next();
    }
}

// voxels (per channel) that block `bx` of a `nblocks`-block launch visits in walk_rows
__device__ __forceinline__ float rows_of_block(const BlkGeom& g, int threads, unsigned bx, unsigned nblocks) {
    const int tpr = g.W * 4;
    const int rpb = tpr >= threads ? 1 : threads / tpr;
    const int rows = g.N * g.D * g.H;
    int chunk = (rows + (int)nblocks - 1) / (int)nblocks;
    chunk = (chunk + rpb - 1) / rpb * rpb;
    const long r0 = (long)bx * chunk;
    long r1 = r0 + chunk < rows ? r0 + chunk : rows;
    return r1 > r0 ? (float)(r1 - r0) * (float)g.W : 0.f;
}

}  // namespace drc_blk
