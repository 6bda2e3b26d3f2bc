// tap2d.hip -- Conv2d(k3, stride 1, dilation d, pad d) (+BN/bias, +residual, +ReLU) with the explicit-wait protocol (gfx950).
//
//   reference: the 3x3 convolutions of the PSMNet feature CNN (submodule.py:60-139, dilation 1 and 2) and of ResNet-50-FPN
//   (backbone/resnet.py Bottleneck conv2, backbone/fpn.py layer blocks); also their stride-1 data gradients.
//
// Same arithmetic and layouts as tapconv.hip's 1x3x3 class, restructured like tapdown.hip: a wave owns R x WT output pixels and
// CT*16 output channels; a phase is one 8-channel half of an input block: the (R+2d) x (WT+2d) input tile is LDS-DMA'd as one
// dense block (rows packed back to back, one or two 1 KiB pieces per tap step, static count), the 9 taps are LDS offsets,
// weights are uncounted raw loads issued before the step's pieces, the consuming step waits with vmcnt(#pieces).  Register
// budget 256 -> two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#define DN_WAVES 4
#define DN_MAXP 18   /* LDS-DMA pieces per phase tile: one or two per tap step */

namespace {

template <int VT, int CT>
__global__ __launch_bounds__(64 * DN_WAVES, 2) void tap2d_kernel(const drc_tapconv_params p) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;

    const int OH = p.OH, OW = p.OW;
    const int n_wt = (OW + p.WT - 1) / p.WT;
    const int n_rt = (OH + p.R - 1) / p.R;
    const int per_cg = p.N * n_rt * n_wt;            // groups of one cout group: (n, row tile, col tile)
    const long groups = (long)(p.cout_pad / 16 / CT) * per_cg;
    const long workers = (long)gridDim.x * DN_WAVES;
    const long wid = (long)blockIdx.x * DN_WAVES + wave;
    long gcur = groups * wid / workers;              // equal contiguous shares
    const long gend = groups * (wid + 1) / workers;
    if (gcur >= gend) return;                        // wave-uniform; no workgroup barrier in this kernel

    const int dil = p.cls[0].sh;                     // tap spacing (dilation), same in h and w
    const int rows_in = p.R + 2 * dil, cols_in = p.WT + 2 * dil;
    const int vox_total = rows_in * cols_in;
    const int units = vox_total * 2;                 // 16-byte units (8 channels x 4 B = 2 units per voxel)
    const int pieces = (units + 63) >> 6;
    const bool two_pieces = pieces > 9;              // 9 tap steps per phase, one or two pieces per step
    const int buf_floats = (two_pieces ? 18 : 9) * 256;   // room for the surplus steps' (duplicate) pieces behind the tile
    float* lds = lds_all + wave * (2 * buf_floats);  // double-buffered
    const int nslots = p.R * p.WT;
    const unsigned mag = ((1u << 20) + cols_in - 1) / cols_in;

    int lane_off[VT];                                // B-fragment offsets (floats) of tap (0,0)
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int s = vt * 16 + j;
        int r = s / p.WT, c = s - r * p.WT;
        if (s >= nslots) { r = 0; c = 0; }
        lane_off[vt] = (r * cols_in + c) * 8 + g * 2;
    }

    const int n_ph = p.cb_in * 2;                    // phases of a group: (channel block, half)
    const int64_t w_half_stride = (int64_t)p.cout_pad * 8;
    const unsigned hs32 = (unsigned)(w_half_stride * 4), ts32 = hs32 * (unsigned)n_ph;   // bytes: per (cb,half), per tap

    struct Group { int n, oh0, ow0, ct0, nr, nc; const float* base; };
    auto decode = [&](long gidx) __attribute__((always_inline)) -> Group {
        Group q;
        int r = (int)(gidx % per_cg);
        q.ct0 = (int)(gidx / per_cg) * CT;
        const int wt = r % n_wt; r /= n_wt;
        const int rt = r % n_rt;
        q.n = r / n_rt;
        q.oh0 = rt * p.R; q.ow0 = wt * p.WT;
        q.nr = OH - q.oh0 < p.R ? OH - q.oh0 : p.R;           // valid output rows / columns of a ragged tile
        q.nc = OW - q.ow0 < p.WT ? OW - q.ow0 : p.WT;
        q.base = p.x + (int64_t)q.n * p.x_n_stride + (int64_t)(q.oh0 + p.cls[0].dh0) * p.x_h_stride + (int64_t)(q.ow0 + p.cls[0].dw0) * 16;
        return q;
    };
    auto phase_base = [&](const Group& G, int ph) __attribute__((always_inline)) -> const float* {
        return G.base + (int64_t)(ph >> 1) * p.x_cb_stride + (ph & 1) * 8;
    };

    // per-lane byte offsets of the LDS-DMA pieces (all lanes active; lanes past the tile and rows / columns past a ragged
    // tile's last needed input re-read a valid neighbour); recomputed only when the staged group's raggedness changes
    unsigned poff[DN_MAXP];
    int poff_nr = -1, poff_nc = -1;
    auto set_poff = [&](int nr, int nc) __attribute__((always_inline)) {
        poff_nr = nr; poff_nc = nc;
#pragma unroll
        for (int q = 0; q < DN_MAXP; ++q) {
            int u = q * 64 + lane;
            u = u < units ? u : units - 1;
            const int v = u >> 1;
            int r = (int)(((unsigned)v * mag) >> 20);
            int m = v - r * cols_in;
            const int rmax = nr - 1 + 2 * dil, mmax = nc - 1 + 2 * dil;    // last input row / column a valid output needs
            r = r < rmax ? r : rmax;
            m = m < mmax ? m : mmax;
            poff[q] = (unsigned)((r * (int)p.x_h_stride + m * 16 + (u & 1) * 4) * 4);
            __builtin_amdgcn_sched_barrier(0);       // one piece at a time
        }
    };
#define DN_STAGE(SBASE, Q, BUFI) \
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR((const char*)(SBASE) + poff[Q]), LDS_PTR(lds + (BUFI) * buf_floats + (Q) * 256), 16, 0, 0)

    // weights: packed [widx = kh*3+kw][cb*2+half][cout_pad][8]; raw asm loads (not counted by the compiler)
    unsigned wlane_off;
    auto load_w = [&](f32x2 (&Wd)[CT], int t, unsigned wph) __attribute__((always_inline)) {
        const unsigned vo = wlane_off + ((unsigned)t * ts32 + wph);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if (ct == 0) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(Wd[0]) : "v"(vo), "s"(p.w));
            if (ct == 1) asm volatile("global_load_dwordx2 %0, %1, %2 offset:512" : "=v"(Wd[CT > 1 ? 1 : 0]) : "v"(vo), "s"(p.w));
            if (ct == 2) asm volatile("global_load_dwordx2 %0, %1, %2 offset:1024" : "=v"(Wd[CT > 2 ? 2 : 0]) : "v"(vo), "s"(p.w));
            if (ct == 3) asm volatile("global_load_dwordx2 %0, %1, %2 offset:1536" : "=v"(Wd[CT > 3 ? 3 : 0]) : "v"(vo), "s"(p.w));
        }
    };
    auto w_phase = [&](int ph) __attribute__((always_inline)) -> unsigned { return (unsigned)ph * hs32; };   // byte offset of phase ph's tap 0

#define DN_CLEAR_ACC()                                                                                 \
    {                                                                                                  \
        float z_;                                                                                      \
        asm volatile("v_mov_b32 %0, 0" : "=v"(z_));                                                    \
        const f32x4 z4_ = {z_, z_, z_, z_};                                                            \
        _Pragma("unroll") for (int vt = 0; vt < VT; ++vt)                                              \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) acc[vt][ct] = z4_;                       \
    }
    f32x4 acc[VT][CT];
    DN_CLEAR_ACC()

    f32x2 wbuf[2][CT], bfr[2][VT];
    int bufsel = 0;

    Group cur = decode(gcur);
    set_poff(cur.nr, cur.nc);
    {
        const float* sb0 = phase_base(cur, 0);
#pragma unroll
        for (int q = 0; q < DN_MAXP; ++q)
            if (q < pieces) DN_STAGE(sb0, q, 0);
    }
    wlane_off = (unsigned)(((cur.ct0 * 16 + j) * 8 + g * 2) * 4);
    load_w(wbuf[0], 0, w_phase(0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // B fragments of tap T = kh*3+kw: tile offset (kh*dil, kw*dil)
#define DN_LOAD_B(T, DST)                                                                              \
    {                                                                                                  \
        constexpr int kh_ = (T) / 3, kw_ = (T) % 3;                                                    \
        int so_ = (kh_ * dil * cols_in + kw_ * dil) * 8;                                               \
        asm volatile("" : "+s"(so_));                                                                  \
        const float* bp_ = buf + so_;                                                                  \
        _Pragma("unroll") for (int vt = 0; vt < VT; ++vt) DST[vt] = *(const f32x2*)(bp_ + lane_off[vt]); \
    }
#define DN_MFMA(V0, V1, K)                                                                             \
    _Pragma("unroll") for (int vt = (V0); vt < (V1); ++vt)                                             \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                              \
            acc[vt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wbuf[sel_][ct][K], bfr[sel_][vt][K], acc[vt][ct], 0, 0, 0);

#define DN_STEP(T)                                                                                     \
    {                                                                                                  \
        constexpr int sel_ = (T) & 1;                                                                  \
        if ((T) > 0) {                                                                                 \
            if (two_pieces) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                           \
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                      \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        DN_MFMA(0, 1, 0)                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if ((T) < 8) load_w(wbuf[sel_ ^ 1], (T) + 1, wph); else load_w(wbuf[sel_ ^ 1], 0, wph_nx);     \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        DN_MFMA(1, VT, 0)                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (two_pieces) {   /* surplus steps re-stage an earlier piece: the count per step stays static */ \
            DN_STAGE(sb_nx, 2 * (T), bufsel ^ 1);                                                      \
            DN_STAGE(sb_nx, 2 * (T) + 1, bufsel ^ 1);                                                  \
        } else {                                                                                       \
            DN_STAGE(sb_nx, (T), bufsel ^ 1);                                                          \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        DN_MFMA(0, 1, 1)                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if ((T) < 8) DN_LOAD_B((T) + 1, bfr[sel_ ^ 1])                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        DN_MFMA(1, VT, 1)                                                                              \
    }

#pragma unroll 1
    for (;;) {
        const bool has_next_group = gcur + 1 < gend;
        Group nxg = cur;
        for (int ph = 0; ph < n_ph; ++ph) {
            const bool last_ph = ph + 1 == n_ph;
            if (last_ph && has_next_group) {
                nxg = decode(gcur + 1);
                if (nxg.nr != poff_nr || nxg.nc != poff_nc) set_poff(nxg.nr, nxg.nc);
            }
            // what the next phase reads: same group's next phase, the next group's phase 0, or (very last phase) a harmless re-stage
            const float* sb_nx = last_ph ? phase_base(nxg, 0) : phase_base(cur, ph + 1);
            const unsigned wph = w_phase(ph), wph_nx = last_ph ? w_phase(0) : w_phase(ph + 1);
            const float* buf = lds + bufsel * buf_floats;
            DN_LOAD_B(0, bfr[0])
            DN_STEP(0) DN_STEP(1) DN_STEP(2) DN_STEP(3) DN_STEP(4) DN_STEP(5) DN_STEP(6) DN_STEP(7) DN_STEP(8)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wbuf[0][ct] = wbuf[1][ct];      // 9 steps: the next phase's first weights sit in set 1
            bufsel ^= 1;
        }

        // ---- epilogue: folded BN, residual, ReLU, store; clear the accumulators
        {
            f32x4 bn_sc[CT], bn_sh[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                bn_sc[ct] = *(const f32x4*)(p.scale + (cur.ct0 + ct) * 16 + g * 4);
                bn_sh[ct] = *(const f32x4*)(p.shift + (cur.ct0 + ct) * 16 + g * 4);
            }
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                const int s = vt * 16 + j;
                const int r = s / p.WT, c = s - r * p.WT;
                const bool valid = (s < nslots) && (r < cur.nr) && (c < cur.nc);
                if (valid) {
                    const int64_t yo = p.y_off0 + (int64_t)cur.n * p.y_n_stride + (int64_t)(cur.oh0 + r) * p.y_h_stride +
                                       (int64_t)(cur.ow0 + c) * 16 + g * 4;
                    const int64_t ro = p.r_off0 + (int64_t)cur.n * p.r_n_stride + (int64_t)(cur.oh0 + r) * p.r_h_stride +
                                       (int64_t)(cur.ow0 + c) * 16 + g * 4;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        f32x4 v = acc[vt][ct] * bn_sc[ct] + bn_sh[ct];
                        if (p.res) v += *(const f32x4*)(p.res + ro + (int64_t)(cur.ct0 + ct) * p.r_cb_stride);
                        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        *(f32x4*)(p.y + yo + (int64_t)(cur.ct0 + ct) * p.y_cb_stride) = v;
                    }
                }
            }
            DN_CLEAR_ACC()
        }
        if (!has_next_group) break;
        ++gcur;
        cur = nxg;
        const unsigned wl = (unsigned)(((cur.ct0 * 16 + j) * 8 + g * 2) * 4);
        if (wl != wlane_off) {   // next cout group: the prefetched step-0 weights used the old lane offset
            wlane_off = wl;
            load_w(wbuf[0], 0, w_phase(0));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
#undef DN_STEP
#undef DN_MFMA
#undef DN_LOAD_B
#undef DN_STAGE
#undef DN_CLEAR_ACC
}

template <int VT, int CT>
int launch(const drc_tapconv_params& p, hipStream_t stream) {
    const int dil = p.cls[0].sh;
    const int vox = (p.R + 2 * dil) * (p.WT + 2 * dil);
    const int pieces = (vox * 2 + 63) / 64;
    const size_t lds = (size_t)2 * (pieces > 9 ? 18 : 9) * 1024 * DN_WAVES;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)tap2d_kernel<VT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    static int occ_blocks = 0;
    static size_t occ_lds = 0;
    if (!occ_blocks || occ_lds != lds) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tap2d_kernel<VT, CT>, 64 * DN_WAVES, lds) != hipSuccess || nb < 1) nb = 1;
        occ_blocks = nb; occ_lds = lds;
    }
    const long groups = (long)(p.cout_pad / 16 / CT) * p.N * ((p.OH + p.R - 1) / p.R) * ((p.OW + p.WT - 1) / p.WT);
    long workers = 256L * DN_WAVES * occ_blocks;
    if (workers > groups) workers = groups;
    dim3 grid((unsigned)((workers + DN_WAVES - 1) / DN_WAVES), 1, 1);
    hipLaunchKernelGGL((tap2d_kernel<VT, CT>), grid, dim3(64 * DN_WAVES), lds, stream, p);
    return (int)hipGetLastError();
}

template <int CT>
int launch_vt(int nvt, const drc_tapconv_params& p, hipStream_t s) {
    switch (nvt) {
        case 1: return launch<1, CT>(p, s);
        case 2: return launch<2, CT>(p, s);
        case 3: return launch<3, CT>(p, s);
        case 4: return launch<4, CT>(p, s);
        case 5: return launch<5, CT>(p, s);
        case 6: return launch<6, CT>(p, s);
        case 7: return launch<7, CT>(p, s);
    }
    return -3;
}

}  // namespace

extern "C" int drc_conv2d_k3_fwd(const drc_tapconv_params* pp, int cout_tiles_per_wave, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD != 1 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 1 || p.out_mul != 1 || k.nd != 1 || k.nh != 3 || k.nw != 3 || k.sh < 1 || k.sh != k.sw ||
        k.wbase != 0 || k.wsh != 3 || k.wsw != 1)
        return -4;
    if (p.R <= 0 || p.WT <= 0 || p.R * p.WT > 112) return -3;
    const int vox = (p.R + 2 * k.sh) * (p.WT + 2 * k.sh);
    const int pieces = (vox * 2 + 63) / 64;
    if (pieces > DN_MAXP || (long)vox * (p.WT + 2 * k.sh) >= (1L << 20)) return -5;
    const int ct = p.cout_pad / 16, CT = cout_tiles_per_wave;
    if ((CT != 1 && CT != 2 && CT != 4) || ct % CT) return -2;
    const int nvt = (p.R * p.WT + 15) / 16;
    if (nvt * CT > 28) return -3;                    // VT*CT accumulator tiles of 4 registers
    hipStream_t s = (hipStream_t)stream;
    return CT == 4 ? launch_vt<4>(nvt, p, s) : CT == 2 ? launch_vt<2>(nvt, p, s) : launch_vt<1>(nvt, p, s);
}
