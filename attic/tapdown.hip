// tapdown.hip -- Conv3d(k3, stride 2, pad 1) (+BN, +residual, +ReLU) on parity-split LDS tiles (gfx950 / CDNA4).
//
//   reference: hourglass.conv1 / conv3 (stackhourglass.py:9-16) and, transposed, the data gradient of the
//   ConvTranspose3d layers (conv5 / conv6) in training.
//
// In padded input coordinates an output voxel o reads i = 2o + k (k = 0,1,2) per dimension: taps k = 0 and k = 2 hit EVEN
// input positions (2o and 2(o+1)), tap k = 1 the ODD position 2o+1.  A wave owns R x WT output voxels of one output slice and
// CT*16 output channels.  Per phase (one input slice 2od+kd, one 8-channel half) it LDS-DMAs the four (row parity, column
// parity) planes of the input rows it needs as four DENSE unit-stride tiles -- a DMA lane can fetch from any address, so the
// de-interleave is free -- and every tap (kh, kw) becomes a unit-stride B fragment of plane (kh&1, kw&1) at offset
// (kh>>1, kw>>1): conflict-free ds_read_b64, where the stride-2 gather of the generic kernel is a 4-way bank conflict.
// All output channels of a 64-channel layer sit in one wave (CT = 4): a staged tile feeds 9 * VT * CT * 2 MFMAs.
// Waits follow tapslide.hip: weights are uncounted raw loads issued before the step's LDS-DMA piece(s); the consuming step
// waits with vmcnt(#pieces); the piece count per step is static (surplus steps re-stage a piece).
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
__global__ __launch_bounds__(64 * DN_WAVES, 2) void tapdown_kernel(const drc_tapconv_params p) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;

    const int OD = p.OD, OH = p.OH, OW = p.OW;
    const int n_wt = (OW + p.WT - 1) / p.WT;
    const int n_rt = (OH + p.R - 1) / p.R;
    const int per_cg = p.N * OD * n_rt * n_wt;       // groups of one cout group: (n, od, row tile, col tile)
    const long groups = (long)(p.cout_pad / 16 / CT) * per_cg;
    const long workers = (long)gridDim.x * DN_WAVES;
    const long wid = (long)blockIdx.x * DN_WAVES + wave;
    long gcur = groups * wid / workers;              // equal contiguous shares
    const long gend = groups * (wid + 1) / workers;
    if (gcur >= gend) return;                        // wave-uniform; no workgroup barrier in this kernel

    // the four parity planes of a phase tile, dense and back to back: (ph,pw) = (0,0) (0,1) (1,0) (1,1)
    const int ce = p.WT + 1, co = p.WT;              // columns of an even / odd column-parity plane
    const int n0 = (p.R + 1) * ce, n1 = (p.R + 1) * co, n2 = p.R * ce, n3 = p.R * co;
    const int vox_total = n0 + n1 + n2 + n3;
    const int units = vox_total * 2;                 // 16-byte units (8 channels x 4 B = 2 units per voxel)
    const int pieces = (units + 63) >> 6;
    const bool two_pieces = pieces > 9;              // 9 tap steps per phase, one or two pieces per step
    const int buf_floats = (two_pieces ? 18 : 9) * 256;   // room for the surplus steps' (duplicate) pieces behind the tile
    float* lds = lds_all + wave * (2 * buf_floats);  // double-buffered
    const int nslots = p.R * p.WT;
    const unsigned mag_e = ((1u << 20) + ce - 1) / ce, mag_o = ((1u << 20) + co - 1) / co;

    int lane_off_e[VT], lane_off_o[VT];              // B-fragment offsets (floats) inside an even / odd column-parity plane
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int s = vt * 16 + j;
        int r = s / p.WT, c = s - r * p.WT;
        if (s >= nslots) { r = 0; c = 0; }
        lane_off_e[vt] = (r * ce + c) * 8 + g * 2;
        lane_off_o[vt] = (r * co + c) * 8 + g * 2;
    }

    const int n_pc = p.cb_in * 2;                    // (channel block, half)
    const int n_ph = n_pc * 3;                       // phases of a group: pc outer, depth tap kd inner
    const int64_t w_half_stride = (int64_t)p.cout_pad * 8;
    const unsigned hs32 = (unsigned)(w_half_stride * 4), ts32 = hs32 * (unsigned)n_pc;   // bytes: per (cb,half), per tap

    struct Group { int n, od, oh0, ow0, ct0, nr, nc; const float* base; };
    auto decode = [&](long gidx) __attribute__((always_inline)) -> Group {
        Group q;
        int r = (int)(gidx % per_cg);
        q.ct0 = (int)(gidx / per_cg) * CT;
        const int wt = r % n_wt; r /= n_wt;
        const int rt = r % n_rt; r /= n_rt;
        q.od = r % OD; q.n = r / OD;
        q.oh0 = rt * p.R; q.ow0 = wt * p.WT;
        q.nr = OH - q.oh0 < p.R ? OH - q.oh0 : p.R;           // valid output rows / columns of a ragged tile
        q.nc = OW - q.ow0 < p.WT ? OW - q.ow0 : p.WT;
        q.base = p.x + (int64_t)q.n * p.x_n_stride + (int64_t)(2 * q.od + p.cls[0].dd0) * p.x_d_stride +
                 (int64_t)(2 * q.oh0 + p.cls[0].dh0) * p.x_h_stride + (int64_t)(2 * q.ow0 + p.cls[0].dw0) * 16;
        return q;
    };
    auto phase_base = [&](const Group& G, int ph) __attribute__((always_inline)) -> const float* {   // ph = pc*3 + kd
        const int pc = ph / 3, kd = ph - pc * 3;
        return G.base + (int64_t)(pc >> 1) * p.x_cb_stride + (pc & 1) * 8 + (int64_t)kd * p.x_d_stride;
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
            int v = u >> 1;
            int ph_, pw_, loc;
            if (v < n0) { ph_ = 0; pw_ = 0; loc = v; }
            else if (v < n0 + n1) { ph_ = 0; pw_ = 1; loc = v - n0; }
            else if (v < n0 + n1 + n2) { ph_ = 1; pw_ = 0; loc = v - n0 - n1; }
            else { ph_ = 1; pw_ = 1; loc = v - n0 - n1 - n2; }
            const int cols = pw_ ? co : ce;
            int r = (int)(((unsigned)loc * (pw_ ? mag_o : mag_e)) >> 20);
            int m = loc - r * cols;
            const int rmax = ph_ ? (nr > 0 ? nr - 1 : 0) : nr, mmax = pw_ ? (nc > 0 ? nc - 1 : 0) : nc;
            r = r < rmax ? r : rmax;
            m = m < mmax ? m : mmax;
            poff[q] = (unsigned)(((2 * r + ph_) * (int)p.x_h_stride + (2 * m + pw_) * 16 + (u & 1) * 4) * 4);
            __builtin_amdgcn_sched_barrier(0);       // one piece at a time: 18 interleaved chains would cost ~100 live registers
        }
    };
#define DN_STAGE(SBASE, Q, BUFI) \
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR((const char*)(SBASE) + poff[Q]), LDS_PTR(lds + (BUFI) * buf_floats + (Q) * 256), 16, 0, 0)

    // weights: packed [widx = (kd*3+kh)*3+kw][cb*2+half][cout_pad][8]; raw asm loads (not counted by the compiler)
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
    auto w_phase = [&](int ph) __attribute__((always_inline)) -> unsigned {   // byte offset of phase ph's tap 0: (kd*9) taps + pc halves
        const int pc = ph / 3, kd = ph - pc * 3;
        return (unsigned)(kd * 9) * ts32 + (unsigned)pc * hs32;
    };

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

    // B fragments of tap T = kh*3+kw: plane (kh&1, kw&1) at offset (kh>>1, kw>>1)
#define DN_LOAD_B(T, DST)                                                                              \
    {                                                                                                  \
        constexpr int kh_ = (T) / 3, kw_ = (T) % 3;                                                    \
        constexpr int pl_ = (kh_ & 1) * 2 + (kw_ & 1);                                                 \
        int so_ = (pl_ == 0 ? 0 : pl_ == 1 ? n0 : pl_ == 2 ? n0 + n1 : n0 + n1 + n2) * 8 + ((kh_ >> 1) * ((kw_ & 1) ? co : ce) + (kw_ >> 1)) * 8; \
        asm volatile("" : "+s"(so_));                                                                  \
        const float* bp_ = buf + so_;                                                                  \
        _Pragma("unroll") for (int vt = 0; vt < VT; ++vt) DST[vt] = *(const f32x2*)(bp_ + ((kw_ & 1) ? lane_off_o[vt] : lane_off_e[vt])); \
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
                    const int64_t yo = p.y_off0 + (int64_t)cur.n * p.y_n_stride + (int64_t)cur.od * p.y_d_stride +
                                       (int64_t)(cur.oh0 + r) * p.y_h_stride + (int64_t)(cur.ow0 + c) * 16 + g * 4;
                    const int64_t ro = p.r_off0 + (int64_t)cur.n * p.r_n_stride + (int64_t)cur.od * p.r_d_stride +
                                       (int64_t)(cur.oh0 + r) * p.r_h_stride + (int64_t)(cur.ow0 + c) * 16 + g * 4;
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
    const int vox = (p.R + 1) * (p.WT + 1) + (p.R + 1) * p.WT + p.R * (p.WT + 1) + p.R * p.WT;
    const int pieces = (vox * 2 + 63) / 64;
    const size_t lds = (size_t)2 * (pieces > 9 ? 18 : 9) * 1024 * DN_WAVES;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)tapdown_kernel<VT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    static int occ_blocks = 0;
    static size_t occ_lds = 0;
    if (!occ_blocks || occ_lds != lds) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tapdown_kernel<VT, CT>, 64 * DN_WAVES, lds) != hipSuccess || nb < 1) nb = 1;
        occ_blocks = nb; occ_lds = lds;
    }
    const long groups = (long)(p.cout_pad / 16 / CT) * p.N * p.OD * ((p.OH + p.R - 1) / p.R) * ((p.OW + p.WT - 1) / p.WT);
    long workers = 256L * DN_WAVES * occ_blocks;
    if (workers > groups) workers = groups;
    dim3 grid((unsigned)((workers + DN_WAVES - 1) / DN_WAVES), 1, 1);
    hipLaunchKernelGGL((tapdown_kernel<VT, CT>), grid, dim3(64 * DN_WAVES), lds, stream, p);
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

extern "C" int drc_conv3d_k3s2_fwd(const drc_tapconv_params* pp, int cout_tiles_per_wave, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 2 || p.out_mul != 1 || k.nd != 3 || k.nh != 3 || k.nw != 3 || k.sd != 1 || k.sh != 1 ||
        k.sw != 1 || k.wbase != 0 || k.wsd != 9 || k.wsh != 3 || k.wsw != 1)
        return -4;
    if (p.R <= 0 || p.WT <= 0 || p.R * p.WT > 112) return -3;
    const int vox = (p.R + 1) * (p.WT + 1) + (p.R + 1) * p.WT + p.R * (p.WT + 1) + p.R * p.WT;
    const int pieces = (vox * 2 + 63) / 64;
    if (pieces > DN_MAXP || (long)vox * (p.WT + 1) >= (1L << 20) || (size_t)2 * (pieces > 9 ? 18 : 9) * 1024 * DN_WAVES > 160 * 1024) return -5;
    const int ct = p.cout_pad / 16, CT = cout_tiles_per_wave;
    if ((CT != 1 && CT != 2 && CT != 4) || ct % CT) return -2;
    const int nvt = (p.R * p.WT + 15) / 16;
    if (nvt * CT > 28) return -3;                    // VT*CT accumulator tiles of 4 registers
    hipStream_t s = (hipStream_t)stream;
    return CT == 4 ? launch_vt<4>(nvt, p, s) : CT == 2 ? launch_vt<2>(nvt, p, s) : launch_vt<1>(nvt, p, s);
}
