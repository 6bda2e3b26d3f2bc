// Variant library source: the brick-staged Winograd kernel (wino3d_brick.hip) in front of the register-patch kernel without LDS
// parking (wino3d_nz.hip), which stays the fallback and keeps serving the fused-cost-volume / weight-packing entry points.
#define drc_conv3d_k3_wino_fwd drc_conv3d_k3_wino_fwd_regpatch
#include "wino3d_nz.hip"
#undef drc_conv3d_k3_wino_fwd
#include "wino3d_brick.hip"

namespace {
template <int VPR>
int launch_brick(const drc_tapconv_params& p, hipStream_t stream) {
    constexpr int CT = 2;
    const int TH = p.OH / 2, TW = p.OW / 2;
    const int nrows_max = (TW - 1 + 64 + TW - 1) / TW;
    const int nslabs_max = (TH - 1 + nrows_max + TH - 1) / TH;
    if (nslabs_max > 3) return 1;
    const int SLOTS = 2 * nrows_max + 2 * nslabs_max;
    const int cps = VPR / 2 > 8 ? 2 : 1;
    if (SLOTS * cps > 32) return 1;
    const size_t lds = (size_t)2 * 8 * CT * 256 * 4 + (size_t)4 * SLOTS * cps * 1024;
    if (lds > 160 * 1024) return 1;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)brick::wino3d_brick_kernel<CT, VPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long tiles = (long)p.N * (p.OD / 2) * TH * TW;
    const long groups = (tiles + 15) / 16;
    const int n_cg = p.cout_pad / 16 / CT;
    long per_cg = 256 / n_cg;
    const long need = (groups + WB_WAVES - 1) / WB_WAVES;
    if (per_cg > need) per_cg = need;
    if (per_cg < 1) per_cg = 1;
    hipLaunchKernelGGL((brick::wino3d_brick_kernel<CT, VPR>), dim3((unsigned)(per_cg * n_cg)), dim3(64 * WB_WAVES), lds, stream, p, SLOTS);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
}  // namespace

extern "C" int drc_conv3d_k3_wino_fwd(const drc_tapconv_params* pp, int cout_tiles_per_wave, void* stream) {
    if (pp && cout_tiles_per_wave == 2 && !getenv("WB_OFF")) {
        const drc_tapconv_params& p = *pp;
        const drc_tap_class& k = p.cls[0];
        const bool plain = p.x && p.n_classes == 1 && p.in_mul == 1 && p.out_mul == 1 && k.nd == 3 && k.nh == 3 && k.nw == 3 && k.sd == 1 && k.sh == 1 &&
                           k.sw == 1 && k.dd0 == 0 && k.dh0 == 0 && k.dw0 == 0 && !((p.OD | p.OH | p.OW) & 1) && p.N > 0 && (p.cout_pad / 16) % 2 == 0 &&
                           (int64_t)p.N * p.x_n_stride * 4 < (1LL << 32);
        if (plain && p.x_h_stride == (int64_t)(p.OW + 2) * 16) {
            int st = 1;
            if (p.OW + 2 == 30) st = launch_brick<30>(p, (hipStream_t)stream);
            else if (p.OW + 2 == 16) st = launch_brick<16>(p, (hipStream_t)stream);
            if (st != 1) return st;
        }
    }
    return drc_conv3d_k3_wino_fwd_regpatch(pp, cout_tiles_per_wave, stream);
}
