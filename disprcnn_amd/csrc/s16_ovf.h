// s16_ovf.h -- range guard of the split-f16 path (round 6).
//
// An RS16 value is hi = fp16(v), lo = fp16(v - hi): representable while |v| <= 65504.  The reference computes in fp32 (config/defaults.py:22,
// submodule.py:19-22) and has no such limit, so every kernel that WRITES split-f16 values (the conv epilogues of convs16*.hip, the converters
// of s16_ops.hip, the RS16 epilogue of deconvdirect.hip) reports a value it had to clamp -- |v| > 65504, Inf or NaN -- by OR-ing 1 into a
// caller-owned device word (drc_s16conv_params.ovf / the converters' `ovf` argument; NULL: no report).  The host reads the word once per
// forward pass and re-runs on the fp32 kernels ("auto") or raises (regressor_math / feature_math = "f16x2"): engine.OverflowGuard.
//
// What is tested is the value AFTER the clamp: |v| >= 65504 there means it was clamped (or sat exactly on the limit) -- a negative
// pre-activation that a ReLU zeroes anyway does not count.  NaN cannot reach a clamp's output (v_med3_f32 / v_min / v_max return a finite
// operand), so the places NaN / Inf can ENTER are tested on their own: the converters test their fp32 input before clamping, the conv kernels
// test the folded BN scale / shift they load once per workgroup (activations and weights are stored finite; an fp32 accumulator of
// <= 64 x 27 products of halfs cannot overflow).
// Cost: one vector compare per value (or per running maximum of a group of values: see_max) and a scalar OR into the wave's running lane
// mask -- no vector register lives across the kernel (a per-lane running maximum cost two VGPRs and pushed the fused-head kernel, which
// sits at 502 of 512, into spills) -- plus one test and at most one atomic per wave at the end of the kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct S16Ovf {
    unsigned long long m;                 // wave-uniform: the lanes that met a value out of range in any step so far
    __device__ __forceinline__ S16Ovf() : m(0ull) {}
    // the lane mask of the values a launch stores: idle lanes / dropped planes hold over-read data, not voxels of the map
    static __device__ __forceinline__ unsigned long long lanes(bool ok) { return __builtin_amdgcn_ballot_w64(ok); }
    // a value after its clamp to [-65504 | 0, 65504]
    __device__ __forceinline__ void see(float clamped, unsigned long long keep = ~0ull) {
        m |= __builtin_amdgcn_ballot_w64(__builtin_fabsf(clamped) >= 65504.f) & keep;
    }
    // the largest |clamped value| of a group (the caller folds the group with fmaxf: v_max3_f32 chains)
    __device__ __forceinline__ void see_max(float mx, unsigned long long keep = ~0ull) {
        m |= __builtin_amdgcn_ballot_w64(mx >= 65504.f) & keep;
    }
    // an fp32 input value before its clamp (converters), a folded BN scale / shift: NaN and Inf count as well
    __device__ __forceinline__ void see_raw(float x, float limit = 65504.f) {
        m |= __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(x) <= limit));
    }
    __device__ __forceinline__ void flush(uint32_t* ovf) const {
        if (ovf != nullptr && m != 0ull && (threadIdx.x & 63u) == 0u) atomicOr(ovf, 1u);
    }
};
