// Shared helpers for the gfx950 kernels of libsvgattn.  CDNA4 only: wave64, MFMA 32x32x16, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/svg_attn.h"

namespace svg {

using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x4 = float __attribute__((ext_vector_type(4)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using bf16x4 = __bf16 __attribute__((ext_vector_type(4)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
using f16x4 = _Float16 __attribute__((ext_vector_type(4)));
using i16x4 = short __attribute__((ext_vector_type(4)));
using i16x8 = short __attribute__((ext_vector_type(8)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
using u32x2 = unsigned __attribute__((ext_vector_type(2)));

extern thread_local int g_last_hip_error;

inline int launch_status() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return SVG_ERR_LAUNCH;
    }
    return SVG_OK;
}

// Element-type traits: storage vectors + the MFMA builtin for that type.
template <typename T>
struct Elt;

template <>
struct Elt<__bf16> {
    using v8 = bf16x8;
    using v4 = bf16x4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    // D = A B + C with D and C in DIFFERENT registers, as inline asm: for a C that stays live hipcc selects the tied form (D == C) of
    // the builtin and copies C first (16 registers per call).  The caller owns the hazards (the compiler does not look inside):
    // see attn_body_pp2 / attn_f8.h mfma_qk_first and tools/asm_hazards.py --asm-mfma.
    static __device__ __forceinline__ f32x16 mfma_keep_c(v8 a, v8 b, const f32x16& c) {
        f32x16 d;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
    }
    static __device__ __forceinline__ float to_float(__bf16 x) { return (float)x; }
    static __device__ __forceinline__ __bf16 from_float(float x) { return (__bf16)x; }
    static __device__ __forceinline__ __bf16 from_double(double x) { return (__bf16)(float)x; }  // torch: double -> float -> bf16
    // acc + a + b in fp32 for a packed pair of 16-bit values: one v_dot2c_f32_bf16 against (1, 1)
    static __device__ __forceinline__ float add_pair(__bf16 a, __bf16 b, float acc) {
        using v2 = __bf16 __attribute__((ext_vector_type(2)));
        const v2 x = {a, b}, one = {(__bf16)1.f, (__bf16)1.f};
        return __builtin_amdgcn_fdot2_f32_bf16(x, one, acc, false);
    }
};

template <>
struct Elt<_Float16> {
    using v8 = f16x8;
    using v4 = f16x4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma_keep_c(v8 a, v8 b, const f32x16& c) {   // (see Elt<__bf16>)
        f32x16 d;
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
    }
    static __device__ __forceinline__ float to_float(_Float16 x) { return (float)x; }
    static __device__ __forceinline__ _Float16 from_float(float x) { return (_Float16)x; }
    static __device__ __forceinline__ float add_pair(_Float16 a, _Float16 b, float acc) {   // v_dot2c_f32_f16 against (1, 1)
        using v2 = _Float16 __attribute__((ext_vector_type(2)));
        const v2 x = {a, b}, one = {(_Float16)1.f, (_Float16)1.f};
        return __builtin_amdgcn_fdot2(x, one, acc, false);
    }
    static __device__ __forceinline__ _Float16 from_double(double x) { return (_Float16)(float)x; }  // torch: double -> float -> half
};

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int kNumXCD = 8;
constexpr int kNumCU = 256;

// Where the four tensors of an attention call live (kernel-argument form of svg_attn_layout_t, strides in ELEMENTS): element
// (bh, s, :) of tensor x starts at x + (bh / hpb) * x_bs + (bh % hpb) * x_hs + s * x_rs.  The contiguous [BH, S, D] layout is
// { hpb = BH, hs = S * D, rs = D }.  Head and batch strides reach every attention body through the policies' *_base(); ROW strides
// other than D exist in the two-phase bodies (attn_m16.h, attn_body_pp2 of attn_core.h) and the online profiler's second form only — the host
// side refuses them elsewhere.
struct AttnLayout {
    int hpb_q, hpb_kv;
    long long q_bs, q_hs, k_bs, k_hs, v_bs, v_hs, o_bs, o_hs;
    int q_rs, k_rs, v_rs, o_rs;
    bool rows_contiguous(int D) const { return q_rs == D && k_rs == D && v_rs == D && o_rs == D; }
};
inline AttnLayout contiguous_layout(int Hq, int Hkv, long long Sq, long long Skv, int D) {
    AttnLayout l;
    l.hpb_q = Hq > 0 ? Hq : 1, l.hpb_kv = Hkv > 0 ? Hkv : 1;
    l.q_bs = l.o_bs = (long long)Hq * Sq * D, l.k_bs = l.v_bs = (long long)Hkv * Skv * D;
    l.q_hs = l.o_hs = Sq * D, l.k_hs = l.v_hs = Skv * D;
    l.q_rs = l.k_rs = l.v_rs = l.o_rs = D;
    return l;
}
// svg_attn_layout_t -> AttnLayout with the checks of the header (16-byte rows, 32-bit LDS-DMA offsets for k / v, 23-bit row strides).
inline int layout_from_abi(const svg_attn_layout_t* a, int Hq, int Hkv, long long Sq, long long Skv, int D, const void* q, const void* k,
                           const void* v, const void* o, AttnLayout& l) {
    if (!a) return SVG_ERR_BAD_ARG;
    const int hq = a->heads_per_batch, hkv = a->kv_heads_per_batch > 0 ? a->kv_heads_per_batch : (int)((long long)hq * Hkv / (Hq > 0 ? Hq : 1));
    if (hq <= 0 || hkv <= 0 || Hq % hq != 0 || Hkv % hkv != 0 || Hq / hq != Hkv / hkv) return SVG_ERR_BAD_ARG;
    l.hpb_q = hq, l.hpb_kv = hkv;
    const svg_tensor_strides_t* s4[4] = {&a->q, &a->k, &a->v, &a->o};
    const void* p4[4] = {q, k, v, o};
    for (int i = 0; i < 4; ++i) {
        const svg_tensor_strides_t& s = *s4[i];
        if (s.row < D || s.head < 0 || s.batch < 0) return SVG_ERR_BAD_ARG;
        if (s.row % 8 != 0 || s.head % 8 != 0 || s.batch % 8 != 0 || ((size_t)p4[i] & 15) != 0) return SVG_ERR_UNSUPPORTED;
        if (s.row >= (1ll << 23)) return SVG_ERR_UNSUPPORTED;
    }
    if (Skv * a->k.row * 2 >= (1ll << 32) || Skv * a->v.row * 2 >= (1ll << 32)) return SVG_ERR_UNSUPPORTED;
    if (Sq >= (1ll << 24) || Skv >= (1ll << 24)) return SVG_ERR_UNSUPPORTED;
    l.q_bs = a->q.batch, l.q_hs = a->q.head, l.q_rs = (int)a->q.row;
    l.k_bs = a->k.batch, l.k_hs = a->k.head, l.k_rs = (int)a->k.row;
    l.v_bs = a->v.batch, l.v_hs = a->v.head, l.v_rs = (int)a->v.row;
    l.o_bs = a->o.batch, l.o_hs = a->o.head, l.o_rs = (int)a->o.row;
    return SVG_OK;
}
__device__ __forceinline__ size_t layout_head_off(long long bs, long long hs, int hpb, int head) {
    const int b = head / hpb;   // (wave-uniform: scalar unit, once per workgroup)
    return (size_t)b * (size_t)bs + (size_t)(head - b * hpb) * (size_t)hs;
}

}  // namespace svg
