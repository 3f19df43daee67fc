// fp8 (e4m3) SVG1 band / dense attention: quantise + placement pre-pass, kernel on attn_body_f8, C entry points.
// BASELINE.json configs[4]; no reference implementation exists (README.md:117), see attn_f8.h.
#include "attn_f8.h"
#include "band_policy.h"

namespace svg {

template <typename T>
using BandF8 = BandPolicy<T, 128, 8, false>;   // 8 waves x 32 rows: 256-row q-tiles (the tiling of the two-phase 16-bit kernel)

#ifndef SVG_F8_PINGPONG
#define SVG_F8_PINGPONG 1     // 1: two-phase ping-pong body (attn_body_f8pp), 0: lock-step body (attn_body_f8)
#endif
template <typename T>
__global__ __launch_bounds__(512, 2) void band_attn_f8_kernel(typename BandF8<T>::Params prm, F8Args fa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#if SVG_F8_PINGPONG
    attn_body_f8pp<T, BandF8<T>>(prm, fa, smem);
#else
    attn_body_f8<T, BandF8<T>>(prm, fa, smem);
#endif
}

// ---- pre-pass 1: per-head absolute maxima of q, k, v (float bits of non-negative values order like unsigned integers) ----
template <typename T>
__global__ __launch_bounds__(256) void f8_amax_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                      unsigned* __restrict__ amax, size_t per_head) {
    const int head = blockIdx.y;
    const size_t n8 = per_head / 8;
    float mq = 0.f, mk = 0.f, mv = 0.f;
    using V8 = typename Elt<T>::v8;
    const V8* q8 = (const V8*)(q + head * per_head);
    const V8* k8 = (const V8*)(k + head * per_head);
    const V8* v8 = (const V8*)(v + head * per_head);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const V8 a = q8[i], b = k8[i], c = v8[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mq = fmaxf(mq, fabsf((float)a[j]));
            mk = fmaxf(mk, fabsf((float)b[j]));
            mv = fmaxf(mv, fabsf((float)c[j]));
        }
    }
    mq = wave_max(mq), mk = wave_max(mk), mv = wave_max(mv);
    if ((threadIdx.x & 63) == 0) {
        atomicMax(amax + 3 * head + 0, __float_as_uint(mq));
        atomicMax(amax + 3 * head + 1, __float_as_uint(mk));
        atomicMax(amax + 3 * head + 2, __float_as_uint(mv));
    }
}

// ---- pre-pass 2: quantise to e4m3 with the head's scales, in LOGICAL token order (the head placement of the 16-bit kernels'
// address arithmetic, ref svg/models/hyvideo/placement.py:34-153, applied here once), V transposed per 64-key tile in the slot
// order of attn_f8.h.  One workgroup = one 64-row tile of one head.
constexpr float kF8Max = 448.f;

__device__ __forceinline__ int f8_phys_row(int logical, bool perm, int vid0, int F, int P, int V) {
    if (perm) {
        const unsigned i = (unsigned)(logical - vid0);
        if (i < (unsigned)V) {
            const unsigned pp = i / (unsigned)F, f = i - pp * (unsigned)F;
            return vid0 + (int)(f * (unsigned)P + pp);
        }
    }
    return logical;
}

__device__ __forceinline__ unsigned f8_pack4(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}

template <typename T>
__global__ __launch_bounds__(256) void f8_quantize_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                          uint8_t* __restrict__ q8, uint8_t* __restrict__ k8, uint8_t* __restrict__ vt8,
                                                          const unsigned* __restrict__ amax, float* __restrict__ scales, int S, int S_pad,
                                                          const int64_t* __restrict__ head_flag, int vid0, int F, int P, int V,
                                                          float scale_log2) {
    constexpr int D = 128;
    __shared__ __attribute__((aligned(16))) T vs[kBN][D + 8];   // V tile, rows padded by 16 B (column reads hit different banks)
    const int head = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const bool perm = head_flag != nullptr && head_flag[head] != 0;
    const float aq = __uint_as_float(amax[3 * head]), ak = __uint_as_float(amax[3 * head + 1]), av = __uint_as_float(amax[3 * head + 2]);
    const float sk = ak > 0.f ? kF8Max / ak : 1.f, sv = av > 0.f ? kF8Max / av : 1.f;
    // q carries the whole softmax scale: q8 = q * (scale_log2 / sk) * 2^-e, with the power of two e chosen so that the head's largest
    // |q8| lands in (224, 448] — then sum(k8 q8) * 2^e IS the exponent argument scale_log2 * q.k, and 2^e is an E8M0 block scale of the
    // MFMA (two-phase body) or one multiply (lock-step body): no per-element scale-and-shift on the VALU
    const float mq_ideal = scale_log2 / sk;
    int e = aq > 0.f ? (int)ceilf(log2f(aq * mq_ideal / kF8Max)) : 0;
    e = max(-120, min(120, e));
    const float sq = mq_ideal * exp2f((float)-e);
    if (tile == 0 && tid == 0) {
        scales[4 * head] = exp2f((float)e);
        scales[4 * head + 1] = 1.f / sv;
        scales[4 * head + 2] = __int_as_float((127 + e) * 0x01010101);
        scales[4 * head + 3] = 0.f;
    }
    const size_t hb = (size_t)head * S * D;
    const size_t ob = (size_t)head * S_pad * D;
    using V8 = typename Elt<T>::v8;
    // q, k: 64 rows x 128 columns = 1024 groups of 8 elements per tensor, 4 per thread; v rows go to LDS
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int id = tid + it * 256;
        const int r = id >> 4, c8 = (id & 15) * 8;
        const int l = tile * kBN + r;
        unsigned w0 = 0, w1 = 0, x0 = 0, x1 = 0;
        V8 vv;
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] = (T)0.f;
        if (l < S) {
            const size_t src = hb + (size_t)f8_phys_row(l, perm, vid0, F, P, V) * D + c8;
            const V8 a = *(const V8*)(q + src), b = *(const V8*)(k + src);
            vv = *(const V8*)(v + src);
            w0 = f8_pack4((float)a[0] * sq, (float)a[1] * sq, (float)a[2] * sq, (float)a[3] * sq);
            w1 = f8_pack4((float)a[4] * sq, (float)a[5] * sq, (float)a[6] * sq, (float)a[7] * sq);
            x0 = f8_pack4((float)b[0] * sk, (float)b[1] * sk, (float)b[2] * sk, (float)b[3] * sk);
            x1 = f8_pack4((float)b[4] * sk, (float)b[5] * sk, (float)b[6] * sk, (float)b[7] * sk);
        }
        const size_t dst = ob + (size_t)l * D + c8;
        *(u32x2*)(q8 + dst) = u32x2{w0, w1};
        *(u32x2*)(k8 + dst) = u32x2{x0, x1};
        *(V8*)(&vs[r][c8]) = vv;
    }
    __syncthreads();
    // V^T: 128 rows (d) x 64 byte positions; thread = (d, 16-byte chunk), 2 per thread.  Position 32 g + 16 b + 4 j + i <-> key
    // 32 b + 8 j + 4 g + i (attn_f8.h)
    uint8_t* vt = vt8 + ob + (size_t)tile * (kBN * D);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int id = tid + it * 256;
        const int d = id >> 2, ch = id & 3;       // chunk ch holds positions 16 ch .. 16 ch + 15: g = ch >> 1, b = ch & 1
        const int gg = ch >> 1, b = ch & 1;
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key0 = 32 * b + 8 * j + 4 * gg;
            w[j] = f8_pack4((float)vs[key0][d] * sv, (float)vs[key0 + 1][d] * sv, (float)vs[key0 + 2][d] * sv, (float)vs[key0 + 3][d] * sv);
        }
        *(u32x4*)(vt + d * 64 + ch * 16) = u32x4{w[0], w[1], w[2], w[3]};
    }
}

// ---- plain quantiser of the gathering body (SVG2): one tensor [H, S, D] -> e4m3 row-major with the head's scale; inv[head * stride]
// receives amax / 448 ----
template <typename T>
__global__ __launch_bounds__(256) void f8_amax1_kernel(const T* __restrict__ x, unsigned* __restrict__ amax, size_t per_head) {
    const int head = blockIdx.y;
    using V8 = typename Elt<T>::v8;
    const V8* x8 = (const V8*)(x + head * per_head);
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per_head / 8; i += (size_t)gridDim.x * 256) {
        const V8 a = x8[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf((float)a[j]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(amax + head, __float_as_uint(m));
}

// mode 0: x * 448 / amax(head), inv = amax / 448.  mode 1 (q of the gathering body): q carries the softmax scale like in the SVG1
// pre-pass — q8 = q * (scale_log2 / sk) * 2^-e with sk the scale of the kv head this q head reads and e the power of two that puts
// the head's largest |q8| into (224, 448]; inv receives the E8M0 scale word (127 + e) * 0x01010101 as float bits.
template <typename T>
__global__ __launch_bounds__(256) void f8_quantize1_kernel(const T* __restrict__ x, uint8_t* __restrict__ y, const unsigned* __restrict__ amax,
                                                           float* __restrict__ inv, int inv_stride, size_t per_head, int mode,
                                                           const unsigned* __restrict__ amax_k, int group, float scale_log2) {
    const int head = blockIdx.y;
    const float a = __uint_as_float(amax[head]);
    float sc = a > 0.f ? kF8Max / a : 1.f;
    float inv_val = 1.f / sc;
    if (mode == 1) {
        const float ak = __uint_as_float(amax_k[head / group]);
        const float sk = ak > 0.f ? kF8Max / ak : 1.f;
        const float ideal = scale_log2 / sk;
        int e = a > 0.f ? (int)ceilf(log2f(a * ideal / kF8Max)) : 0;
        e = max(-120, min(120, e));
        sc = ideal * exp2f((float)-e);
        inv_val = __int_as_float((127 + e) * 0x01010101);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) inv[(size_t)head * inv_stride] = inv_val;
    using V8 = typename Elt<T>::v8;
    const V8* x8 = (const V8*)(x + head * per_head);
    u32x2* y8 = (u32x2*)(y + head * per_head);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per_head / 8; i += (size_t)gridDim.x * 256) {
        const V8 v = x8[i];
        y8[i] = u32x2{f8_pack4((float)v[0] * sc, (float)v[1] * sc, (float)v[2] * sc, (float)v[3] * sc),
                      f8_pack4((float)v[4] * sc, (float)v[5] * sc, (float)v[6] * sc, (float)v[7] * sc)};
    }
}

size_t f8g_ws_bytes(int Hq, int Hkv, int Sq, int Skv) {
    return (size_t)Hq * Sq * 128 + 2 * (size_t)Hkv * Skv * 128 + (size_t)(Hq + 2 * Hkv) * (sizeof(unsigned) + sizeof(float)) + 512;
}

// quantise q [Hq, Sq, 128], k, v [Hkv, Skv, 128] into `ws` and fill `fa` (declared in band_policy.h; the kernel that consumes it
// lives in attention.hip next to the variable-block policy)
template <typename T>
static int f8g_quantize_t(const void* q, const void* k, const void* v, int Hq, int Hkv, int Sq, int Skv, float sm_scale, void* ws,
                          F8GArgs* fa, hipStream_t st) {
    constexpr int D = 128;
    uint8_t* q8 = (uint8_t*)ws;
    uint8_t* k8 = q8 + (size_t)Hq * Sq * D;
    uint8_t* v8 = k8 + (size_t)Hkv * Skv * D;
    unsigned* amax = (unsigned*)(((uintptr_t)(v8 + (size_t)Hkv * Skv * D) + 63) & ~(uintptr_t)63);   // [Hq] q, [Hkv] k, [Hkv] v
    float* q_inv = (float*)(amax + Hq + 2 * Hkv);
    float* kv_inv = q_inv + Hq;
    if (hipMemsetAsync(amax, 0, (size_t)(Hq + 2 * Hkv) * sizeof(unsigned), st) != hipSuccess) return SVG_ERR_LAUNCH;
    const size_t phq = (size_t)Sq * D, phk = (size_t)Skv * D;
    hipLaunchKernelGGL(f8_amax1_kernel<T>, dim3(64, Hq), dim3(256), 0, st, (const T*)q, amax, phq);
    hipLaunchKernelGGL(f8_amax1_kernel<T>, dim3(64, Hkv), dim3(256), 0, st, (const T*)k, amax + Hq, phk);
    hipLaunchKernelGGL(f8_amax1_kernel<T>, dim3(64, Hkv), dim3(256), 0, st, (const T*)v, amax + Hq + Hkv, phk);
    const unsigned* no_k = nullptr;
    hipLaunchKernelGGL(f8_quantize1_kernel<T>, dim3(256, Hq), dim3(256), 0, st, (const T*)q, q8, amax, q_inv, 1, phq, 1,
                       (const unsigned*)(amax + Hq), Hq / Hkv, sm_scale * 1.4426950408889634f);
    hipLaunchKernelGGL(f8_quantize1_kernel<T>, dim3(256, Hkv), dim3(256), 0, st, (const T*)k, k8, amax + Hq, kv_inv, 2, phk, 0, no_k, 1, 0.f);
    hipLaunchKernelGGL(f8_quantize1_kernel<T>, dim3(256, Hkv), dim3(256), 0, st, (const T*)v, v8, amax + Hq + Hkv, kv_inv + 1, 2, phk, 0, no_k,
                       1, 0.f);
    fa->q8 = q8, fa->k8 = k8, fa->v8 = v8, fa->q_inv = q_inv, fa->kv_inv = kv_inv;
    return launch_status();
}

int f8g_quantize(const void* q, const void* k, const void* v, int Hq, int Hkv, int Sq, int Skv, int dtype, float sm_scale, void* ws,
                 F8GArgs* fa, hipStream_t st) {
    if (dtype == SVG_DTYPE_BF16) return f8g_quantize_t<__bf16>(q, k, v, Hq, Hkv, Sq, Skv, sm_scale, ws, fa, st);
    if (dtype == SVG_DTYPE_F16) return f8g_quantize_t<_Float16>(q, k, v, Hq, Hkv, Sq, Skv, sm_scale, ws, fa, st);
    return SVG_ERR_UNSUPPORTED;
}

static size_t f8_ws_bytes(int BH, int S) {
    const size_t S_pad = (size_t)(S + kBN - 1) / kBN * kBN;
    return 3 * (size_t)BH * S_pad * 128 + (size_t)BH * (3 * sizeof(unsigned) + 4 * sizeof(float)) + 256;
}

// what = 1: pre-pass only, 2: attention on a workspace the pre-pass has filled (same BH, S, perm), 3: both
template <typename T>
static int run_f8(const void* q, const void* k, const void* v, void* o, int BH, int S, float sm_scale, const svg_band_mask_t* mask,
                  const svg_perm_desc_t* perm, void* ws, const BandOpts& opts, int what, hipStream_t st) {
    using Pol = BandF8<T>;
    constexpr int D = 128;
    const int S_pad = (S + kBN - 1) / kBN * kBN;
    uint8_t* q8 = (uint8_t*)ws;
    uint8_t* k8 = q8 + (size_t)BH * S_pad * D;
    uint8_t* vt8 = k8 + (size_t)BH * S_pad * D;
    unsigned* amax = (unsigned*)(((uintptr_t)(vt8 + (size_t)BH * S_pad * D) + 63) & ~(uintptr_t)63);
    float* scales = (float*)(amax + 3 * BH);
    if (what & 1) {
    if (hipMemsetAsync(amax, 0, (size_t)BH * 3 * sizeof(unsigned), st) != hipSuccess) return SVG_ERR_LAUNCH;
    hipLaunchKernelGGL(f8_amax_kernel<T>, dim3(64, BH), dim3(256), 0, st, (const T*)q, (const T*)k, (const T*)v, amax, (size_t)S * D);
    const bool has_perm = perm && perm->head_perm_flag;
    hipLaunchKernelGGL(f8_quantize_kernel<T>, dim3(S_pad / kBN, BH), dim3(256), 0, st, (const T*)q, (const T*)k, (const T*)v, q8, k8, vt8,
                       amax, scales, S, S_pad, has_perm ? perm->head_perm_flag : nullptr, has_perm ? perm->vid0 : 0,
                       has_perm ? perm->num_frame : 1, has_perm ? perm->frame_size : 1,
                       has_perm ? perm->num_frame * perm->frame_size : 0, sm_scale * 1.4426950408889634f);
    }
    if (!(what & 2)) return launch_status();
    const typename Pol::Params p = make_band_params<Pol, T>(q, k, v, o, BH, S, sm_scale, mask, perm, opts);
    F8Args fa{q8, k8, vt8, scales, S_pad};
    auto kern = band_attn_f8_kernel<T>;
    if (const int rc = configure_lds((const void*)kern, attn_f8_lds_bytes<D, 8, 4>()); rc != SVG_OK) return rc;
    constexpr int kLds = attn_f8_lds_bytes<D, 8, 4>();
    hipLaunchKernelGGL(kern, dim3(p.nqt * BH), dim3(512), kLds, st, p, fa);
    return launch_status();
}

}  // namespace svg

using namespace svg;

extern "C" size_t svg_band_attention_fp8_workspace_bytes(int32_t BH, int32_t S, int32_t D) {
    if (BH <= 0 || S <= 0 || D != 128) return 0;
    return f8_ws_bytes(BH, S);
}

static int f8_entry(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D, int32_t dtype, float sm_scale,
                    const svg_band_mask_t* mask, const svg_perm_desc_t* perm, void* workspace, size_t workspace_bytes, int what,
                    void* stream) {
    if (!q || !k || !v || !o || !mask || !workspace || BH <= 0 || S <= 0) return SVG_ERR_BAD_ARG;
    if (D != 128) return SVG_ERR_UNSUPPORTED;
    if (mask->real_len < 0 || mask->real_len > S || mask->band < 0 || mask->band > S + 1) return SVG_ERR_BAD_ARG;
    if (mask->colfull_lo > mask->colfull_hi || mask->rowfull_lo > mask->rowfull_hi) return SVG_ERR_BAD_ARG;
    if (perm && perm->head_perm_flag) {
        if (perm->num_frame <= 0 || perm->frame_size <= 0 || perm->vid0 < 0 ||
            (int64_t)perm->vid0 + (int64_t)perm->num_frame * perm->frame_size > S)
            return SVG_ERR_BAD_ARG;
    }
    if (workspace_bytes < f8_ws_bytes(BH, S)) return SVG_ERR_WORKSPACE;
    if (dtype == SVG_DTYPE_BF16) return run_f8<__bf16>(q, k, v, o, BH, S, sm_scale, mask, perm, workspace, BandOpts(), what, (hipStream_t)stream);
    if (dtype == SVG_DTYPE_F16) return run_f8<_Float16>(q, k, v, o, BH, S, sm_scale, mask, perm, workspace, BandOpts(), what, (hipStream_t)stream);
    return SVG_ERR_UNSUPPORTED;
}

extern "C" int svg_band_attention_fp8(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                      int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    return f8_entry(q, k, v, o, BH, S, D, dtype, sm_scale, mask, perm, workspace, workspace_bytes, 3, stream);
}

extern "C" int svg_band_attention_fp8_stage(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                            int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                            void* workspace, size_t workspace_bytes, int32_t stage, void* stream) {
    if (stage != 1 && stage != 2) return SVG_ERR_BAD_ARG;
    return f8_entry(q, k, v, o, BH, S, D, dtype, sm_scale, mask, perm, workspace, workspace_bytes, stage, stream);
}
