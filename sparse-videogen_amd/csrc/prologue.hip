// Pre-attention prologue for gfx950: in-place QK RMSNorm / LayerNorm over head_dim and rotary embedding (three variants),
// optionally fused into ONE pass over Q and K.  HBM-bound: every element of Q and K is read once and written once
// (4 * D bytes per token-head at bf16); the cos / sin tables are read once per token position and reused for all heads,
// both tensors and the whole batch.
// ref: svg/kernels/csrc/ops.h:19-78 (layer_norm_forward, rms_norm_forward), :80-260 (apply_qk_rope_inplace_cossin,
//      _txtlast, _complex); semantics = the torch references of the reference's own tests
//      (svg/kernels/test/test_rms_norm.py:27-36, test_layer_norm.py:24-29, test_apply_rope.py:24-37,
//      test_apply_rope_txtlast.py:24-37, test_apply_rope_complex.py:25-36).  The CUDA sources behind ops.h
//      (include/norm/*.cuh, include/rope/*.cuh) are not part of the reference checkout.
#include "svg_common.h"

#pragma clang fp contract(off)  // the torch references round every product before the add; no fma contraction here

namespace svg {

enum : int { kNormNone = 0, kNormRms = 1, kNormLayer = 2 };
enum : int { kRopeNone = 0, kRopeCosSin = 1, kRopeComplex = 2 };

struct PrologueParams {
    void* q;
    void* k;
    const void* q_src;   // null: in place.  Otherwise the input in token-major layout [bsz, S, H, D] (the projection output);
    const void* k_src;   // the result is written head-major [bsz, H, S, D] to q / k: the transpose rides along for free
    int Hq, Hkv, S;
    int norm, rope;
    const void* qw;   // [D] norm weight for q (dtype of q), may be null (= ones)
    const void* qb;   // [D] LayerNorm bias for q, may be null (= zeros)
    const void* kw;
    const void* kb;
    float eps;
    const float* cs;  // cos [rope_hi - rope_lo, D]   (complex: real part [.., D/2])
    const float* sn;  // sin                           (complex: imaginary part)
    int rope_lo, rope_hi;  // positions [rope_lo, rope_hi) are rotated with table row (pos - rope_lo)
    float q_scale;         // factor folded into the LAST rounding of q (1: none; an exact no-op then).  The attention kernels take a q
                           // that carries sm_scale * log2(e) (svg_band_attention_prescaled): rotated positions are rounded once either
                           // way — the factor multiplies the fp32 RoPE result in front of that rounding
};

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {  // sum over the LPR consecutive lanes that hold one row
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// T(v * oscale) with the product ROUNDED TO fp32 FIRST, like torch's `(x.float() * s).to(T)`.  Without the fence hipcc picks, element by
// element, between v_mul_f32 + v_cvt_f16_f32 and v_fma_mixlo_f16 (one instruction from the fp32 operands to the fp16 result) for T = fp16, and
// the two do not agree bit for bit: two kernels that state the same arithmetic then differ in 0.1 % of the elements (found in round 6: the
// fused Wan prologue against the unfused sequence, fp16, q_scale != 1).  bf16 has no such instruction; the fence is free there.
template <typename E>
__device__ __forceinline__ auto scaled_round(float v, float oscale) {
    float t = v * oscale;
    asm volatile("" : "+v"(t));
    return E::from_float(t);
}

// One lane = 8 consecutive channels of one (batch, head, position) row; a wave covers 64 / (D / 8) consecutive positions;
// the workgroup (4 waves) walks over all heads of Q and then of K for its positions, kUnroll rows in flight per lane.
template <typename T, int D>
__global__ __launch_bounds__(256) void qk_prologue_kernel(PrologueParams p) {
    using E = Elt<T>;
    using V8 = typename E::v8;
    constexpr int LPR = D / 8;
    constexpr int RPW = 64 / LPR;
    constexpr int kUnroll = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, c = lane - sub * LPR;
    const int pos = (blockIdx.x * 4 + wave) * RPW + sub;
    const int b = blockIdx.y;
    const bool valid = pos < p.S;
    const int spos = valid ? pos : 0;   // invalid lanes load row 0 and store nothing (the shuffles need every lane)

    const bool rot = p.rope != kRopeNone && pos >= p.rope_lo && pos < p.rope_hi;
    float cs[8], sn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 1.f, sn[j] = 0.f;
    if (rot) {
        const size_t r = (size_t)(pos - p.rope_lo);
        if (p.rope == kRopeCosSin) {
            const f32x4* pc = (const f32x4*)(p.cs + r * D + c * 8);
            const f32x4* ps = (const f32x4*)(p.sn + r * D + c * 8);
            const f32x4 c0 = pc[0], c1 = pc[1], s0 = ps[0], s1 = ps[1];
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[j] = c0[j], cs[4 + j] = c1[j], sn[j] = s0[j], sn[4 + j] = s1[j];
        } else {  // complex: 4 (real, imag) pairs for this lane's 4 channel pairs
            const f32x4 fr = *(const f32x4*)(p.cs + r * (D / 2) + c * 4);
            const f32x4 fi = *(const f32x4*)(p.sn + r * (D / 2) + c * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[j] = fr[j], sn[j] = fi[j];
        }
    }

    auto run = [&](T* base, const T* src, int H, const T* wgt, const T* bias, const float oscale) {
        float w[8], bs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = 1.f, bs[j] = 0.f;
        if (p.norm != kNormNone) {
            if (wgt) {
                const V8 wv = *(const V8*)(wgt + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = E::to_float(wv[j]);
            }
            if (bias) {
                const V8 bv = *(const V8*)(bias + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) bs[j] = E::to_float(bv[j]);
            }
        }
        T* row0 = base + (((size_t)b * H) * p.S + spos) * D + c * 8;
        const size_t hstride = (size_t)p.S * D;
        // input rows: in place, or token-major [bsz, S, H, D] where the heads of one position are D apart
        const T* in0 = src ? src + (((size_t)b * p.S + spos) * H) * D + c * 8 : row0;
        const size_t in_hstride = src ? (size_t)D : hstride;
        for (int h0 = 0; h0 < H; h0 += kUnroll) {
            V8 xin[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u)   // (heads past H re-read the last head; nothing is stored for them)
                xin[u] = *(const V8*)(in0 + (size_t)min(h0 + u, H - 1) * in_hstride);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = E::to_float(xin[u][j]);
                if (p.norm == kNormRms) {
                    // diffusers RMSNorm / the reference's "replica": fp32 variance, normalised value rounded to the tensor
                    // dtype, THEN multiplied by the weight (rounded again)
                    float ss = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
                    ss = group_sum<LPR>(ss);
                    const float inv = 1.0f / sqrtf(ss / (float)D + p.eps);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float n = E::to_float(E::from_float(x[j] * inv));
                        x[j] = E::to_float(E::from_float(w[j] * n));
                    }
                } else if (p.norm == kNormLayer) {
                    // torch layer_norm: fp32 statistics (biased variance), one rounding at the end
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) s += x[j];
                    const float mean = group_sum<LPR>(s) / (float)D;
                    float vs = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) vs += (x[j] - mean) * (x[j] - mean);
                    const float inv = 1.0f / sqrtf(group_sum<LPR>(vs) / (float)D + p.eps);
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = E::to_float(E::from_float((x[j] - mean) * inv * w[j] + bs[j]));
                }
                V8 out;
                if (rot && p.rope == kRopeCosSin) {
                    // out = x * cos + rotate(x) * sin in fp32, rotate(x)[2i] = -x[2i+1], rotate(x)[2i+1] = x[2i]
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float a = x[2 * i], bq = x[2 * i + 1];
                        out[2 * i] = scaled_round<E>(a * cs[2 * i] + (-bq) * sn[2 * i], oscale);
                        out[2 * i + 1] = scaled_round<E>(bq * cs[2 * i + 1] + a * sn[2 * i + 1], oscale);
                    }
                } else if (rot) {
                    // (x[2i] + i x[2i+1]) * (fr + i fi) in fp64 (the reference multiplies complex128 by complex64)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double a = (double)x[2 * i], bq = (double)x[2 * i + 1];
                        const double fr = (double)cs[i], fi = (double)sn[i];
                        out[2 * i] = E::from_double((a * fr - bq * fi) * (double)oscale);
                        out[2 * i + 1] = E::from_double((a * fi + bq * fr) * (double)oscale);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) out[j] = scaled_round<E>(x[j], oscale);
                }
                if (valid && h0 + u < H) *(V8*)(row0 + (size_t)(h0 + u) * hstride) = out;
            }
        }
    };
    if (p.q) run((T*)p.q, (const T*)p.q_src, p.Hq, (const T*)p.qw, (const T*)p.qb, p.q_scale);
    if (p.k) run((T*)p.k, (const T*)p.k_src, p.Hkv, (const T*)p.kw, (const T*)p.kb, 1.f);
}

template <typename T>
static int launch_prologue_t(const PrologueParams& p, int bsz, int D, hipStream_t st) {
    auto go = [&](auto d_c) -> int {
        constexpr int DD = decltype(d_c)::value;
        constexpr int RPB = 4 * (64 / (DD / 8));
        hipLaunchKernelGGL((qk_prologue_kernel<T, DD>), dim3((p.S + RPB - 1) / RPB, bsz), dim3(256), 0, st, p);
        return launch_status();
    };
    switch (D) {
        case 32: return go(std::integral_constant<int, 32>{});
        case 64: return go(std::integral_constant<int, 64>{});
        case 128: return go(std::integral_constant<int, 128>{});
        case 256: return go(std::integral_constant<int, 256>{});
        default: return SVG_ERR_UNSUPPORTED;
    }
}

static int launch_prologue(const PrologueParams& p, int bsz, int D, int dtype, hipStream_t st) {
    if (bsz <= 0 || p.S <= 0) return SVG_ERR_BAD_ARG;
    if (bsz > 65535) return SVG_ERR_UNSUPPORTED;
    if (dtype == SVG_DTYPE_BF16) return launch_prologue_t<__bf16>(p, bsz, D, st);
    if (dtype == SVG_DTYPE_F16) return launch_prologue_t<_Float16>(p, bsz, D, st);
    return SVG_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Wan 2.1 prologue in ONE pass: RMSNorm ACROSS ALL HEADS (the reference's Triton form: fp32 x * rstd * w over the H * D row, one rounding) ->
// rotary embedding -> head-major transpose, for q and k, plus the plain transpose of v in the same launch.
// ref: svg/models/wan/attention.py:99-148 — get_qk_norm (triton_rmsnorm_forward, :105-120), get_transpose_qkv (:122-133: three
//      `.transpose(1, 2).contiguous()` copies), get_rotary_emb (:135-140: _kernels.apply_qk_rope_inplace_cossin_complex) — three passes over q
//      and k and one over v there (and in this library until round 6: svg_rmsnorm_forward + svg_qk_norm_rope_transpose).
// One wave per token row: lane l owns the 16-byte chunks l, l + 64, ... of the row (exactly svg_rmsnorm_forward's layout and summation order:
// the same rstd, bit for bit), which live in registers between the statistics and the output; chunk c of the row is chunk c % (D / 8) of
// head c / (D / 8), and 64 % (D / 8) == 0, so ALL of a lane's chunks sit at the same channels of their heads: one rotary-table read per
// lane and row.  The normalised value is rounded to T before the rotation, as the unfused sequence leaves it in memory: the result equals
// svg_rmsnorm_forward -> svg_qk_norm_rope_transpose bit for bit.  HBM traffic: every tensor read once, written once.
// ---------------------------------------------------------------------------------------------------------------------------------
struct RmsAllParams {
    const void* q_in;    // [bsz, S, H * D] token-major projection outputs (k_in, v_in may be null)
    const void* k_in;
    const void* v_in;
    void* q_out;         // [bsz, H, S, D]
    void* k_out;
    void* v_out;
    const void* qw;      // [H * D] weights (dtype w_dt), may be null (= ones)
    const void* kw;
    int w_dt;
    int H, S, bsz;
    float eps;
    int rope;
    const float* cs;
    const float* sn;
    int rope_lo, rope_hi;
    float q_scale;
};

template <typename T, int D, int NCH>
__global__ __launch_bounds__(256) void rmsall_rope_transpose_kernel(RmsAllParams p) {
    using E = Elt<T>;
    using V8 = typename E::v8;
    constexpr int LPR = D / 8;
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)p.bsz * p.S) return;
    const int b = (int)(row / p.S), pos = (int)(row - (long long)b * p.S);
    const int N = p.H * D, nchunks = N / 8;
    const int cih = lane % LPR;                      // the lane's chunk inside a head (the same for all its chunks)
    const bool rot = p.rope != kRopeNone && pos >= p.rope_lo && pos < p.rope_hi;
    float cs[8], sn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 1.f, sn[j] = 0.f;
    if (rot) {
        const size_t r = (size_t)(pos - p.rope_lo);
        if (p.rope == kRopeCosSin) {
            const f32x4* pc = (const f32x4*)(p.cs + r * D + cih * 8);
            const f32x4* ps = (const f32x4*)(p.sn + r * D + cih * 8);
            const f32x4 c0 = pc[0], c1 = pc[1], s0 = ps[0], s1 = ps[1];
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[j] = c0[j], cs[4 + j] = c1[j], sn[j] = s0[j], sn[4 + j] = s1[j];
        } else {
            const f32x4 fr = *(const f32x4*)(p.cs + r * (D / 2) + cih * 4);
            const f32x4 fi = *(const f32x4*)(p.sn + r * (D / 2) + cih * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[j] = fr[j], sn[j] = fi[j];
        }
    }
    auto out_ptr = [&](void* base, int c) -> T* {   // chunk c of the token row -> its place in the head-major tensor
        const int head = c / LPR;
        return (T*)base + (((size_t)b * p.H + head) * p.S + pos) * D + cih * 8;
    };
    auto run = [&](const T* in, T* out, const void* wgt, const float oscale) {
        const T* src = in + (size_t)row * N;
        V8 xin[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) xin[i] = *(const V8*)(src + (size_t)c * 8);
        }
        float x[NCH][8];
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[i][j] = (lane + 64 * i < nchunks) ? E::to_float(xin[i][j]) : 0.f;
        // svg_rmsnorm_forward (glue.hip, row_glue_body<NCH, 2>): the same sum, the same order
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s2 += x[i][j] * x[i][j];
        const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)N + p.eps);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c >= nchunks) continue;
            float w[8];
            if (wgt) {
                if (p.w_dt == SVG_DTYPE_F32) {
                    const f32x4* pw = (const f32x4*)((const float*)wgt + (size_t)c * 8);
                    const f32x4 a = pw[0], d = pw[1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = a[j], w[4 + j] = d[j];
                } else if (p.w_dt == SVG_DTYPE_BF16) {
                    const bf16x8 a = *(const bf16x8*)((const __bf16*)wgt + (size_t)c * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = (float)a[j];
                } else {
                    const f16x8 a = *(const f16x8*)((const _Float16*)wgt + (size_t)c * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = (float)a[j];
                }
            }
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = x[i][j] * rstd;
                y[j] = E::to_float(E::from_float(wgt ? xh * w[j] : xh));      // the rounding svg_rmsnorm_forward stores
            }
            V8 o;
            if (rot && p.rope == kRopeCosSin) {
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const float a = y[2 * k2], bq = y[2 * k2 + 1];
                    o[2 * k2] = scaled_round<E>(a * cs[2 * k2] + (-bq) * sn[2 * k2], oscale);
                    o[2 * k2 + 1] = scaled_round<E>(bq * cs[2 * k2 + 1] + a * sn[2 * k2 + 1], oscale);
                }
            } else if (rot) {
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const double a = (double)y[2 * k2], bq = (double)y[2 * k2 + 1];
                    const double fr = (double)cs[k2], fi = (double)sn[k2];
                    o[2 * k2] = E::from_double((a * fr - bq * fi) * (double)oscale);
                    o[2 * k2 + 1] = E::from_double((a * fi + bq * fr) * (double)oscale);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = scaled_round<E>(y[j], oscale);
            }
            *(V8*)out_ptr(out, c) = o;
        }
    };
    // One wave takes its row through q, k and v in turn.  Measured at Wan 720p and NOT shipped (same boxes: the three-pass sequence took 1.77 ms on
    // each): the three tensors as three waves (grid.y) 1.13 ms, that plus the rows of a workgroup staged through LDS so that a store instruction
    // writes 1 KB of one head instead of 256-byte pieces of four heads 1.17 ms — against 1.04 ms for this form.
    if (p.q_in) run((const T*)p.q_in, (T*)p.q_out, p.qw, p.q_scale);
    if (p.k_in) run((const T*)p.k_in, (T*)p.k_out, p.kw, 1.f);
    if (p.v_in) {   // plain transpose
        const T* src = (const T*)p.v_in + (size_t)row * N;
        V8 xin[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) xin[i] = *(const V8*)(src + (size_t)c * 8);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) *(V8*)out_ptr(p.v_out, c) = xin[i];
        }
    }
}

template <typename T, int D>
static int launch_rmsall(const RmsAllParams& p, hipStream_t st) {
    const int need = (p.H * D / 8 + 63) / 64;
    const long long rows = (long long)p.bsz * p.S;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define SVG_RMSALL(NC)                                                                              \
    if (need <= NC) {                                                                               \
        hipLaunchKernelGGL((rmsall_rope_transpose_kernel<T, D, NC>), grid, block, 0, st, p);        \
        return launch_status();                                                                     \
    }
    SVG_RMSALL(2) SVG_RMSALL(4) SVG_RMSALL(6) SVG_RMSALL(10) SVG_RMSALL(16)
#undef SVG_RMSALL
    return SVG_ERR_UNSUPPORTED;
}

}  // namespace svg

using namespace svg;

extern "C" int svg_rmsnorm_rope_transpose(const void* q_in, const void* k_in, const void* v_in, void* q_out, void* k_out, void* v_out,
                                          int32_t bsz, int32_t H, int32_t S, int32_t D, int32_t dtype, const void* q_weight,
                                          const void* k_weight, int32_t w_dtype, float eps, int32_t rope_kind, const float* cos_or_real,
                                          const float* sin_or_imag, int32_t rope_lo, int32_t rope_hi, float q_scale, void* stream) {
    if (!q_in && !k_in && !v_in) return SVG_ERR_BAD_ARG;
    if ((q_in && !q_out) || (k_in && !k_out) || (v_in && !v_out) || q_in == q_out || (k_in && k_in == k_out) || (v_in && v_in == v_out))
        return SVG_ERR_BAD_ARG;
    if (bsz <= 0 || H <= 0 || S <= 0 || !(q_scale > 0.f)) return SVG_ERR_BAD_ARG;
    if ((int64_t)bsz * S > 0x7fffffffll * 4) return SVG_ERR_UNSUPPORTED;
    if (rope_kind < 0 || rope_kind > 2) return SVG_ERR_BAD_ARG;
    if (rope_kind != kRopeNone && (!cos_or_real || !sin_or_imag || rope_lo < 0 || rope_hi > S || rope_lo > rope_hi)) return SVG_ERR_BAD_ARG;
    if (w_dtype != SVG_DTYPE_BF16 && w_dtype != SVG_DTYPE_F16 && w_dtype != SVG_DTYPE_F32) return SVG_ERR_UNSUPPORTED;
    if ((int64_t)H * D > 8192) return SVG_ERR_UNSUPPORTED;
    RmsAllParams p;
    p.q_in = q_in, p.k_in = k_in, p.v_in = v_in, p.q_out = q_out, p.k_out = k_out, p.v_out = v_out;
    p.qw = q_weight, p.kw = k_weight, p.w_dt = w_dtype, p.H = H, p.S = S, p.bsz = bsz, p.eps = eps;
    p.rope = rope_kind, p.cs = cos_or_real, p.sn = sin_or_imag, p.rope_lo = rope_lo, p.rope_hi = rope_hi, p.q_scale = q_scale;
    hipStream_t st = (hipStream_t)stream;
#define SVG_RMSALL_D(T)                                            \
    switch (D) {   /* the head sizes of the models that normalise across heads (Wan 2.1: 128); 64 for completeness */ \
        case 64: return launch_rmsall<T, 64>(p, st);               \
        case 128: return launch_rmsall<T, 128>(p, st);             \
        default: return SVG_ERR_UNSUPPORTED;                       \
    }
    if (dtype == SVG_DTYPE_BF16) { SVG_RMSALL_D(__bf16) }
    if (dtype == SVG_DTYPE_F16) { SVG_RMSALL_D(_Float16) }
#undef SVG_RMSALL_D
    return SVG_ERR_UNSUPPORTED;
}

extern "C" int svg_qk_norm_rope_qscale(void* q, void* k, int32_t bsz, int32_t Hq, int32_t Hkv, int32_t S, int32_t D, int32_t dtype,
                                       int32_t norm_kind, const void* q_weight, const void* q_bias, const void* k_weight,
                                       const void* k_bias, float eps, int32_t rope_kind, const float* cos_or_real,
                                       const float* sin_or_imag, int32_t rope_lo, int32_t rope_hi, float q_scale, void* stream) {
    if (!(q_scale > 0.f)) return SVG_ERR_BAD_ARG;
    if ((!q && !k) || (q && Hq <= 0) || (k && Hkv <= 0)) return SVG_ERR_BAD_ARG;
    if (norm_kind < 0 || norm_kind > 2 || rope_kind < 0 || rope_kind > 2) return SVG_ERR_BAD_ARG;
    if (rope_kind != kRopeNone) {
        if (!cos_or_real || !sin_or_imag || rope_lo < 0 || rope_hi > S || rope_lo > rope_hi) return SVG_ERR_BAD_ARG;
    }
    PrologueParams p;
    p.q_src = nullptr, p.k_src = nullptr;
    p.q = q, p.k = k, p.Hq = Hq, p.Hkv = Hkv, p.S = S, p.norm = norm_kind, p.rope = rope_kind;
    p.qw = q_weight, p.qb = q_bias, p.kw = k_weight, p.kb = k_bias, p.eps = eps;
    p.cs = cos_or_real, p.sn = sin_or_imag, p.rope_lo = rope_lo, p.rope_hi = rope_hi;
    p.q_scale = q_scale;
    return launch_prologue(p, bsz, D, dtype, (hipStream_t)stream);
}

extern "C" int svg_qk_norm_rope(void* q, void* k, int32_t bsz, int32_t Hq, int32_t Hkv, int32_t S, int32_t D, int32_t dtype,
                                int32_t norm_kind, const void* q_weight, const void* q_bias, const void* k_weight,
                                const void* k_bias, float eps, int32_t rope_kind, const float* cos_or_real,
                                const float* sin_or_imag, int32_t rope_lo, int32_t rope_hi, void* stream) {
    return svg_qk_norm_rope_qscale(q, k, bsz, Hq, Hkv, S, D, dtype, norm_kind, q_weight, q_bias, k_weight, k_bias, eps, rope_kind,
                                   cos_or_real, sin_or_imag, rope_lo, rope_hi, 1.f, stream);
}

extern "C" int svg_qk_norm_rope_transpose_qscale(const void* q_in, const void* k_in, void* q_out, void* k_out, int32_t bsz,
                                                 int32_t Hq, int32_t Hkv, int32_t S, int32_t D, int32_t dtype, int32_t norm_kind,
                                                 const void* q_weight, const void* q_bias, const void* k_weight,
                                                 const void* k_bias, float eps, int32_t rope_kind, const float* cos_or_real,
                                                 const float* sin_or_imag, int32_t rope_lo, int32_t rope_hi, float q_scale,
                                                 void* stream) {
    if (!(q_scale > 0.f)) return SVG_ERR_BAD_ARG;
    if ((!q_in && !k_in) || (q_in && (!q_out || Hq <= 0)) || (k_in && (!k_out || Hkv <= 0))) return SVG_ERR_BAD_ARG;
    if (q_in == q_out || (k_in && k_in == k_out)) return SVG_ERR_BAD_ARG;   // the layouts differ: not an in-place operation
    if (norm_kind < 0 || norm_kind > 2 || rope_kind < 0 || rope_kind > 2) return SVG_ERR_BAD_ARG;
    if (rope_kind != kRopeNone) {
        if (!cos_or_real || !sin_or_imag || rope_lo < 0 || rope_hi > S || rope_lo > rope_hi) return SVG_ERR_BAD_ARG;
    }
    PrologueParams p;
    p.q = q_in ? q_out : nullptr, p.k = k_in ? k_out : nullptr, p.q_src = q_in, p.k_src = k_in;
    p.Hq = Hq, p.Hkv = Hkv, p.S = S, p.norm = norm_kind, p.rope = rope_kind;
    p.qw = q_weight, p.qb = q_bias, p.kw = k_weight, p.kb = k_bias, p.eps = eps;
    p.cs = cos_or_real, p.sn = sin_or_imag, p.rope_lo = rope_lo, p.rope_hi = rope_hi;
    p.q_scale = q_scale;
    return launch_prologue(p, bsz, D, dtype, (hipStream_t)stream);
}

extern "C" int svg_qk_norm_rope_transpose(const void* q_in, const void* k_in, void* q_out, void* k_out, int32_t bsz, int32_t Hq,
                                          int32_t Hkv, int32_t S, int32_t D, int32_t dtype, int32_t norm_kind,
                                          const void* q_weight, const void* q_bias, const void* k_weight, const void* k_bias,
                                          float eps, int32_t rope_kind, const float* cos_or_real, const float* sin_or_imag,
                                          int32_t rope_lo, int32_t rope_hi, void* stream) {
    return svg_qk_norm_rope_transpose_qscale(q_in, k_in, q_out, k_out, bsz, Hq, Hkv, S, D, dtype, norm_kind, q_weight, q_bias,
                                             k_weight, k_bias, eps, rope_kind, cos_or_real, sin_or_imag, rope_lo, rope_hi, 1.f, stream);
}

// Rows of a flat [m, n] tensor are independent: view it as [H, S, n] with the largest H in {16, 8, 4, 2, 1} dividing m so that
// every lane keeps several rows in flight (the kernel walks "heads" with kUnroll loads outstanding).
static inline int rows_as_heads(int64_t m) {
    for (int h = 16; h > 1; h >>= 1)
        if (m % h == 0) return h;
    return 1;
}

// The five entry points of the reference extension (`_kernels`), same argument meaning.
extern "C" int svg_rms_norm_forward(void* x, const void* weight, int64_t m, int32_t n, int32_t dtype, float eps, void* stream) {
    if (!x || !weight || m <= 0 || m > 0x7fffffff) return SVG_ERR_BAD_ARG;
    const int h = rows_as_heads(m);
    return svg_qk_norm_rope(x, nullptr, 1, h, 0, (int32_t)(m / h), n, dtype, kNormRms, weight, nullptr, nullptr, nullptr, eps,
                            kRopeNone, nullptr, nullptr, 0, 0, stream);
}
extern "C" int svg_layer_norm_forward(void* x, const void* weight, const void* bias, int64_t m, int32_t n, int32_t dtype,
                                      void* stream) {
    if (!x || !weight || !bias || m <= 0 || m > 0x7fffffff) return SVG_ERR_BAD_ARG;
    const int h = rows_as_heads(m);
    return svg_qk_norm_rope(x, nullptr, 1, h, 0, (int32_t)(m / h), n, dtype, kNormLayer, weight, bias, nullptr, nullptr, 1e-5f,
                            kRopeNone, nullptr, nullptr, 0, 0, stream);
}
extern "C" int svg_apply_qk_rope_inplace_cossin(void* q, void* k, const float* cos_cache, const float* sin_cache, int32_t bsz,
                                                int32_t Hq, int32_t Hkv, int32_t S, int32_t D, int32_t dtype,
                                                int32_t len_text_prompt, void* stream) {
    if (!q || !k || len_text_prompt < 0 || len_text_prompt >= S) return SVG_ERR_BAD_ARG;
    return svg_qk_norm_rope(q, k, bsz, Hq, Hkv, S, D, dtype, kNormNone, nullptr, nullptr, nullptr, nullptr, 0.f, kRopeCosSin,
                            cos_cache, sin_cache, len_text_prompt, S, stream);
}
extern "C" int svg_apply_qk_rope_inplace_cossin_txtlast(void* q, void* k, const float* cos_cache, const float* sin_cache,
                                                        int32_t bsz, int32_t Hq, int32_t Hkv, int32_t S, int32_t D,
                                                        int32_t dtype, int32_t len_text_prompt, void* stream) {
    if (!q || !k || len_text_prompt < 0 || len_text_prompt >= S) return SVG_ERR_BAD_ARG;
    return svg_qk_norm_rope(q, k, bsz, Hq, Hkv, S, D, dtype, kNormNone, nullptr, nullptr, nullptr, nullptr, 0.f, kRopeCosSin,
                            cos_cache, sin_cache, 0, S - len_text_prompt, stream);
}
extern "C" int svg_apply_qk_rope_inplace_cossin_complex(void* q, void* k, const float* freqs_real, const float* freqs_imag,
                                                        int32_t bsz, int32_t Hq, int32_t Hkv, int32_t S, int32_t D,
                                                        int32_t dtype, int32_t len_text_prompt, void* stream) {
    if (!q || !k || len_text_prompt < 0 || len_text_prompt >= S) return SVG_ERR_BAD_ARG;
    return svg_qk_norm_rope(q, k, bsz, Hq, Hkv, S, D, dtype, kNormNone, nullptr, nullptr, nullptr, nullptr, 0.f, kRopeComplex,
                            freqs_real, freqs_imag, len_text_prompt, S, stream);
}
