// Transformer-block glue around the attention layer-call: LayerNorm over the hidden size in fp32,
// modulate  y = x * (1 + scale) + shift,  gate-residual  y = residual + x * gate,  and the fused LayerNorm + modulate.
// HBM-bound row operations: one wave per row, the row lives in registers between the statistics and the output pass.
// ref: svg/kernels/triton/layernorm.py:13-216 (triton_layernorm_forward), svg/kernels/triton/modulate.py:13-164
//      (triton_modulate_shift_forward, triton_modulate_gate_residual_forward); call sites svg/models/wan/custom_models.py:37-111.
#include "svg_common.h"

#pragma clang fp contract(off)

namespace svg {

struct GlueParams {
    const void* x;
    const void* r;       // residual (gate-residual only)
    void* y;
    const void* w;       // LayerNorm weight / bias [N] (null: no affine)
    const void* b;
    const float* scale;  // [M / rows_per_batch, N] fp32 (null: no modulate)
    const float* shift;
    const float* gate;
    int M, N, rows_per_batch;
    int x_dt, r_dt, y_dt, w_dt;
    float eps;
    int do_ln;           // 0: none, 1: LayerNorm, 2: RMSNorm (the reference's Triton form: fp32 x * rstd * w, ONE rounding; rmsnorm.py:8-48)
    float pad_cols;      // 0: LayerNorm.  > 0: the reference's Triton kernels' variance, which counts the zero padding of the row up to
                         // the next power of two — (0 - mean)^2 for N2 - N columns (svg/kernels/triton/layernorm.py:35-41): opt-in
};

__device__ __forceinline__ void load8(const void* base, size_t elem, int dt, float (&v)[8]) {
    if (dt == SVG_DTYPE_F32) {
        const f32x4* p = (const f32x4*)((const float*)base + elem);
        const f32x4 a = p[0], c = p[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = a[j], v[4 + j] = c[j];
    } else if (dt == SVG_DTYPE_BF16) {
        const bf16x8 a = *(const bf16x8*)((const __bf16*)base + elem);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
    } else {
        const f16x8 a = *(const f16x8*)((const _Float16*)base + elem);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
    }
}
__device__ __forceinline__ void store8(void* base, size_t elem, int dt, const float (&v)[8]) {
    if (dt == SVG_DTYPE_F32) {
        f32x4 a, c;
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = v[j], c[j] = v[4 + j];
        f32x4* p = (f32x4*)((float*)base + elem);
        p[0] = a, p[1] = c;
    } else if (dt == SVG_DTYPE_BF16) {
        bf16x8 a;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (__bf16)v[j];
        *(bf16x8*)((__bf16*)base + elem) = a;
    } else {
        f16x8 a;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (_Float16)v[j];
        *(f16x8*)((_Float16*)base + elem) = a;
    }
}

// one wave per row; lane l owns the 8-element chunks l, l + 64, ... (NCH of them at most)
// (The modulated forms take scale / shift from an LDS copy when the launch allows it: row_glue_lds_kernel below.)
// LN: p.do_ln as a compile-time constant — with the three normalisation forms behind run-time branches of one kernel the hidden-size-5120
// instance needed 182 registers (two waves per SIMD) where the LayerNorm-only kernel of round 1 had 114 (four): svg_layernorm_modulate_forward
// fell from 0.41 to 0.62 ms at Wan 720p (profiles/r01f_bench_glue.json, r05n_bench.json hbm_kernels).
// MOD_LDS: scale / shift come from the workgroup's LDS copy `mod` ((1 + scale) and shift as 16-byte pieces, piece h of chunk c at
// (h * nchunks + c) for (1 + scale), behind 2 * nchunks pieces for shift: consecutive lanes read consecutive pieces).
template <int NCH, int LN, bool MOD_LDS>
__device__ __forceinline__ void row_glue_body(const GlueParams& p, int row, int lane, const f32x4* mod) {
    const int nchunks = p.N / 8;
    const size_t rbase = (size_t)row * p.N;
    float x[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) load8(p.x, rbase + (size_t)c * 8, p.x_dt, x[i]);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[i][j] = 0.f;
        }
    }
    if constexpr (LN == 2) {
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s2 += x[i][j] * x[i][j];      // (chunks behind the row hold zeros)
        const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)p.N + p.eps);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
                float w[8];
                if (p.w) load8(p.w, (size_t)c * 8, p.w_dt, w);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = x[i][j] * rstd;
                    x[i][j] = p.w ? xh * w[j] : xh;
                }
            }
        }
    } else if constexpr (LN == 1) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += x[i][j];
        const float mean = wave_sum(s) / (float)p.N;
        float vs = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) vs += (x[i][j] - mean) * (x[i][j] - mean);
            }
        }
        // (pad_cols == 0: + 0 * mean^2 is exact, the result is bit for bit what it was without the term)
        const float rstd = 1.0f / sqrtf((wave_sum(vs) + p.pad_cols * (mean * mean)) / (float)p.N + p.eps);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
                float w[8], b[8];
                if (p.w) {
                    load8(p.w, (size_t)c * 8, p.w_dt, w);
                    load8(p.b, (size_t)c * 8, p.w_dt, b);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (x[i][j] - mean) * rstd;
                    x[i][j] = p.w ? xh * w[j] + b[j] : xh;
                }
            }
        }
    }
    const size_t mrow = (size_t)(row / p.rows_per_batch) * p.N;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c >= nchunks) continue;
        if constexpr (MOD_LDS) {
            const f32x4 s0 = mod[c], s1 = mod[nchunks + c], h0 = mod[2 * nchunks + c], h1 = mod[3 * nchunks + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[i][j] = x[i][j] * s0[j] + h0[j], x[i][4 + j] = x[i][4 + j] * s1[j] + h1[j];
            store8(p.y, rbase + (size_t)c * 8, p.y_dt, x[i]);
            __builtin_amdgcn_sched_barrier(0);   // (or every chunk's four LDS reads are hoisted to the top: 16 registers each, 218 at hidden size 5120)
            continue;
        } else if (p.scale) {
            float sc[8], sh[8];
            load8(p.scale, mrow + (size_t)c * 8, SVG_DTYPE_F32, sc);
            load8(p.shift, mrow + (size_t)c * 8, SVG_DTYPE_F32, sh);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[i][j] = x[i][j] * (1.0f + sc[j]) + sh[j];
        }
        store8(p.y, rbase + (size_t)c * 8, p.y_dt, x[i]);
    }
}

template <int NCH, int LN>
__global__ __launch_bounds__(256) void row_glue_kernel(GlueParams p) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    row_glue_body<NCH, LN, false>(p, row, threadIdx.x & 63, nullptr);
}

// The modulated forms (round 5): a row's scale and shift are 64 bytes of fp32 per 16-byte chunk of x — three times the row's own traffic through the
// vector memory path, every row again: svg_modulate_shift_forward alone took 0.397 ms at Wan's [75600, 5120] where LayerNorm alone takes 0.314 (a
// copy 0.308).  Here a workgroup of 8 waves copies (1 + scale) and shift of its batch to LDS (8 N bytes) once for its 8 rows, one wave per row as
// above: an eighth of that traffic.  The same numbers: 1 + scale is formed once per workgroup instead of per row.
constexpr int kGlueLdsWaves = 8;
template <int NCH, int LN>
__global__ __launch_bounds__(kGlueLdsWaves * 64) void row_glue_lds_kernel(GlueParams p) {
    extern __shared__ __attribute__((aligned(16))) char glue_smem[];
    f32x4* mod = (f32x4*)glue_smem;
    const int nchunks = p.N / 8, batch = blockIdx.y;
    const f32x4* sc = (const f32x4*)(p.scale + (size_t)batch * p.N);
    const f32x4* sh = (const f32x4*)(p.shift + (size_t)batch * p.N);
    for (int i = threadIdx.x; i < 2 * nchunks; i += kGlueLdsWaves * 64) {    // piece i of the row: chunk i / 2, half i % 2
        const f32x4 a = sc[i];
        mod[(i & 1) * nchunks + (i >> 1)] = f32x4{1.0f + a[0], 1.0f + a[1], 1.0f + a[2], 1.0f + a[3]};
        mod[2 * nchunks + (i & 1) * nchunks + (i >> 1)] = sh[i];
    }
    __syncthreads();
    // (one row per wave, no row loop: with a grid-stride loop around the body hipcc needs 218 registers at hidden size 5120 instead of 116)
    const int row = batch * p.rows_per_batch + blockIdx.x * kGlueLdsWaves + (threadIdx.x >> 6);
    if (row < (batch + 1) * p.rows_per_batch) row_glue_body<NCH, LN, true>(p, row, threadIdx.x & 63, mod);
}

// y = residual + x * gate, 8 elements per thread
__global__ __launch_bounds__(256) void gate_residual_kernel(GlueParams p) {
    const size_t chunk = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int cpr = p.N / 8;
    if (chunk >= (size_t)p.M * cpr) return;
    const int row = (int)(chunk / cpr);
    const int c = (int)(chunk - (size_t)row * cpr);
    float r[8], x[8], g[8];
    load8(p.r, chunk * 8, p.r_dt, r);
    load8(p.x, chunk * 8, p.x_dt, x);
    load8(p.gate, (size_t)(row / p.rows_per_batch) * p.N + (size_t)c * 8, SVG_DTYPE_F32, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = r[j] + x[j] * g[j];
    store8(p.y, chunk * 8, p.y_dt, r);
}

static bool dt_ok(int dt) { return dt == SVG_DTYPE_BF16 || dt == SVG_DTYPE_F16 || dt == SVG_DTYPE_F32; }

static int launch_row_glue(const GlueParams& p, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.N % 8 != 0 || p.N > 8192 || p.rows_per_batch <= 0) return p.N > 8192 ? SVG_ERR_UNSUPPORTED : SVG_ERR_BAD_ARG;
    if (!dt_ok(p.x_dt) || !dt_ok(p.y_dt)) return SVG_ERR_UNSUPPORTED;
    const int need = (p.N / 8 + 63) / 64;
    const dim3 grid((p.M + 3) / 4), block(256);
    // modulated, a workgroup full of rows per batch, and the LDS copy fits three times into a CU: the LDS form
    const int lds_bytes = 2 * p.N * (int)sizeof(float);
    // (batches ride grid.y: more than 65535 of them — tiny batches of a huge M — take the plain form below, as before round 5)
    if (p.scale && p.rows_per_batch >= kGlueLdsWaves && lds_bytes <= 49152 && p.M / p.rows_per_batch <= 65535) {
        const int batches = p.M / p.rows_per_batch;
        const int per_batch = (p.rows_per_batch + kGlueLdsWaves - 1) / kGlueLdsWaves;
        const dim3 g2(per_batch, batches), b2(kGlueLdsWaves * 64);
#define SVG_GLUE_LDS(NC)                                                                                      \
    if (need <= NC) {                                                                                         \
        if (p.do_ln == 2) hipLaunchKernelGGL((row_glue_lds_kernel<NC, 2>), g2, b2, lds_bytes, st, p);         \
        else if (p.do_ln == 1) hipLaunchKernelGGL((row_glue_lds_kernel<NC, 1>), g2, b2, lds_bytes, st, p);    \
        else hipLaunchKernelGGL((row_glue_lds_kernel<NC, 0>), g2, b2, lds_bytes, st, p);                      \
        return launch_status();                                                                               \
    }
        SVG_GLUE_LDS(2) SVG_GLUE_LDS(4) SVG_GLUE_LDS(6) SVG_GLUE_LDS(8) SVG_GLUE_LDS(10) SVG_GLUE_LDS(12) SVG_GLUE_LDS(16)
#undef SVG_GLUE_LDS
    }
#define SVG_GLUE(NC)                                                                                          \
    if (need <= NC) {                                                                                         \
        if (p.do_ln == 2) hipLaunchKernelGGL((row_glue_kernel<NC, 2>), grid, block, 0, st, p);                \
        else if (p.do_ln == 1) hipLaunchKernelGGL((row_glue_kernel<NC, 1>), grid, block, 0, st, p);           \
        else hipLaunchKernelGGL((row_glue_kernel<NC, 0>), grid, block, 0, st, p);                             \
        return launch_status();                                                                               \
    }
    SVG_GLUE(2) SVG_GLUE(4) SVG_GLUE(6) SVG_GLUE(8) SVG_GLUE(10) SVG_GLUE(12) SVG_GLUE(16)
#undef SVG_GLUE
    return SVG_ERR_UNSUPPORTED;
}

}  // namespace svg

using namespace svg;

// columns the reference's Triton kernels pad a row of N with: next_power_of_2(N) - N (layernorm.py:76)
static float reference_pad_cols(int N, int reference_padding) {
    if (!reference_padding || N <= 0) return 0.f;
    int n2 = 1;
    while (n2 < N) n2 <<= 1;
    return (float)(n2 - N);
}

extern "C" int svg_layernorm_forward_ex(const void* x, void* y, const void* weight, const void* bias, int64_t M, int32_t N,
                                        int32_t x_dtype, int32_t y_dtype, int32_t w_dtype, float eps, int32_t reference_padding,
                                        void* stream) {
    if (!x || !y || (weight == nullptr) != (bias == nullptr) || M <= 0 || M > 0x7fffffff) return SVG_ERR_BAD_ARG;
    if (weight && !dt_ok(w_dtype)) return SVG_ERR_UNSUPPORTED;
    GlueParams p{};
    p.x = x, p.y = y, p.w = weight, p.b = bias, p.M = (int)M, p.N = N, p.rows_per_batch = (int)M;
    p.x_dt = x_dtype, p.y_dt = y_dtype, p.w_dt = w_dtype, p.eps = eps, p.do_ln = 1;
    p.pad_cols = reference_pad_cols(N, reference_padding);
    return launch_row_glue(p, (hipStream_t)stream);
}

extern "C" int svg_layernorm_forward(const void* x, void* y, const void* weight, const void* bias, int64_t M, int32_t N,
                                     int32_t x_dtype, int32_t y_dtype, int32_t w_dtype, float eps, void* stream) {
    return svg_layernorm_forward_ex(x, y, weight, bias, M, N, x_dtype, y_dtype, w_dtype, eps, 0, stream);
}

// RMSNorm over the last dimension of [M, N] as the reference's Triton kernel computes it (svg/kernels/triton/rmsnorm.py:8-48, the
// q / k normalisation of its Wan processors, wan/attention.py:105-120): fp32 statistics, y = T(x * rstd * w) — ONE rounding, unlike
// diffusers' RMSNorm (and svg_rms_norm_forward, the head_dim-wide QK norm of the other models), which round before the weight.
extern "C" int svg_rmsnorm_forward(const void* x, void* y, const void* weight, int64_t M, int32_t N, int32_t x_dtype, int32_t y_dtype,
                                   int32_t w_dtype, float eps, void* stream) {
    if (!x || !y || M <= 0 || M > 0x7fffffff) return SVG_ERR_BAD_ARG;
    if (weight && !dt_ok(w_dtype)) return SVG_ERR_UNSUPPORTED;
    GlueParams p{};
    p.x = x, p.y = y, p.w = weight, p.M = (int)M, p.N = N, p.rows_per_batch = (int)M;
    p.x_dt = x_dtype, p.y_dt = y_dtype, p.w_dt = w_dtype, p.eps = eps, p.do_ln = 2;
    return launch_row_glue(p, (hipStream_t)stream);
}

extern "C" int svg_modulate_shift_forward(const void* x, void* y, const float* scale, const float* shift, int64_t M, int32_t N,
                                          int64_t rows_per_batch, int32_t x_dtype, int32_t y_dtype, void* stream) {
    if (!x || !y || !scale || !shift || M <= 0 || M > 0x7fffffff || rows_per_batch <= 0 || M % rows_per_batch != 0) return SVG_ERR_BAD_ARG;
    GlueParams p{};
    p.x = x, p.y = y, p.scale = scale, p.shift = shift, p.M = (int)M, p.N = N, p.rows_per_batch = (int)rows_per_batch;
    p.x_dt = x_dtype, p.y_dt = y_dtype, p.do_ln = 0;
    return launch_row_glue(p, (hipStream_t)stream);
}

extern "C" int svg_layernorm_modulate_forward_ex(const void* x, void* y, const void* weight, const void* bias, const float* scale,
                                                 const float* shift, int64_t M, int32_t N, int64_t rows_per_batch, int32_t x_dtype,
                                                 int32_t y_dtype, int32_t w_dtype, float eps, int32_t reference_padding, void* stream) {
    if (!x || !y || (weight == nullptr) != (bias == nullptr) || (scale == nullptr) != (shift == nullptr)) return SVG_ERR_BAD_ARG;
    if (M <= 0 || M > 0x7fffffff || rows_per_batch <= 0 || M % rows_per_batch != 0) return SVG_ERR_BAD_ARG;
    if (weight && !dt_ok(w_dtype)) return SVG_ERR_UNSUPPORTED;
    GlueParams p{};
    p.x = x, p.y = y, p.w = weight, p.b = bias, p.scale = scale, p.shift = shift, p.M = (int)M, p.N = N;
    p.rows_per_batch = (int)rows_per_batch, p.x_dt = x_dtype, p.y_dt = y_dtype, p.w_dt = w_dtype, p.eps = eps, p.do_ln = 1;
    p.pad_cols = reference_pad_cols(N, reference_padding);
    return launch_row_glue(p, (hipStream_t)stream);
}

extern "C" int svg_layernorm_modulate_forward(const void* x, void* y, const void* weight, const void* bias, const float* scale,
                                              const float* shift, int64_t M, int32_t N, int64_t rows_per_batch, int32_t x_dtype,
                                              int32_t y_dtype, int32_t w_dtype, float eps, void* stream) {
    return svg_layernorm_modulate_forward_ex(x, y, weight, bias, scale, shift, M, N, rows_per_batch, x_dtype, y_dtype, w_dtype, eps, 0,
                                             stream);
}

extern "C" int svg_modulate_gate_residual_forward(const void* residual, const void* x, const float* gate, void* y, int64_t M,
                                                  int32_t N, int64_t rows_per_batch, int32_t r_dtype, int32_t x_dtype,
                                                  int32_t y_dtype, void* stream) {
    if (!residual || !x || !gate || !y || M <= 0 || M > 0x7fffffff || N <= 0 || N % 8 != 0) return SVG_ERR_BAD_ARG;
    if (rows_per_batch <= 0 || M % rows_per_batch != 0) return SVG_ERR_BAD_ARG;
    if (!dt_ok(r_dtype) || !dt_ok(x_dtype) || !dt_ok(y_dtype)) return SVG_ERR_UNSUPPORTED;
    GlueParams p{};
    p.r = residual, p.x = x, p.gate = gate, p.y = y, p.M = (int)M, p.N = N, p.rows_per_batch = (int)rows_per_batch;
    p.r_dt = r_dtype, p.x_dt = x_dtype, p.y_dt = y_dtype;
    const size_t chunks = (size_t)M * (N / 8);
    hipLaunchKernelGGL(gate_residual_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status();
}
