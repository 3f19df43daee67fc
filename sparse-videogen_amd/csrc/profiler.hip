// Online profiler of SVG1 (`sample_mse`): for R sampled query rows per head compute the dense attention output
// ("golden") and the outputs under the two candidate masks, return the per-head mean squared errors.
// ref: svg/models/hyvideo/attention.py:376-399 (wan/attention.py:211-234, cog/attention.py:120-145) and the
// profiling masks of get_attention_mask (hyvideo/utils.py:47-93, wan/utils.py:63-110, cog/utils.py:61-88).
//
// The reference materialises two [10000, S] fp32 masks (4.8 GB each) and runs three full softmaxes in torch.
// Here the masks are analytic predicates and the three variants are three workgroup roles of ONE launch that
// share attn_core.h: grid = (kv_chunks, BH, 3); every workgroup streams its chunk of K/V once (split-KV), emits
// un-normalised fp32 partials (O, m, l) per sampled row, and a small second kernel merges the chunks, normalises
// and reduces the MSE.  K/V of a head are therefore read once from HBM and twice more from L2/MALL.
#include "attn_core.h"

namespace svg {

struct ProfVariant {
    int coord;        // 0: indices as stored (frame-major), 1: token-major (p*F + f) inside the video range
    int origin;       // subtracted from the coordinate before blocking
    int span;         // band domain: 0 <= x - origin < span
    int band_blocks;  // |floor(x/128) - floor(y/128)| < band_blocks
    int sink_cols;    // y < sink_cols always visible (in the variant's coordinate)
    int text_lo, text_hi;  // rows / cols in [text_lo, text_hi) are all-ones (empty when lo >= hi)
};

constexpr int kProfNW = 4;      // waves per workgroup (rows 64.. of the 128-row tile simply do not exist)
constexpr int kProfMaxRows = 64;

template <typename T, int D>
struct ProfilePolicy {
    static constexpr bool kFixup = true;
    static constexpr bool kPartialOut = true;
    static constexpr int kAbl = 0;
    static constexpr bool kSetPrio = false;
    static constexpr bool kSkew = false;
    static constexpr int NW = kProfNW;

    struct Params {
        const T* q;
        const T* k;
        const T* v;
        int S, BH, R, n_chunks, tiles_per_chunk;
        float scale_log2;   // log2(e) when emulate (scale folded into score_fixup), else scale*log2(e)
        float fix_scale;    // sm_scale when emulate
        int emulate;
        const int64_t* rows;
        int vid0, F, P, V;
        ProfVariant var[2];
        float* part;        // [3][BH][n_chunks][kProfMaxRows][D + 4]
    };
    struct Ctx {
        int head, variant, chunk, t0, nT;
        ProfVariant pv;
    };
    struct KvCursor {};

    static __device__ __forceinline__ bool init(const Params& p, Ctx& c, char*) {
        c.chunk = blockIdx.x;
        c.head = blockIdx.y;
        c.variant = blockIdx.z;
        const int ntiles = (p.S + kBN - 1) / kBN;
        c.t0 = c.chunk * p.tiles_per_chunk;
        c.nT = max(0, min(p.tiles_per_chunk, ntiles - c.t0));
        c.pv = p.var[c.variant == 2 ? 1 : 0];
        return true;
    }
    static __device__ __forceinline__ const T* q_base(const Params& p, const Ctx& c) { return p.q + (size_t)c.head * p.S * D; }
    static __device__ __forceinline__ const T* k_base(const Params& p, const Ctx& c) { return p.k + (size_t)c.head * p.S * D; }
    static __device__ __forceinline__ const T* v_base(const Params& p, const Ctx& c) { return p.v + (size_t)c.head * p.S * D; }

    // logical q index of a sampled row = its physical row
    static __device__ __forceinline__ int q_phys(const Params& p, const Ctx&, int row) {
        return row < p.R ? (int)p.rows[row] : -1;
    }
    static __device__ __forceinline__ int q_logical(const Ctx&, int row) { return row; }  // resolved in allowed()
    static __device__ __forceinline__ int tile_key0(const Ctx& c, int t) { return (c.t0 + t) * kBN; }
    static __device__ __forceinline__ void kv_cursor_init(const Params&, const Ctx&, KvCursor&, int) {}
    static __device__ __forceinline__ int kv_phys(const Params& p, const Ctx& c, KvCursor&, int t, int row) {
        const int l = (c.t0 + t) * kBN + row;
        return l < p.S ? l : 0;  // masked by allowed()
    }
    static __device__ __forceinline__ int classify(const Params& p, const Ctx& c, int k0, int wrow0) {
        if (wrow0 >= p.R) return TILE_SKIP;
        if (c.variant == 0) return (k0 + kBN <= p.S) ? TILE_FULL : TILE_PARTIAL;
        return TILE_PARTIAL;
    }
    static __device__ __forceinline__ int coord(const Params& p, const ProfVariant& pv, int i) {
        if (pv.coord == 1) {
            const unsigned r = (unsigned)(i - p.vid0);
            if (r < (unsigned)p.V) {
                const unsigned f = r / (unsigned)p.P;
                const unsigned pp = r - f * (unsigned)p.P;
                return p.vid0 + (int)(pp * (unsigned)p.F + f);
            }
        }
        return i;
    }
    static __device__ __forceinline__ bool allowed(const Params& p, const Ctx& c, int qrow, int k) {
        if (k >= p.S) return false;
        if (c.variant == 0) return true;
        const ProfVariant& pv = c.pv;
        const int q = qrow < p.R ? (int)p.rows[qrow] : 0;
        const bool tq = (unsigned)(q - pv.text_lo) < (unsigned)(pv.text_hi - pv.text_lo);
        const bool tk = (unsigned)(k - pv.text_lo) < (unsigned)(pv.text_hi - pv.text_lo);
        const int x = coord(p, pv, q) - pv.origin;
        const int y = coord(p, pv, k) - pv.origin;
        const bool dom = ((unsigned)x < (unsigned)pv.span) & ((unsigned)y < (unsigned)pv.span);
        const int db = (x >> 7) - (y >> 7);
        const bool band = (db < pv.band_blocks) & (-db < pv.band_blocks);
        const bool sink = y < pv.sink_cols;
        return tq | tk | (dom & (band | sink));
    }
    static __device__ __forceinline__ float score_fixup(const Params& p, float s) {
        if (!p.emulate) return s;
        // torch: (q @ k^T) rounds to the input dtype, "/ sqrt(D)" rounds again (attention.py:383)
        const float r1 = Elt<T>::to_float(Elt<T>::from_float(s));
        return Elt<T>::to_float(Elt<T>::from_float(r1 * p.fix_scale));
    }
    static __device__ __forceinline__ void store_partial(const Params& p, const Ctx& c, int row, int g, const f32x16* acc,
                                                         float m, float l) {
        if (row >= p.R) return;
        float* dst = p.part + ((((size_t)c.variant * p.BH + c.head) * p.n_chunks + c.chunk) * kProfMaxRows + row) * (D + 4);
#pragma unroll
        for (int db = 0; db < D / 32; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 v4 = {acc[db][rq * 4 + 0], acc[db][rq * 4 + 1], acc[db][rq * 4 + 2], acc[db][rq * 4 + 3]};
                *(f32x4*)(dst + 32 * db + 8 * rq + 4 * g) = v4;
            }
        if (g == 0) {
            dst[D] = m;
            dst[D + 1] = l;
        }
    }
    static __device__ __forceinline__ T* o_base(const Params&, const Ctx&) { return nullptr; }
};

template <typename T, int D>
__global__ __launch_bounds__(kProfNW * 64, 2) void profile_attn_kernel(typename ProfilePolicy<T, D>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body<T, D, kProfNW, ProfilePolicy<T, D>>(prm, smem, nullptr);
}

// merge the split-KV partials, normalise, and reduce the two MSEs of one head.  grid = (BH), block = 256
template <typename T, int D>
__global__ __launch_bounds__(256) void profile_combine_kernel(const float* __restrict__ part, float* __restrict__ out_mse, int BH,
                                                              int R, int n_chunks, int emulate) {
    __shared__ float red[2][4];
    const int h = blockIdx.x, tid = threadIdx.x;
    float sq[2] = {0.f, 0.f};
    bool bad[2] = {false, false};
    for (int e = tid; e < R * D; e += 256) {
        const int row = e / D, d = e - row * D;
        float o[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const float* base = part + ((((size_t)v * BH + h) * n_chunks) * kProfMaxRows + row) * (D + 4);
            const size_t cs = (size_t)kProfMaxRows * (D + 4);
            float M = -INFINITY;
            for (int c = 0; c < n_chunks; ++c) M = fmaxf(M, base[c * cs + D]);
            float L = 0.f, acc = 0.f;
            for (int c = 0; c < n_chunks; ++c) {
                const float mc = base[c * cs + D];
                const float w = (mc == -INFINITY) ? 0.f : exp2f(mc - M);
                L += base[c * cs + D + 1] * w;
                acc += base[c * cs + d] * w;
            }
            // a row whose mask admits no key is NaN in the reference (softmax over all -inf)
            o[v] = (L > 0.f) ? acc / L : __builtin_nanf("");
            if (emulate) o[v] = Elt<T>::to_float(Elt<T>::from_float(o[v]));
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            float diff = o[v + 1] - o[0];
            if (emulate) {
                diff = Elt<T>::to_float(Elt<T>::from_float(diff));
                float s2 = diff * diff;
                s2 = Elt<T>::to_float(Elt<T>::from_float(s2));
                sq[v] += s2;
            } else {
                sq[v] += diff * diff;
            }
            bad[v] |= (diff != diff);
        }
    }
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        float s = bad[v] ? __builtin_nanf("") : sq[v];
        s = wave_sum(s);
        if ((tid & 63) == 0) red[v][tid >> 6] = s;
    }
    __syncthreads();
    if (tid < 2) {
        float s = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
        s = s / (float)(R * D);
        if (emulate) s = Elt<T>::to_float(Elt<T>::from_float(s));
        out_mse[(size_t)tid * BH + h] = s;
    }
}

static int prof_chunks(int BH, int S) {
    const int ntiles = (S + kBN - 1) / kBN;
    int n = (4 * kNumCU + 3 * BH - 1) / (3 * BH);
    n = n < 1 ? 1 : n;
    n = n > ntiles ? ntiles : n;
    n = n > 64 ? 64 : n;
    return n;
}

template <typename T, int D>
static int run_profile(const void* q, const void* k, const void* v, const int64_t* rows, int R, int BH, int S, float sm_scale,
                       const svg_profile_desc_t* pd, float* out_mse, void* ws, hipStream_t st) {
    using Pol = ProfilePolicy<T, D>;
    typename Pol::Params p;
    p.q = (const T*)q, p.k = (const T*)k, p.v = (const T*)v;
    p.S = S, p.BH = BH, p.R = R;
    p.n_chunks = prof_chunks(BH, S);
    const int ntiles = (S + kBN - 1) / kBN;
    p.tiles_per_chunk = (ntiles + p.n_chunks - 1) / p.n_chunks;
    p.emulate = pd->emulate_bf16;
    p.scale_log2 = p.emulate ? 1.4426950408889634f : sm_scale * 1.4426950408889634f;
    p.fix_scale = sm_scale;
    p.rows = rows;
    p.vid0 = pd->vid0, p.F = pd->num_frame, p.P = pd->frame_size, p.V = pd->num_frame * pd->frame_size;
    for (int i = 0; i < 2; ++i) {
        p.var[i].coord = pd->variant[i].coord;
        p.var[i].origin = pd->variant[i].origin;
        p.var[i].span = pd->variant[i].span;
        p.var[i].band_blocks = pd->variant[i].band_blocks;
        p.var[i].sink_cols = pd->variant[i].sink_cols;
        p.var[i].text_lo = pd->variant[i].text_lo;
        p.var[i].text_hi = pd->variant[i].text_hi;
    }
    p.part = (float*)ws;
    const int lds = attn_lds_bytes<D, kProfNW>();
    auto kern = profile_attn_kernel<T, D>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return SVG_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(p.n_chunks, BH, 3), dim3(kProfNW * 64), lds, st, p);
    hipLaunchKernelGGL((profile_combine_kernel<T, D>), dim3(BH), dim3(256), 0, st, (const float*)ws, out_mse, BH, R,
                       p.n_chunks, p.emulate);
    return launch_status();
}

}  // namespace svg

using namespace svg;

extern "C" size_t svg_sample_mse_workspace_bytes(int32_t BH, int32_t R, int32_t D, int32_t S) {
    if (BH <= 0 || R <= 0 || D <= 0 || S <= 0) return 0;
    return (size_t)3 * BH * prof_chunks(BH, S) * kProfMaxRows * (D + 4) * sizeof(float);
}

extern "C" int svg_sample_mse(const void* q, const void* k, const void* v, const int64_t* rows, int32_t R, int32_t BH,
                              int32_t S, int32_t D, int32_t dtype, float sm_scale, const svg_profile_desc_t* prof,
                              float* out_mse, void* workspace, size_t workspace_bytes, void* stream) {
    if (!q || !k || !v || !rows || !prof || !out_mse || !workspace) return SVG_ERR_BAD_ARG;
    if (R <= 0 || BH <= 0 || S <= 0) return SVG_ERR_BAD_ARG;
    if (R > kProfMaxRows) return SVG_ERR_UNSUPPORTED;
    if (workspace_bytes < svg_sample_mse_workspace_bytes(BH, R, D, S)) return SVG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SVG_DTYPE_BF16) {
        if (D == 128) return run_profile<__bf16, 128>(q, k, v, rows, R, BH, S, sm_scale, prof, out_mse, workspace, st);
        if (D == 64) return run_profile<__bf16, 64>(q, k, v, rows, R, BH, S, sm_scale, prof, out_mse, workspace, st);
    } else if (dtype == SVG_DTYPE_F16) {
        if (D == 128) return run_profile<_Float16, 128>(q, k, v, rows, R, BH, S, sm_scale, prof, out_mse, workspace, st);
        if (D == 64) return run_profile<_Float16, 64>(q, k, v, rows, R, BH, S, sm_scale, prof, out_mse, workspace, st);
    }
    return SVG_ERR_UNSUPPORTED;
}
