// SVG1 band attention / dense attention on the one-wave-per-SIMD body (attn_w4.h): kernels and launchers.
// A translation unit of its own: the body is large (two unrolled tile iterations x three instantiations) and compiles in
// parallel with attention.hip.
#include "attn_w4.h"
#include "band_policy.h"

namespace svg {

template <typename T, int D>
using BandW4 = BandPolicy<T, D, 4, false, 0, 2>;   // 4 waves x 2 row blocks: 256-row q-tiles

template <typename T, int D>
__global__ __launch_bounds__(256, 1) void band_attn_w4_kernel(typename BandW4<T, D>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_w4<T, D, BandW4<T, D>>(prm, smem, nullptr);
}

#ifdef SVG_ABLATIONS
// diagnostics build only: the same kernel with the per-phase cycle trace (svg_band_attention variant 32, svg_debug_pp_trace)
template <typename T, int D, int ABL>
__global__ __launch_bounds__(256, 1) void band_attn_w4_trace_kernel(typename BandW4<T, D>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_w4<T, D, BandW4<T, D>, true, ABL>(prm, smem, nullptr);
}
#endif

template <typename T, int D>
static int run_w4(const void* q, const void* k, const void* v, void* o, int BH, int S, float sm_scale, const svg_band_mask_t* mask,
                  const svg_perm_desc_t* perm, const BandOpts& opts, hipStream_t st) {
    using Pol = BandW4<T, D>;
    const typename Pol::Params p = make_band_params<Pol, T>(q, k, v, o, BH, S, sm_scale, mask, perm, opts);
    if (opts.trace) {
#ifdef SVG_ABLATIONS
        if constexpr (D == 128 && std::is_same<T, __bf16>::value) {
#define SVG_W4_TRACE(A) case A: return launch_attn(band_attn_w4_trace_kernel<T, D, A>, p, dim3(p.nqt * BH), 256, attn_w4_lds_bytes<D>(), st);
            switch (opts.trace_abl) {
                SVG_W4_TRACE(0) SVG_W4_TRACE(1) SVG_W4_TRACE(2) SVG_W4_TRACE(3) SVG_W4_TRACE(4) SVG_W4_TRACE(5) SVG_W4_TRACE(6)
                default: return SVG_ERR_UNSUPPORTED;
            }
#undef SVG_W4_TRACE
        }
#endif
        return SVG_ERR_UNSUPPORTED;   // diagnostics builds only (-DSVG_ABLATIONS), bf16 / D = 128
    }
    return launch_attn(band_attn_w4_kernel<T, D>, p, dim3(p.nqt * BH), 256, attn_w4_lds_bytes<D>(), st);
}

#define SVG_W4_TD(FN, ...)                                                                      \
    if (dtype == SVG_DTYPE_BF16 && D == 128) return FN<__bf16, 128>(__VA_ARGS__);               \
    if (dtype == SVG_DTYPE_BF16 && D == 64) return FN<__bf16, 64>(__VA_ARGS__);                 \
    if (dtype == SVG_DTYPE_F16 && D == 128) return FN<_Float16, 128>(__VA_ARGS__);              \
    if (dtype == SVG_DTYPE_F16 && D == 64) return FN<_Float16, 64>(__VA_ARGS__);                \
    return SVG_ERR_UNSUPPORTED;

int run_band_w4(const void* q, const void* k, const void* v, void* o, int BH, int S, int D, int dtype, float sm_scale,
                const svg_band_mask_t* mask, const svg_perm_desc_t* perm, const BandOpts& opts, hipStream_t st) {
    SVG_W4_TD(run_w4, q, k, v, o, BH, S, sm_scale, mask, perm, opts, st)
}

#undef SVG_W4_TD

int w4_read_trace(uint64_t* out104) {
#ifdef SVG_ABLATIONS
    hipError_t e = hipMemcpyFromSymbol(out104, HIP_SYMBOL(g_pp_trace), 104 * sizeof(uint64_t));
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return SVG_ERR_LAUNCH;
    }
    return SVG_OK;
#else
    (void)out104;
    return SVG_ERR_UNSUPPORTED;
#endif
}

}  // namespace svg
