// Library bookkeeping (error strings, build info) and the small map-density reduction.
#include "svg_common.h"

namespace svg {

// ref: density_calculation, svg/kmeans_utils.py:13-31.  One workgroup per head.
__global__ __launch_bounds__(256) void map_density_kernel(const uint8_t* __restrict__ map, const int32_t* __restrict__ q_sizes,
                                                          const int32_t* __restrict__ k_sizes, float* __restrict__ out, int QB,
                                                          int KB) {
    __shared__ double red[2][4];
    const int h = blockIdx.x, tid = threadIdx.x;
    const uint8_t* m = map + (size_t)h * QB * KB;
    const int32_t* qs = q_sizes + (size_t)h * QB;
    const int32_t* ks = k_sizes + (size_t)h * KB;
    double num = 0.0, den = 0.0;
    for (int idx = tid; idx < QB * KB; idx += 256) {
        const int i = idx / KB, j = idx - i * KB;
        const double blk = (double)qs[i] * (double)ks[j];
        den += blk;
        if (m[idx]) num += blk;
    }
    for (int o = 32; o >= 1; o >>= 1) {
        num += __shfl_xor(num, o);
        den += __shfl_xor(den, o);
    }
    if ((tid & 63) == 0) red[0][tid >> 6] = num, red[1][tid >> 6] = den;
    __syncthreads();
    if (tid == 0) {
        const double n = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const double d = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        out[h] = (float)(n / d);
    }
}

// BSR (indptr, indices) of MB x NB uniform blocks -> the dense per-head block map + size arrays the variable-block kernel
// takes, with one extra leading block-row / block-column for the text tokens (always active).  grid = (MB + 1), block = 256.
__global__ __launch_bounds__(256) void bsr_to_map_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                         uint8_t* __restrict__ map, int32_t* __restrict__ q_sizes,
                                                         int32_t* __restrict__ k_sizes, int MB, int NB, int R, int Cc, int text,
                                                         int heads) {
    const int i = blockIdx.x, tid = threadIdx.x;   // block-row i of the extended map (0 = text)
    const int KB = NB + 1, QB = MB + 1;
    for (int h = 0; h < heads; ++h) {
        uint8_t* row = map + ((size_t)h * QB + i) * KB;
        for (int j = tid; j < KB; j += 256) row[j] = (i == 0 || j == 0) ? 1 : 0;
        if (tid == 0) q_sizes[(size_t)h * QB + i] = i == 0 ? text : R;
        if (i == 0)
            for (int j = tid; j < KB; j += 256) k_sizes[(size_t)h * KB + j] = j == 0 ? text : Cc;
    }
    __syncthreads();
    if (i > 0) {
        const int lo = indptr[i - 1], hi = indptr[i];
        for (int h = 0; h < heads; ++h) {
            uint8_t* row = map + ((size_t)h * QB + i) * KB;
            for (int p = lo + tid; p < hi; p += 256) {
                const int j = indices[p];
                if (j >= 0 && j < NB) row[1 + j] = 1;
            }
        }
    }
}

// Shader-clock probe: one wave sleeps until `stop[0]` becomes non-zero (or `max_ticks` of the constant 100 MHz counter have
// passed — it can never hang a queue) and reports how far the shader-clock counter (s_memtime) and the 100 MHz counter
// (wall_clock64) advanced meanwhile: sclk = 100 MHz * d_shader / d_wall, the clock the chip actually sustained while whatever ran
// beside this wave was running (bench.py: the timed steps).  The wave issues one s_sleep per ~64 clocks: no measurable load.
__global__ __launch_bounds__(64) void clock_probe_kernel(const int32_t* __restrict__ stop, unsigned long long* __restrict__ out,
                                                         long long max_ticks) {
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
    while (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && (long long)(wall_clock64() - w0) < max_ticks)
        __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) {
        out[0] = __builtin_amdgcn_s_memtime() - c0;
        out[1] = wall_clock64() - w0;
    }
}

}  // namespace svg

extern "C" int svg_debug_clock_probe(const int32_t* stop_flag, uint64_t* out2, int32_t max_ms, void* stream) {
    if (!stop_flag || !out2 || max_ms <= 0) return SVG_ERR_BAD_ARG;
    hipLaunchKernelGGL(svg::clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stop_flag, (unsigned long long*)out2,
                       (long long)max_ms * 100000LL);
    return svg::launch_status();
}

extern "C" int svg_bsr_to_block_map(const int32_t* indptr, const int32_t* indices, int32_t MB, int32_t NB, int32_t row_block,
                                    int32_t col_block, int32_t len_text, int32_t heads, uint8_t* block_map, int32_t* q_sizes,
                                    int32_t* k_sizes, void* stream) {
    if (!indptr || !indices || !block_map || !q_sizes || !k_sizes) return SVG_ERR_BAD_ARG;
    if (MB <= 0 || NB <= 0 || row_block <= 0 || col_block <= 0 || len_text < 0 || heads <= 0) return SVG_ERR_BAD_ARG;
    hipLaunchKernelGGL(svg::bsr_to_map_kernel, dim3(MB + 1), dim3(256), 0, (hipStream_t)stream, indptr, indices, block_map, q_sizes,
                       k_sizes, MB, NB, row_block, col_block, len_text, heads);
    return svg::launch_status();
}

extern "C" const char* svg_strerror(int code) {
    switch (code) {
        case SVG_OK: return "ok";
        case SVG_ERR_BAD_ARG: return "bad argument (null pointer, negative size or inconsistent geometry)";
        case SVG_ERR_UNSUPPORTED: return "unsupported head_dim / dtype / size";
        case SVG_ERR_WORKSPACE: return "workspace too small";
        case SVG_ERR_LAUNCH: return "HIP launch failed (see svg_last_hip_error)";
        default: return "unknown error";
    }
}

extern "C" int svg_last_hip_error(void) { return svg::g_last_hip_error; }

extern "C" int svg_abi_version(void) { return SVG_ABI_VERSION; }

extern "C" const char* svg_build_info(void) { return "libsvgattn gfx950 wave64 mfma16x16x32+32x32x16 built " __DATE__ " " __TIME__; }

extern "C" int svg_map_density(const uint8_t* block_map, const int32_t* q_sizes, const int32_t* k_sizes, float* out,
                               int32_t BH, int32_t QB, int32_t KB, void* stream) {
    if (!block_map || !q_sizes || !k_sizes || !out || BH <= 0 || QB <= 0 || KB <= 0) return SVG_ERR_BAD_ARG;
    hipLaunchKernelGGL(svg::map_density_kernel, dim3(BH), dim3(256), 0, (hipStream_t)stream, block_map, q_sizes, k_sizes, out,
                       QB, KB);
    return svg::launch_status();
}
