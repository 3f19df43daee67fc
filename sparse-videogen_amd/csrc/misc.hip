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

}  // namespace svg

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

extern "C" const char* svg_build_info(void) { return "libsvgattn gfx950 wave64 mfma32x32x16 built " __DATE__ " " __TIME__; }

extern "C" int svg_map_density(const uint8_t* block_map, const int32_t* q_sizes, const int32_t* k_sizes, float* out,
                               int32_t BH, int32_t QB, int32_t KB, void* stream) {
    if (!block_map || !q_sizes || !k_sizes || !out || BH <= 0 || QB <= 0 || KB <= 0) return SVG_ERR_BAD_ARG;
    hipLaunchKernelGGL(svg::map_density_kernel, dim3(BH), dim3(256), 0, (hipStream_t)stream, block_map, q_sizes, k_sizes, out,
                       QB, KB);
    return svg::launch_status();
}
