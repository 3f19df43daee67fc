// Layout transformation ("placement") kernels: frame-major <-> token-major row permutation per head.
// ref: svg/models/hyvideo/placement.py:34-153 (forward), :285-387 (inverse), svg/models/cog/placement.py:33-100
// (text-first variant).  Pure HBM-bandwidth work: every row is read once and written once, 16 B per lane.
//
// Tiling: one workgroup owns TP consecutive patch positions p of one head.  In token-major order those are one
// contiguous chunk of TP*F rows; in frame-major order they are F chunks of TP contiguous rows.  Both sides of the
// copy therefore move whole multi-row segments, not isolated rows.
#include "svg_common.h"

namespace svg {

constexpr int kPlaceThreads = 256;
constexpr int kPlaceTP = 16;  // patch positions per workgroup

// grid = (video_blocks + text_blocks, BH, n_tensors)
__global__ __launch_bounds__(kPlaceThreads) void placement_kernel(const void* s0, const void* s1, const void* s2, void* d0,
                                                                   void* d1, void* d2, const int64_t* __restrict__ best,
                                                                   int S, int row_bytes, int vid0, int F, int P,
                                                                   int video_blocks, int inverse) {
    const int bh = blockIdx.y;
    const char* src = (const char*)(blockIdx.z == 0 ? s0 : blockIdx.z == 1 ? s1 : s2);
    char* dst = (char*)(blockIdx.z == 0 ? d0 : blockIdx.z == 1 ? d1 : d2);
    const size_t head_off = (size_t)bh * S * row_bytes;
    src += head_off;
    dst += head_off;
    const int lpr = row_bytes >> 4;  // lanes (16 B each) per row
    const int rpp = kPlaceThreads / lpr;  // rows per pass
    const int sub = threadIdx.x / lpr;
    const int col = (threadIdx.x - sub * lpr) << 4;
    const bool temporal = best[bh] != 0;
    const int V = F * P;

    if ((int)blockIdx.x < video_blocks) {
        const int p0 = blockIdx.x * kPlaceTP;
        const int np = min(kPlaceTP, P - p0);
        const int nrows = np * F;
        // token-major local row r <-> (pl = r / F, f = r % F)
        for (int r = sub; r < nrows; r += rpp) {
            const int pl = r / F;
            const int f = r - pl * F;
            const int tm = vid0 + (p0 + pl) * F + f;  // token-major row
            const int fm = vid0 + f * P + p0 + pl;    // frame-major row
            int srow, drow;
            if (!temporal) {
                // straight copy; use the token-major enumeration of rows for both sides (covers the same set)
                srow = drow = tm;
            } else if (!inverse) {
                srow = fm;
                drow = tm;
            } else {
                srow = tm;
                drow = fm;
            }
            const uint4 val = *(const uint4*)(src + (size_t)srow * row_bytes + col);
            *(uint4*)(dst + (size_t)drow * row_bytes + col) = val;
        }
    } else {
        // text rows: [0, vid0) and [vid0 + V, S), copied unchanged
        const int tb = blockIdx.x - video_blocks;
        const int ntext = S - V;
        const int rows_per_block = rpp * 8;
        const int r0 = tb * rows_per_block;
        for (int r = r0 + sub; r < min(ntext, r0 + rows_per_block); r += rpp) {
            const int row = r < vid0 ? r : r + V;
            const uint4 val = *(const uint4*)(src + (size_t)row * row_bytes + col);
            *(uint4*)(dst + (size_t)row * row_bytes + col) = val;
        }
    }
}

}  // namespace svg

extern "C" int svg_head_placement(const void* const* src, void* const* dst, int32_t n_tensors,
                                  const int64_t* best_mask_idx, int32_t BH, int32_t S, int32_t D, int32_t dtype,
                                  int32_t context_length, int32_t num_frame, int32_t frame_size, int32_t text_first,
                                  int32_t inverse, void* stream) {
    using namespace svg;
    if (!src || !dst || !best_mask_idx || n_tensors < 1 || n_tensors > 3) return SVG_ERR_BAD_ARG;
    for (int i = 0; i < n_tensors; ++i)
        if (!src[i] || !dst[i]) return SVG_ERR_BAD_ARG;
    if (BH <= 0 || S <= 0 || D <= 0 || context_length < 0 || num_frame <= 0 || frame_size <= 0) return SVG_ERR_BAD_ARG;
    if (dtype != SVG_DTYPE_BF16 && dtype != SVG_DTYPE_F16) return SVG_ERR_UNSUPPORTED;
    if ((int64_t)num_frame * frame_size + context_length != S) return SVG_ERR_BAD_ARG;
    const int row_bytes = D * 2;
    if (row_bytes % 16 != 0 || row_bytes > 16 * kPlaceThreads || kPlaceThreads % (row_bytes / 16) != 0)
        return SVG_ERR_UNSUPPORTED;
    const int vid0 = text_first ? context_length : 0;
    const int video_blocks = (frame_size + kPlaceTP - 1) / kPlaceTP;
    const int rpp = kPlaceThreads / (row_bytes / 16);
    const int text_blocks = (context_length + rpp * 8 - 1) / (rpp * 8);
    dim3 grid(video_blocks + text_blocks, BH, n_tensors);
    hipLaunchKernelGGL(placement_kernel, grid, dim3(kPlaceThreads), 0, (hipStream_t)stream, src[0],
                       n_tensors > 1 ? src[1] : nullptr, n_tensors > 2 ? src[2] : nullptr, dst[0],
                       n_tensors > 1 ? dst[1] : nullptr, n_tensors > 2 ? dst[2] : nullptr, best_mask_idx, S, row_bytes,
                       vid0, num_frame, frame_size, video_blocks, inverse);
    return launch_status();
}
