// Token permutation kernels: row gather / row scatter and the stable counting-sort argsort of cluster labels.
// ref: svg/kernels/triton/permute.py:12-75 (gather / scatter), :113 (torch.argsort of labels).
// Gather/scatter are pure HBM-bandwidth kernels (16 B per lane, a row = D*2 B is one contiguous segment).
#include "svg_common.h"

namespace svg {

constexpr int kPermThreads = 256;

// SCATTER = false: y[bh, s] = x[bh, idx[bh, s]];  SCATTER = true: y[bh, idx[bh, s]] = x[bh, s]
template <bool SCATTER>
__global__ __launch_bounds__(kPermThreads) void permute_rows_kernel(const char* __restrict__ x,
                                                                     const int32_t* __restrict__ idx,
                                                                     char* __restrict__ y, int S, int row_bytes,
                                                                     int rows_per_block) {
    const int bh = blockIdx.y;
    const size_t head_off = (size_t)bh * S * row_bytes;
    x += head_off;
    y += head_off;
    idx += (size_t)bh * S;
    const int lpr = row_bytes >> 4;
    const int rpp = kPermThreads / lpr;
    const int sub = threadIdx.x / lpr;
    const int col = (threadIdx.x - sub * lpr) << 4;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(S, r0 + rows_per_block);
    for (int r = r0 + sub; r < r1; r += rpp) {
        const int j = idx[r];
        const int srow = SCATTER ? r : j;
        const int drow = SCATTER ? j : r;
        const uint4 val = *(const uint4*)(x + (size_t)srow * row_bytes + col);
        *(uint4*)(y + (size_t)drow * row_bytes + col) = val;
    }
}

// ---------------- stable counting sort of labels (one wave per chunk of kSortChunk points) ----------------
constexpr int kSortChunk = 1024;

__global__ __launch_bounds__(64) void sort_hist_kernel(const int32_t* __restrict__ labels, int32_t* __restrict__ chunk_hist,
                                                       int N, int K, int nchunks) {
    extern __shared__ int32_t hist[];
    const int b = blockIdx.y, c = blockIdx.x, lane = threadIdx.x;
    for (int k = lane; k < K; k += 64) hist[k] = 0;
    __syncthreads();
    const int32_t* lb = labels + (size_t)b * N;
    const int n0 = c * kSortChunk;
    for (int i = n0 + lane; i < min(N, n0 + kSortChunk); i += 64) {
        const int l = lb[i];
        if ((unsigned)l < (unsigned)K) atomicAdd(&hist[l], 1);
    }
    __syncthreads();
    int32_t* out = chunk_hist + ((size_t)b * nchunks + c) * K;
    for (int k = lane; k < K; k += 64) out[k] = hist[k];
}

// grid = (B), block = 1024: turns per-chunk histograms into per-chunk starting offsets (in place) and writes counts.
__global__ __launch_bounds__(1024) void sort_scan_kernel(int32_t* __restrict__ chunk_hist, int32_t* __restrict__ counts, int K,
                                                         int nchunks) {
    __shared__ int32_t wave_tot[16];
    __shared__ int32_t carry_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int32_t* h = chunk_hist + (size_t)b * nchunks * K;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += 1024) {
        const int k = k0 + tid;
        int total = 0;
        if (k < K)
            for (int c = 0; c < nchunks; ++c) total += h[(size_t)c * K + k];
        if (k < K && counts) counts[(size_t)b * K + k] = total;
        // block exclusive scan of `total`
        int incl = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_tot[w] = incl;
        __syncthreads();
        int wbase = 0;
        for (int i = 0; i < w; ++i) wbase += wave_tot[i];
        const int carry = carry_s;
        int base = carry + wbase + incl - total;  // exclusive prefix over clusters
        if (k < K) {
            int run = base;
            for (int c = 0; c < nchunks; ++c) {
                const int v = h[(size_t)c * K + k];
                h[(size_t)c * K + k] = run;
                run += v;
            }
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + wbase + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void sort_scatter_kernel(const int32_t* __restrict__ labels,
                                                          const int32_t* __restrict__ chunk_off,
                                                          int32_t* __restrict__ sorted_idx, int N, int K, int nchunks) {
    extern __shared__ int32_t running[];
    const int b = blockIdx.y, c = blockIdx.x, lane = threadIdx.x;
    const int32_t* off = chunk_off + ((size_t)b * nchunks + c) * K;
    for (int k = lane; k < K; k += 64) running[k] = off[k];
    __syncthreads();
    const int32_t* lb = labels + (size_t)b * N;
    int32_t* out = sorted_idx + (size_t)b * N;
    const int n0 = c * kSortChunk;
    const int n1 = min(N, n0 + kSortChunk);
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int nbits = 32 - __builtin_clz((unsigned)max(K - 1, 1));
    // the chunk's labels: all kSortChunk / 64 loads in flight at once (the batches below depend on each other through `running` only;
    // loading inside the loop paid one global-load latency per batch: 0.10 ms per call at Wan 720p, profiles/r05f_svg2_kernel_trace.txt)
    constexpr int kBatches = kSortChunk / 64;
    int lreg[kBatches];
#pragma unroll
    for (int u = 0; u < kBatches; ++u) {
        const int i = n0 + u * 64 + lane;
        lreg[u] = i < n1 ? lb[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < kBatches; ++u) {
        const int i0 = n0 + u * 64;
        if (i0 >= n1) break;
        const int i = i0 + lane;
        int l = lreg[u];
        if ((unsigned)l >= (unsigned)K) l = -1;
        const bool valid = l >= 0;
        // lanes with this lane's label (match-any), one ballot per label BIT: log2(K) independent steps.  (The first form looped over the
        // DISTINCT labels of the batch — leader, broadcast, ballot, clear —, a serial scalar chain of up to 64 rounds: 12k cycles per batch
        // on random labels, 0.10 ms per call at Wan 720p; profiles/r05f_svg2_kernel_trace.txt.)
        unsigned long long m = __ballot(valid);
        for (int bit = 0; bit < nbits; ++bit) {
            const bool one = (l >> bit) & 1;
            const unsigned long long bal = __ballot(valid && one);
            m &= one ? bal : ~bal;
        }
        const int rank = __popcll(m & lt_mask), cnt = __popcll(m);
        int base = 0;
        if (valid) base = running[l];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): reads done before the leader's update below
        if (valid) {
            out[base + rank] = i;
            if (rank == 0) running[l] = base + cnt;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

}  // namespace svg

using namespace svg;

static int check_rows(const void* x, const int32_t* idx, void* y, int32_t BH, int32_t S, int32_t D, int32_t dtype) {
    if (!x || !idx || !y || BH <= 0 || S <= 0 || D <= 0) return SVG_ERR_BAD_ARG;
    if (dtype != SVG_DTYPE_BF16 && dtype != SVG_DTYPE_F16) return SVG_ERR_UNSUPPORTED;
    const int row_bytes = D * 2;
    if (row_bytes % 16 != 0 || row_bytes > 16 * kPermThreads || kPermThreads % (row_bytes / 16) != 0)
        return SVG_ERR_UNSUPPORTED;
    return SVG_OK;
}

extern "C" int svg_permute_rows(const void* x, const int32_t* idx, void* y, int32_t BH, int32_t S, int32_t D,
                                int32_t dtype, void* stream) {
    int rc = check_rows(x, idx, y, BH, S, D, dtype);
    if (rc) return rc;
    const int rows_per_block = 256;
    dim3 grid((S + rows_per_block - 1) / rows_per_block, BH);
    hipLaunchKernelGGL(permute_rows_kernel<false>, grid, dim3(kPermThreads), 0, (hipStream_t)stream, (const char*)x, idx,
                       (char*)y, S, D * 2, rows_per_block);
    return launch_status();
}

extern "C" int svg_inverse_permute_rows(const void* x, const int32_t* idx, void* y, int32_t BH, int32_t S, int32_t D,
                                        int32_t dtype, void* stream) {
    int rc = check_rows(x, idx, y, BH, S, D, dtype);
    if (rc) return rc;
    const int rows_per_block = 256;
    dim3 grid((S + rows_per_block - 1) / rows_per_block, BH);
    hipLaunchKernelGGL(permute_rows_kernel<true>, grid, dim3(kPermThreads), 0, (hipStream_t)stream, (const char*)x, idx,
                       (char*)y, S, D * 2, rows_per_block);
    return launch_status();
}

extern "C" size_t svg_argsort_workspace_bytes(int32_t B, int32_t N, int32_t K) {
    if (B <= 0 || N <= 0 || K <= 0) return 0;
    const size_t nchunks = (N + kSortChunk - 1) / kSortChunk;
    return (size_t)B * nchunks * K * sizeof(int32_t);
}

extern "C" int svg_argsort_labels(const int32_t* labels, int32_t* sorted_idx, int32_t* counts, int32_t B, int32_t N,
                                  int32_t K, void* workspace, size_t workspace_bytes, void* stream) {
    if (!labels || !sorted_idx || !workspace || B <= 0 || N <= 0 || K <= 0) return SVG_ERR_BAD_ARG;
    if (K > 8192) return SVG_ERR_UNSUPPORTED;
    if (workspace_bytes < svg_argsort_workspace_bytes(B, N, K)) return SVG_ERR_WORKSPACE;
    const int nchunks = (N + kSortChunk - 1) / kSortChunk;
    int32_t* hist = (int32_t*)workspace;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sort_hist_kernel, dim3(nchunks, B), dim3(64), K * sizeof(int32_t), st, labels, hist, N, K, nchunks);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(B), dim3(1024), 0, st, hist, counts, K, nchunks);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(nchunks, B), dim3(64), K * sizeof(int32_t), st, labels, hist, sorted_idx,
                       N, K, nchunks);
    return launch_status();
}
