// Top-p block selection of SVG2 on device.
// ref: identify_dynamic_map + weighted_softmax, svg/kmeans_utils.py:852-896.
// One workgroup per (head, query-cluster) row; KC <= 4096 probabilities live in LDS.  The reference does this with
// ~8 torch launches and a full [B,H,QC,KC] sort; here it is one launch and no global temporaries.
//
// Arithmetic follows the reference's dtypes step by step so that the produced map is reproducible:
//   scores  = bf16(bf16(qc . kc) / sqrt(D))            (matmul output and the division both round to the input dtype)
//   probs   = fp32 weighted softmax (weights = k cluster sizes), clamp(sum, 1e-12), rounded to the input dtype
//   order   = descending by prob, ties -> lower cluster index (stable sort)
//   cumsum  = sequential, fp32 accumulator, every prefix rounded to the input dtype (torch cumsum on bf16)
//   keep[r] = r == 0 || !(cumsum[r-1] > p_dtype) || r < preserve_length
#include "svg_common.h"

namespace svg {

constexpr int kDynThreads = 256;

template <typename T, int D>
__global__ __launch_bounds__(kDynThreads) void dynmap_kernel(const T* __restrict__ qc, const T* __restrict__ kc,
                                                             const int32_t* __restrict__ k_sizes, uint8_t* __restrict__ out,
                                                             int QC, int KC, float sqrt_d, float top_p,
                                                             int preserve) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* prob = (float*)smem;              // [KC]  probabilities (already rounded to T), later reused
    float* sorted = prob + KC;               // [KC]  probabilities in descending order
    uint8_t* keep = (uint8_t*)(sorted + KC); // [KC]  keep flag per sorted position
    int* ranks_w = (int*)smem + 2 * KC + ((KC + 3) / 4);  // [KC] rank of cluster j in the descending order
    __shared__ float qrow[D];
    __shared__ float red[4];
    const int row = blockIdx.x, bh = blockIdx.y, tid = threadIdx.x;
    const T* q = qc + ((size_t)bh * QC + row) * D;
    const T* kb = kc + (size_t)bh * KC * D;
    const int32_t* ks = k_sizes + (size_t)bh * KC;
    if (tid < D) qrow[tid] = Elt<T>::to_float(q[tid]);
    __syncthreads();

    // 1. scores (kept in prob[]), running max
    float lmax = -INFINITY;
    for (int j = tid; j < KC; j += kDynThreads) {
        const T* kr = kb + (size_t)j * D;
        float acc = 0.f;
#pragma unroll 4
        for (int d0 = 0; d0 < D; d0 += 8) {
            const typename Elt<T>::v8 v = *(const typename Elt<T>::v8*)(kr + d0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(qrow[d0 + e], Elt<T>::to_float(v[e]), acc);
        }
        float s = Elt<T>::to_float(Elt<T>::from_float(acc));
        s = Elt<T>::to_float(Elt<T>::from_float(s / sqrt_d));
        prob[j] = s;
        lmax = fmaxf(lmax, s);
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    // 2. weighted exp, sum
    float lsum = 0.f;
    for (int j = tid; j < KC; j += kDynThreads) {
        const float we = (float)ks[j] * expf(prob[j] - gmax);
        prob[j] = we;
        lsum += we;
    }
    lsum = wave_sum(lsum);
    if ((tid & 63) == 0) red[tid >> 6] = lsum;
    __syncthreads();
    const float gsum = fmaxf(red[0] + red[1] + red[2] + red[3], 1e-12f);
    for (int j = tid; j < KC; j += kDynThreads) prob[j] = Elt<T>::to_float(Elt<T>::from_float(prob[j] / gsum));
    __syncthreads();
    // 3. rank of every element in the stable descending order
    for (int j = tid; j < KC; j += kDynThreads) {
        const float pj = prob[j];
        int rank = 0;
        for (int i = 0; i < KC; ++i) {
            const float pi = prob[i];
            rank += (pi > pj) | ((pi == pj) & (i < j));
        }
        sorted[rank] = pj;
        ranks_w[j] = rank;
    }
    __syncthreads();
    // 4. sequential cumsum (one lane), keep flags per sorted position
    if (tid == 0) {
        const float p_cmp = Elt<T>::to_float(Elt<T>::from_float(top_p));
        float acc = 0.f, prev_cum = 0.f;
        for (int r = 0; r < KC; ++r) {
            bool rm = (r > 0) && (prev_cum > p_cmp);
            if (r < preserve) rm = false;
            keep[r] = rm ? 0 : 1;
            acc += sorted[r];
            prev_cum = Elt<T>::to_float(Elt<T>::from_float(acc));
        }
    }
    __syncthreads();
    // 5. scatter back to cluster order
    uint8_t* orow = out + ((size_t)bh * QC + row) * KC;
    const int* ranks = (const int*)smem + 2 * KC + ((KC + 3) / 4);
    for (int j = tid; j < KC; j += kDynThreads) orow[j] = keep[ranks[j]];
}

}  // namespace svg

using namespace svg;

extern "C" int svg_identify_dynamic_map(const void* qc, const void* kc, const int32_t* k_sizes, uint8_t* out_map, int32_t BH,
                                        int32_t QC, int32_t KC, int32_t D, int32_t dtype, float top_p,
                                        int32_t preserve_length, void* stream) {
    if (!qc || !kc || !k_sizes || !out_map || BH <= 0 || QC <= 0 || KC <= 0) return SVG_ERR_BAD_ARG;
    if (KC > 4096) return SVG_ERR_UNSUPPORTED;
    // LDS: prob[KC] f32, sorted[KC] f32, keep[KC] u8 (padded to 4), ranks[KC] i32
    const size_t lds = (size_t)KC * 4 * 2 + ((KC + 3) / 4) * 4 + (size_t)KC * 4;
    const float inv = sqrtf((float)D);  // scores are divided by sqrt(D) like the reference
    dim3 grid(QC, BH);
    hipStream_t st = (hipStream_t)stream;
#define SVG_DYN(T, DD)                                                                                                   \
    hipLaunchKernelGGL((dynmap_kernel<T, DD>), grid, dim3(kDynThreads), lds, st, (const T*)qc, (const T*)kc, k_sizes, out_map, \
                       QC, KC, inv, top_p, preserve_length)
    if (dtype == SVG_DTYPE_BF16 && D == 128) SVG_DYN(__bf16, 128);
    else if (dtype == SVG_DTYPE_BF16 && D == 64) SVG_DYN(__bf16, 64);
    else if (dtype == SVG_DTYPE_F16 && D == 128) SVG_DYN(_Float16, 128);
    else if (dtype == SVG_DTYPE_F16 && D == 64) SVG_DYN(_Float16, 64);
    else return SVG_ERR_UNSUPPORTED;
#undef SVG_DYN
    return launch_status();
}
