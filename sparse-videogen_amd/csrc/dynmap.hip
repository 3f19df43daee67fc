// Top-p block selection of SVG2 on device.
// ref: identify_dynamic_map + weighted_softmax, svg/kmeans_utils.py:852-896.
// One workgroup per (head, query-cluster) row; KC <= 4096 probabilities live in LDS.  The reference does this with
// ~8 torch launches and a full [B,H,QC,KC] sort; here it is one launch and no global temporaries.
//
// Arithmetic follows the reference's dtypes step by step, with every quantity the reference ROUNDS computed exactly enough
// that the rounding is order-independent — so the map is bit-reproducible (tests: == the oracle's exact mode, bit for bit):
//   scores  = bf16(bf16(qc . kc) / sqrt(D))            (matmul output and the division both round to the input dtype; the dot
//                                                       product is accumulated in fp64: products of 16-bit values are exact and
//                                                       128 fp64 additions stay 2^-45 away from a bf16 rounding boundary, whereas
//                                                       an fp32 accumulation depends on its order — torch's, cuBLAS's and a plain
//                                                       loop disagree on ~0.05 % of the scores)
//   probs   = weighted softmax (weights = k cluster sizes) in fp64, clamp(sum, 1e-12), rounded to the input dtype (the reference
//             computes it in fp32; the fp64 value rounds to the same 16-bit number except when fp32 error straddles a boundary)
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
    // LDS: keys[N2] u64 (sort keys; N2 = KC rounded up to a power of two), prob[KC] f64
    const int N2 = 1 << (32 - __builtin_clz(max(KC, 2) - 1));
    unsigned long long* keys = (unsigned long long*)smem;
    double* prob = (double*)(keys + N2);
    __shared__ double qrow[D];   // the query centroid as doubles: converted once, read as 8-byte broadcasts
    __shared__ double red[4];
    __shared__ int cut_s;
    const int row = blockIdx.x, bh = blockIdx.y, tid = threadIdx.x;
    const T* q = qc + ((size_t)bh * QC + row) * D;
    const T* kb = kc + (size_t)bh * KC * D;
    const int32_t* ks = k_sizes + (size_t)bh * KC;
    if (tid < D) qrow[tid] = (double)Elt<T>::to_float(q[tid]);
    __syncthreads();

    // 1. scores (kept in prob[]), running max.  A thread works on kJB key centroids at once: every q value is read from LDS once
    //    per kJB dot products and the kJB row loads are in flight together; each dot product still accumulates d = 0 .. D - 1 in
    //    order, in fp64 (the order-independent rounding argument of the header does not even need that, but it keeps the sums
    //    identical to the previous form).
    float lmax = -INFINITY;
    constexpr int kJB = 4;
    for (int j0 = tid; j0 < KC; j0 += kJB * kDynThreads) {
        const T* kr[kJB];
        double acc[kJB];
#pragma unroll
        for (int u = 0; u < kJB; ++u) {
            const int j = j0 + u * kDynThreads;
            kr[u] = kb + (size_t)(j < KC ? j : j0) * D;    // (columns behind KC re-read a valid row; nothing is stored for them)
            acc[u] = 0.0;
        }
#pragma unroll 2
        for (int d0 = 0; d0 < D; d0 += 8) {
            typename Elt<T>::v8 v[kJB];
#pragma unroll
            for (int u = 0; u < kJB; ++u) v[u] = *(const typename Elt<T>::v8*)(kr[u] + d0);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double qd = qrow[d0 + e];
#pragma unroll
                for (int u = 0; u < kJB; ++u) acc[u] = __builtin_fma(qd, (double)Elt<T>::to_float(v[u][e]), acc[u]);
            }
#ifdef SVG_DYN_ABL
            if ((SVG_DYN_ABL) & 1) break;
#endif
        }
#pragma unroll
        for (int u = 0; u < kJB; ++u) {
            const int j = j0 + u * kDynThreads;
            if (j < KC) {
                float s = Elt<T>::to_float(Elt<T>::from_double(acc[u]));
                s = Elt<T>::to_float(Elt<T>::from_float(s / sqrt_d));
                prob[j] = (double)s;
                lmax = fmaxf(lmax, s);
            }
        }
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) red[tid >> 6] = (double)lmax;
    __syncthreads();
    const double gmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    // 2. weighted exp, sum (fp64; the partial sums are combined in a fixed order)
    double lsum = 0.0;
    for (int j = tid; j < KC; j += kDynThreads) {
        const double we = (double)ks[j] * exp(prob[j] - gmax);
        prob[j] = we;
        lsum += we;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) lsum += __shfl_xor(lsum, o);
    if ((tid & 63) == 0) red[tid >> 6] = lsum;
    __syncthreads();
    const double gsum = fmax(red[0] + red[1] + red[2] + red[3], 1e-12);
    // 3. stable descending order: sort the unique keys (inverted probability bits, cluster index) ascending — probabilities
    //    are >= 0, so their bit patterns order like the values; ties fall back to the lower index like a stable sort.
    //    Bitonic network in LDS: O(N log^2 N) compare-exchanges instead of the N^2 rank counting this kernel started with.
    for (int j = tid; j < N2; j += kDynThreads) {
        unsigned long long key = ~0ull;
        if (j < KC) {
            const float pj = Elt<T>::to_float(Elt<T>::from_double(prob[j] / gsum));
            key = ((unsigned long long)(0xffffffffu - __float_as_uint(pj)) << 32) | (unsigned)j;
        }
        keys[j] = key;
    }
    __syncthreads();
#ifdef SVG_DYN_ABL
    if (!((SVG_DYN_ABL) & 2))
#endif
    for (int k = 2; k <= N2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < N2 / 2; t += kDynThreads) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const unsigned long long a = keys[lo], b2 = keys[hi];
                const bool asc = (lo & k) == 0;
                if ((a > b2) == asc) keys[lo] = b2, keys[hi] = a;
            }
            __syncthreads();
        }
    }
    // 4. cumulative sum in sorted order: sequential, fp32 accumulator, every prefix rounded to the input dtype (torch cumsum
    //    on a 16-bit tensor).  The prefix is non-decreasing, so everything after the first position whose previous prefix
    //    exceeds p is dropped: one lane walks only that far.
    if (tid == 0) {
        const float p_cmp = Elt<T>::to_float(Elt<T>::from_float(top_p));
        float acc = 0.f, prev_cum = 0.f;
        int r = 0;
        bool done = false;
        // (the same sequence of fp32 additions as a one-by-one walk; the keys of a block of 8 are loaded together, so the walk pays
        //  one LDS round trip per 8 positions instead of one per position — N2 is a power of two >= 2, entries behind KC hold the
        //  probability 0 sentinel and are never reached: the r < KC test comes first)
#ifdef SVG_DYN_ABL
        if ((SVG_DYN_ABL) & 4) done = true, r = KC / 4;
#endif
        for (int r0 = 0; r0 < KC && !done; r0 += 8) {
            unsigned long long kq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kq[u] = keys[min(r0 + u, N2 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (done) continue;
                r = r0 + u;
                if (r >= KC || (r > 0 && prev_cum > p_cmp)) {   // position r (and all later ones) is removed unless r < preserve
                    done = true;
                    continue;
                }
                acc += __uint_as_float(0xffffffffu - (unsigned)(kq[u] >> 32));
                prev_cum = Elt<T>::to_float(Elt<T>::from_float(acc));
                r = r0 + u + 1;
            }
        }
        cut_s = max(r, preserve);
    }
    __syncthreads();
    // 5. scatter back to cluster order
    const int cut = cut_s;
    uint8_t* orow = out + ((size_t)bh * QC + row) * KC;
    for (int r = tid; r < KC; r += kDynThreads) orow[(unsigned)keys[r]] = r < cut ? 1 : 0;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Second form (round 5; KC <= 1024, the production sizes): one workgroup per 16 query clusters.
//   1. scores of 16 rows x KC columns on v_mfma_f64_16x16x4_f64 (fp64 accumulation as above: the rounding argument of the header holds
//      for ANY summation order; a lane group g4 contracts d = (D / 4) g4 + [0, D / 4)), rounded like the first form and kept in LDS as
//      16-bit patterns;  2. one WAVE per row: weighted softmax in fp64, 32-bit sort keys ((0xffff - probability bits) << 16 | cluster), a
//      bitonic sort of 1024 keys in the wave's registers — 16 keys per lane, partner lanes by ds_bpermute, no barrier, no LDS traffic —,
//      the sequential cumulative sum by one lane, scatter.
// The first form (dynmap_kernel: one workgroup per row, fp64 FMA chains on the vector pipe, a bitonic network in LDS with 55 barrier
// rounds) took 0.80 ms at Wan 2.1 720p (40 x 300 x 1000): 0.36 ms of it the dot products, 0.35 the sort (profiles/r05f_dynmap_ablation.txt).
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kDyn2Rows = 16;
constexpr int kDyn2E = 16;                  // keys per lane
constexpr int kDyn2N = 64 * kDyn2E;         // 1024
#ifndef SVG_DYN2_WAVES
#define SVG_DYN2_WAVES 8
#endif
constexpr int kDyn2Waves = SVG_DYN2_WAVES;   // waves per workgroup: the key blocks of step 1 and the rows of step 2 are dealt round-robin

template <int E>
__device__ __forceinline__ void wave_bitonic_sort(unsigned (&key)[E], int lane) {
    constexpr int N = 64 * E;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= E) {
                const int lj = j / E;
                const bool lower = (lane & lj) == 0;
                const bool asc = (k >= N) ? true : ((lane & (k / E)) == 0);
                const bool keep_min = lower == asc;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const unsigned other = (unsigned)__shfl_xor((int)key[e], lj);
                    const unsigned lo = key[e] < other ? key[e] : other, hi = key[e] < other ? other : key[e];
                    key[e] = keep_min ? lo : hi;
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & j) != 0) continue;
                    const bool asc = (k >= N) ? true : (k >= E ? ((lane & (k / E)) == 0) : ((e & k) == 0));
                    const unsigned a = key[e], b = key[e | j];
                    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
                    key[e] = asc ? lo : hi;
                    key[e | j] = asc ? hi : lo;
                }
            }
        }
    }
}

template <typename T, int D>
__global__ __launch_bounds__(64 * kDyn2Waves) void dynmap16_kernel(const T* __restrict__ qc, const T* __restrict__ kc, const int32_t* __restrict__ k_sizes,
                                                       uint8_t* __restrict__ out, int QC, int KC, float sqrt_d, float top_p, int preserve) {
    constexpr int CH = D / 4;                           // contraction elements of a lane group
    __shared__ unsigned short sc16[kDyn2Rows][kDyn2N];  // rounded scores as bit patterns of T
    __shared__ unsigned sorted[kDyn2Waves][kDyn2N];     // a wave's sorted keys (for the walk and the scatter)
    const int bh = blockIdx.y, row0 = blockIdx.x * kDyn2Rows;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int n16 = lane & 15, g4 = lane >> 4;
    const T* kb = kc + (size_t)bh * KC * D;
    const int32_t* ks = k_sizes + (size_t)bh * KC;

    // ---- 1. scores ----
    double qd[CH];
    {
        const int qr = min(row0 + n16, QC - 1);
        const T* qrow = qc + ((size_t)bh * QC + qr) * D + CH * g4;
#pragma unroll
        for (int c = 0; c < CH; c += 8) {
            const typename Elt<T>::v8 v = *(const typename Elt<T>::v8*)(qrow + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) qd[c + e] = (double)Elt<T>::to_float(v[e]);
        }
    }
    const int nkb = (KC + 15) / 16;
    for (int kblk = wave; kblk < nkb; kblk += kDyn2Waves) {
        const int kr = min(kblk * 16 + n16, KC - 1);
        const T* krow = kb + (size_t)kr * D + CH * g4;
        typename Elt<T>::v8 kv[CH / 8];
#pragma unroll
        for (int c = 0; c < CH / 8; ++c) kv[c] = *(const typename Elt<T>::v8*)(krow + 8 * c);
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < CH; ++c)   // A: query cluster n16 (rows of the product), B: key cluster n16 (columns); both at d = CH g4 + c
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qd[c], (double)Elt<T>::to_float(kv[c / 8][c % 8]), acc, 0, 0, 0);
        // lane (g4, n16) holds the product's rows 4 i + g4 (query clusters), column n16 (key cluster) in acc[i] — measured: tools/probe_dynmap.hip
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s2 = Elt<T>::to_float(Elt<T>::from_double(acc[i]));
            const T r2 = Elt<T>::from_float(s2 / sqrt_d);
            sc16[4 * i + g4][kblk * 16 + n16] = __builtin_bit_cast(unsigned short, r2);
        }
    }
    __syncthreads();

    // ---- 2. one wave per row ----
    unsigned* srt = sorted[wave];
    for (int rr = wave; rr < kDyn2Rows; rr += kDyn2Waves) {
        const int row = row0 + rr;
        if (row >= QC) break;                                 // (wave-uniform)
        float sv[kDyn2E];
        float lmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < kDyn2E; ++e) {
            const int j = lane * kDyn2E + e;
            sv[e] = Elt<T>::to_float(__builtin_bit_cast(T, sc16[rr][j]));
            if (j < KC) lmax = fmaxf(lmax, sv[e]);
        }
        const double gmax = (double)wave_max(lmax);
        double we[kDyn2E];
        double lsum = 0.0;
#pragma unroll
        for (int e = 0; e < kDyn2E; ++e) {
            const int j = lane * kDyn2E + e;
            we[e] = j < KC ? (double)ks[j] * exp((double)sv[e] - gmax) : 0.0;
            lsum += we[e];
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) lsum += __shfl_xor(lsum, o);
        const double gsum = fmax(lsum, 1e-12);
        unsigned key[kDyn2E];
#pragma unroll
        for (int e = 0; e < kDyn2E; ++e) {
            const int j = lane * kDyn2E + e;
            const T pj = Elt<T>::from_double(we[e] / gsum);
            key[e] = j < KC ? (((0xffffu - (unsigned)__builtin_bit_cast(unsigned short, pj)) << 16) | (unsigned)j) : 0xffffffffu;
        }
        wave_bitonic_sort<kDyn2E>(key, lane);
#pragma unroll
        for (int e = 0; e < kDyn2E; e += 4) {
            const u32x4 k4 = {key[e], key[e + 1], key[e + 2], key[e + 3]};
            *(u32x4*)(srt + lane * kDyn2E + e) = k4;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        int cut = 0;
        if (lane == 0) {   // the cumulative sum of the first form, on 32-bit keys
            const float p_cmp = Elt<T>::to_float(Elt<T>::from_float(top_p));
            float acc = 0.f, prev_cum = 0.f;
            int r = 0;
            bool done = false;
            for (int r0 = 0; r0 < KC && !done; r0 += 8) {
                unsigned kq[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) kq[u] = srt[min(r0 + u, kDyn2N - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (done) continue;
                    r = r0 + u;
                    if (r >= KC || (r > 0 && prev_cum > p_cmp)) {
                        done = true;
                        continue;
                    }
                    acc += Elt<T>::to_float(__builtin_bit_cast(T, (unsigned short)(0xffffu - (kq[u] >> 16))));   // (the key carries T's bit pattern)
                    prev_cum = Elt<T>::to_float(Elt<T>::from_float(acc));
                    r = r0 + u + 1;
                }
            }
            cut = max(r, preserve);
        }
        cut = __shfl(cut, 0);
        uint8_t* orow = out + ((size_t)bh * QC + row) * KC;
#pragma unroll
        for (int e = 0; e < kDyn2E; ++e) {
            const int r = lane * kDyn2E + e;
            if (r < KC) orow[key[e] & 0xffffu] = r < cut ? 1 : 0;
        }
        __builtin_amdgcn_wave_barrier();   // (the next row overwrites srt; the walk above has finished: same wave, program order)
    }
}

}  // namespace svg

using namespace svg;

extern "C" int svg_identify_dynamic_map(const void* qc, const void* kc, const int32_t* k_sizes, uint8_t* out_map, int32_t BH,
                                        int32_t QC, int32_t KC, int32_t D, int32_t dtype, float top_p,
                                        int32_t preserve_length, void* stream) {
    if (!qc || !kc || !k_sizes || !out_map || BH <= 0 || QC <= 0 || KC <= 0) return SVG_ERR_BAD_ARG;
    if (KC > 4096) return SVG_ERR_UNSUPPORTED;
    // LDS: keys[N2] u64 (N2 = KC rounded up to a power of two) + prob[KC] f64
    int n2 = 2;
    while (n2 < KC) n2 <<= 1;
    const size_t lds = (size_t)n2 * 8 + (size_t)KC * 8;
    const float inv = sqrtf((float)D);  // scores are divided by sqrt(D) like the reference
    dim3 grid(QC, BH);
    hipStream_t st = (hipStream_t)stream;
    // 256 < KC <= 1024: the second form (dynmap16_kernel); smaller and larger maps keep the first one
#ifndef SVG_DYNMAP_FIRST_FORM
    if (KC > 256 && KC <= kDyn2N) {   // (small maps: the first form sorts the next power of two of KC, this one always 1024 keys)
        dim3 grid16((QC + kDyn2Rows - 1) / kDyn2Rows, BH);
#define SVG_DYN16(T, DD) \
    hipLaunchKernelGGL((dynmap16_kernel<T, DD>), grid16, dim3(64 * kDyn2Waves), 0, st, (const T*)qc, (const T*)kc, k_sizes, out_map, QC, KC, inv, top_p, preserve_length)
        if (dtype == SVG_DTYPE_BF16 && D == 128) SVG_DYN16(__bf16, 128);
        else if (dtype == SVG_DTYPE_BF16 && D == 64) SVG_DYN16(__bf16, 64);
        else if (dtype == SVG_DTYPE_F16 && D == 128) SVG_DYN16(_Float16, 128);
        else if (dtype == SVG_DTYPE_F16 && D == 64) SVG_DYN16(_Float16, 64);
        else return SVG_ERR_UNSUPPORTED;
#undef SVG_DYN16
        return launch_status();
    }
#endif
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
