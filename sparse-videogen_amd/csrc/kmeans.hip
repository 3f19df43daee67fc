// flash-kmeans for gfx950: one Lloyd iteration of batched Euclidean k-means, entirely on device.
// ref: batch_kmeans_Euclid / _euclid_iter, svg/kmeans_utils.py:629-643,684-733
//      assign kernel  _euclid_assign_kernel             svg/kmeans_utils.py:464-554
//      centroid update (sorted, chunked)                svg/kmeans_utils.py:258-322,375-421
//
//   assign : S^T[c][n] = <C[c], X[n]> on MFMA 32x32x16 (A = centroid tile from LDS, B = point fragments in
//            registers, exactly the S^T half of attn_core.h), dist = max(0, xsq[n] + csq[c] - 2 S), running
//            (min, argmin) per lane, lowest index wins exact ties like the reference's strict '<' across chunks and
//            first-index argmin inside a chunk.  The N x K distance matrix is never materialised.
//   update : stable counting sort of the labels (permute.hip) -> every cluster is a contiguous run of
//            sorted_idx; one workgroup per (batch, cluster) gathers its rows and reduces them in a FIXED order
//            (fp32), so centroids are bit-reproducible run to run (the reference's atomics are not).
#include <algorithm>

#include "attn_core.h"

extern "C" size_t svg_argsort_workspace_bytes(int32_t B, int32_t N, int32_t K);
extern "C" int svg_argsort_labels(const int32_t* labels, int32_t* sorted_idx, int32_t* counts, int32_t B, int32_t N,
                                  int32_t K, void* workspace, size_t workspace_bytes, void* stream);

namespace svg {

// ---- ||x||^2 exactly like the reference: (x**2) rounded to the input dtype, summed in fp32, rounded to dtype ----
// ref: svg/kmeans_utils.py:704 `x_sq = (x**2).sum(dim=-1)` on a bf16 tensor, :512 cast to fp32 in the kernel
template <typename T>
__global__ __launch_bounds__(256) void xsq_kernel(const T* __restrict__ x, float* __restrict__ xsq, long long rows, int D) {
    const int lpr = D / 8;                       // lanes per row (8 elements = 16 B each)
    const int rpb = 256 / lpr;                   // rows per block pass
    const int sub = threadIdx.x / lpr, li = threadIdx.x - sub * lpr;
    for (long long r = (long long)blockIdx.x * rpb + sub; r < rows; r += (long long)gridDim.x * rpb) {
        const typename Elt<T>::v8 v = *(const typename Elt<T>::v8*)(x + r * D + li * 8);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = Elt<T>::to_float(v[j]);
            s += Elt<T>::to_float(Elt<T>::from_float(f * f));
        }
        for (int o = lpr >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (li == 0) xsq[r] = Elt<T>::to_float(Elt<T>::from_float(s));
    }
}

// ---- ||c||^2: products rounded to the input dtype, fp32 sum (ref: svg/kmeans_utils.py:531) ----
template <typename T>
__global__ __launch_bounds__(256) void csq_kernel(const T* __restrict__ c, float* __restrict__ csq, long long rows, int D) {
    const int lpr = D / 8;
    const int rpb = 256 / lpr;
    const int sub = threadIdx.x / lpr, li = threadIdx.x - sub * lpr;
    for (long long r = (long long)blockIdx.x * rpb + sub; r < rows; r += (long long)gridDim.x * rpb) {
        const typename Elt<T>::v8 v = *(const typename Elt<T>::v8*)(c + r * D + li * 8);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = Elt<T>::to_float(v[j]);
            s += Elt<T>::to_float(Elt<T>::from_float(f * f));
        }
        for (int o = lpr >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (li == 0) csq[r] = s;
    }
}

#ifdef SVG_KMEANS_TRACE
static __device__ unsigned long long g_km_trace[8 * 8];   // diagnostics build: per wave of workgroup (7, 0): cycles in { init, mfma, epilogue, stage, barrier }, tiles
#endif
// ---- assignment: grid = (ceil(N / (NW*32)), B), block = NW*64 ----
template <typename T, int D, int NW>
__global__ __launch_bounds__(NW * 64, 2) void kmeans_assign_kernel(const T* __restrict__ x, const T* __restrict__ cent,
                                                                    const float* __restrict__ xsq,
                                                                    const float* __restrict__ csq, int32_t* __restrict__ labels,
                                                                    int N, int K, long long x_bs /* elements between the batches of x */) {
    using E = Elt<T>;
    using V8 = typename E::v8;
    using L = LdsLayout<D>;
    constexpr int NT = NW * 64;
    constexpr int KS = D / 16;
    constexpr int NCH = (kBN * L::kCPR) / NT;
    constexpr int kStage = L::kKBytes + kBN * 4;  // centroid tile + its 64 squared norms
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int g = lane >> 5, ql = lane & 31;
    const int n = blockIdx.x * (NW * 32) + wave * 32 + ql;
    const T* xb = x + (size_t)b * (size_t)x_bs;
    const T* cb = cent + (size_t)b * K * D;
    const float* csqb = csq + (size_t)b * K;

    V8 xf[KS];
    {
        const T* xrow = xb + (size_t)(n < N ? n : 0) * D + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *(const V8*)(xrow + ks * 16);
    }
    (void)xsq;   // |x|^2 does not change the argmin; the parameter stays for the callers that keep the distance form

    int srow[NCH], scol[NCH], k_dst[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = tid + i * NT;
        srow[i] = id / L::kCPR;
        scol[i] = id - srow[i] * L::kCPR;
        k_dst[i] = L::k_off(srow[i], scol[i]);
    }
    u32x4 creg[NCH];
    float sreg = 0.f;
    bool sreg_ok = false;
    const int nT = (K + kBN - 1) / kBN;
    auto issue = [&](int t) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = t * kBN + srow[i];
            creg[i] = *(const u32x4*)(cb + (size_t)(c < K ? c : 0) * D + scol[i] * 8);
        }
        if (tid < kBN) {   // |c|^2, RAW: the multiply sits in write() — used here it makes hipcc wait (vmcnt(0)) for the loads it has just issued
            const int c = t * kBN + tid;
            sreg = csqb[c < K ? c : 0];
            sreg_ok = c < K;
        }
    };
    auto write = [&](int buf) {
        char* base = smem + buf * kStage;
#pragma unroll
        for (int i = 0; i < NCH; ++i) *(u32x4*)(base + k_dst[i]) = creg[i];
        // -|c|^2 / 2 (the accumulator's start value, see below); centroids behind K can never win
        if (tid < kBN) *(float*)(base + L::kKBytes + tid * 4) = sreg_ok ? -0.5f * sreg : -INFINITY;
    };

    issue(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf[ks]));
    write(0);
    if (nT > 1) issue(1);
    __syncthreads();

    const int ksw0 = (D == 128) ? (ql & 15) : ((ql >> 1) & 7);
    // argmin_c |x - c|^2 = argmax_c (x.c - |c|^2 / 2): the accumulators start at -|c|^2 / 2, so an element of the epilogue is a
    // compare and two selects (the distance form — fma, clamp, bounds test, compare, two selects — made this kernel VALU-bound:
    // 192 VALU against 16 MFMAs per tile).  Ties keep the lowest index, like argmin.
    // Round 5 (profiles/r05r_kmeans_assign_trace.txt, r05j_ab_kmeans_assign.txt; labels bit-identical throughout).  The per-phase trace (-DSVG_KMEANS_TRACE) found
    // the MFMA phase at 1050 - 1240 cycles for 512 of matrix work (one fragment pair in flight) and the loads of tile t + 2 waited out in the iteration that
    // issued them (the -|c|^2 / 2 multiply consumed its load at once): both fixed below — 3040 -> 2640 cycles per tile, 0.55 -> 0.525 ms per call.  What bounds
    // the kernel now is the arg-max epilogue: per SIMD and tile four waves need 2048 cycles of matrix pipe and ~4000 of vector issue (3 VALU per score).
    // Measured and removed: a second score set (168 registers: slower), a half-tile pipeline, 128-centroid stages (no gain: neither overlap nor the
    // barrier was the problem).
    float best = -INFINITY;
    int best_idx = 0;
#ifdef SVG_KMEANS_TRACE
    unsigned long long tr[5] = {0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
#define KM_TICK(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tr[i] += now_ - tl; tl = now_; }
#else
#define KM_TICK(i)
#endif
    for (int t = 0; t < nT; ++t) {
        const int buf = t & 1;
        const char* kbuf = smem + buf * kStage;
        const float* h_t = (const float*)(kbuf + L::kKBytes);
        f32x16 s[2];
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 h4 = *(const f32x4*)(h_t + 32 * bb + 8 * rq + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[bb][rq * 4 + j] = h4[j];
            }
        float tb;
        int ti;
        KM_TICK(0)
        // centroid fragments two k-steps ahead of their MFMAs (hipcc's own schedule kept ONE pair in flight and waited for it in front of
        // every MFMA: 1050 - 1240 cycles for 512 cycles of matrix work per tile, profiles/r05r_kmeans_assign_trace.txt)
        constexpr int kAhead = 2;
        V8 af[kAhead + 1][2];
        auto afetch = [&](int ks) {
            const int cch = ((2 * ks + g) ^ ksw0) << 4;
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) af[ks % (kAhead + 1)][bb] = *(const V8*)(kbuf + (32 * bb + ql) * L::kRowBytes + cch);
        };
#pragma unroll
        for (int ks = 0; ks < kAhead; ++ks) afetch(ks);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + kAhead < KS) afetch(ks + kAhead);
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) s[bb] = E::mfma(af[ks % (kAhead + 1)][bb], xf[ks], s[bb]);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef SVG_KMEANS_TRACE
        asm volatile("" : "+v"(s[0]), "+v"(s[1]));
#endif
        KM_TICK(1)
        // four independent chains of eight elements (ascending index inside a chain), merged in index order with the same strict compare:
        // the result of the one 32-step chain — lowest index wins ties — at a quarter of its dependent latency (3 dependent VALU per step:
        // 665 - 860 cycles per tile in the trace)
        float cb_[4];
        int ci_[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const int bb = ch >> 1, r0 = (ch & 1) * 8;
            cb_[ch] = s[bb][r0];
            ci_[ch] = 32 * bb + 8 * (r0 >> 2) + (r0 & 3) + 4 * g;
#pragma unroll
            for (int r = r0 + 1; r < r0 + 8; ++r) {
                const int c = 32 * bb + 8 * (r >> 2) + (r & 3);   // + 4 g: tile-local centroid index, ascending in (bb, r)
                const bool upd = s[bb][r] > cb_[ch];
                cb_[ch] = upd ? s[bb][r] : cb_[ch];
                ci_[ch] = upd ? c + 4 * g : ci_[ch];
            }
        }
        tb = cb_[0];
        ti = ci_[0];
#pragma unroll
        for (int ch = 1; ch < 4; ++ch) {
            const bool upd = cb_[ch] > tb;
            tb = upd ? cb_[ch] : tb;
            ti = upd ? ci_[ch] : ti;
        }
        const bool upd = tb > best;
        best = upd ? tb : best;
        best_idx = upd ? t * kBN + ti : best_idx;
#ifdef SVG_KMEANS_TRACE
        asm volatile("" : "+v"(best), "+v"(best_idx));
#endif
        KM_TICK(2)
        if (t + 1 < nT) write(buf ^ 1);
        if (t + 2 < nT) issue(t + 2);
        KM_TICK(3)
        __syncthreads();
        KM_TICK(4)
    }
#ifdef SVG_KMEANS_TRACE
    if (blockIdx.x == 7 && blockIdx.y == 0 && lane == 0) {
        for (int i = 0; i < 5; ++i) g_km_trace[wave * 8 + i] = tr[i];
        g_km_trace[wave * 8 + 5] = (unsigned long long)nT;
    }
#endif
    // the two lanes of a point cover disjoint centroid subsets: merge, lowest index wins ties
    const float ob = __shfl_xor(best, 32);
    const int oi = __shfl_xor(best_idx, 32);
    if (ob > best || (ob == best && oi < best_idx)) best = ob, best_idx = oi;
    if (g == 0 && n < N) labels[(size_t)b * N + n] = best_idx;
}

// ---- centroid update: grid = (groups, B), block = 256 = 16 row slots x 16 lanes (D = 128) / 32 x 8 (D = 64); a workgroup walks clusters
// k = blockIdx.x, + gridDim.x, ... of its batch.  (Until round 5 one workgroup per cluster: at K = 1000 that is 40000 workgroups of ~75 rows whose
// lifetime is three dependent latencies — count / offset, row indices, rows — 0.36 ms per call at Wan 720p where the K = 300 side, HBM-bound, takes
// 0.16.  Now the next cluster's count, offset and first four row indices per slot are fetched while the current one is summed.  The order of every
// sum is unchanged: the same bits.)
template <typename T, int D>
__global__ __launch_bounds__(256) void kmeans_update_kernel(const T* __restrict__ x, const T* __restrict__ c_old,
                                                            T* __restrict__ c_new, const int32_t* __restrict__ sorted_idx,
                                                            const int32_t* __restrict__ offsets /* [B][nchunks][K], chunk 0 */,
                                                            const int32_t* __restrict__ counts, float* __restrict__ shift,
                                                            int N, int K, size_t off_batch_stride, long long x_bs) {
    constexpr int LPR = D / 8;
    constexpr int SLOTS = 256 / LPR;
    __shared__ float red[SLOTS][D + 4];
    __shared__ float nrm[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int slot = tid / LPR, li = tid - slot * LPR;
    const T* xb = x + (size_t)b * (size_t)x_bs;
    const int32_t* sidx_b = sorted_idx + (size_t)b * N;
    using V8 = typename Elt<T>::v8;
    int k = blockIdx.x;
    if (k >= K) return;
    int cnt = counts[(size_t)b * K + k];
    int start = offsets[(size_t)b * off_batch_stride + k];
    int pre[4];     // the slot's first four row indices of the cluster (rows slot, slot + SLOTS, ...)
#pragma unroll
    for (int u = 0; u < 4; ++u) pre[u] = (slot + u * SLOTS < cnt) ? sidx_b[start + slot + u * SLOTS] : 0;
    for (;;) {
        const int kn = k + (int)gridDim.x;
        int cnt_n = 0, start_n = 0;
        if (kn < K) {
            cnt_n = counts[(size_t)b * K + kn];
            start_n = offsets[(size_t)b * off_batch_stride + kn];
        }
        const int32_t* sidx = sidx_b + start;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        // four rows of a slot in flight (index loads, then row loads); summed in the order a one-row loop sums them
        int r = slot;
        {
#pragma clang fp reassociate(off)
            if (r + 3 * SLOTS < cnt) {     // the first batch: its indices are here already
                V8 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *(const V8*)(xb + (size_t)pre[u] * D + li * 8);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += Elt<T>::to_float(v[u][j]);
                r += 4 * SLOTS;
                for (; r + 3 * SLOTS < cnt; r += 4 * SLOTS) {
                    int row[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) row[u] = sidx[r + u * SLOTS];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = *(const V8*)(xb + (size_t)row[u] * D + li * 8);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] += Elt<T>::to_float(v[u][j]);
                }
                for (; r < cnt; r += SLOTS) {
                    const V8 w = *(const V8*)(xb + (size_t)sidx[r] * D + li * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += Elt<T>::to_float(w[j]);
                }
            } else {                       // at most three rows for this slot, all of them prefetched
                V8 v[3];
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (slot + u * SLOTS < cnt) v[u] = *(const V8*)(xb + (size_t)pre[u] * D + li * 8);
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (slot + u * SLOTS < cnt) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] += Elt<T>::to_float(v[u][j]);
                    }
            }
        }
        // the next cluster's first indices: in flight during the reduction below
        if (kn < K) {
#pragma unroll
            for (int u = 0; u < 4; ++u) pre[u] = (slot + u * SLOTS < cnt_n) ? sidx_b[start_n + slot + u * SLOTS] : 0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[slot][li * 8 + j] = acc[j];
        __syncthreads();
        float d2 = 0.f;
        if (tid < D) {
            float s = 0.f;
            for (int sl = 0; sl < SLOTS; ++sl) s += red[sl][tid];  // fixed order -> deterministic
            const size_t o = ((size_t)b * K + k) * D + tid;
            const float oldv = Elt<T>::to_float(c_old[o]);
            // ref: svg/kmeans_utils.py:416-421 sums / clamp(count,1), empty cluster keeps the old centroid, cast to x.dtype
            const T nv = cnt > 0 ? Elt<T>::from_float(s / (float)cnt) : c_old[o];
            c_new[o] = nv;
            const float df = Elt<T>::to_float(Elt<T>::from_float(Elt<T>::to_float(nv) - oldv));
            d2 = df * df;
        }
        d2 = wave_sum(d2);
        if ((tid & 63) == 0) nrm[tid >> 6] = d2;
        __syncthreads();
        if (tid == 0) {
            const float nr = sqrtf(nrm[0] + nrm[1] + nrm[2] + nrm[3]);
            atomicMax((int*)(shift + b), __float_as_int(nr));  // non-negative floats order like ints
        }
        if (kn >= K) break;
        k = kn, cnt = cnt_n, start = start_n;
        // (red / nrm of this cluster are read before the barriers above; the next writes to red come after the second one, the next write to
        //  nrm after the next cluster's first barrier: tid 0 has read nrm by then only if it passed that barrier too — it has: same barrier)
    }
}

// the two halves of an iteration (the reference's euclid_assign_triton / triton_centroid_update_sorted_euclid, svg/kmeans_utils.py:562-627,
// 375-421) as host functions of their own: svg_kmeans_iter runs one after the other, svg_kmeans_assign / svg_kmeans_update expose them
template <typename T, int D>
static int run_kmeans_assign(const void* x, const float* xsq, const void* c_in, int32_t* labels, int B, int N, int K, void* ws,
                             hipStream_t st, long long x_bs) {
    constexpr int NW = 8;
    float* csq = (float*)ws;
    hipLaunchKernelGGL((csq_kernel<T>), dim3(std::min<long long>(2048, ((long long)B * K + 15) / 16)), dim3(256), 0, st,
                       (const T*)c_in, csq, (long long)B * K, D);
    constexpr int lds = 2 * (LdsLayout<D>::kKBytes + kBN * 4);
    auto kern = kmeans_assign_kernel<T, D, NW>;
    hipLaunchKernelGGL(kern, dim3((N + NW * 32 - 1) / (NW * 32), B), dim3(NW * 64), lds, st, (const T*)x, (const T*)c_in, xsq,
                       csq, labels, N, K, x_bs);
    return launch_status();
}

template <typename T, int D>
static int run_kmeans_update(const void* x, const void* c_in, void* c_out, const int32_t* labels, int32_t* counts, int32_t* sorted_idx,
                             float* shift, int B, int N, int K, void* ws, hipStream_t st, long long x_bs) {
    const size_t csq_bytes = ((size_t)B * K * sizeof(float) + 255) / 256 * 256;
    char* sort_ws = (char*)ws + csq_bytes;
    const size_t sort_bytes = svg_argsort_workspace_bytes(B, N, K);
    int rc = svg_argsort_labels(labels, sorted_idx, counts, B, N, K, sort_ws, sort_bytes, (void*)st);
    if (rc) return rc;
    (void)hipMemsetAsync(shift, 0, (size_t)B * sizeof(float), st);
    const int nchunks = (N + 1023) / 1024;
    // workgroups per batch: three per CU over the whole launch.  Measured at Wan 720p (40 heads, K = 300 and 1000 on two streams, the 2-iteration
    // stage): 6 per head 3.49 ms, 8 3.27, 12 3.14, 16 3.05, 20 3.02, 26 3.13, 32 3.17, 52 3.22, 100 3.36, one per cluster 3.37; the kernel of
    // round 4 (one workgroup per cluster, no prefetch) 3.39 (profiles/r05zw_kmeans_update_groups.txt).  SVG_KMEANS_UPDATE_GROUPS: A/B builds
    int groups = std::max(1, std::min(K, (3 * kNumCU + B - 1) / B));
#ifdef SVG_KMEANS_UPDATE_GROUPS_ENV
    if (const char* e = getenv("SVG_KMEANS_UPDATE_GROUPS")) groups = std::max(1, std::min(K, atoi(e)));
#endif
    hipLaunchKernelGGL((kmeans_update_kernel<T, D>), dim3(groups, B), dim3(256), 0, st, (const T*)x, (const T*)c_in, (T*)c_out,
                       sorted_idx, (const int32_t*)sort_ws, counts, shift, N, K, (size_t)nchunks * K, x_bs);
    return launch_status();
}

template <typename T, int D>
static int run_kmeans_iter(const void* x, const float* xsq, const void* c_in, void* c_out, int32_t* labels, int32_t* counts,
                           int32_t* sorted_idx, float* shift, int B, int N, int K, void* ws, size_t ws_bytes,
                           hipStream_t st, long long x_bs) {
    (void)ws_bytes;
    if (const int rc = run_kmeans_assign<T, D>(x, xsq, c_in, labels, B, N, K, ws, st, x_bs); rc != SVG_OK) return rc;
    return run_kmeans_update<T, D>(x, c_in, c_out, labels, counts, sorted_idx, shift, B, N, K, ws, st, x_bs);
}

// ---- the Lloyd loop on the device (svg_kmeans_loop): commit of one iteration's result under the reference's stopping rule ----
// state[0..1]: "the reference's loop has left" before iteration `it` (ping-pong by iteration parity: every thread of the commit
// kernel reads the flag of the iterations before, one thread writes the flag for the next), state[2]: n_iters, state[3]: which
// centroid buffer holds the result (0 initial, 1 / 2 the two work buffers).
__global__ __launch_bounds__(256) void kmeans_commit_kernel(const int32_t* __restrict__ labels, const int32_t* __restrict__ sorted_idx,
                                                            const int32_t* __restrict__ counts, int32_t* __restrict__ labels_r,
                                                            int32_t* __restrict__ sorted_r, int32_t* __restrict__ counts_r,
                                                            const float* __restrict__ shift, int32_t* __restrict__ state, long long bn,
                                                            long long bk, int B, float tol, int it, int sel_cur, int sel_out) {
    const bool stopped = state[it & 1] != 0;
    if (!stopped) {   // ref svg/kmeans_utils.py:716-733: the labels / sizes of the iteration the loop is in are the result
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < bn; i += (long long)gridDim.x * 256) {
            labels_r[i] = labels[i];
            sorted_r[i] = sorted_idx[i];
        }
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < bk; i += (long long)gridDim.x * 256) counts_r[i] = counts[i];
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        float mx = 0.f;
        bool nan = false;   // fmaxf drops a NaN operand; the reference's `shift.max()` propagates it, and `NaN < tol` is False (:723)
        for (int b = threadIdx.x; b < B; b += 64) {
            const float sft = shift[b];
            nan |= sft != sft;
            mx = fmaxf(mx, sft);
        }
        mx = wave_max(mx);
        nan = __builtin_amdgcn_ballot_w64(nan) != 0ull;
        if (threadIdx.x == 0) {
            const bool conv = !nan && mx < tol;   // `center_shift < tol` (:723): a NaN shift never converges, the loop runs to max_iters
            if (!stopped) {
                state[2] += 1;
                state[3] = conv ? sel_cur : sel_out;   // converged: the OLD centroids are returned; otherwise the new ones become current
            }
            state[(it + 1) & 1] = (stopped || conv) ? 1 : 0;
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void kmeans_select_kernel(const T* __restrict__ c0, const T* __restrict__ c1, const T* __restrict__ c2,
                                                            T* __restrict__ out, const int32_t* __restrict__ state, long long n8) {
    const int sel = state[3];
    const T* src = sel == 0 ? c0 : (sel == 1 ? c1 : c2);
    using V8 = typename Elt<T>::v8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256)
        ((V8*)out)[i] = ((const V8*)src)[i];
}

}  // namespace svg

using namespace svg;

#ifdef SVG_KMEANS_TRACE
extern "C" int svg_debug_kmeans_trace(uint64_t* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_km_trace), sizeof(uint64_t) * 64) == hipSuccess ? SVG_OK : SVG_ERR_LAUNCH;
}
#endif

extern "C" int svg_kmeans_xsq(const void* x, float* xsq, int32_t B, int32_t N, int32_t D, int32_t dtype, void* stream) {
    if (!x || !xsq || B <= 0 || N <= 0) return SVG_ERR_BAD_ARG;
    if (D != 64 && D != 128) return SVG_ERR_UNSUPPORTED;
    const long long rows = (long long)B * N;
    const int rpb = 256 / (D / 8);
    dim3 grid((unsigned)std::min<long long>(4096, (rows + rpb - 1) / rpb));
    if (dtype == SVG_DTYPE_BF16)
        hipLaunchKernelGGL((xsq_kernel<__bf16>), grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, xsq, rows, D);
    else if (dtype == SVG_DTYPE_F16)
        hipLaunchKernelGGL((xsq_kernel<_Float16>), grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, xsq, rows, D);
    else
        return SVG_ERR_UNSUPPORTED;
    return launch_status();
}

extern "C" size_t svg_kmeans_workspace_bytes(int32_t B, int32_t N, int32_t K, int32_t D) {
    (void)D;
    if (B <= 0 || N <= 0 || K <= 0) return 0;
    const size_t csq_bytes = ((size_t)B * K * sizeof(float) + 255) / 256 * 256;
    return csq_bytes + svg_argsort_workspace_bytes(B, N, K);
}

// one Lloyd iteration on x whose batches are x_bs elements apart (rows contiguous inside a batch): svg_kmeans_iter (x_bs = N * D) and the
// iterations of svg_kmeans_loop[_strided]
static int kmeans_iter_impl(const void* x, long long x_bs, const float* xsq, const void* centroids_in, void* centroids_out,
                            int32_t* labels, int32_t* counts, int32_t* sorted_idx, float* shift, int32_t B, int32_t N,
                            int32_t K, int32_t D, int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    // (xsq may be NULL: the argmax form of the assignment does not use |x|^2 — see kmeans_assign_kernel; the parameter stays in the
    //  signature because the reference's distance form carries it, svg/kmeans_utils.py:704)
    if (!x || !centroids_in || !centroids_out || !labels || !counts || !sorted_idx || !shift || !workspace)
        return SVG_ERR_BAD_ARG;
    if (B <= 0 || N <= 0 || K <= 0) return SVG_ERR_BAD_ARG;
    if (K > 8192) return SVG_ERR_UNSUPPORTED;
    if (workspace_bytes < svg_kmeans_workspace_bytes(B, N, K, D)) return SVG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SVG_DTYPE_BF16) {
        if (D == 128)
            return run_kmeans_iter<__bf16, 128>(x, xsq, centroids_in, centroids_out, labels, counts, sorted_idx, shift, B, N,
                                                K, workspace, workspace_bytes, st, x_bs);
        if (D == 64)
            return run_kmeans_iter<__bf16, 64>(x, xsq, centroids_in, centroids_out, labels, counts, sorted_idx, shift, B, N, K,
                                               workspace, workspace_bytes, st, x_bs);
    } else if (dtype == SVG_DTYPE_F16) {
        if (D == 128)
            return run_kmeans_iter<_Float16, 128>(x, xsq, centroids_in, centroids_out, labels, counts, sorted_idx, shift, B, N,
                                                  K, workspace, workspace_bytes, st, x_bs);
        if (D == 64)
            return run_kmeans_iter<_Float16, 64>(x, xsq, centroids_in, centroids_out, labels, counts, sorted_idx, shift, B, N,
                                                 K, workspace, workspace_bytes, st, x_bs);
    }
    return SVG_ERR_UNSUPPORTED;
}

extern "C" int svg_kmeans_iter(const void* x, const float* xsq, const void* centroids_in, void* centroids_out,
                               int32_t* labels, int32_t* counts, int32_t* sorted_idx, float* shift, int32_t B, int32_t N,
                               int32_t K, int32_t D, int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return kmeans_iter_impl(x, (long long)N * D, xsq, centroids_in, centroids_out, labels, counts, sorted_idx, shift, B, N, K, D, dtype,
                            workspace, workspace_bytes, stream);
}

#define SVG_KM_DISPATCH(CALL)                                                                  \
    if (dtype == SVG_DTYPE_BF16) {                                                             \
        if (D == 128) return CALL(__bf16, 128);                                                \
        if (D == 64) return CALL(__bf16, 64);                                                  \
    } else if (dtype == SVG_DTYPE_F16) {                                                       \
        if (D == 128) return CALL(_Float16, 128);                                              \
        if (D == 64) return CALL(_Float16, 64);                                                \
    }                                                                                          \
    return SVG_ERR_UNSUPPORTED;

// labels[b, n] = argmin_k |x[b, n] - centroids[b, k]|^2 (lowest index on ties): the assignment half of svg_kmeans_iter on its own.
extern "C" int svg_kmeans_assign(const void* x, const void* centroids, int32_t* labels, int32_t B, int32_t N, int32_t K, int32_t D,
                                 int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !centroids || !labels || !workspace || B <= 0 || N <= 0 || K <= 0) return SVG_ERR_BAD_ARG;
    if (K > 8192) return SVG_ERR_UNSUPPORTED;
    if (workspace_bytes < svg_kmeans_workspace_bytes(B, N, K, D)) return SVG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
#define SVG_KM_ASSIGN(T, DD) run_kmeans_assign<T, DD>(x, nullptr, centroids, labels, B, N, K, workspace, st, (long long)N * DD)
    SVG_KM_DISPATCH(SVG_KM_ASSIGN)
#undef SVG_KM_ASSIGN
}

// The update half on GIVEN labels: stable sort of the labels, per-cluster fp32 means in a fixed order, empty clusters keep
// centroids_in, shift[b] = the largest centre movement (as svg_kmeans_iter).
extern "C" int svg_kmeans_update(const void* x, const int32_t* labels, const void* centroids_in, void* centroids_out, int32_t* counts,
                                 int32_t* sorted_idx, float* shift, int32_t B, int32_t N, int32_t K, int32_t D, int32_t dtype,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !labels || !centroids_in || !centroids_out || !counts || !sorted_idx || !shift || !workspace) return SVG_ERR_BAD_ARG;
    if (B <= 0 || N <= 0 || K <= 0) return SVG_ERR_BAD_ARG;
    if (K > 8192) return SVG_ERR_UNSUPPORTED;
    if (workspace_bytes < svg_kmeans_workspace_bytes(B, N, K, D)) return SVG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
#define SVG_KM_UPDATE(T, DD) run_kmeans_update<T, DD>(x, centroids_in, centroids_out, labels, counts, sorted_idx, shift, B, N, K, workspace, st, (long long)N * DD)
    SVG_KM_DISPATCH(SVG_KM_UPDATE)
#undef SVG_KM_UPDATE
}
#undef SVG_KM_DISPATCH

extern "C" size_t svg_kmeans_loop_workspace_bytes(int32_t B, int32_t N, int32_t K, int32_t D) {
    const size_t it = svg_kmeans_workspace_bytes(B, N, K, D);
    if (it == 0) return 0;
    // per-iteration scratch of svg_kmeans_iter + scratch labels / sorted indices / counts / shift of one iteration + the state words
    const size_t a = ((size_t)B * N * 4 + 255) / 256 * 256, c = ((size_t)B * K * 4 + 255) / 256 * 256, sh = ((size_t)B * 4 + 255) / 256 * 256;
    return (it + 255) / 256 * 256 + 2 * a + c + sh + 256;
}

// svg_kmeans_loop (x_bs = N * D) and svg_kmeans_loop_strided
static int kmeans_loop_impl(const void* x, long long x_bs, const float* xsq, const void* c_init, void* c_work_a, void* c_work_b, int32_t* labels,
                            int32_t* counts, int32_t* sorted_idx, void* centroids_out, int32_t* n_iters, int32_t B, int32_t N,
                            int32_t K, int32_t D, int32_t dtype, int32_t max_iters, float tol, void* workspace,
                            size_t workspace_bytes, void* stream) {
    if (!x || !c_init || !c_work_a || !c_work_b || !labels || !counts || !sorted_idx || !centroids_out || !n_iters || !workspace)
        return SVG_ERR_BAD_ARG;   // (xsq may be NULL, see svg_kmeans_iter)
    if (B <= 0 || N <= 0 || K <= 0 || max_iters <= 0) return SVG_ERR_BAD_ARG;
    if (K > 8192 || (D != 64 && D != 128) || (dtype != SVG_DTYPE_BF16 && dtype != SVG_DTYPE_F16)) return SVG_ERR_UNSUPPORTED;
    if (workspace_bytes < svg_kmeans_loop_workspace_bytes(B, N, K, D)) return SVG_ERR_WORKSPACE;
    if (c_init == c_work_a || c_init == c_work_b || c_work_a == c_work_b || centroids_out == c_work_a || centroids_out == c_work_b)
        return SVG_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t it_bytes = (svg_kmeans_workspace_bytes(B, N, K, D) + 255) / 256 * 256;
    const size_t a = ((size_t)B * N * 4 + 255) / 256 * 256, c = ((size_t)B * K * 4 + 255) / 256 * 256, sh = ((size_t)B * 4 + 255) / 256 * 256;
    char* w = (char*)workspace;
    void* it_ws = w;
    int32_t* t_labels = (int32_t*)(w + it_bytes);
    int32_t* t_sorted = (int32_t*)(w + it_bytes + a);
    int32_t* t_counts = (int32_t*)(w + it_bytes + 2 * a);
    float* t_shift = (float*)(w + it_bytes + 2 * a + c);
    int32_t* state = (int32_t*)(w + it_bytes + 2 * a + c + sh);
    if (hipMemsetAsync(state, 0, 4 * sizeof(int32_t), st) != hipSuccess) return SVG_ERR_LAUNCH;
    const void* cur = c_init;
    int sel_cur = 0;
    const long long bn = (long long)B * N, bk = (long long)B * K;
    const unsigned grid = (unsigned)std::min<long long>(2048, (bn + 255) / 256);
    for (int it = 0; it < max_iters; ++it) {
        void* out = (it & 1) ? c_work_b : c_work_a;
        const int sel_out = (it & 1) ? 2 : 1;
        const int rc = kmeans_iter_impl(x, x_bs, xsq, cur, out, t_labels, t_counts, t_sorted, t_shift, B, N, K, D, dtype, it_ws, it_bytes, stream);
        if (rc != SVG_OK) return rc;
        hipLaunchKernelGGL(kmeans_commit_kernel, dim3(grid), dim3(256), 0, st, t_labels, t_sorted, t_counts, labels, sorted_idx, counts,
                           t_shift, state, bn, bk, B, tol, it, sel_cur, sel_out);
        cur = out, sel_cur = sel_out;
    }
    const long long n8 = bk * D / 8;
    const unsigned g2 = (unsigned)std::min<long long>(1024, (n8 + 255) / 256);
    if (dtype == SVG_DTYPE_BF16)
        hipLaunchKernelGGL((kmeans_select_kernel<__bf16>), dim3(g2), dim3(256), 0, st, (const __bf16*)c_init, (const __bf16*)c_work_a,
                           (const __bf16*)c_work_b, (__bf16*)centroids_out, state, n8);
    else
        hipLaunchKernelGGL((kmeans_select_kernel<_Float16>), dim3(g2), dim3(256), 0, st, (const _Float16*)c_init, (const _Float16*)c_work_a,
                           (const _Float16*)c_work_b, (_Float16*)centroids_out, state, n8);
    if (hipMemcpyAsync(n_iters, state + 2, sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess) return SVG_ERR_LAUNCH;
    return launch_status();
}

extern "C" int svg_kmeans_loop(const void* x, const float* xsq, const void* c_init, void* c_work_a, void* c_work_b, int32_t* labels,
                               int32_t* counts, int32_t* sorted_idx, void* centroids_out, int32_t* n_iters, int32_t B, int32_t N,
                               int32_t K, int32_t D, int32_t dtype, int32_t max_iters, float tol, void* workspace,
                               size_t workspace_bytes, void* stream) {
    return kmeans_loop_impl(x, (long long)N * D, xsq, c_init, c_work_a, c_work_b, labels, counts, sorted_idx, centroids_out, n_iters, B, N, K, D,
                            dtype, max_iters, tol, workspace, workspace_bytes, stream);
}

extern "C" int svg_kmeans_loop_strided(const void* x, int64_t x_batch_stride, const void* c_init, void* c_work_a, void* c_work_b,
                                       int32_t* labels, int32_t* counts, int32_t* sorted_idx, void* centroids_out, int32_t* n_iters,
                                       int32_t B, int32_t N, int32_t K, int32_t D, int32_t dtype, int32_t max_iters, float tol,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    if (x_batch_stride < (int64_t)N * D || x_batch_stride % 8 != 0 || ((size_t)x & 15) != 0) return x_batch_stride < (int64_t)N * D ? SVG_ERR_BAD_ARG : SVG_ERR_UNSUPPORTED;
    return kmeans_loop_impl(x, (long long)x_batch_stride, nullptr, c_init, c_work_a, c_work_b, labels, counts, sorted_idx, centroids_out, n_iters,
                            B, N, K, D, dtype, max_iters, tol, workspace, workspace_bytes, stream);
}
