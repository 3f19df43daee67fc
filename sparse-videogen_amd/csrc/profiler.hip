// Online profiler of SVG1 (`sample_mse`): for R sampled query rows per head compute the dense attention output
// ("golden") and the outputs under the two candidate masks, return the per-head mean squared errors.
// ref: svg/models/hyvideo/attention.py:376-399 (wan/attention.py:211-234, cog/attention.py:120-145) and the
// profiling masks of get_attention_mask (hyvideo/utils.py:47-93, wan/utils.py:63-110, cog/utils.py:61-88).
//
// The reference materialises two [10000, S] fp32 masks (4.8 GB each) and runs three full softmaxes in torch.
// Here the masks are analytic predicates and the three variants are three wave roles of ONE workgroup that shares
// attn_core.h: grid = (BH, kv_chunks); waves 0-1 compute the golden rows, waves 2-3 the rows under mask 1, waves 4-5 under
// mask 0 (waves 6-7 only help staging) on the SAME staged K/V tiles, so a chunk of K/V is read and staged once (split-KV);
// every role emits un-normalised fp32 partials (O, m, l) per sampled row, and a small second kernel merges the chunks,
// normalises and reduces the MSE.  (The first version ran the roles as three workgroups, grid.z = 3: K/V staged three times,
// 1.26 ms per call at Hunyuan 720p.)
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

constexpr int kProfNW = 8;      // waves per workgroup: 3 roles x 2 waves x 32 sampled rows, the last two waves have no rows
constexpr int kProfMaxRows = 64;
constexpr int kProfRoleRows = 3 * kProfMaxRows;   // workgroup rows [64 v, 64 v + 64) belong to role v

template <typename T, int D>
struct ProfilePolicy {
    static constexpr bool kFixup = true;
    static constexpr bool kPartialOut = true;
    static constexpr bool kIntervalMask = false;
    static constexpr int kShadow128 = 1;
    static constexpr bool kFastPartial = true;     // token-major mask: tiles inside one frame row block, see classify()   // the profiling masks are general element predicates (allowed())
    static constexpr int kAbl = 0;
    static constexpr bool kSetPrio = false;
    static constexpr bool kSkew = false;
    static constexpr int kRowBlocks = 1;
    static constexpr int kSubTiles = 1;
    static constexpr int kPrefetch = 1;
    // (225 registers: ONE workgroup per CU, so the 504 workgroups of a HunyuanVideo call run in two rounds of ~0.25 ms, each tile waiting out the
    //  latency of loads issued one tile ahead.  Round 5 tried a second staging register set (loads of tile t + 3 in flight during tile t, 241
    //  registers): as plain loads hipcc's waitcnt pass drains both sets (vmcnt(0)) in front of the older set's LDS writes — nothing gained —, and
    //  as inline-asm loads with hand-placed waits the compiler copies the destination registers ahead of the wait across the loop's parity
    //  branches: memory faults.  Removed; what shipped is the fast predicate on frame-major tiles in classify() below.)
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
        const int32_t* skip; // device flag (or nullptr): non-zero = this call is not needed (dense step), every kernel returns at once
    };
    struct Ctx {
        int head, variant, chunk, t0, nT;
        ProfVariant pv;
        // per-lane: this lane's sampled row in mask coordinates; per-wave: block range of the wave's rows
        int qx, qtext;
        int xlo_blk, xhi_blk, any_text;
        // per-tile scratch written by classify(): token-major decomposition of the tile's first key
        mutable int tk0, f0, p0;
        // token-major fast tiles (TILE_PARTIAL_FAST): the keys of the tile are y = ybase + off * F in mask coordinates (one frame,
        // no text keys), and this lane's row sees y in [fa0, fa0 + falen) u [0, fblen): band blocks n domain, sink columns n domain
        mutable int ybase;
        int fa0, g4F, ystride;   // ystride: coordinate step between consecutive keys of a fast tile (F token-major, 1 frame-major)
        unsigned falen, fblen;
    };
    struct KvCursor {};
    static __device__ __forceinline__ bool wave_active(const Ctx&, int wrow0) { return wrow0 < kProfRoleRows; }

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

    static __device__ __forceinline__ bool init(const Params& p, Ctx& c, char*) {
        if (p.skip && p.skip[0] != 0) return false;
        // chunk-major dispatch: the sampled rows are low (< sample_mse_max_row), so the frame-major mask only has work in the first
        // chunks of every head, on the expensive element predicate; dispatching all heads' chunk 0, 1, ... first puts the long
        // workgroups at the front of the launch instead of into its tail
        c.chunk = blockIdx.y;
        c.head = blockIdx.x;
        // two waves per role; a wave lives on SIMD (wave % 4): the golden rows (waves 0-1) and the token-major mask (waves 2-3, the
        // expensive predicate) get a SIMD each, the frame-major mask (waves 4-5, skips most tiles) shares with the golden waves
        const int role = (int)(threadIdx.x >> 7);
        c.variant = role == 0 ? 0 : (role == 1 ? 2 : 1);
        const int ntiles = (p.S + kBN - 1) / kBN;
        c.t0 = c.chunk * p.tiles_per_chunk;
        c.nT = max(0, min(p.tiles_per_chunk, ntiles - c.t0));
        c.pv = p.var[c.variant == 2 ? 1 : 0];
        // this lane's query row (rows >= R do not exist) in the coordinates of the variant's mask
        const int row = (threadIdx.x >> 6) * 32 + (threadIdx.x & 31);
        const bool have = exists(p, row);
        const int q = have ? (int)p.rows[row & (kProfMaxRows - 1)] : 0;
        c.qx = coord(p, c.pv, q) - c.pv.origin;
        c.qtext = have && ((unsigned)(q - c.pv.text_lo) < (unsigned)(c.pv.text_hi - c.pv.text_lo));
        const int blk = c.qx >> 7;
        int lo = have ? blk : (1 << 28), hi = have ? blk : -(1 << 28), anyt = c.qtext;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o));
            hi = max(hi, __shfl_xor(hi, o));
            anyt |= __shfl_xor(anyt, o);
        }
        c.xlo_blk = lo, c.xhi_blk = hi, c.any_text = anyt;
        c.tk0 = 0, c.f0 = 0, c.p0 = 0;
        c.ystride = c.pv.coord == 1 ? p.F : 1;
        c.ybase = 0, c.g4F = 4 * (int)((threadIdx.x >> 5) & 1) * c.ystride;
        {
            const int x = c.qx, span = c.pv.span;
            const bool xdom = have && ((unsigned)x < (unsigned)span);
            const int ylo = ((x >> 7) - c.pv.band_blocks + 1) * 128, yhi = ((x >> 7) + c.pv.band_blocks) * 128;
            const int a0 = max(ylo, 0), a1 = min(yhi, span);
            c.fa0 = a0, c.falen = xdom ? (unsigned)max(a1 - a0, 0) : 0u;
            c.fblen = (xdom && c.pv.sink_cols > 0) ? (unsigned)min(c.pv.sink_cols, span) : 0u;
            if (c.qtext) c.fa0 = -(1 << 30), c.falen = 0xFFFFFFFFu;   // a text row sees every key
        }
        return true;
    }
    static __device__ __forceinline__ const T* q_base(const Params& p, const Ctx& c) { return p.q + (size_t)c.head * p.S * D; }
    static __device__ __forceinline__ const T* k_base(const Params& p, const Ctx& c) { return p.k + (size_t)c.head * p.S * D; }
    static __device__ __forceinline__ const T* v_base(const Params& p, const Ctx& c) { return p.v + (size_t)c.head * p.S * D; }

    // workgroup row -> sampled row of its role
    static __device__ __forceinline__ bool exists(const Params& p, int row) {
        return row < kProfRoleRows && (row & (kProfMaxRows - 1)) < p.R;
    }
    static __device__ __forceinline__ int q_phys(const Params& p, const Ctx&, int row) {
        return exists(p, row) ? (int)p.rows[row & (kProfMaxRows - 1)] : -1;
    }
    static __device__ __forceinline__ int q_logical(const Ctx&, int row) { return row; }
    static __device__ __forceinline__ int tile_key0(const Ctx& c, int t) { return (c.t0 + t) * kBN; }
    static __device__ __forceinline__ void kv_cursor_init(const Params&, const Ctx&, KvCursor&, int) {}
    static __device__ __forceinline__ int kv_phys(const Params& p, const Ctx& c, KvCursor&, int t, int row) {
        const int l = (c.t0 + t) * kBN + row;
        return l < p.S ? l : 0;  // masked by allowed()
    }
    static __device__ __forceinline__ int classify(const Params& p, const Ctx& c, int k0, int wrow0) {
        if (!exists(p, wrow0)) return TILE_SKIP;
        if (c.variant == 0) return (k0 + kBN <= p.S) ? TILE_FULL : TILE_PARTIAL;
        const ProfVariant& pv = c.pv;
        const int k1 = min(k0 + kBN, p.S);
        if (pv.coord == 0) {
            // frame-major mask: drop tiles no row of this wave can see (most of the sequence for the spatial mask)
            const bool text_keys = (k0 < pv.text_hi) && (k1 > pv.text_lo);
            if (!c.any_text && !text_keys) {
                const int y0 = k0 - pv.origin, y1 = k1 - 1 - pv.origin;
                const bool outside = (y1 < 0) || (y0 >= pv.span);
                const bool far = ((y0 >> 7) - c.xhi_blk >= pv.band_blocks) || (c.xlo_blk - (y1 >> 7) >= pv.band_blocks);
                const bool no_sink = y0 >= pv.sink_cols;
                if (outside || (far && no_sink && y0 >= 0)) return TILE_SKIP;
            }
            // frame-major mask, all 64 keys inside the band domain and none of them a text column: the keys are consecutive coordinates
            // y = ybase + off, and the element predicate collapses to the two interval tests of allowed_fast like the token-major tiles
            // below.  (Without this every tile the spatial role does not skip paid the general predicate, ~25 instructions per element:
            // the workgroups of a head's first chunks — where the sampled rows' bands lie — ran 2.5x as long as the others and set the
            // launch time: 0.66 ms at HunyuanVideo 720p with the average wave alive for 40 % of it, profiles/r05a_profiler_kernel_trace.txt.)
            if (!text_keys && k0 + kBN <= p.S && k0 - pv.origin >= 0 && k0 + kBN - pv.origin <= pv.span) {
                c.tk0 = k0;
                c.ybase = k0 - pv.origin;
                return TILE_PARTIAL_FAST;
            }
        } else {
            // token-major mask: one division per tile instead of one per element (keys of a tile are consecutive)
            const int i0 = max(k0 - p.vid0, 0);
            c.f0 = (int)((unsigned)i0 / (unsigned)p.P);
            c.p0 = i0 - c.f0 * p.P;
            // all 64 keys inside the video, inside one frame, none of them a text column: the general predicate collapses to
            // two interval tests on y = ybase + off * F (6 instead of ~25 instructions per element)
            const bool text_keys = (k0 < pv.text_hi) && (k0 + kBN > pv.text_lo);
            if (k0 >= p.vid0 && k0 + kBN <= p.vid0 + p.V && k0 + kBN <= p.S && c.p0 + kBN <= p.P && !text_keys) {
                c.tk0 = k0;
                c.ybase = p.vid0 + c.p0 * p.F + c.f0 - pv.origin;
                return TILE_PARTIAL_FAST;
            }
        }
        c.tk0 = k0;
        return TILE_PARTIAL;
    }
    static __device__ __forceinline__ bool allowed(const Params& p, const Ctx& c, int, int k) {
        if (c.variant == 0) return k < p.S;
        const ProfVariant& pv = c.pv;
        const bool tk = (unsigned)(k - pv.text_lo) < (unsigned)(pv.text_hi - pv.text_lo);
        int yc = k;
        if (pv.coord == 1) {
            const int off = k - c.tk0;  // 0..63, P >= 64 (checked on the host): at most one wrap into the next frame
            int pp = c.p0 + off, f = c.f0;
            const bool wrap = pp >= p.P;
            pp = wrap ? pp - p.P : pp;
            f = wrap ? f + 1 : f;
            const bool in_video = (unsigned)(k - p.vid0) < (unsigned)p.V;
            yc = in_video ? p.vid0 + pp * p.F + f : k;
        }
        const int x = c.qx, y = yc - pv.origin;
        const bool dom = ((unsigned)x < (unsigned)pv.span) & ((unsigned)y < (unsigned)pv.span);
        const int db = (x >> 7) - (y >> 7);
        const bool band = (db < pv.band_blocks) & (-db < pv.band_blocks);
        const bool sink = y < pv.sink_cols;
        return (k < p.S) & (bool)(c.qtext | tk | (dom & (band | sink)));
    }
    static __device__ __forceinline__ bool allowed_fast(const Params& p, const Ctx& c, int off) {
        const int y = c.ybase + c.g4F + off * c.ystride;   // off: key offset inside the tile without the lane's 4 g
        return ((unsigned)(y - c.fa0) < c.falen) | ((unsigned)y < c.fblen);
    }
    static __device__ __forceinline__ float score_fixup(const Params& p, float s) {
        if (!p.emulate) return s;
        // torch: (q @ k^T) rounds to the input dtype, "/ sqrt(D)" rounds again (attention.py:383)
        const float r1 = Elt<T>::to_float(Elt<T>::from_float(s));
        return Elt<T>::to_float(Elt<T>::from_float(r1 * p.fix_scale));
    }
    static __device__ __forceinline__ void store_partial(const Params& p, const Ctx& c, int row, int g, const f32x16* acc,
                                                         float m, float l) {
        if (!exists(p, row)) return;
        float* dst = p.part + ((((size_t)c.variant * p.BH + c.head) * p.n_chunks + c.chunk) * kProfMaxRows +
                               (row & (kProfMaxRows - 1))) * (D + 4);
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
    static __device__ __forceinline__ void notify(const Params&, const Ctx&) {}
};

// (The two-phase ping-pong body was tried here and is slower, 1.89 ms vs 1.22 ms: it walks every wave through every tile, and a
//  tile a masked role does not need still pays the element predicate, which the lock-step body skips.)
#ifdef SVG_PROF_TRACE
static __device__ unsigned long long g_prof_trace[2048 * 4];   // diagnostics build: per workgroup { start, end (s_memtime), hw id, chunk << 16 | head }
#endif
template <typename T, int D>
__global__ __launch_bounds__(kProfNW * 64, 2) void profile_attn_kernel(typename ProfilePolicy<T, D>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef SVG_PROF_TRACE
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    attn_body<T, D, kProfNW, ProfilePolicy<T, D>>(prm, smem, nullptr);
#ifdef SVG_PROF_TRACE
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
        if (wg < 2048) {
            g_prof_trace[wg * 4 + 0] = t0;
            g_prof_trace[wg * 4 + 1] = __builtin_amdgcn_s_memtime();
            g_prof_trace[wg * 4 + 2] = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
            g_prof_trace[wg * 4 + 3] = ((unsigned long long)blockIdx.y << 16) | blockIdx.x;
        }
    }
#endif
}

// merge the split-KV partials, normalise, and reduce squared errors.  grid = (kProfRowGroups, BH), block = 256:
// row group rg of head h -> sq_part[h][rg][0..1] (sum of squared errors of the two masks), [2..3] NaN flags
constexpr int kProfRowGroups = 32;   // (8 until round 5: 192 workgroups read 51 MB of partials in 0.112 ms; 32: 768 workgroups)

template <typename T, int D>
__global__ __launch_bounds__(256) void profile_combine_kernel(const float* __restrict__ part, float* __restrict__ sq_part, int BH,
                                                              int R, int n_chunks, int emulate, const int32_t* __restrict__ skip) {
    __shared__ float red[4][4];
    if (skip && skip[0] != 0) return;
    const int rg = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const int rows_per = (R + kProfRowGroups - 1) / kProfRowGroups;
    const int r0 = rg * rows_per, r1 = min(R, r0 + rows_per);
    float sq[2] = {0.f, 0.f};
    float bad[2] = {0.f, 0.f};
    constexpr int DS = D + 4;
    for (int e = r0 * D + tid; e < r1 * D; e += 256) {
        const int row = e / D, d = e - row * D;
        float o[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const float* base = part + ((((size_t)v * BH + h) * n_chunks) * kProfMaxRows + row) * DS;
            const size_t cs = (size_t)kProfMaxRows * DS;
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
                sq[v] += Elt<T>::to_float(Elt<T>::from_float(diff * diff));
            } else {
                sq[v] += diff * diff;
            }
            if (diff != diff) bad[v] = 1.f, sq[v] = 0.f;
        }
    }
    float vals[4] = {bad[0] > 0.f ? 0.f : sq[0], bad[1] > 0.f ? 0.f : sq[1], bad[0], bad[1]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float s = wave_sum(vals[i]);
        if ((tid & 63) == 0) red[i][tid >> 6] = s;
    }
    __syncthreads();
    if (tid < 4) sq_part[((size_t)h * kProfRowGroups + rg) * 4 + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
}

// grid = (1), block = 256: out_mse[v][h] = mean over rows and D (NaN when any sampled row had no visible key)
template <typename T>
__global__ __launch_bounds__(256) void profile_finalize_kernel(const float* __restrict__ sq_part, float* __restrict__ out_mse,
                                                               int BH, int R, int D, int emulate, const int32_t* __restrict__ skip) {
    if (skip && skip[0] != 0) return;
    for (int i = threadIdx.x; i < 2 * BH; i += 256) {
        const int v = i / BH, h = i - v * BH;
        float s = 0.f, bad = 0.f;
        for (int rg = 0; rg < kProfRowGroups; ++rg) {
            s += sq_part[((size_t)h * kProfRowGroups + rg) * 4 + v];
            bad += sq_part[((size_t)h * kProfRowGroups + rg) * 4 + 2 + v];
        }
        s = s / (float)(R * D);
        if (emulate) s = Elt<T>::to_float(Elt<T>::from_float(s));
        out_mse[(size_t)v * BH + h] = bad > 0.f ? __builtin_nanf("") : s;
    }
}

static int prof_chunks(int BH, int S) {
    const int ntiles = (S + kBN - 1) / kBN;
    int n = (2 * kNumCU) / BH;   // two 512-thread workgroups fit a CU: one round of the launch
    n = n < 1 ? 1 : n;
    n = n > ntiles ? ntiles : n;
    n = n > 64 ? 64 : n;
#ifdef SVG_PROF_CHUNKS_ENV
    if (const char* e = getenv("SVG_PROF_CHUNKS")) n = std::max(1, std::min(atoi(e), ntiles));   // (A/B builds only)
#endif
    return n;
}

template <typename T, int D>
static int run_profile(const void* q, const void* k, const void* v, const int64_t* rows, int R, int BH, int S, float sm_scale,
                       const svg_profile_desc_t* pd, float* out_mse, void* ws, const int32_t* skip, hipStream_t st) {
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
    p.skip = skip;
    const int lds = attn_lds_bytes<D, kProfNW>();
    auto kern = profile_attn_kernel<T, D>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return SVG_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(BH, p.n_chunks), dim3(kProfNW * 64), lds, st, p);
    float* sq_part = (float*)ws + (size_t)3 * BH * p.n_chunks * kProfMaxRows * (D + 4);
    hipLaunchKernelGGL((profile_combine_kernel<T, D>), dim3(kProfRowGroups, BH), dim3(256), 0, st, (const float*)ws, sq_part, BH,
                       R, p.n_chunks, p.emulate, skip);
    hipLaunchKernelGGL((profile_finalize_kernel<T>), dim3(1), dim3(256), 0, st, (const float*)sq_part, out_mse, BH, R, D,
                       p.emulate, skip);
    return launch_status();
}

}  // namespace svg

using namespace svg;

#ifdef SVG_PROF_TRACE
extern "C" int svg_debug_prof_trace(uint64_t* out, int n_workgroups) {
    if (!out || n_workgroups <= 0 || n_workgroups > 2048) return SVG_ERR_BAD_ARG;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof_trace), (size_t)n_workgroups * 4 * sizeof(uint64_t)) == hipSuccess ? SVG_OK : SVG_ERR_LAUNCH;
}
#endif

extern "C" size_t svg_sample_mse_workspace_bytes(int32_t BH, int32_t R, int32_t D, int32_t S) {
    if (BH <= 0 || R <= 0 || D <= 0 || S <= 0) return 0;
    return ((size_t)3 * BH * prof_chunks(BH, S) * kProfMaxRows * (D + 4) + (size_t)BH * kProfRowGroups * 4) * sizeof(float);
}

extern "C" int svg_sample_mse_flagged(const void* q, const void* k, const void* v, const int64_t* rows, int32_t R, int32_t BH,
                                      int32_t S, int32_t D, int32_t dtype, float sm_scale, const svg_profile_desc_t* prof,
                                      float* out_mse, void* workspace, size_t workspace_bytes, const int32_t* skip_flag,
                                      void* stream) {
    if (!q || !k || !v || !rows || !prof || !out_mse || !workspace) return SVG_ERR_BAD_ARG;
    if (R <= 0 || BH <= 0 || S <= 0) return SVG_ERR_BAD_ARG;
    if (R > kProfMaxRows) return SVG_ERR_UNSUPPORTED;
    for (int i = 0; i < 2; ++i)
        if (prof->variant[i].coord == 1 && prof->frame_size < kBN) return SVG_ERR_UNSUPPORTED;  // one-wrap stepping
    if (workspace_bytes < svg_sample_mse_workspace_bytes(BH, R, D, S)) return SVG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SVG_DTYPE_BF16) {
        if (D == 128) return run_profile<__bf16, 128>(q, k, v, rows, R, BH, S, sm_scale, prof, out_mse, workspace, skip_flag, st);
        if (D == 64) return run_profile<__bf16, 64>(q, k, v, rows, R, BH, S, sm_scale, prof, out_mse, workspace, skip_flag, st);
    } else if (dtype == SVG_DTYPE_F16) {
        if (D == 128) return run_profile<_Float16, 128>(q, k, v, rows, R, BH, S, sm_scale, prof, out_mse, workspace, skip_flag, st);
        if (D == 64) return run_profile<_Float16, 64>(q, k, v, rows, R, BH, S, sm_scale, prof, out_mse, workspace, skip_flag, st);
    }
    return SVG_ERR_UNSUPPORTED;
}

extern "C" int svg_sample_mse(const void* q, const void* k, const void* v, const int64_t* rows, int32_t R, int32_t BH,
                              int32_t S, int32_t D, int32_t dtype, float sm_scale, const svg_profile_desc_t* prof,
                              float* out_mse, void* workspace, size_t workspace_bytes, void* stream) {
    return svg_sample_mse_flagged(q, k, v, rows, R, BH, S, D, dtype, sm_scale, prof, out_mse, workspace, workspace_bytes, nullptr,
                                  stream);
}
