// Online profiler of SVG1 (`sample_mse`): for R sampled query rows per head compute the dense attention output
// ("golden") and the outputs under the two candidate masks, return the per-head mean squared errors.
// ref: svg/models/hyvideo/attention.py:376-399 (wan/attention.py:211-234, cog/attention.py:120-145) and the
// profiling masks of get_attention_mask (hyvideo/utils.py:47-93, wan/utils.py:63-110, cog/utils.py:61-88).
//
// The reference materialises two [10000, S] fp32 masks (4.8 GB each) and runs three full softmaxes in torch.
// Here the masks are analytic predicates (ProfilePolicy below: tile classification + element predicate) and K / V are streamed once per
// call: grid = (BH, kv_chunks), every workgroup emits un-normalised fp32 partials (O, m, l) per sampled row, output and chunk (split-KV),
// profile_combine_kernel merges the chunks, normalises and reduces the squared errors, profile_finalize_kernel writes mse[2][BH].
// Two forms of the attention kernel:
//   profile16_kernel     (bf16, round 5)  ONE score tile for the golden rows and both masks, 4 waves x 16 rows on 16x16x32 MFMAs, two
//                        workgroups per CU — see the note above the kernel;
//   profile_attn_kernel  (fp16; -DSVG_PROF_FIRST_FORM: both types)  the three outputs as three wave roles of one workgroup on attn_core.h's
//                        lock-step body: waves 0-1 the golden rows, 2-3 the rows under mask 1, 4-5 under mask 0 on the SAME staged K / V
//                        tiles, each role its own scores.  (Before that: three workgroups, grid.z = 3, K / V staged three times, 1.26 ms per
//                        call at HunyuanVideo 720p; the role form 0.78 -> 0.60 ms; the shipped one 0.32.)
#include <type_traits>
#include "attn_core.h"
#include "attn_m16.h"

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
        AttnLayout lay;     // strides of q, k, v (svg_sample_mse_strided; row strides other than D: the second form only)
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
    static __device__ __forceinline__ const T* q_base(const Params& p, const Ctx& c) { return p.q + layout_head_off(p.lay.q_bs, p.lay.q_hs, p.lay.hpb_q, c.head); }
    static __device__ __forceinline__ const T* k_base(const Params& p, const Ctx& c) { return p.k + layout_head_off(p.lay.k_bs, p.lay.k_hs, p.lay.hpb_kv, c.head); }
    static __device__ __forceinline__ const T* v_base(const Params& p, const Ctx& c) { return p.v + layout_head_off(p.lay.v_bs, p.lay.v_hs, p.lay.hpb_kv, c.head); }

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
static __device__ unsigned long long g_prof_phase[2048 * 8];   // second form, wave 0: ticks in { DMA wait + barrier, scores + softmax, P V, tiles }
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Second form (round 5): ONE score tile for all three outputs.  The golden rows and the rows under the two masks are the SAME 64 sampled
// query rows against the SAME keys: S = Q K^T is computed once per tile, exponentiated once against the golden rows' running maximum (a masked
// softmax may use any reference >= its own maximum: numerator and denominator scale together), and the two masks only select which
// probabilities enter their P V and their row sum.  Workgroup = 4 waves x 16 sampled rows on v_mfma_f32_16x16x32 (the fragment layouts, LDS
// images, swizzle and LDS-DMA staging of attn_m16.h); per tile and wave 16 MFMAs for S^T and 16 + 2 per output that sees the tile (a V^T fragment
// read feeds up to three MFMAs; the row sums ride the matrix pipe).  ~225 registers, 72 KiB of LDS: two workgroups per CU.
//   * the sampled rows are ranked by their coordinate under the second mask, so a wave's 16 rows sit close in it and its tiles outside their
//     common band are skipped (in sampling order 16 rows' bands cover nearly every tile);
//   * a tile's class under each mask and its first coordinate come from a per-wave LDS table (128 tiles per fill, a tile per lane): as scalar
//     code per tile and wave the classification was ~250 of a tile's ~950 instructions;
//   * the class is wave-uniform: one branch per mask and tile, the element predicate inside it branch-free (one interval test without sink
//     columns); the running maximum, the exponential of its step and the rescaling of the 108 accumulators only in the tiles where it moves.
// HunyuanVideo 720p: 0.345 ms against 0.60 ms for the first form (three roles x two waves of the lock-step body, each role its own S, ONE
// workgroup per CU in two rounds); with the golden output alone 0.31 ms, which is the time of its loads alone: removing the rounding, the
// exponentials, the QK or the P V MFMAs changes nothing, removing the loads gives 0.18 ms, the loads without any arithmetic take 0.309 ms whether
// an instruction moves 16 x 64 B or 1 KiB contiguous, and two tiles of K in flight instead of one gain 1.4 % (profiles/r05zj_profiler_bound_ablations.txt).
// Partials (O, m, l per sampled row, output and KV chunk) in the layout of the first form: profile_combine_kernel is unchanged.
// -DSVG_P16_ABL=<bits> (timing only, results wrong by construction): 1 no mask predicates, 2 no P V of the masked outputs, 4 no wait on the staged
// tile, 8 no loads, 16 no arithmetic in the tile loop.
#ifndef SVG_P16_ABL
#define SVG_P16_ABL 0
#endif
using p16_i32x4 = int __attribute__((ext_vector_type(4)));
constexpr int kP16Win = 128;                                          // tiles per fill of a wave's class table
constexpr int p16_lds_bytes(int D) { return 2 * 2 * kBN * D * 2 + 4 * kP16Win * 16; }   // two stages of a K and a V image + the four waves' tables
// torch: (q @ k^T) rounds to the input dtype, "/ sqrt(D)" rounds again (ProfilePolicy::score_fixup) — for a pair of scores: bf16 converts two
// values per instruction and widens back with a shift / a mask
template <typename T>
__device__ __forceinline__ void p16_fixup_pair(float& a, float& b, float fs) {
    if constexpr (std::is_same_v<T, __bf16>) {
        using v2 = __bf16 __attribute__((ext_vector_type(2)));
        auto rnd = [](float& x, float& y) {
            const v2 r = {(__bf16)x, (__bf16)y};
            const uint32_t w = __builtin_bit_cast(uint32_t, r);
            x = __builtin_bit_cast(float, w << 16);
            y = __builtin_bit_cast(float, w & 0xffff0000u);
        };
        rnd(a, b);
        a *= fs, b *= fs;
        rnd(a, b);
    } else {
        a = Elt<T>::to_float(Elt<T>::from_float(Elt<T>::to_float(Elt<T>::from_float(a)) * fs));
        b = Elt<T>::to_float(Elt<T>::from_float(Elt<T>::to_float(Elt<T>::from_float(b)) * fs));
    }
}
template <typename T, int D>
__global__ __launch_bounds__(256, 2) void profile16_kernel(typename ProfilePolicy<T, D>::Params prm) {
    using Pol = ProfilePolicy<T, D>;
    using E = Elt<T>;
    using M = Mfma16<T>;
    using V8 = typename E::v8;
    constexpr int KS = D / 32, NDB = D / 16;
    constexpr int kImg = kBN * D * 2, kStage = 2 * kImg;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (prm.skip && prm.skip[0] != 0) return;
#ifdef SVG_PROF_TRACE
    const unsigned long long tr_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long tr_wait = 0, tr_sm = 0, tr_pv = 0, tr_qk = 0, tr_fix = 0, tr_mark = tr_t0;
#define P16_MARK(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tr_mark; tr_mark = now_; }
#else
#define P16_MARK(acc)
#endif
    const int head = blockIdx.x, chunk = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int g4 = lane >> 4, n16 = lane & 15;
    const int ntiles = (prm.S + kBN - 1) / kBN;
    const int t0 = chunk * prm.tiles_per_chunk;
    const int nT = max(0, min(prm.tiles_per_chunk, ntiles - t0));
    const T* __restrict__ qb = prm.q + layout_head_off(prm.lay.q_bs, prm.lay.q_hs, prm.lay.hpb_q, head);
    const T* __restrict__ kb_ = prm.k + layout_head_off(prm.lay.k_bs, prm.lay.k_hs, prm.lay.hpb_kv, head);
    const T* __restrict__ vb = prm.v + layout_head_off(prm.lay.v_bs, prm.lay.v_hs, prm.lay.hpb_kv, head);

    // The sampled rows in the order of their coordinate under the second mask (token-major in every model of the reference).  The squared
    // errors are summed over the rows: their order is free.  Rank by counting, through LDS.
    {
        const int mine = lane < prm.R ? (int)prm.rows[lane] : 0;
        const int key = lane < prm.R ? Pol::coord(prm, prm.var[1], mine) : 0x7fffffff;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int kj = __builtin_amdgcn_readlane(key, j);
            rank += (kj < key || (kj == key && j < lane)) ? 1 : 0;
        }
        if (wave == 0) ((int*)smem)[rank] = mine;
        __syncthreads();
    }
    // this lane's sampled row (rows >= R do not exist: they read row 0 and are never stored)
    const int r = wave * 16 + n16;
    const bool have = r < prm.R;
    const int qrow = have ? ((const int*)smem)[r] : 0;
    __syncthreads();    // (the staging below reuses the bytes)
    V8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const V8*)(qb + (size_t)qrow * prm.lay.q_rs + ks * 32 + g4 * 8);

    // the two masks' per-lane / per-wave state, as ProfilePolicy::init computes it (output 1 <- var[0], output 2 <- var[1])
    typename Pol::Ctx mc[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        typename Pol::Ctx& c = mc[v];
        c.head = head, c.variant = v + 1, c.chunk = chunk, c.t0 = t0, c.nT = nT;
        c.pv = prm.var[v];
        c.qx = Pol::coord(prm, c.pv, qrow) - c.pv.origin;
        c.qtext = have && ((unsigned)(qrow - c.pv.text_lo) < (unsigned)(c.pv.text_hi - c.pv.text_lo));
        const int blk = c.qx >> 7;
        int lo = have ? blk : (1 << 28), hi = have ? blk : -(1 << 28), anyt = c.qtext;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o));
            hi = max(hi, __shfl_xor(hi, o));
            anyt |= __shfl_xor(anyt, o);
        }
        c.xlo_blk = __builtin_amdgcn_readfirstlane(lo), c.xhi_blk = __builtin_amdgcn_readfirstlane(hi);
        c.any_text = __builtin_amdgcn_readfirstlane(anyt);
        c.tk0 = 0, c.f0 = 0, c.p0 = 0;
        c.ystride = c.pv.coord == 1 ? prm.F : 1;
        c.ybase = 0, c.g4F = 4 * g4 * c.ystride;     // (this layout: a lane's keys of a 16-key block are 4 g4 + [0, 4))
        const int x = c.qx, span = c.pv.span;
        const bool xdom = have && ((unsigned)x < (unsigned)span);
        const int ylo = ((x >> 7) - c.pv.band_blocks + 1) * 128, yhi = ((x >> 7) + c.pv.band_blocks) * 128;
        const int a0 = max(ylo, 0), a1 = min(yhi, span);
        c.fa0 = a0, c.falen = xdom ? (unsigned)max(a1 - a0, 0) : 0u;
        c.fblen = (xdom && c.pv.sink_cols > 0) ? (unsigned)min(c.pv.sink_cols, span) : 0u;
        if (c.qtext) c.fa0 = -(1 << 30), c.falen = 0xFFFFFFFFu;
    }

    // ---- LDS-DMA staging: wave w moves key group w (16 keys) of a tile, every 64-byte d-block, K and V ----
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned vsw = (unsigned)((lane >> 4) & 1) << 1;
    const unsigned col_v = ((lane & 3) ^ vsw) * 16u;
    const int krow = 16 * wave + (lane >> 2);
    const unsigned k_rsb = (unsigned)prm.lay.k_rs * 2u, v_rsb = (unsigned)prm.lay.v_rs * 2u;
    auto dma_tile = [&](int t) {
        const int l = (t0 + t) * kBN + krow;
        const unsigned nphys = (unsigned)(l < prm.S ? l : 0);    // rows behind the sequence: masked below (the last tile is never FULL)
#pragma unroll
        for (int j = 0; j < D / 32; ++j) {
            const unsigned st = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((t & 1) * kStage) + (unsigned)(j * (kBN * 64) + wave * 1024));
            // row * (row stride in bytes, a kernel argument: svg_attn_layout_t) + the d-block + the lane's 16-byte column
            const unsigned ko = __umul24(nphys, k_rsb) + ((unsigned)(j * 64) | col_v);
            const unsigned vo = __umul24(nphys, v_rsb) + ((unsigned)(j * 64) | col_v);
            asm volatile("s_mov_b32 m0, %0\n\t"
                         "s_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %3\n\t"
                         "s_add_u32 m0, m0, %5\n\t"
                         "s_nop 0\n\t"
                         "global_load_lds_dwordx4 %2, %4"
                         :
                         : "s"(st), "v"(ko), "v"(vo), "s"(kb_), "s"(vb), "n"(kImg)
                         : "memory", "scc");
        }
    };
    const int k_lane = n16 * 64 + ((g4 ^ (((n16 >> 2) & 1) << 1)) << 4);
    const int v_lane0 = kImg + (4 * g4 + (n16 >> 2)) * 64 + (((g4 & 1) * 16) + 4 * (n16 & 3)) * 2;
    const int v_lane1 = v_lane0 ^ 32;
    auto kfrag = [&](const char* st, int kblk, int ks) -> V8 { return *(const V8*)(st + k_lane + ks * (kBN * 64) + kblk * 1024); };
    auto vfrag = [&](const char* st, int kc, int db) -> V8 {
        const char* vbase = st + ((db & 1) ? v_lane1 : v_lane0) + (db >> 1) * (kBN * 64) + (32 * kc) * 64;
        const i16x4 lo = lds_read_tr16(vbase);
        const i16x4 hi = lds_read_tr16(vbase + 16 * 64);
        const i16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(V8, both);
    };

    // ---- the wave's class table: { class under mask 0, first coordinate, class under mask 1, first coordinate } per tile of the window ----
    int* const tab = (int*)(smem + 2 * kStage) + wave * (kP16Win * 4);
    auto fill_classes = [&](int win0) {
#pragma unroll
        for (int i = 0; i < kP16Win / 64; ++i) {
            const int tl = win0 + lane + 64 * i;
            p16_i32x4 e = {TILE_SKIP, 0, TILE_SKIP, 0};
            if (tl < nT) {
                const int kk = (t0 + tl) * kBN;
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    typename Pol::Ctx x = mc[v];
                    int c = Pol::classify(prm, x, kk, wave * 16);
                    // a fast tile (64 keys at coordinates ybase + [0, 63] * ystride, none of them text) farther than the band from every row of the wave
                    if (c == TILE_PARTIAL_FAST && !x.any_text && x.ybase >= x.pv.sink_cols) {
                        const int b0 = x.ybase >> 7, b1 = (x.ybase + 63 * x.ystride) >> 7;
                        if (b0 - x.xhi_blk >= x.pv.band_blocks || x.xlo_blk - b1 >= x.pv.band_blocks) c = TILE_SKIP;
                    }
                    e[2 * v] = c, e[2 * v + 1] = x.ybase;
                }
            }
            *(p16_i32x4*)(tab + (lane + 64 * i) * 4) = e;
        }
    };

    // row sums on the matrix pipe (an all-ones A fragment: every row of D is the column sum of P^T — of the P the numerator sees, rounded to T), as
    // attn_m16.h ships it: 2 MFMAs per output and tile instead of 16 additions and a cross-lane reduction at the end
    float m_run = -INFINITY;
    f32x4 acc_l[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    V8 ones8;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones8[i] = E::from_float(1.f);
    f32x4 acc[3][NDB];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int db = 0; db < NDB; ++db) acc[o][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float c_log2 = prm.scale_log2;

    if (nT > 0) dma_tile(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (the q fragments are consumed here as far as the compiler's counter bookkeeping goes: left pending into the loop, their first use there gets
    //  an s_waitcnt vmcnt(0) — which waits for the tile prefetch just issued)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    __syncthreads();
    for (int t = 0; t < nT; ++t) {
        P16_MARK(tr_wait)
        if (!(SVG_P16_ABL & 8) && t + 1 < nT) dma_tile(t + 1);      // into the stage every wave finished reading one barrier ago
        const char* st = smem + (t & 1) * kStage;
        const int k0 = (t0 + t) * kBN;
        if ((t & (kP16Win - 1)) == 0) fill_classes(t);
        int cls[3];     // (a wave without sampled rows — R < 64 — walks the tiles like the others and stores nothing)
        cls[0] = (k0 + kBN <= prm.S) ? TILE_FULL : TILE_PARTIAL;
        {
            const p16_i32x4 e = *(const p16_i32x4*)(tab + (t & (kP16Win - 1)) * 4);
            cls[1] = __builtin_amdgcn_readfirstlane(e[0]), mc[0].ybase = __builtin_amdgcn_readfirstlane(e[1]);
            cls[2] = __builtin_amdgcn_readfirstlane(e[2]), mc[1].ybase = __builtin_amdgcn_readfirstlane(e[3]);
        }
        if (!(SVG_P16_ABL & 16)) {
            // ---- S^T = K Q^T (lane: query row n16, keys 16 kb + 4 g4 + [0, 4)) ----
            f32x4 sc[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) sc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            {   // K fragments one 32-wide d-block ahead (all sixteen at once is what the scheduler does on its own: 64 registers)
                V8 kf[2][4];
#pragma unroll
                for (int b = 0; b < 4; ++b) kf[0][b] = kfrag(st, b, 0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 1 < KS) {
#pragma unroll
                        for (int b = 0; b < 4; ++b) kf[(ks + 1) & 1][b] = kfrag(st, b, ks + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) sc[b] = M::mfma(kf[ks & 1][b], qf[ks], sc[b]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            P16_MARK(tr_qk)
            // torch's rounding of the scores (emulate), the keys behind the sequence, the lane's maximum: uniform branches, packed pairs
            if (prm.emulate) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    float s0 = sc[b][0], s1 = sc[b][1], s2 = sc[b][2], s3 = sc[b][3];
                    p16_fixup_pair<T>(s0, s1, prm.fix_scale);
                    p16_fixup_pair<T>(s2, s3, prm.fix_scale);
                    sc[b] = f32x4{s0, s1, s2, s3};
                }
            }
            if (cls[0] != TILE_FULL) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sc[b][j] = (k0 + 16 * b + 4 * g4 + j < prm.S) ? sc[b][j] : -INFINITY;
            }
            float mx = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
#pragma unroll
            for (int b = 1; b < 4; ++b) mx = fmaxf(fmaxf(mx, sc[b][0]), fmaxf(fmaxf(sc[b][1], sc[b][2]), sc[b][3]));
            // the running maximum moves in a few tiles per row (a record among ~S / 64 tile maxima): only then the cross-lane reduction, the
            // exponential of the step and the rescaling of the accumulators — the same numbers as updating every tile (a step of 0 is exact)
            if (__any(mx * c_log2 > m_run)) {
                asm volatile("" ::: "memory");   // (keeps the branch: x * 1 is exact, so the optimiser would rather multiply every tile)
                const float m_new = fmaxf(m_run, quad_group_max(mx) * c_log2);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // (m_new is finite: a tile holds at least one key of the sequence)
                m_run = m_new;
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    acc_l[o][0] *= alpha;
#pragma unroll
                    for (int db = 0; db < NDB; ++db)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[o][db][j] *= alpha;
                }
            }
            // The masked outputs share the golden rows' reference: a mask whose best key scores far below the row's overall maximum sees small
            // probabilities.  The reference sits kBias binades under the maximum — bf16: 60 (p <= 2^60; sums of 2^17 keys times |v| stay below
            // 2^90), a masked row only loses keys more than 186 binades = 129 in logit under the overall maximum; fp16: 15 (p <= 2^15 < 65504),
            // 39 binades = 27 in logit.  Below that its sum is 0 and the MSE NaN like a row without visible keys.  A power of two: every
            // number of the kernel scales exactly, the results are bit for bit those without it.
            constexpr float kBias = std::is_same_v<T, __bf16> ? 60.f : 15.f;
            const float m_use = m_run - kBias;
            P16_MARK(tr_fix)
            // ---- probabilities (once), the masks' selections ----
            V8 pf[3][2];
            float pr[16];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pr[4 * b + j] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[b][j], c_log2, -m_use));
                    pf[0][b >> 1][4 * (b & 1) + j] = E::from_float(pr[4 * b + j]);
                }
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                auto select = [&](auto pred) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float pm = pred(16 * b + j) ? pr[4 * b + j] : 0.f;    // key offset inside the tile without the lane's 4 g4
                            pf[v + 1][b >> 1][4 * (b & 1) + j] = E::from_float(pm);
                        }
                };
                if (SVG_P16_ABL & 1) {
                    if (cls[v + 1] != TILE_SKIP) select([&](int) { return true; });
                } else if (cls[v + 1] == TILE_PARTIAL_FAST) {
                    // allowed_fast with the lane's part hoisted; a mask without sink columns (every model but the 5-D Cog masks) has one interval
                    const int yb = mc[v].ybase + mc[v].g4F - mc[v].fa0, ys = mc[v].ystride;
                    const unsigned fal = mc[v].falen, fbl = mc[v].fblen;
                    if (mc[v].pv.sink_cols > 0) {
                        const int fa0 = mc[v].fa0;
                        select([&](int off) { return ((unsigned)(yb + off * ys) < fal) | ((unsigned)(yb + fa0 + off * ys) < fbl); });
                    } else {
                        select([&](int off) { return (unsigned)(yb + off * ys) < fal; });
                    }
                } else if (cls[v + 1] != TILE_SKIP) {
                    (void)Pol::classify(prm, mc[v], k0, wave * 16);    // (the general predicate's per-tile decomposition: tk0, f0, p0)
                    int kb = k0 + 4 * g4;
                    asm volatile("" : "+v"(kb));   // (opaque: or the optimiser computes the two masks' common sub-tests for every tile, ahead of the branch)
                    select([&](int off) { return Pol::allowed(prm, mc[v], 0, kb + off); });
                }
            }
            P16_MARK(tr_sm)
            // ---- O^T += V^T P^T for every output that sees the tile ----
            {   // V^T fragments in groups of four 16-wide d-blocks, two groups in flight; a masked output's MFMAs of a group behind ONE uniform branch
                // (if-then around in-place accumulators: no copies.  Whole-pass forms per combination cost 40 register moves per tile and spills.)
                constexpr int G = 4, NG = 2 * NDB / G;
                const bool s1 = !(SVG_P16_ABL & 2) && cls[1] != TILE_SKIP, s2 = !(SVG_P16_ABL & 2) && cls[2] != TILE_SKIP;
                V8 vf[2][G];
#pragma unroll
                for (int i = 0; i < G; ++i) vf[0][i] = vfrag(st, 0, i);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + 1 < NG) {
#pragma unroll
                        for (int i = 0; i < G; ++i) vf[(g + 1) & 1][i] = vfrag(st, ((g + 1) * G + i) / NDB, ((g + 1) * G + i) % NDB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < G; ++i) acc[0][(g * G + i) % NDB] = M::mfma(vf[g & 1][i], pf[0][(g * G + i) / NDB], acc[0][(g * G + i) % NDB]);
                    if ((g * G) % NDB == 0) acc_l[0] = M::mfma(ones8, pf[0][(g * G) / NDB], acc_l[0]);
                    if (s1) {
                        if ((g * G) % NDB == 0) acc_l[1] = M::mfma(ones8, pf[1][(g * G) / NDB], acc_l[1]);
#pragma unroll
                        for (int i = 0; i < G; ++i) acc[1][(g * G + i) % NDB] = M::mfma(vf[g & 1][i], pf[1][(g * G + i) / NDB], acc[1][(g * G + i) % NDB]);
                    }
                    if (s2) {
                        if ((g * G) % NDB == 0) acc_l[2] = M::mfma(ones8, pf[2][(g * G) / NDB], acc_l[2]);
#pragma unroll
                        for (int i = 0; i < G; ++i) acc[2][(g * G + i) % NDB] = M::mfma(vf[g & 1][i], pf[2][(g * G + i) / NDB], acc[2][(g * G + i) % NDB]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            P16_MARK(tr_pv)
        }
        if (!(SVG_P16_ABL & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#ifdef SVG_PROF_TRACE
    if (threadIdx.x == 0) {
        const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
        if (wg < 2048) {
            g_prof_trace[wg * 4 + 0] = tr_t0;
            g_prof_trace[wg * 4 + 1] = __builtin_amdgcn_s_memtime();
            g_prof_trace[wg * 4 + 2] = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
            g_prof_trace[wg * 4 + 3] = ((unsigned long long)blockIdx.y << 16) | blockIdx.x;
            g_prof_phase[wg * 8 + 0] = tr_wait, g_prof_phase[wg * 8 + 1] = tr_sm, g_prof_phase[wg * 8 + 2] = tr_pv, g_prof_phase[wg * 8 + 3] = nT;
            g_prof_phase[wg * 8 + 4] = tr_qk, g_prof_phase[wg * 8 + 5] = tr_fix, g_prof_phase[wg * 8 + 6] = 0, g_prof_phase[wg * 8 + 7] = 0;
        }
    }
#endif
#undef P16_MARK
    // ---- partials: output 0 golden, 1 under var[0], 2 under var[1] (store_partial's layout; the row sum is the same in the four lanes of a row) ----
    if (have) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float* dst = prm.part + ((((size_t)o * prm.BH + head) * prm.n_chunks + chunk) * kProfMaxRows + r) * (D + 4);
#pragma unroll
            for (int db = 0; db < NDB; ++db) *(f32x4*)(dst + 16 * db + 4 * g4) = acc[o][db];
            if (g4 == 0) {
                dst[D] = m_run;
                dst[D + 1] = acc_l[o][0];
            }
        }
    }
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
    const size_t cs = (size_t)kProfMaxRows * DS;
    // the chunks' weights 2^(m_c - M) and the denominators once per (output, row) — a chunk per lane — instead of per element (every one of a row's
    // D threads re-read its chunks' m twice and l once: 0.033 ms of a 0.355 ms call); the sums in the same order as before: the same bits
    constexpr int kRowsPer = (kProfMaxRows + kProfRowGroups - 1) / kProfRowGroups;
    __shared__ float wsh[3][kRowsPer][64], lws[3][kRowsPer][64], lsh[3][kRowsPer];
    const int nrows = max(r1 - r0, 0);
    for (int pi = tid >> 6; pi < 3 * nrows; pi += 4) {
        const int v = pi / nrows, rr = pi - v * nrows, c = tid & 63;
        const float* base = part + ((((size_t)v * BH + h) * n_chunks) * kProfMaxRows + r0 + rr) * DS;
        const float mc = c < n_chunks ? base[c * cs + D] : -INFINITY;
        const float M = wave_max(mc);
        const float w = (mc == -INFINITY) ? 0.f : exp2f(mc - M);
        wsh[v][rr][c] = w;
        lws[v][rr][c] = c < n_chunks ? base[c * cs + D + 1] * w : 0.f;
    }
    __syncthreads();
    if (tid < 3 * nrows) {
        const int v = tid / nrows, rr = tid - v * nrows;
        float L = 0.f;
        for (int c = 0; c < n_chunks; ++c) L += lws[v][rr][c];
        lsh[v][rr] = L;
    }
    __syncthreads();
    for (int e = r0 * D + tid; e < r1 * D; e += 256) {
        const int row = e / D, d = e - row * D;
        float o[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const float* base = part + ((((size_t)v * BH + h) * n_chunks) * kProfMaxRows + row) * DS;
            const float L = lsh[v][row - r0];
            float acc = 0.f;
            for (int c = 0; c < n_chunks; ++c) acc += base[c * cs + d] * wsh[v][row - r0][c];
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
    if (const char* e = getenv("SVG_PROF_CHUNKS")) n = std::max(1, std::min(std::min(atoi(e), ntiles), 64));   // (A/B builds only)
#endif
    return n;
}

template <typename T, int D>
static int run_profile(const void* q, const void* k, const void* v, const int64_t* rows, int R, int BH, int S, float sm_scale,
                       const svg_profile_desc_t* pd, float* out_mse, void* ws, const int32_t* skip, const AttnLayout* lay, hipStream_t st) {
    using Pol = ProfilePolicy<T, D>;
    typename Pol::Params p;
    p.q = (const T*)q, p.k = (const T*)k, p.v = (const T*)v;
    p.lay = lay ? *lay : contiguous_layout(BH, BH, S, S, D);
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
    // bf16: the second form.  fp16: the first form — the second one exponentiates the masked rows against the golden rows' maximum, and an fp16
    // probability only reaches 39 binades (27 in logit) below it: a sampled row whose best in-mask key sits further down would lose its row sum
    // (NaN for the whole head, where the reference is finite).  bf16 probabilities reach 186 binades (129 in logit); the first form keeps a
    // running maximum per output.  -DSVG_PROF_FIRST_FORM: the first form for both (A/B builds).
#ifdef SVG_PROF_FIRST_FORM
    constexpr bool kSecondForm = false;
#else
    constexpr bool kSecondForm = std::is_same_v<T, __bf16>;
#endif
    if (!kSecondForm && !p.lay.rows_contiguous(D)) return SVG_ERR_UNSUPPORTED;   // row strides: the second form only (svg_attn_layout_t)
    if constexpr (kSecondForm) {   // one score tile for the three outputs, two workgroups per CU
        constexpr int lds16 = p16_lds_bytes(D);
        auto kern16 = profile16_kernel<T, D>;
        hipError_t e16 = hipFuncSetAttribute((const void*)kern16, hipFuncAttributeMaxDynamicSharedMemorySize, lds16);
        if (e16 != hipSuccess) {
            g_last_hip_error = (int)e16;
            return SVG_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern16, dim3(BH, p.n_chunks), dim3(256), lds16, st, p);
    } else {   // three roles x two waves of the lock-step body, each role its own scores and running maximum
        const int lds = attn_lds_bytes<D, kProfNW>();
        auto kern = profile_attn_kernel<T, D>;
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            g_last_hip_error = (int)e;
            return SVG_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, dim3(BH, p.n_chunks), dim3(kProfNW * 64), lds, st, p);
    }
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
extern "C" int svg_debug_prof_phase(uint64_t* out, int n_workgroups) {
    if (!out || n_workgroups <= 0 || n_workgroups > 2048) return SVG_ERR_BAD_ARG;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof_phase), (size_t)n_workgroups * 8 * sizeof(uint64_t)) == hipSuccess ? SVG_OK : SVG_ERR_LAUNCH;
}
#endif

extern "C" size_t svg_sample_mse_workspace_bytes(int32_t BH, int32_t R, int32_t D, int32_t S) {
    if (BH <= 0 || R <= 0 || D <= 0 || S <= 0) return 0;
    return ((size_t)3 * BH * prof_chunks(BH, S) * kProfMaxRows * (D + 4) + (size_t)BH * kProfRowGroups * 4) * sizeof(float);
}

// svg_sample_mse[_flagged] (abi_layout == nullptr: contiguous [BH, S, D] tensors) and svg_sample_mse_strided
static int sample_mse_entry(const void* q, const void* k, const void* v, const int64_t* rows, int32_t R, int32_t BH, int32_t S, int32_t D,
                            int32_t dtype, float sm_scale, const svg_profile_desc_t* prof, float* out_mse, void* workspace,
                            size_t workspace_bytes, const int32_t* skip_flag, const svg_attn_layout_t* abi_layout, void* stream) {
    if (!q || !k || !v || !rows || !prof || !out_mse || !workspace) return SVG_ERR_BAD_ARG;
    if (R <= 0 || BH <= 0 || S <= 0) return SVG_ERR_BAD_ARG;
    if (R > kProfMaxRows) return SVG_ERR_UNSUPPORTED;
    for (int i = 0; i < 2; ++i)
        if (prof->variant[i].coord == 1 && prof->frame_size < kBN) return SVG_ERR_UNSUPPORTED;  // one-wrap stepping
    if (workspace_bytes < svg_sample_mse_workspace_bytes(BH, R, D, S)) return SVG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    AttnLayout lay_storage;
    const AttnLayout* lay = nullptr;
    if (abi_layout) {   // (o is not a tensor of this call: q stands in for the alignment test)
        svg_attn_layout_t a = *abi_layout;
        a.o = a.q;
        if (const int rc = layout_from_abi(&a, BH, BH, S, S, D, q, k, v, q, lay_storage); rc != SVG_OK) return rc;
        lay = &lay_storage;
    }
    if (dtype == SVG_DTYPE_BF16) {
        if (D == 128) return run_profile<__bf16, 128>(q, k, v, rows, R, BH, S, sm_scale, prof, out_mse, workspace, skip_flag, lay, st);
        if (D == 64) return run_profile<__bf16, 64>(q, k, v, rows, R, BH, S, sm_scale, prof, out_mse, workspace, skip_flag, lay, st);
    } else if (dtype == SVG_DTYPE_F16) {
        if (D == 128) return run_profile<_Float16, 128>(q, k, v, rows, R, BH, S, sm_scale, prof, out_mse, workspace, skip_flag, lay, st);
        if (D == 64) return run_profile<_Float16, 64>(q, k, v, rows, R, BH, S, sm_scale, prof, out_mse, workspace, skip_flag, lay, st);
    }
    return SVG_ERR_UNSUPPORTED;
}

extern "C" int svg_sample_mse_flagged(const void* q, const void* k, const void* v, const int64_t* rows, int32_t R, int32_t BH,
                                      int32_t S, int32_t D, int32_t dtype, float sm_scale, const svg_profile_desc_t* prof,
                                      float* out_mse, void* workspace, size_t workspace_bytes, const int32_t* skip_flag,
                                      void* stream) {
    return sample_mse_entry(q, k, v, rows, R, BH, S, D, dtype, sm_scale, prof, out_mse, workspace, workspace_bytes, skip_flag, nullptr, stream);
}

extern "C" int svg_sample_mse_strided(const void* q, const void* k, const void* v, const int64_t* rows, int32_t R, int32_t BH,
                                      int32_t S, int32_t D, int32_t dtype, float sm_scale, const svg_profile_desc_t* prof,
                                      float* out_mse, void* workspace, size_t workspace_bytes, const int32_t* skip_flag,
                                      const svg_attn_layout_t* layout, void* stream) {
    if (!layout) return SVG_ERR_BAD_ARG;
    return sample_mse_entry(q, k, v, rows, R, BH, S, D, dtype, sm_scale, prof, out_mse, workspace, workspace_bytes, skip_flag, layout, stream);
}

extern "C" int svg_sample_mse(const void* q, const void* k, const void* v, const int64_t* rows, int32_t R, int32_t BH,
                              int32_t S, int32_t D, int32_t dtype, float sm_scale, const svg_profile_desc_t* prof,
                              float* out_mse, void* workspace, size_t workspace_bytes, void* stream) {
    return svg_sample_mse_flagged(q, k, v, rows, R, BH, S, D, dtype, sm_scale, prof, out_mse, workspace, workspace_bytes, nullptr,
                                  stream);
}
