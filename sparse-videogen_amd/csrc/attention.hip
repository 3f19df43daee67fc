// SVG1 band (block-sparse) attention, dense attention and SVG2 variable-block attention for gfx950.
// The MFMA / LDS / online-softmax machinery is attn_core.h; this file supplies the two scheduling policies
// (which KV tiles a workgroup visits, where rows live in HBM, which elements are masked) and the C ABI.
#include <algorithm>

#include "attn_core.h"

namespace svg {

// =====================================================================================================
// Band policy: analytic mask family (see svg_band_mask_t in svg_attn.h)
// =====================================================================================================
template <typename T, int D, int NW, bool SKEW, int ABL = 0, int RB = 1, int SUBS = 1>
struct BandPolicy {
    static constexpr int kSubTiles = SUBS;   // 64-key tiles per LDS stage / barrier
    static constexpr int kPrefetch = (ABL == 12) ? 3 : (ABL == 13 ? 2 : 1);  // operand ring depth (k-steps / MFMA steps ahead)
    static constexpr bool kFixup = false;
    static constexpr bool kPartialOut = false;
    static constexpr bool kIntervalMask = true;   // row_intervals() describes the mask (two-phase body)
    static constexpr bool kFastPartial = false;
    static constexpr int kShadow128 = 1;   // two-phase body, D = 128: probability steps in the MFMA shadow (measured best)
    static constexpr int kAbl = ABL;  // > 0 only for the ablation variants (timing experiments)
    static constexpr bool kSetPrio = false;  // measured: s_setprio around the MFMA clusters costs 2 % here
    static constexpr bool kSkew = SKEW;
    static constexpr int kRowBlocks = RB;    // 32-row blocks per wave
    static constexpr int kWR = 32 * RB;      // rows per wave
    static constexpr int BM = NW * kWR;

    struct Params {
        const T* q;
        const T* k;
        const T* v;
        T* o;
        int S, BH, nqt;
        float scale_log2;
        int real_len, band, cf_lo, cf_hi, rf_lo, rf_hi;
        const int64_t* head_flag;
        int vid0, F, P, V;
        int q64, r64;          // 64 / F, 64 % F: tile-to-tile step of the (patch, frame) decomposition
        int q128, r128;        // the same for a 128-row step (two tiles per stage)
        int sp64, sp128;       // physical-row step of a token-major head: q + r * P (the patch index advances by q, the frame by r)
        int wrap_phys;         // 1 - F * P: correction when the frame index wraps
        int heavy_lo, n_heavy; // q-tiles [heavy_lo, heavy_lo + n_heavy) of every head see ALL keys (text rows): scheduled first
        // Row regions: q-tiles never straddle rowfull_lo / rowfull_hi / real_len, so every q-tile is homogeneous (band rows, full
        // rows or rows behind real_len).  Region r = rows [reg_lo[r], reg_hi[r]), its first q-tile is reg_t0[r].
        int reg_lo[4], reg_hi[4], reg_t0[4];
        // completion counters (or nullptr): every wave of a workgroup adds 1 to done[head] after its last store, so done[h] ==
        // 8 * (q-tiles of a head) means head h of O is complete and visible — a consumer on another stream (svg_wait_counters) can
        // start exchanging it while the launch is still working on the next heads (dispatch is head-major)
        int32_t* done;
        int done_nseg, done_tps;   // counters per head: segment of q-tile qt (row order) = min(qt / done_tps, done_nseg - 1)
    };
    struct Ctx {
        int head, qt, q0, q_end, nT, perm;
        int seg_lo[3], seg_n[3];
        int fk_lo, fk_hi;  // per WAVE: tiles with first key in [fk_lo, fk_hi] are FULL for this wave's 32 rows (fast path)
    };
    struct KvCursor {
        int physv, f, prev_k0;  // token-major head: frame f and physical row vid0 + f * P + pp of this thread's row in the previous tile,
    };                          // with (row - vid0) = pp * F + f

    static __device__ __forceinline__ int phys_row(const Params& p, const Ctx& c, int logical) {
        if (c.perm) {
            const unsigned i = (unsigned)(logical - p.vid0);
            if (i < (unsigned)p.V) {
                const unsigned pp = i / (unsigned)p.F;
                const unsigned f = i - pp * (unsigned)p.F;
                return p.vid0 + (int)(f * (unsigned)p.P + pp);
            }
        }
        return logical;
    }

    static __device__ __forceinline__ bool init(const Params& p, Ctx& c, char*) {
        // Work mapping.  The hardware hands dispatch id b to XCD b % 8 and each XCD schedules its share on its own 32 CUs, so the
        // load has to be balanced across XCDs by construction.
        //  * longest-processing-time-first: the few q-tiles that contain text rows visit every KV tile, all of them on the
        //    masked path (~11x the time of a band tile at Hunyuan 720p).  They take the first dispatch ids, un-swizzled:
        //    first so that they do not form the tail of the launch, round-robin so that every XCD gets its share (with the
        //    swizzle below applied to them they all landed on XCD 0, which then ran 16 % longer than the other seven).
        //  * the remaining q-tiles: every XCD gets 32 neighbouring q-tiles of the same 256-tile window, so their KV windows
        //    overlap in that XCD's L2 while the whole chip stays within one or two heads (KV working set fits the 256 MiB
        //    Infinity Cache).
        int qt;
        const int nh = p.BH * p.n_heavy;
        const int b = blockIdx.x;
        if (b >= p.nqt * p.BH) return false;   // (the device-switched launch is sized for the larger of its two masks)
        if (b < nh) {
            c.head = b / p.n_heavy;
            qt = p.heavy_lo + (b - c.head * p.n_heavy);
        } else {
            const int b2 = b - nh;
            const int full = ((p.nqt * p.BH - nh) / (kNumXCD * 32)) * (kNumXCD * 32);
            int w2 = b2;
            if (b2 < full) {
                const int xcd = b2 % kNumXCD, s = b2 / kNumXCD;
                w2 = (s / 32) * (kNumXCD * 32) + xcd * 32 + (s % 32);
            }
            const int nl = p.nqt - p.n_heavy;
            c.head = w2 / nl;
            const int r = w2 - c.head * nl;
            qt = r < p.heavy_lo ? r : r + p.n_heavy;
        }
        // (explicit selects: a run-time index into the kernel-argument arrays would go through scratch)
        const bool r1 = qt >= p.reg_t0[1], r2 = qt >= p.reg_t0[2], r3 = qt >= p.reg_t0[3];
        const int rlo = r3 ? p.reg_lo[3] : r2 ? p.reg_lo[2] : r1 ? p.reg_lo[1] : p.reg_lo[0];
        const int rhi = r3 ? p.reg_hi[3] : r2 ? p.reg_hi[2] : r1 ? p.reg_hi[1] : p.reg_hi[0];
        const int rt0 = r3 ? p.reg_t0[3] : r2 ? p.reg_t0[2] : r1 ? p.reg_t0[1] : 0;
        c.qt = qt;
        c.q0 = rlo + (qt - rt0) * BM;
        c.q_end = min(rhi, c.q0 + BM);
        c.perm = (p.head_flag != nullptr) && (p.head_flag[c.head] != 0);

        // ---- KV schedule: up to three key intervals -> sorted, merged, tile-aligned ranges ----
        // (explicit scalars, no runtime-indexed arrays: keeps everything in SGPRs, no scratch)
        constexpr int BIG = 1 << 28;
        int alo = BIG, ahi = BIG, blo = BIG, bhi = BIG, clo = BIG, chi = BIG;
        const int real = p.real_len;
        if (c.q0 < real) {
            const int qr1 = min(c.q_end, real);
            if (c.q0 < p.rf_hi && qr1 > p.rf_lo) {
                alo = 0, ahi = (real + kBN - 1) / kBN;
            } else {
                alo = max(0, c.q0 - p.band + 1) / kBN;
                ahi = (min(real, qr1 - 1 + p.band) + kBN - 1) / kBN;
                const int ch = min(p.cf_hi, real);
                if (ch > p.cf_lo) blo = p.cf_lo / kBN, bhi = (ch + kBN - 1) / kBN;
            }
        }
        if (c.q_end > real) clo = real / kBN, chi = (p.S + kBN - 1) / kBN;
#define SVG_CSWAP(x, xh, y, yh) if (y < x) { int t_ = x; x = y; y = t_; t_ = xh; xh = yh; yh = t_; }
        SVG_CSWAP(alo, ahi, blo, bhi)
        SVG_CSWAP(blo, bhi, clo, chi)
        SVG_CSWAP(alo, ahi, blo, bhi)
#undef SVG_CSWAP
        if (blo < BIG && blo <= ahi) {
            ahi = max(ahi, bhi);
            blo = clo, bhi = chi, clo = BIG, chi = BIG;
            if (blo < BIG && blo <= ahi) ahi = max(ahi, bhi), blo = BIG, bhi = BIG;
        } else if (clo < BIG && clo <= bhi) {
            bhi = max(bhi, chi), clo = BIG, chi = BIG;
        }
        c.seg_lo[0] = alo, c.seg_n[0] = ahi - alo;
        c.seg_lo[1] = blo, c.seg_n[1] = bhi - blo;
        c.seg_lo[2] = clo, c.seg_n[2] = chi - clo;
        c.nT = c.seg_n[0] + c.seg_n[1] + c.seg_n[2];
        // fast-path classification: inside the band, away from its edges, every (row, key) pair of a wave x tile
        // rectangle is allowed; those tiles (98-99 % of all) are recognised with two scalar compares.
        const int w0 = c.q0 + wave_id() * kWR, w1 = min(w0 + kWR, c.q_end);
        c.fk_lo = 1, c.fk_hi = 0;
        if (w0 < c.q_end && w1 <= real) {
            const bool full_rows = w0 >= p.rf_lo && w1 <= p.rf_hi;   // a wave of full (text) rows: every key tile below real_len
            c.fk_lo = full_rows ? 0 : max(w1 - p.band, 0);
            c.fk_hi = min(full_rows ? real : w0 + p.band - kBN, min(real, p.S) - kBN);
        }
        return true;
    }

    static __device__ __forceinline__ const T* q_base(const Params& p, const Ctx& c) { return p.q + (size_t)c.head * p.S * D; }
    static __device__ __forceinline__ const T* k_base(const Params& p, const Ctx& c) { return p.k + (size_t)c.head * p.S * D; }
    static __device__ __forceinline__ const T* v_base(const Params& p, const Ctx& c) { return p.v + (size_t)c.head * p.S * D; }
    static __device__ __forceinline__ T* o_base(const Params& p, const Ctx& c) { return p.o + (size_t)c.head * p.S * D; }

    static __device__ __forceinline__ int q_logical(const Ctx& c, int row) { return c.q0 + row; }
    static __device__ __forceinline__ bool wave_active(const Ctx& c, int wrow0) { return c.q0 + wrow0 < c.q_end; }
    static __device__ __forceinline__ int q_phys(const Params& p, const Ctx& c, int row) {
        const int l = c.q0 + row;
        return l < c.q_end ? phys_row(p, c, l) : -1;
    }
    static __device__ __forceinline__ int tile_key0(const Ctx& c, int t) {
        // selects, not branches: this runs once per tile on the scalar unit of every wave
        const int n01 = c.seg_n[0] + c.seg_n[1];
        const int a = c.seg_lo[0] + t, b = c.seg_lo[1] + (t - c.seg_n[0]), d = c.seg_lo[2] + (t - n01);
        const int bd = t < n01 ? b : d;
        return (t < c.seg_n[0] ? a : bd) * kBN;
    }
    static __device__ __forceinline__ void kv_cursor_init(const Params&, const Ctx&, KvCursor& cu, int) {
        cu.physv = 0, cu.f = 0, cu.prev_k0 = -(1 << 30);
    }
    static __device__ __forceinline__ int kv_phys(const Params& p, const Ctx& c, KvCursor& cu, int t, int row) {
        const int k0 = tile_key0(c, t);
        const int l = k0 + row;
        if (!c.perm) return l < p.S ? l : 0;
        // token-major head: physical row = vid0 + f * P + pp with (l - vid0) = pp * F + f.  Consecutive tiles advance by 64 rows,
        // so the frame and the physical row are stepped (5 VALU, no multiply) instead of divided (~30 VALU); segment jumps re-divide.
        // A tile that lies inside the video range (scalar test) needs neither the range selects nor the bounds test.
        int f, physv;
        // a cursor advances by one stage per call: 64 keys, or 128 with two tiles per stage (each chunk keeps its sub-tile)
        constexpr int kStep = kBN * SUBS;
        const int delta = __builtin_amdgcn_readfirstlane(k0 - cu.prev_k0);
        if (delta == kStep) {
            f = cu.f + (SUBS == 1 ? p.r64 : p.r128);
            physv = cu.physv + (SUBS == 1 ? p.sp64 : p.sp128);
            const bool wrap = f >= p.F;
            f = wrap ? f - p.F : f;
            physv = wrap ? physv + p.wrap_phys : physv;
        } else {
            const int i = l - p.vid0;
            const int a = i >= 0 ? i : -i - 1;              // floor division also for rows in front of the video
            const int qd = (int)((unsigned)a / (unsigned)p.F);
            const int pp = i >= 0 ? qd : -qd - 1;
            f = i - pp * p.F;
            physv = p.vid0 + f * p.P + pp;
        }
        cu.physv = physv, cu.f = f, cu.prev_k0 = k0;
        if (k0 >= p.vid0 && k0 + kStep <= p.vid0 + p.V) return physv;
        const bool in_video = (unsigned)(l - p.vid0) < (unsigned)p.V;
        const int phys = in_video ? physv : l;
        return l < p.S ? phys : 0;
    }

    static __device__ __forceinline__ int classify(const Params& p, const Ctx& c, int k0, int wrow0) {
        if (k0 >= c.fk_lo && k0 <= c.fk_hi) return TILE_FULL;
        const int w0 = c.q0 + wrow0;
        if (w0 >= c.q_end) return TILE_SKIP;
        const int w1 = min(w0 + kWR, c.q_end);      // rows [w0, w1)
        const int k1 = min(k0 + kBN, p.S);          // keys [k0, k1)
        const int real = p.real_len;
        // ---- every pair allowed? ----
        bool all = false;
        if (k0 + kBN <= p.S) {
            if (w1 <= real && k1 <= real) {
                const bool band_all = (k1 - 1 - w0 < p.band) && (w1 - 1 - k0 < p.band);
                const bool col_all = (k0 >= p.cf_lo && k1 <= p.cf_hi);
                const bool row_all = (w0 >= p.rf_lo && w1 <= p.rf_hi);
                all = band_all || col_all || row_all;
            } else if (w0 >= real && k0 >= real) {
                all = true;
            }
        }
        if (all) return TILE_FULL;
        // ---- any pair allowed? ----
        bool any = false;
        if (w0 < real && k0 < real) {
            const int w1r = min(w1, real), k1r = min(k1, real);
            const bool band_any = (k0 - (w1r - 1) < p.band) && (w0 - (k1r - 1) < p.band);
            const bool col_any = (k0 < p.cf_hi && k1r > p.cf_lo);
            const bool row_any = (w0 < p.rf_hi && w1r > p.rf_lo);
            any = band_any || col_any || row_any;
        }
        if (w1 > real && k1 > real) any = true;
        return any ? TILE_PARTIAL : TILE_SKIP;
    }
    static __device__ __forceinline__ bool allowed(const Params& p, const Ctx&, int q, int k) {
        const bool rq = q < p.real_len, rk = k < p.real_len;
        const bool in_band = (unsigned)(q - k + p.band - 1) < (unsigned)(2 * p.band - 1);
        const bool colf = (unsigned)(k - p.cf_lo) < (unsigned)(p.cf_hi - p.cf_lo);
        const bool rowf = (unsigned)(q - p.rf_lo) < (unsigned)(p.rf_hi - p.rf_lo);
        // bitwise on purpose: branch-free, one v_cndmask per element in the caller
        return ((rq & rk) & (in_band | colf | rowf)) | ((!rq & !rk) & (k < p.S));
    }
    // The same predicate as two key intervals of one query row, [a0, a0 + alen) u [b0, b0 + blen) (unsigned lengths, 0 = empty):
    // the two-phase body keeps them per lane across the tile loop, so a masked element costs 2 x (add, compare) + or + select.
    //   real row, not a full row:  band n [0, real)  u  full columns n [0, real)
    //   full row (text):           [0, real)
    //   row behind real_len:       [real_len, S)
    static __device__ __forceinline__ void row_intervals(const Params& p, const Ctx&, int q, int& a0, unsigned& alen, int& b0,
                                                         unsigned& blen) {
        const int real = p.real_len;
        const bool rq = q < real;
        const bool rowf = (unsigned)(q - p.rf_lo) < (unsigned)(p.rf_hi - p.rf_lo);
        const int band_lo = max(q - p.band + 1, 0), band_hi = min(q + p.band, real);
        const int lo = rq ? (rowf ? 0 : band_lo) : real;
        const int hi = rq ? (rowf ? real : band_hi) : p.S;
        a0 = lo, alen = (unsigned)max(hi - lo, 0);
        const int ch = min(p.cf_hi, real);
        b0 = p.cf_lo, blen = (rq && !rowf) ? (unsigned)max(ch - p.cf_lo, 0) : 0u;
    }
    static __device__ __forceinline__ float score_fixup(const Params&, float s) { return s; }
    static __device__ __forceinline__ void notify(const Params& p, const Ctx& c) {
        if (p.done) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __threadfence();
            if ((threadIdx.x & 63) == 0) atomicAdd(p.done + c.head * p.done_nseg + min(c.qt / p.done_tps, p.done_nseg - 1), 1);
        }
    }
};

template <typename T, int D, int NW, bool SKEW, int ABL = 0>
__global__ __launch_bounds__(NW * 64, 2) void band_attn_kernel(typename BandPolicy<T, D, NW, SKEW, ABL>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body<T, D, NW, BandPolicy<T, D, NW, SKEW, ABL>>(prm, smem, nullptr);
}

// 8 waves x 32 rows, two 64-key tiles per LDS stage: one barrier and one staging round per 128 keys
template <typename T, int D>
__global__ __launch_bounds__(512, 2) void band_attn_s2_kernel(typename BandPolicy<T, D, 8, false, 0, 1, 2>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body<T, D, 8, BandPolicy<T, D, 8, false, 0, 1, 2>>(prm, smem, nullptr);
}

// 4 waves x 64 query rows, ONE wave per SIMD with the whole 512-entry register file: every K / V fragment read from LDS
// feeds two MFMAs (LDS operand traffic per FLOP halves), and there is one workgroup of 256 threads per CU.
template <typename T, int D, int ABL = 0>
__global__ __launch_bounds__(256, 1) void band_attn_r64_kernel(typename BandPolicy<T, D, 4, false, ABL, 2>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body<T, D, 4, BandPolicy<T, D, 4, false, ABL, 2>>(prm, smem, nullptr);
}

template <typename T, int D>
__global__ __launch_bounds__(512, 2) void band_attn_pipe_kernel(typename BandPolicy<T, D, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pipe<T, D, 8, BandPolicy<T, D, 8, false>>(prm, smem, nullptr);
}

// ping-pong schedule (attn_body_pp): the two waves of a SIMD alternate matrix and memory / VALU clusters
template <typename T, int D>
__global__ __launch_bounds__(512, 2) void band_attn_pp_kernel(typename BandPolicy<T, D, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pp<T, D, BandPolicy<T, D, 8, false>>(prm, smem, nullptr);
}

template <typename T, int D, int ABL>
__global__ __launch_bounds__(512, 2) void band_attn_pp_trace_kernel(typename BandPolicy<T, D, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pp<T, D, BandPolicy<T, D, 8, false>, true, ABL>(prm, smem, nullptr);
}

// two-phase ping-pong schedule (attn_body_pp2)
template <typename T, int D>
__global__ __launch_bounds__(512, 2) void band_attn_pp2_kernel(typename BandPolicy<T, D, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pp2<T, D, BandPolicy<T, D, 8, false>>(prm, smem, nullptr);
}
// Device-side switch between two masks: `flag[0] != 0` selects prm_alt (the dense warm-up mask, no layout
// transformation) — the dense / sparse decision of attention_core_logic (hyvideo/attention.py:491-496) without reading the
// timestep back to the host.
template <typename T, int D>
__global__ __launch_bounds__(512, 2) void band_attn_pp2_switch_kernel(typename BandPolicy<T, D, 8, false>::Params prm,
                                                                      typename BandPolicy<T, D, 8, false>::Params prm_alt,
                                                                      const int32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (flag[0] != 0) attn_body_pp2<T, D, BandPolicy<T, D, 8, false>>(prm_alt, smem, nullptr);
    else attn_body_pp2<T, D, BandPolicy<T, D, 8, false>>(prm, smem, nullptr);
}
template <typename T, int D, int ABL>
__global__ __launch_bounds__(512, 2) void band_attn_pp2_trace_kernel(typename BandPolicy<T, D, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pp2<T, D, BandPolicy<T, D, 8, false>, true, ABL>(prm, smem, nullptr);
}

// =====================================================================================================
// Variable-block policy (SVG2): q rows of block-row i attend the kv rows of the active block-cols.
// The active, non-empty column blocks of the workgroup's block-row are compacted into a run list in LDS
// (start, inclusive prefix of lengths); KV tiles are cut from the *concatenation* of the runs, so tiles are
// always full except the last one — no per-cluster padding waste on the key side.
// =====================================================================================================
constexpr int kVbMaxKB = 4096;
constexpr int kVbFull = 256;   // mixed tiling: full 256-row tiles go to the 8-wave kernel, the rest of a block-row to 128-row tiles

template <typename T, int D, int NW>
struct VarblockPolicy {
    static constexpr bool kFixup = false;
    static constexpr bool kPartialOut = false;
    static constexpr bool kIntervalMask = true;
    static constexpr bool kFastPartial = false;
    static constexpr int kShadow128 = 2;   // the vector phase also resolves rows through the run list and the index arrays
    static constexpr int kAbl = 0;
    static constexpr bool kSetPrio = false;
    static constexpr bool kSkew = false;
    static constexpr int kRowBlocks = 1;
    static constexpr int kSubTiles = 1;
    static constexpr int kPrefetch = 1;
    static constexpr int BM = NW * 32;

    struct Params {
        const T* q;
        const T* k;
        const T* v;
        T* o;
        int Hq, Hkv, group, Sq, Skv, QB, KB, max_tiles, kb_cap;
        int tile_mode;              // 0: ceil(n / BM) tiles per block-row; 1: only its full 256-row tiles; 2: its rows after them
        float scale_log2;
        const uint8_t* block_map;   // [Hkv, QB, KB]
        const int32_t* q_off;       // [Hkv, QB + 1] exclusive prefix of q_sizes
        const int32_t* k_off;       // [Hkv, KB + 1]
        const int32_t* tile_off;    // [Hkv, QB + 1] exclusive prefix of ceil(q_size / BM)
        const int32_t* order;       // longest-first launch order (or nullptr): [0] = #workgroups, then (hq, block-row << 16 | sub-tile)
        const int32_t* q_row_idx;   // [Hq, Sq] or null
        const int32_t* kv_row_idx;  // [Hkv, Skv] or null
    };
    struct Ctx {
        int hq, hkv, q0, q_end, nT, total;  // q rows [q0, q_end) in permuted coordinates; total = active keys
        const int32_t* run_start;           // LDS: permuted start position of run j
        const int32_t* run_pref;            // LDS: inclusive prefix of run lengths (run_pref[j] = end of run j in compact coords)
        const int32_t* qidx;
        const int32_t* kidx;
        int nruns;
    };
    struct KvCursor {
        int j;
    };

    static __device__ __forceinline__ bool init(const Params& p, Ctx& c, char* plds) {
        int i, sub;
        if (p.order) {   // 1-D grid in longest-first order (varblock_order_kernel)
            const int b = blockIdx.x;
            if (b >= p.order[0]) return false;
            c.hq = p.order[2 + 2 * b];
            const int e = p.order[3 + 2 * b];
            i = e >> 16, sub = e & 0xFFFF;
            c.hkv = c.hq / p.group;
        } else {
            c.hq = blockIdx.y;
            c.hkv = c.hq / p.group;
            const int32_t* toff = p.tile_off + (size_t)c.hkv * (p.QB + 1);
            const int w = blockIdx.x;
            if (w >= toff[p.QB]) return false;
            // block-row i with tile_off[i] <= w < tile_off[i+1]
            int a = 0, bnd = p.QB;
            while (bnd - a > 1) {
                const int mid = (a + bnd) >> 1;
                if (toff[mid] <= w) a = mid; else bnd = mid;
            }
            i = a;
            sub = w - toff[i];
        }
        const int32_t* qoff = p.q_off + (size_t)c.hkv * (p.QB + 1);
        const int base = qoff[i] + (p.tile_mode == 2 ? ((qoff[i + 1] - qoff[i]) / kVbFull) * kVbFull : 0);
        c.q0 = base + sub * BM;
        c.q_end = min(qoff[i + 1], c.q0 + BM);
        c.qidx = p.q_row_idx ? p.q_row_idx + (size_t)c.hq * p.Sq : nullptr;
        c.kidx = p.kv_row_idx ? p.kv_row_idx + (size_t)c.hkv * p.Skv : nullptr;

        // ---- compact the active non-empty column blocks of row i into the LDS run list ----
        int32_t* run_start = (int32_t*)plds;
        int32_t* run_pref = run_start + p.kb_cap;
        int32_t* wave_cnt = run_pref + p.kb_cap;      // [NW] counts, [NW] lengths
        const uint8_t* mrow = p.block_map + ((size_t)c.hkv * p.QB + i) * p.KB;
        const int32_t* koff = p.k_off + (size_t)c.hkv * (p.KB + 1);
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        constexpr int NT = NW * 64;
        int base_cnt = 0, base_len = 0;
        for (int j0 = 0; j0 < p.KB; j0 += NT) {
            const int j = j0 + tid;
            int len = 0, st = 0;
            if (j < p.KB && mrow[j]) {
                st = koff[j];
                len = koff[j + 1] - st;
            }
            const int flag = len > 0;
            int icnt = flag, ilen = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t1 = __shfl_up(icnt, o), t2 = __shfl_up(ilen, o);
                if (lane >= o) icnt += t1, ilen += t2;
            }
            __syncthreads();  // previous round's readers of wave_cnt are done
            if (lane == 63) wave_cnt[wv] = icnt, wave_cnt[NW + wv] = ilen;
            __syncthreads();
            int wc = base_cnt, wl = base_len;
            for (int x = 0; x < wv; ++x) wc += wave_cnt[x], wl += wave_cnt[NW + x];
            if (flag) {
                run_start[wc + icnt - 1] = st;
                run_pref[wc + icnt - 1] = wl + ilen;
            }
            for (int x = 0; x < NW; ++x) base_cnt += wave_cnt[x], base_len += wave_cnt[NW + x];
        }
        __syncthreads();
        c.nruns = base_cnt;
        c.total = base_len;
        c.run_start = run_start;
        c.run_pref = run_pref;
        c.nT = (c.total + kBN - 1) / kBN;
        return true;
    }

    static __device__ __forceinline__ const T* q_base(const Params& p, const Ctx& c) { return p.q + (size_t)c.hq * p.Sq * D; }
    static __device__ __forceinline__ const T* k_base(const Params& p, const Ctx& c) { return p.k + (size_t)c.hkv * p.Skv * D; }
    static __device__ __forceinline__ const T* v_base(const Params& p, const Ctx& c) { return p.v + (size_t)c.hkv * p.Skv * D; }
    static __device__ __forceinline__ T* o_base(const Params& p, const Ctx& c) { return p.o + (size_t)c.hq * p.Sq * D; }

    static __device__ __forceinline__ int q_logical(const Ctx& c, int row) { return c.q0 + row; }
    static __device__ __forceinline__ bool wave_active(const Ctx& c, int wrow0) { return c.q0 + wrow0 < c.q_end; }
    static __device__ __forceinline__ int q_phys(const Params&, const Ctx& c, int row) {
        const int l = c.q0 + row;
        if (l >= c.q_end) return -1;
        return c.qidx ? c.qidx[l] : l;
    }
    static __device__ __forceinline__ int tile_key0(const Ctx&, int t) { return t * kBN; }
    static __device__ __forceinline__ void kv_cursor_init(const Params&, const Ctx&, KvCursor& cu, int) { cu.j = 0; }
    static __device__ __forceinline__ int kv_phys(const Params&, const Ctx& c, KvCursor& cu, int t, int row) {
        const int pos = t * kBN + row;  // compact coordinate
        if (pos >= c.total) return 0;   // masked by allowed(): reads row 0
        int j = cu.j;
        while (c.run_pref[j] <= pos) ++j;  // tiles advance monotonically: amortised O(1)
        cu.j = j;
        const int len_before = j > 0 ? c.run_pref[j - 1] : 0;
        const int perm = c.run_start[j] + (pos - len_before);
        return c.kidx ? c.kidx[perm] : perm;
    }
    static __device__ __forceinline__ int classify(const Params&, const Ctx& c, int k0, int wrow0) {
        if (c.q0 + wrow0 >= c.q_end) return TILE_SKIP;
        return (k0 + kBN <= c.total) ? TILE_FULL : TILE_PARTIAL;
    }
    static __device__ __forceinline__ bool allowed(const Params&, const Ctx& c, int, int k) { return k < c.total; }
    static __device__ __forceinline__ void row_intervals(const Params&, const Ctx& c, int, int& a0, unsigned& alen, int& b0,
                                                         unsigned& blen) {
        a0 = 0, alen = (unsigned)c.total, b0 = 0, blen = 0u;
    }
    static __device__ __forceinline__ void notify(const Params&, const Ctx&) {}
    static __device__ __forceinline__ float score_fixup(const Params&, float s) { return s; }
};

template <typename T, int D, int NW>
__global__ __launch_bounds__(NW * 64, 2) void varblock_attn_kernel(typename VarblockPolicy<T, D, NW>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body<T, D, NW, VarblockPolicy<T, D, NW>>(prm, smem, smem + attn_lds_bytes<D, NW>());
}

// two-phase ping-pong body for the variable-block policy (256-row q tiles)
template <typename T, int D>
__global__ __launch_bounds__(512, 2) void varblock_attn_pp2_kernel(typename VarblockPolicy<T, D, 8>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pp2<T, D, VarblockPolicy<T, D, 8>>(prm, smem, smem + attn_pp2_lds_bytes<D>());
}

// the same kernel with the launch timeline of svg_debug_wg_trace (variant 5)
template <typename T, int D>
__global__ __launch_bounds__(512, 2) void varblock_attn_pp2_trace_kernel(typename VarblockPolicy<T, D, 8>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pp2<T, D, VarblockPolicy<T, D, 8>, true>(prm, smem, smem + attn_pp2_lds_bytes<D>());
}

static inline int vb_policy_lds(int kb_cap) { return (2 * kb_cap + 32) * (int)sizeof(int32_t); }

// plan: exclusive prefix sums of q_sizes, k_sizes and of the per-block-row tile counts.  grid = (Hkv), block = 256
__global__ __launch_bounds__(256) void varblock_plan_kernel(const int32_t* __restrict__ q_sizes,
                                                            const int32_t* __restrict__ k_sizes, int32_t* __restrict__ q_off,
                                                            int32_t* __restrict__ k_off, int32_t* __restrict__ tile_off,
                                                            int32_t* __restrict__ tile_off2, int QB, int KB, int BM) {
    __shared__ int32_t wtot[4];
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // div > 0: ceil(v / div); div == -1: number of full kVbFull-row tiles; div == -2: 128-row tiles of the remainder
    auto scan = [&](const int32_t* in, int32_t* out, int n, int div) {
        int carry = 0;
        for (int i0 = 0; i0 < n; i0 += 256) {
            const int i = i0 + tid;
            int v = i < n ? in[i] : 0;
            if (div > 0) v = (v + div - 1) / div;
            else if (div == -1) v = v / kVbFull;
            else if (div == -2) v = (v % kVbFull + 127) / 128;
            int incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            __syncthreads();
            if (lane == 63) wtot[wv] = incl;
            __syncthreads();
            int wb = carry;
            for (int x = 0; x < wv; ++x) wb += wtot[x];
            if (i < n) out[i] = wb + incl - v;
            carry += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        }
        if (tid == 0) out[n] = carry;
        __syncthreads();
    };
    scan(q_sizes + (size_t)h * QB, q_off + (size_t)h * (QB + 1), QB, 0);
    scan(k_sizes + (size_t)h * KB, k_off + (size_t)h * (KB + 1), KB, 0);
    if (BM > 0) {
        scan(q_sizes + (size_t)h * QB, tile_off + (size_t)h * (QB + 1), QB, BM);
    } else {  // mixed tiling
        scan(q_sizes + (size_t)h * QB, tile_off + (size_t)h * (QB + 1), QB, -1);
        scan(q_sizes + (size_t)h * QB, tile_off2 + (size_t)h * (QB + 1), QB, -2);
    }
}

// Longest-first launch order of the 256-row variable-block kernel.  The work of a workgroup is the number of active keys of its
// block-row (top-p keeps between a few and all key clusters); in block-row order the last round of the launch ends with whatever
// rows come last (modelled makespan 2.4 % over the ideal at Wan 720p, 0.5 % longest-first).  The order stays head-major — a global
// longest-first order interleaves all heads and their K/V (1.5 GB at Wan 720p) no longer stay in the Infinity Cache: 38.4 ms
// instead of 33.6 — and is longest-first inside every kv head.  A counting sort on (head, 64-key tile count / 16), in three small
// launches: histogram (one wave per block-row), scan, scatter.
constexpr int kVbBuckets = 64;   // per kv head
__global__ __launch_bounds__(256) void varblock_work_kernel(const uint8_t* __restrict__ block_map, const int32_t* __restrict__ k_sizes,
                                                            const int32_t* __restrict__ tile_off, int32_t* __restrict__ work,
                                                            int32_t* __restrict__ hist, int Hkv, int QB, int KB, int group) {
    // (bucket = head-major key: h * kVbBuckets + descending work class)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= Hkv * QB) return;
    const int h = row / QB, i = row - h * QB;
    const uint8_t* m = block_map + (size_t)row * KB;
    const int32_t* ks = k_sizes + (size_t)h * KB;
    int keys = 0;
    for (int j = lane; j < KB; j += 64) keys += m[j] ? ks[j] : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) keys += __shfl_xor(keys, o);
    if (lane == 0) {
        const int tiles = (keys + kBN - 1) / kBN;
        const int bucket = h * kVbBuckets + (kVbBuckets - 1 - min(tiles / 16, kVbBuckets - 1));   // descending work inside the head
        work[row] = bucket;
        const int32_t* toff = tile_off + (size_t)h * (QB + 1);
        const int n = (toff[i + 1] - toff[i]) * group;
        if (n > 0) atomicAdd(hist + bucket, n);
    }
}
__global__ __launch_bounds__(256) void varblock_scan_kernel(int32_t* __restrict__ hist, int32_t* __restrict__ order, int nb) {
    __shared__ int32_t part[256];
    const int tid = threadIdx.x;
    const int per = (nb + 255) / 256, lo = min(tid * per, nb), hi = min(lo + per, nb);
    int sum = 0;
    for (int x = lo; x < hi; ++x) sum += hist[x];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int x = 0; x < 256; ++x) {
            const int t = part[x];
            part[x] = run;
            run += t;
        }
        order[0] = run;   // number of workgroups
    }
    __syncthreads();
    int run = part[tid];
    for (int x = lo; x < hi; ++x) {   // the histogram becomes the scatter cursor of each bucket
        const int t = hist[x];
        hist[x] = run;
        run += t;
    }
}
__global__ __launch_bounds__(256) void varblock_scatter_kernel(const int32_t* __restrict__ tile_off, const int32_t* __restrict__ work,
                                                               int32_t* __restrict__ cursor, int32_t* __restrict__ order, int Hkv,
                                                               int QB, int group) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= Hkv * QB) return;
    const int h = row / QB, i = row - h * QB;
    const int32_t* toff = tile_off + (size_t)h * (QB + 1);
    const int nsub = toff[i + 1] - toff[i];
    if (nsub <= 0) return;
    int pos = atomicAdd(cursor + work[row], nsub * group);
    for (int g = 0; g < group; ++g)
        for (int sub = 0; sub < nsub; ++sub, ++pos) {
            order[2 + 2 * pos] = h * group + g;
            order[3 + 2 * pos] = (i << 16) | sub;
        }
}

thread_local int g_last_hip_error = 0;
static thread_local bool g_band_pipe = false;  // set per call from `variant` bit 2
static thread_local bool g_band_pp = false;    // set per call from `variant` bit 5
static thread_local bool g_band_pp_trace = false;  // `variant` bit 6: ping-pong schedule with the cycle trace (bf16, D = 128)
static thread_local bool g_band_pp2 = false;       // `variant` bit 7: two-phase ping-pong schedule (attn_body_pp2)
static thread_local int32_t* g_band_done = nullptr;     // svg_band_attention_notify: per-head completion counters of this call
static thread_local int g_band_done_nseg = 1;           // ... split into this many row segments per head
static thread_local bool g_vb_trace = false;            // svg_varblock_attention variant 5: variant 3 with the launch timeline (bf16, D = 128)
static thread_local bool g_vb_block_row_order = false;  // svg_varblock_attention variant 4: two-phase kernel in block-row order (A/B)
static thread_local int g_band_pp_abl = 0;         // `variant` bits 8..11 together with bit 6: ablation of the traced kernel

template <typename K, typename Prm>
static int launch_attn(K kernel, const Prm& prm, dim3 grid, int threads, int lds, hipStream_t st) {
    static thread_local const void* configured[16];
    static thread_local int nconf = 0;
    bool seen = false;
    for (int i = 0; i < nconf; ++i) seen |= (configured[i] == (const void*)kernel);
    if (!seen) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            g_last_hip_error = (int)e;
            return SVG_ERR_LAUNCH;
        }
        if (nconf < 16) configured[nconf++] = (const void*)kernel;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, st, prm);
    return launch_status();
}

template <typename Pol, typename T>
static typename Pol::Params make_band_params(const void* q, const void* k, const void* v, void* o, int BH, int S, float sm_scale,
                                             const svg_band_mask_t* mask, const svg_perm_desc_t* perm) {
    typename Pol::Params p;
    p.q = (const T*)q, p.k = (const T*)k, p.v = (const T*)v, p.o = (T*)o;
    p.S = S, p.BH = BH, p.nqt = (S + Pol::BM - 1) / Pol::BM;
    p.scale_log2 = sm_scale * 1.4426950408889634f;
    p.real_len = mask->real_len, p.band = mask->band;
    p.cf_lo = mask->colfull_lo, p.cf_hi = mask->colfull_hi, p.rf_lo = mask->rowfull_lo, p.rf_hi = mask->rowfull_hi;
    p.done = g_band_done, p.done_nseg = 1, p.done_tps = 1 << 30;   // (done_tps is set once nqt is known, below)
    p.head_flag = nullptr, p.vid0 = 0, p.F = 1, p.P = 1, p.V = 0;
    if (perm && perm->head_perm_flag) {
        p.head_flag = perm->head_perm_flag;
        p.vid0 = perm->vid0, p.F = perm->num_frame, p.P = perm->frame_size, p.V = perm->num_frame * perm->frame_size;
    }
    p.q64 = kBN / p.F, p.r64 = kBN % p.F;
    p.q128 = 2 * kBN / p.F, p.r128 = 2 * kBN % p.F;
    p.sp64 = p.q64 + p.r64 * p.P, p.sp128 = p.q128 + p.r128 * p.P;
    p.wrap_phys = 1 - p.F * p.P;
    // row regions (see Params): cut at rowfull_lo, rowfull_hi (inside [0, real_len)) and real_len; unused slots are empty regions
    // behind the last tile.  A q-tile of full rows visits every key tile on the unmasked fast path (with the text rows sharing a
    // tile with band rows or rows behind real_len, all 1861 tiles of it took the per-element masked path: 9.5 ms instead of 3.2).
    {
        const int real = std::min(std::max(p.real_len, 0), S);
        const bool has_rf = p.rf_hi > p.rf_lo && p.rf_lo < real && p.band <= S;
        const int a = has_rf ? std::max(p.rf_lo, 0) : 0, b = has_rf ? std::min(p.rf_hi, real) : 0;
        const int cuts[5] = {0, a, b, real, S};
        int nreg = 0, t0 = 0, heavy_reg = -1;
        for (int i = 0; i < 4; ++i) {
            if (cuts[i + 1] <= cuts[i]) continue;
            p.reg_lo[nreg] = cuts[i], p.reg_hi[nreg] = cuts[i + 1], p.reg_t0[nreg] = t0;
            if (has_rf && i == 1) heavy_reg = nreg;
            t0 += (cuts[i + 1] - cuts[i] + Pol::BM - 1) / Pol::BM;
            ++nreg;
        }
        p.nqt = t0;
        p.done_nseg = std::max(1, std::min(g_band_done_nseg, t0));
        p.done_tps = (t0 + p.done_nseg - 1) / p.done_nseg;
        for (int i = nreg; i < 4; ++i) p.reg_lo[i] = S, p.reg_hi[i] = S, p.reg_t0[i] = 1 << 30;
        p.heavy_lo = 0, p.n_heavy = 0;
        if (heavy_reg >= 0) {
            p.heavy_lo = p.reg_t0[heavy_reg];
            p.n_heavy = (heavy_reg + 1 < nreg ? p.reg_t0[heavy_reg + 1] : p.nqt) - p.heavy_lo;
            if (p.n_heavy >= p.nqt) p.heavy_lo = 0, p.n_heavy = 0;
        }
    }
    return p;
}

template <typename T, int D, int NW, bool SKEW, int ABL = 0, int RB = 1, int SUBS = 1>
static int run_band(const void* q, const void* k, const void* v, void* o, int BH, int S, float sm_scale,
                    const svg_band_mask_t* mask, const svg_perm_desc_t* perm, hipStream_t st) {
    using Pol = BandPolicy<T, D, NW, SKEW, ABL, RB, SUBS>;
    const typename Pol::Params p = make_band_params<Pol, T>(q, k, v, o, BH, S, sm_scale, mask, perm);
    if constexpr (RB == 2) {
        return launch_attn(band_attn_r64_kernel<T, D, ABL>, p, dim3(p.nqt * BH), 256, attn_lds_bytes<D, 4, 2, 2>(), st);
    } else if constexpr (SUBS == 2) {
        return launch_attn(band_attn_s2_kernel<T, D>, p, dim3(p.nqt * BH), 512, attn_lds_bytes<D, 8, 2, 1, 2>(), st);
    } else {
        if constexpr (NW == 8 && !SKEW && ABL == 0) {
            if constexpr (D == 128 && std::is_same<T, __bf16>::value) {
                if (g_band_pp2 && g_band_pp_trace) {
#define SVG_PP_TRACE(A) case A: return launch_attn(band_attn_pp2_trace_kernel<T, D, A>, p, dim3(p.nqt * BH), 512, attn_pp2_lds_bytes<D>(), st);
                    switch (g_band_pp_abl) {
                        SVG_PP_TRACE(0) SVG_PP_TRACE(1) SVG_PP_TRACE(2) SVG_PP_TRACE(4) SVG_PP_TRACE(5) SVG_PP_TRACE(6) SVG_PP_TRACE(7)
                        default: return SVG_ERR_UNSUPPORTED;
                    }
#undef SVG_PP_TRACE
                }
                if (g_band_pp && g_band_pp_trace) {
#define SVG_PP_TRACE(A) case A: return launch_attn(band_attn_pp_trace_kernel<T, D, A>, p, dim3(p.nqt * BH), 512, attn_pp_lds_bytes<D>(), st);
                    switch (g_band_pp_abl) {
                        SVG_PP_TRACE(0) SVG_PP_TRACE(1) SVG_PP_TRACE(2) SVG_PP_TRACE(3) SVG_PP_TRACE(4) SVG_PP_TRACE(5)
                        SVG_PP_TRACE(6) SVG_PP_TRACE(7)
                        default: return SVG_ERR_UNSUPPORTED;
                    }
#undef SVG_PP_TRACE
                }
            }
            if (g_band_pp2)
                return launch_attn(band_attn_pp2_kernel<T, D>, p, dim3(p.nqt * BH), 512, attn_pp2_lds_bytes<D>(), st);
            if (g_band_pp)
                return launch_attn(band_attn_pp_kernel<T, D>, p, dim3(p.nqt * BH), 512, attn_pp_lds_bytes<D>(), st);
            if (g_band_pipe)
                return launch_attn(band_attn_pipe_kernel<T, D>, p, dim3(p.nqt * BH), 512, attn_lds_bytes<D, 8, 3>(), st);
        }
        return launch_attn(band_attn_kernel<T, D, NW, SKEW, ABL>, p, dim3(p.nqt * BH), NW * 64,
                           attn_lds_bytes<D, NW, attn_stages<NW, Pol>()>(), st);
    }
}

}  // namespace svg

using namespace svg;

namespace svg {
template <typename T, int D>
static int run_band_switch(const void* q, const void* k, const void* v, void* o, int BH, int S, float sm_scale,
                           const svg_band_mask_t* mask, const svg_perm_desc_t* perm, const svg_band_mask_t* alt_mask,
                           const int32_t* flag, hipStream_t st) {
    using Pol = BandPolicy<T, D, 8, false>;
    const typename Pol::Params a = make_band_params<Pol, T>(q, k, v, o, BH, S, sm_scale, mask, perm);
    const typename Pol::Params b = make_band_params<Pol, T>(q, k, v, o, BH, S, sm_scale, alt_mask, nullptr);
    auto kern = band_attn_pp2_switch_kernel<T, D>;
    static thread_local bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, attn_pp2_lds_bytes<D>());
        if (e != hipSuccess) {
            g_last_hip_error = (int)e;
            return SVG_ERR_LAUNCH;
        }
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3(std::max(a.nqt, b.nqt) * BH), dim3(512), attn_pp2_lds_bytes<D>(), st, a, b, flag);
    return launch_status();
}
}  // namespace svg

// one wave: spin until every counter has reached `target` (see BandPolicy::Params::done)
__global__ __launch_bounds__(64) void wait_counters_kernel(const int32_t* __restrict__ counters, int n, int target) {
    for (int i = threadIdx.x; i < n; i += 64) {
        while (__hip_atomic_load(counters + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(32);
    }
}

extern "C" int32_t svg_band_attention_notify_target(int32_t S, const svg_band_mask_t* mask) {
    if (!mask || S <= 0) return -1;
    using Pol = svg::BandPolicy<__bf16, 128, 8, false>;
    const auto p = svg::make_band_params<Pol, __bf16>(nullptr, nullptr, nullptr, nullptr, 1, S, 1.f, mask, nullptr);
    return p.nqt * 8;   // every wave of every q-tile of a head reports once
}

extern "C" int32_t svg_band_attention_notify_layout(int32_t S, const svg_band_mask_t* mask, int32_t nseg, int32_t* row_bounds,
                                                    int32_t* targets) {
    if (!mask || S <= 0 || nseg <= 0 || !row_bounds || !targets) return -1;
    using Pol = svg::BandPolicy<__bf16, 128, 8, false>;
    g_band_done_nseg = nseg;
    const auto p = svg::make_band_params<Pol, __bf16>(nullptr, nullptr, nullptr, nullptr, 1, S, 1.f, mask, nullptr);
    g_band_done_nseg = 1;
    auto tile_row = [&](int t) {   // first row of q-tile t (row order), S behind the last tile
        if (t >= p.nqt) return S;
        int r = 0;
        for (int i = 1; i < 4; ++i)
            if (t >= p.reg_t0[i]) r = i;
        return p.reg_lo[r] + (t - p.reg_t0[r]) * Pol::BM;
    };
    for (int sgm = 0; sgm < p.done_nseg; ++sgm) {
        const int t_lo = sgm * p.done_tps, t_hi = (sgm == p.done_nseg - 1) ? p.nqt : std::min(p.nqt, (sgm + 1) * p.done_tps);
        row_bounds[sgm] = tile_row(t_lo);
        targets[sgm] = (t_hi - t_lo) * 8;
    }
    row_bounds[p.done_nseg] = S;
    return p.done_nseg;   // segments actually used (<= nseg)
}

extern "C" int svg_band_attention_notify(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                         int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                         int32_t* done_per_head, void* stream) {
    return svg_band_attention_notify_seg(q, k, v, o, BH, S, D, dtype, sm_scale, mask, perm, done_per_head, 1, stream);
}

extern "C" int svg_band_attention_notify_seg(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                             int32_t dtype, float sm_scale, const svg_band_mask_t* mask,
                                             const svg_perm_desc_t* perm, int32_t* done, int32_t nseg, void* stream) {
    if (!done || nseg <= 0) return SVG_ERR_BAD_ARG;
    g_band_done = done, g_band_done_nseg = nseg;
    const int rc = svg_band_attention(q, k, v, o, BH, S, D, dtype, sm_scale, mask, perm, 0, stream);
    g_band_done = nullptr, g_band_done_nseg = 1;
    return rc;
}

extern "C" int svg_wait_counters(const int32_t* counters, int32_t n, int32_t target, void* stream) {
    if (!counters || n <= 0) return SVG_ERR_BAD_ARG;
    hipLaunchKernelGGL(wait_counters_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counters, n, target);
    return launch_status();
}

extern "C" int svg_band_attention_switch(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                         int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                         const svg_band_mask_t* alt_mask, const int32_t* use_alt_flag, void* stream) {
    if (!q || !k || !v || !o || !mask || !alt_mask || !use_alt_flag || BH <= 0 || S <= 0) return SVG_ERR_BAD_ARG;
    for (const svg_band_mask_t* m : {mask, alt_mask}) {
        if (m->real_len < 0 || m->real_len > S || m->band < 1 || m->band > S + 1) return SVG_ERR_BAD_ARG;
        if (m->colfull_lo > m->colfull_hi || m->rowfull_lo > m->rowfull_hi) return SVG_ERR_BAD_ARG;
    }
    if (perm && perm->head_perm_flag) {
        if (perm->num_frame <= 0 || perm->frame_size <= 0 || perm->vid0 < 0 ||
            (int64_t)perm->vid0 + (int64_t)perm->num_frame * perm->frame_size > S)
            return SVG_ERR_BAD_ARG;
    }
    if ((int64_t)BH * S * D >= (1ll << 40) || (int64_t)S * D * 2 >= (1ll << 32)) return SVG_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#define SVG_SW_ARGS q, k, v, o, BH, S, sm_scale, mask, perm, alt_mask, use_alt_flag, st
    if (dtype == SVG_DTYPE_BF16) {
        if (D == 128) return run_band_switch<__bf16, 128>(SVG_SW_ARGS);
        if (D == 64) return run_band_switch<__bf16, 64>(SVG_SW_ARGS);
    } else if (dtype == SVG_DTYPE_F16) {
        if (D == 128) return run_band_switch<_Float16, 128>(SVG_SW_ARGS);
        if (D == 64) return run_band_switch<_Float16, 64>(SVG_SW_ARGS);
    }
#undef SVG_SW_ARGS
    return SVG_ERR_UNSUPPORTED;
}

extern "C" int svg_debug_wg_trace(uint64_t* out, int n_workgroups) {
    if (!out || n_workgroups < 0 || n_workgroups > kWgTraceMax) return SVG_ERR_BAD_ARG;
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_trace), (size_t)n_workgroups * 6 * sizeof(uint64_t));
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return SVG_ERR_LAUNCH;
    }
    return SVG_OK;
}

extern "C" int svg_debug_pp_trace(uint64_t* out104) {
    if (!out104) return SVG_ERR_BAD_ARG;
    hipError_t e = hipMemcpyFromSymbol(out104, HIP_SYMBOL(g_pp_trace), 104 * sizeof(uint64_t));
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return SVG_ERR_LAUNCH;
    }
    return SVG_OK;
}

extern "C" int svg_band_attention(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                  int32_t dtype, float sm_scale, const svg_band_mask_t* mask,
                                  const svg_perm_desc_t* perm, int32_t variant, void* stream) {
    if (!q || !k || !v || !o || !mask || BH <= 0 || S <= 0) return SVG_ERR_BAD_ARG;
    if (mask->real_len < 0 || mask->real_len > S || mask->band < 1 || mask->band > S + 1) return SVG_ERR_BAD_ARG;
    if (mask->colfull_lo > mask->colfull_hi || mask->rowfull_lo > mask->rowfull_hi) return SVG_ERR_BAD_ARG;
    if (perm && perm->head_perm_flag) {
        if (perm->num_frame <= 0 || perm->frame_size <= 0 || perm->vid0 < 0 ||
            (int64_t)perm->vid0 + (int64_t)perm->num_frame * perm->frame_size > S)
            return SVG_ERR_BAD_ARG;
    }
    if ((int64_t)BH * S * D >= (1ll << 40)) return SVG_ERR_UNSUPPORTED;
    if ((int64_t)S * D * 2 >= (1ll << 32)) return SVG_ERR_UNSUPPORTED;   // the LDS-DMA requests carry 32-bit byte offsets per head
    hipStream_t st = (hipStream_t)stream;
    // bits 8..11: ablation experiments (timing only, results are wrong): bf16, D = 128, 8 waves, lock-step schedule
    const int abl = (variant >> 8) & 15;
    g_band_pp_abl = (variant & 64) ? abl : 0;
    if (abl && !(variant & 64) && dtype == SVG_DTYPE_BF16 && D == 128) {
        switch (abl) {
            case 1: return run_band<__bf16, 128, 8, false, 1>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            case 2: return run_band<__bf16, 128, 8, false, 2>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            case 3: return run_band<__bf16, 128, 8, false, 3>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            case 4: return run_band<__bf16, 128, 8, false, 4>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            case 5: return run_band<__bf16, 128, 8, false, 5>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            case 6: return run_band<__bf16, 128, 8, false, 6>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            case 7: return run_band<__bf16, 128, 8, false, 7>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            case 8: return run_band<__bf16, 128, 8, false, 8>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            case 9: return run_band<__bf16, 128, 8, false, 9>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            case 10: return run_band<__bf16, 128, 8, false, 10>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            case 11: return run_band<__bf16, 128, 8, false, 11>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            case 12: return run_band<__bf16, 128, 8, false, 12>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            case 13: return run_band<__bf16, 128, 8, false, 13>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
            default: return SVG_ERR_UNSUPPORTED;
        }
    }
    g_band_pipe = (variant & 4) != 0;      // bit 2: software-pipelined schedule (attn_body_pipe)
    g_band_pp = (variant & 32) != 0;       // bit 5: ping-pong schedule (attn_body_pp)
    g_band_pp2 = (variant & 128) != 0;     // bit 7: two-phase ping-pong schedule (attn_body_pp2)
    g_band_pp_trace = (variant & 64) != 0; // bit 6 (with bit 5): cycle trace of one workgroup, read with svg_debug_pp_trace
    // default schedule: two-phase ping-pong (attn_body_pp2); bit 12: the previous default — lock-step, 8 waves x 32 rows,
    // two 64-key tiles per LDS stage (one barrier / staging round per 128 keys)
    if (variant == 0) {
        g_band_pp2 = true;
        if (dtype == SVG_DTYPE_BF16 && D == 128) return run_band<__bf16, 128, 8, false>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
        if (dtype == SVG_DTYPE_BF16 && D == 64) return run_band<__bf16, 64, 8, false>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
        if (dtype == SVG_DTYPE_F16 && D == 128) return run_band<_Float16, 128, 8, false>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
        if (dtype == SVG_DTYPE_F16 && D == 64) return run_band<_Float16, 64, 8, false>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
        return SVG_ERR_UNSUPPORTED;
    }
    if (variant & 4096) {
        if (dtype == SVG_DTYPE_BF16 && D == 128) return run_band<__bf16, 128, 8, false, 0, 1, 2>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
        if (dtype == SVG_DTYPE_BF16 && D == 64) return run_band<__bf16, 64, 8, false, 0, 1, 2>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
        if (dtype == SVG_DTYPE_F16 && D == 128) return run_band<_Float16, 128, 8, false, 0, 1, 2>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
        if (dtype == SVG_DTYPE_F16 && D == 64) return run_band<_Float16, 64, 8, false, 0, 1, 2>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
        return SVG_ERR_UNSUPPORTED;
    }
    if (variant & 8) {                     // bit 3: 4 waves x 64 rows, one wave per SIMD (512 registers)
        if (dtype == SVG_DTYPE_BF16 && D == 128) return run_band<__bf16, 128, 4, false, 0, 2>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
        if (dtype == SVG_DTYPE_BF16 && D == 64) return run_band<__bf16, 64, 4, false, 0, 2>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
        if (dtype == SVG_DTYPE_F16 && D == 128) return run_band<_Float16, 128, 4, false, 0, 2>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
        if (dtype == SVG_DTYPE_F16 && D == 64) return run_band<_Float16, 64, 4, false, 0, 2>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
        return SVG_ERR_UNSUPPORTED;
    }
    const bool w4 = (variant & 1) != 0;    // bit 0: 4 waves x 32 rows, 2 WG / CU; bit 4 (16): 8 waves, one tile per stage
    const bool prio = (variant & 2) != 0;  // bit 1: skewed two-group schedule (experimental; measured 3 % slower than lock-step)
#define SVG_BAND_RUN(T, DD, NWW)                                                                                  \
    return prio ? run_band<T, DD, NWW, true>(q, k, v, o, BH, S, sm_scale, mask, perm, st)                         \
                : run_band<T, DD, NWW, false>(q, k, v, o, BH, S, sm_scale, mask, perm, st);
#define SVG_BAND_DISPATCH(T)                                                                                      \
    if (D == 128) { if (w4) { SVG_BAND_RUN(T, 128, 4) } else { SVG_BAND_RUN(T, 128, 8) } }                        \
    if (D == 64) { if (w4) { SVG_BAND_RUN(T, 64, 4) } else { SVG_BAND_RUN(T, 64, 8) } }
    if (dtype == SVG_DTYPE_BF16) {
        SVG_BAND_DISPATCH(__bf16)
    } else if (dtype == SVG_DTYPE_F16) {
        SVG_BAND_DISPATCH(_Float16)
    }
#undef SVG_BAND_RUN
#undef SVG_BAND_DISPATCH
    return SVG_ERR_UNSUPPORTED;
}

extern "C" size_t svg_varblock_workspace_bytes(int32_t Hq, int32_t Hkv, int32_t QB, int32_t KB, int32_t Sq) {
    if (Hq <= 0 || Hkv <= 0 || QB <= 0 || KB <= 0 || Sq <= 0) return 0;
    // plan (prefix sums) + longest-first order: per-block-row bucket, histogram / cursors, (count, pad, entries[2 * max workgroups])
    const size_t plan = (size_t)Hkv * (3 * (size_t)(QB + 1) + (size_t)(KB + 1));
    const size_t order = (size_t)Hkv * QB + (size_t)Hkv * kVbBuckets + 2 + 2 * ((size_t)Sq / 256 + QB) * Hq;
    return (plan + order) * sizeof(int32_t);
}

namespace svg {
// NW = 4 / 8: uniform tiling (128- / 256-row q tiles).  NW = 0: mixed tiling — the full 256-row tiles of every block-row run
// on the 8-wave kernel, its remaining rows on 128-row tiles of the 4-wave kernel (two workgroups per CU).  k-means clusters are
// ragged (Wan 720p bench: mean 252 rows, sigma 161): uniform 256-row tiles keep 68 % of the processed rows real, uniform
// 128-row tiles 79 % but run the slower 4-wave schedule everywhere; mixed keeps 79 % with most rows on the 8-wave kernel.
template <typename T, int D, int NW>
static int run_varblock(const void* q, const void* k, const void* v, void* o, int Hq, int Hkv, int Sq, int Skv,
                        float sm_scale, const uint8_t* block_map, const int32_t* q_sizes, const int32_t* k_sizes, int QB,
                        int KB, const int32_t* q_row_idx, const int32_t* kv_row_idx, void* ws, hipStream_t st) {
    int32_t* q_off = (int32_t*)ws;
    int32_t* tile_off = q_off + (size_t)Hkv * (QB + 1);
    int32_t* k_off = tile_off + (size_t)Hkv * (QB + 1);
    int32_t* tile_off2 = k_off + (size_t)Hkv * (KB + 1);
    hipLaunchKernelGGL(varblock_plan_kernel, dim3(Hkv), dim3(256), 0, st, q_sizes, k_sizes, q_off, k_off, tile_off, tile_off2, QB,
                       KB, (NW < 0 ? -NW : NW) * 32);
    auto launch = [&](auto nw_c, int mode, const int32_t* toff, int max_tiles) -> int {
        constexpr int W = decltype(nw_c)::value;
        using Pol = VarblockPolicy<T, D, W>;
        typename Pol::Params p;
        p.q = (const T*)q, p.k = (const T*)k, p.v = (const T*)v, p.o = (T*)o;
        p.Hq = Hq, p.Hkv = Hkv, p.group = Hq / Hkv, p.Sq = Sq, p.Skv = Skv, p.QB = QB, p.KB = KB;
        p.max_tiles = max_tiles;
        p.tile_mode = mode;
        p.kb_cap = (KB + 63) / 64 * 64;
        p.scale_log2 = sm_scale * 1.4426950408889634f;
        p.block_map = block_map, p.q_off = q_off, p.k_off = k_off, p.tile_off = toff;
        p.q_row_idx = q_row_idx, p.kv_row_idx = kv_row_idx;
        p.order = nullptr;
        if constexpr (NW == -8) {
            const int group = Hq / Hkv;
            if (!g_vb_block_row_order && QB < 32768 && Sq / 256 + 1 < 65536) {   // packing of (block-row, sub-tile) in one word
                int32_t* work = tile_off2 + (size_t)Hkv * (QB + 1);
                int32_t* hist = work + (size_t)Hkv * QB;
                const int nb = Hkv * kVbBuckets;
                int32_t* order = hist + nb;
                if (hipMemsetAsync(hist, 0, (size_t)nb * sizeof(int32_t), st) != hipSuccess) return SVG_ERR_LAUNCH;
                hipLaunchKernelGGL(varblock_work_kernel, dim3((Hkv * QB + 3) / 4), dim3(256), 0, st, block_map, k_sizes, toff, work, hist,
                                   Hkv, QB, KB, group);
                hipLaunchKernelGGL(varblock_scan_kernel, dim3(1), dim3(256), 0, st, hist, order, nb);
                hipLaunchKernelGGL(varblock_scatter_kernel, dim3((Hkv * QB + 255) / 256), dim3(256), 0, st, toff, work, hist, order, Hkv,
                                   QB, group);
                p.order = order;
                if constexpr (D == 128 && std::is_same<T, __bf16>::value) {
                    if (g_vb_trace)
                        return launch_attn(varblock_attn_pp2_trace_kernel<T, D>, p, dim3(p.max_tiles * Hq), 512,
                                           attn_pp2_lds_bytes<D>() + vb_policy_lds(p.kb_cap), st);
                }
                return launch_attn(varblock_attn_pp2_kernel<T, D>, p, dim3(p.max_tiles * Hq), 512,
                                   attn_pp2_lds_bytes<D>() + vb_policy_lds(p.kb_cap), st);
            }
            return launch_attn(varblock_attn_pp2_kernel<T, D>, p, dim3(p.max_tiles, Hq), 512,
                               attn_pp2_lds_bytes<D>() + vb_policy_lds(p.kb_cap), st);
        } else
            return launch_attn(varblock_attn_kernel<T, D, W>, p, dim3(p.max_tiles, Hq), W * 64,
                               attn_lds_bytes<D, W>() + vb_policy_lds(p.kb_cap), st);
    };
    if constexpr (NW == 0) {
        int rc = SVG_OK;
        if (Sq >= kVbFull) rc = launch(std::integral_constant<int, 8>{}, 1, tile_off, Sq / kVbFull);
        if (rc != SVG_OK) return rc;
        return launch(std::integral_constant<int, 4>{}, 2, tile_off2, 2 * QB);
    } else if constexpr (NW == -8) {   // two-phase ping-pong body, 256-row q tiles
        return launch(std::integral_constant<int, 8>{}, 0, tile_off, Sq / 256 + QB);
    } else {
        return launch(std::integral_constant<int, NW>{}, 0, tile_off, Sq / (NW * 32) + QB);
    }
}
}  // namespace svg

extern "C" int svg_varblock_attention(const void* q, const void* k, const void* v, void* o, int32_t Hq, int32_t Hkv,
                                      int32_t Sq, int32_t Skv, int32_t D, int32_t dtype, float sm_scale,
                                      const uint8_t* block_map, const int32_t* q_sizes, const int32_t* k_sizes, int32_t QB,
                                      int32_t KB, const int32_t* q_row_idx, const int32_t* kv_row_idx, void* workspace,
                                      size_t workspace_bytes, int32_t variant, void* stream) {
    if (!q || !k || !v || !o || !block_map || !q_sizes || !k_sizes || !workspace) return SVG_ERR_BAD_ARG;
    if (Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 || Sq <= 0 || Skv <= 0 || QB <= 0 || KB <= 0) return SVG_ERR_BAD_ARG;
    if (KB > kVbMaxKB) return SVG_ERR_UNSUPPORTED;
    if ((int64_t)Skv * D * 2 >= (1ll << 32) || (int64_t)Sq * D * 2 >= (1ll << 32)) return SVG_ERR_UNSUPPORTED;
    if (workspace_bytes < svg_varblock_workspace_bytes(Hq, Hkv, QB, KB, Sq)) return SVG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    // variant 0: 4 waves, 128-row q tiles; 1: 8 waves, 256-row q tiles; 2: mixed (full 256-row tiles on 8 waves, rest on 4)
    g_vb_block_row_order = (variant == 4);
    g_vb_trace = (variant == 5);
#define SVG_VB_ARGS q, k, v, o, Hq, Hkv, Sq, Skv, sm_scale, block_map, q_sizes, k_sizes, QB, KB, q_row_idx, kv_row_idx, workspace, st
#define SVG_VB_DISPATCH(T)                                                                       \
    if (D == 128) {                                                                              \
        if (variant == 2) return run_varblock<T, 128, 0>(SVG_VB_ARGS);                           \
        if (variant >= 3) return run_varblock<T, 128, -8>(SVG_VB_ARGS);                          \
        return variant == 1 ? run_varblock<T, 128, 8>(SVG_VB_ARGS) : run_varblock<T, 128, 4>(SVG_VB_ARGS); \
    }                                                                                            \
    if (D == 64) {                                                                               \
        if (variant == 2) return run_varblock<T, 64, 0>(SVG_VB_ARGS);                            \
        if (variant >= 3) return run_varblock<T, 64, -8>(SVG_VB_ARGS);                           \
        return variant == 1 ? run_varblock<T, 64, 8>(SVG_VB_ARGS) : run_varblock<T, 64, 4>(SVG_VB_ARGS);   \
    }
    if (variant < -1 || variant > 5) return SVG_ERR_BAD_ARG;
    // -1 (auto): 256-row q tiles with the two-phase ping-pong body once the average block-row is large enough to fill them
    // (Wan 720p, 252-row clusters: 40.4 ms; lock-step 8 waves 45.5, 4 waves 47.7, mixed 46.9), 128-row tiles otherwise
    if (variant == -1) variant = ((int64_t)Sq >= (int64_t)160 * QB) ? 3 : 0;
    if (dtype == SVG_DTYPE_BF16) {
        SVG_VB_DISPATCH(__bf16)
    } else if (dtype == SVG_DTYPE_F16) {
        SVG_VB_DISPATCH(_Float16)
    }
#undef SVG_VB_ARGS
#undef SVG_VB_DISPATCH
    return SVG_ERR_UNSUPPORTED;
}
