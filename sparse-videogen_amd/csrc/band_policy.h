// SVG1 band policy (analytic mask family, see svg_band_mask_t in svg_attn.h) for the attention bodies of attn_core.h / attn_w4.h:
// which KV tiles a workgroup visits, where rows live in HBM, which elements are masked — plus the host-side parameter builder and
// the launch helper shared by the translation units that instantiate band kernels (attention.hip, attention_w4.hip).
#pragma once
#include <algorithm>

#include "attn_core.h"

namespace svg {

// per-call options of a band launch (no process- or thread-global state)
struct BandOpts {
    int32_t* done = nullptr;   // completion counters (svg_band_attention_notify*), or nullptr
    int done_nseg = 1;         // counters per head
    bool prescaled = false;    // q carries sm_scale * log2(e) (svg_band_attention_prescaled; two-phase body only)
    bool trace = false;        // diagnostics builds: the traced kernel (svg_debug_pp_trace)
    int trace_abl = 0;         // ... and its timing ablation
    bool strided = false;      // `lay` describes the tensors (svg_band_attention_strided); otherwise contiguous [BH, S, D]
    AttnLayout lay{};
};

// =====================================================================================================
// Band policy: analytic mask family (see svg_band_mask_t in svg_attn.h)
// =====================================================================================================
template <typename T, int D, int NW, bool SKEW, int ABL = 0, int RB = 1, int SUBS = 1>
struct BandPolicy {
    static constexpr int kHeadDim = D;
    static constexpr int kSubTiles = SUBS;   // 64-key tiles per LDS stage / barrier
    static constexpr int kPrefetch = (ABL == 12) ? 3 : (ABL == 13 ? 2 : 1);  // operand ring depth (k-steps / MFMA steps ahead)
    static constexpr bool kFixup = false;
    static constexpr bool kPartialOut = false;
    static constexpr bool kIntervalMask = true;   // row_intervals() describes the mask (two-phase body)
    static constexpr bool kFastPartial = false;
    static constexpr int kShadow128 = 1;   // two-phase body, D = 128: probability steps in the MFMA shadow (measured best)
    static constexpr int kAbl = ABL;  // > 0 only for the ablation variants (timing experiments)
    static constexpr bool kSetPrio = false;  // measured: s_setprio around the MFMA clusters costs 2 % here
    static constexpr bool kOneBarrier = false;   // two-phase body: two barriers per tile (one measured neutral for this policy)
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
        int32_t* done;             // int32 [BH * done_nseg + BH]: segment counters, then one hidden counter per head (see notify)
        int done_nseg, done_tps;   // counters per head: segment of q-tile qt (row order) = min(qt / done_tps, done_nseg - 1)
        AttnLayout lay;            // strides of q, k, v, o (contiguous [BH, S, D] unless the call came through a *_strided entry point)
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
#ifdef SVG_BAND_NO_XCD_SWIZZLE      // A/B builds: dispatch id = work id (the hardware's round-robin then hands NEIGHBOURING q-tiles to DIFFERENT XCDs)
            if (false) {
#else
            if (b2 < full) {
#endif
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
        // (Round 4 tried a cyclic sweep start here — the 32 q-tiles an XCD works on together start at a common key tile and wrap, so that
        //  the group reads at most two different key tiles at any time — to cut the L2 <-> fabric traffic.  Measured: 87.9 GB per launch
        //  with all 32 on the same tile, 68.4 GB staggered by one tile, against 32.7 GB for the plain sweep, and 1 - 4.5 % more time: the
        //  L2 does not merge requests for a line that is still in flight.  Removed; profiles/r04z_rotate_traffic.txt, r04c_ab_rotate_stagger.txt.)
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

    static __device__ __forceinline__ const T* q_base(const Params& p, const Ctx& c) { return p.q + layout_head_off(p.lay.q_bs, p.lay.q_hs, p.lay.hpb_q, c.head); }
    static __device__ __forceinline__ const T* k_base(const Params& p, const Ctx& c) { return p.k + layout_head_off(p.lay.k_bs, p.lay.k_hs, p.lay.hpb_kv, c.head); }
    static __device__ __forceinline__ const T* v_base(const Params& p, const Ctx& c) { return p.v + layout_head_off(p.lay.v_bs, p.lay.v_hs, p.lay.hpb_kv, c.head); }
    static __device__ __forceinline__ T* o_base(const Params& p, const Ctx& c) { return p.o + layout_head_off(p.lay.o_bs, p.lay.o_hs, p.lay.hpb_q, c.head); }
    // row strides in elements (attn_m16.h; every other body addresses rows at stride D and the host refuses anything else for it)
    static __device__ __forceinline__ int q_rs(const Params& p) { return p.lay.q_rs; }
    static __device__ __forceinline__ int k_rs(const Params& p) { return p.lay.k_rs; }
    static __device__ __forceinline__ int v_rs(const Params& p) { return p.lay.v_rs; }
    static __device__ __forceinline__ int o_rs(const Params& p) { return p.lay.o_rs; }

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
    // Stateful form of tile_key0 for bodies that walk the tiles in order (attn_body_w4): ~3 scalar instructions per tile instead of
    // the ~10 of the three-way select above.  Behind the last tile the cursor holds harmless values.
    struct TileCur {
        int k0, left;       // first key of the tile, tiles left in its segment (this one included)
        int k1, left1;      // the segments behind it (scalars that shift down: a run-time index into the segment arrays of
        int k2, left2;      //  Ctx would send them through scratch)
    };
    static __device__ __forceinline__ void tile_cur_shift(TileCur& tc) {
        tc.k0 = tc.k1, tc.left = tc.left1;
        tc.k1 = tc.k2, tc.left1 = tc.left2;
        tc.left2 = 0;
    }
    static __device__ __forceinline__ void tile_cur_init(const Ctx& c, TileCur& tc) {
        tc.k0 = c.seg_lo[0] * kBN, tc.left = c.seg_n[0];
        tc.k1 = c.seg_lo[1] * kBN, tc.left1 = c.seg_n[1];
        tc.k2 = c.seg_lo[2] * kBN, tc.left2 = c.seg_n[2];
        if (tc.left <= 0) tile_cur_shift(tc);
        if (tc.left <= 0) tile_cur_shift(tc);
    }
    static __device__ __forceinline__ void tile_cur_next(const Ctx& c, TileCur& tc) {
        tile_cur_step(tc);
        if (tile_cur_ended(tc)) tile_cur_fix(c, tc);   // (rare: at most twice per q-tile)
    }
    // the same in two halves, for a caller that folds the rare case into a branch of its own
    static __device__ __forceinline__ void tile_cur_step(TileCur& tc) {
        tc.k0 += kBN;
        --tc.left;
    }
    static __device__ __forceinline__ bool tile_cur_ended(const TileCur& tc) { return tc.left <= 0; }
    static __device__ __forceinline__ void tile_cur_fix(const Ctx&, TileCur& tc) {
        tile_cur_shift(tc);
        if (tc.left <= 0) tile_cur_shift(tc);
    }
    // Branch-free row stepping for bodies that keep their phases straight-line (attn_body_w4): rows_seek places a lane on row
    // `row` of the tile that starts at key k0 (exact, with the division of a token-major head), rows_step advances it by one tile
    // inside a segment, rows_phys gives the physical row (0 for rows behind the sequence: those keys are masked).
    static constexpr bool kRowStep = true;
    struct RowState {
        int l, f, physv;   // logical row; token-major head: frame f and physical row vid0 + f * P + pp of (l - vid0) = pp * F + f
    };
    static __device__ __forceinline__ void rows_seek(const Params& p, const Ctx& c, RowState& st, int k0, int row) {
        st.l = k0 + row;
        const int i = st.l - p.vid0;
        const int a = i >= 0 ? i : -i - 1;              // floor division also for rows in front of the video
        const int qd = (int)((unsigned)a / (unsigned)p.F);
        const int pp = i >= 0 ? qd : -qd - 1;
        st.f = i - pp * p.F;
        st.physv = p.vid0 + st.f * p.P + pp;
    }
    static __device__ __forceinline__ void rows_step(const Params& p, const Ctx&, RowState& st) {
        st.l += kBN;
        int f = st.f + p.r64, physv = st.physv + p.sp64;
        const bool wrap = f >= p.F;
        st.f = wrap ? f - p.F : f;
        st.physv = wrap ? physv + p.wrap_phys : physv;
    }
    static __device__ __forceinline__ int rows_phys(const Params& p, const Ctx& c, const RowState& st) {
        const bool in_video = (unsigned)(st.l - p.vid0) < (unsigned)p.V;
        const int phys = (c.perm && in_video) ? st.physv : st.l;
        return st.l < p.S ? phys : 0;
    }
    // wave-uniform: every (row, key) pair of this wave's rows x the tile at key k0 is allowed (the per-wave interval of init())
    static __device__ __forceinline__ bool fast_full(const Ctx& c, int k0) { return k0 >= c.fk_lo && k0 <= c.fk_hi; }
    static __device__ __forceinline__ void kv_cursor_init(const Params&, const Ctx&, KvCursor& cu, int) {
        cu.physv = 0, cu.f = 0, cu.prev_k0 = -(1 << 30);
    }
    static __device__ __forceinline__ int kv_phys(const Params& p, const Ctx& c, KvCursor& cu, int t, int row) {
        return kv_phys_at(p, c, cu, tile_key0(c, t), row);
    }
    // the same for a tile given by its first key (a TileCur)
    static __device__ __forceinline__ int kv_phys_at(const Params& p, const Ctx& c, KvCursor& cu, int k0, int row) {
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
        const bool in_band = (p.band > 0) & ((unsigned)(q - k + p.band - 1) < (unsigned)(2 * p.band - 1));   // band 0: |q - k| < 0 admits nothing
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
    // Completion counters (svg_band_attention_notify*).  Contract: counter (head, s) reaching its target means the PHYSICAL rows
    // [row_bounds[s], row_bounds[s + 1]) of the head are complete and visible.  Segments are cut in logical (q-tile) order; a head
    // run with the fused layout permutation scatters the rows of a logical segment over every frame, so such a head is released
    // at head granularity: its waves count in a hidden per-head counter (done[BH * nseg + head]) and the last arriver raises
    // all segment counters of the head to their targets at once.
    static __device__ __forceinline__ void notify(const Params& p, const Ctx& c) {
        if (p.done) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __threadfence();
            if ((threadIdx.x & 63) == 0) {
                int32_t* cnt = p.done + c.head * p.done_nseg;
                if (!c.perm) {
                    atomicAdd(cnt + min(c.qt / p.done_tps, p.done_nseg - 1), 1);
                } else if (atomicAdd(p.done + p.BH * p.done_nseg + c.head, 1) == p.nqt * NW - 1) {
                    __threadfence();
                    for (int sgm = 0; sgm < p.done_nseg; ++sgm) {
                        const int t_lo = sgm * p.done_tps;
                        const int t_hi = (sgm == p.done_nseg - 1) ? p.nqt : min(p.nqt, (sgm + 1) * p.done_tps);
                        atomicAdd(cnt + sgm, (t_hi - t_lo) * NW);
                    }
                }
            }
        }
    }
};

// hipFuncAttributeMaxDynamicSharedMemorySize of `kernel` on the CURRENT device raised to at least `lds` bytes.  A cache of the
// driver call, keyed by (device, kernel) and remembering the largest size configured so far: the variable-block kernels ask for
// more LDS when KB grows, and a second GPU driven from the same thread needs its own attribute.
inline int configure_lds(const void* kernel, int lds) {
    struct Entry {
        const void* kernel;
        int device, lds;
    };
    static thread_local Entry table[64];
    static thread_local int n = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    Entry* e = nullptr;
    for (int i = 0; i < n; ++i)
        if (table[i].kernel == kernel && table[i].device == dev) e = &table[i];
    if (e && e->lds >= lds) return SVG_OK;
    const hipError_t err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) {
        g_last_hip_error = (int)err;
        return SVG_ERR_LAUNCH;
    }
    if (e) e->lds = lds;
    else if (n < 64) table[n++] = Entry{kernel, dev, lds};   // (a full table only costs the driver call again)
    return SVG_OK;
}

template <typename K, typename Prm>
inline int launch_attn(K kernel, const Prm& prm, dim3 grid, int threads, int lds, hipStream_t st) {
    const int rc = configure_lds((const void*)kernel, lds);
    if (rc != SVG_OK) return rc;
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, st, prm);
    return launch_status();
}

template <typename Pol, typename T>
inline typename Pol::Params make_band_params(const void* q, const void* k, const void* v, void* o, int BH, int S, float sm_scale,
                                             const svg_band_mask_t* mask, const svg_perm_desc_t* perm, const BandOpts& opts = BandOpts()) {
    typename Pol::Params p;
    p.q = (const T*)q, p.k = (const T*)k, p.v = (const T*)v, p.o = (T*)o;
    p.lay = opts.strided ? opts.lay : contiguous_layout(BH, BH, S, S, Pol::kHeadDim);
    p.S = S, p.BH = BH, p.nqt = (S + Pol::BM - 1) / Pol::BM;
    p.scale_log2 = sm_scale * 1.4426950408889634f;
    p.real_len = mask->real_len, p.band = mask->band;
    p.cf_lo = mask->colfull_lo, p.cf_hi = mask->colfull_hi, p.rf_lo = mask->rowfull_lo, p.rf_hi = mask->rowfull_hi;
    p.done = opts.done, p.done_nseg = 1, p.done_tps = 1 << 30;   // (done_tps is set once nqt is known, below)
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
        p.done_nseg = std::max(1, std::min(opts.done_nseg, t0));
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

// fp8 gathering body (attn_f8.h): quantiser in attention_f8.hip, kernel next to the variable-block policy in attention.hip
struct F8GArgs;
size_t f8g_ws_bytes(int Hq, int Hkv, int Sq, int Skv);
int f8g_quantize(const void* q, const void* k, const void* v, int Hq, int Hkv, int Sq, int Skv, int dtype, float sm_scale, void* ws,
                 F8GArgs* fa, hipStream_t st);

// 4 waves x 64 rows, one wave per SIMD (attn_body_w4, attention_w4.hip)
int run_band_w4(const void* q, const void* k, const void* v, void* o, int BH, int S, int D, int dtype, float sm_scale,
                const svg_band_mask_t* mask, const svg_perm_desc_t* perm, const BandOpts& opts, hipStream_t st);
int w4_read_trace(uint64_t* out104);   // per-phase cycle trace of the last traced w4 launch (diagnostics builds)
// waves that report per 256-row q-tile of the kernel `variant` selects (completion-counter targets); -1: no counters
int band_waves_per_tile(int variant);

}  // namespace svg
