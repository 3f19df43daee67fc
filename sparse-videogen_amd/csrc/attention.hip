// SVG1 band (block-sparse) attention, dense attention and SVG2 variable-block attention for gfx950.
// The MFMA / LDS / online-softmax machinery is attn_core.h; this file supplies the two scheduling policies
// (which KV tiles a workgroup visits, where rows live in HBM, which elements are masked) and the C ABI.
#include <algorithm>

#include "attn_core.h"
#include "attn_f8.h"
#include "attn_m16.h"
#include "band_policy.h"

namespace svg {

// lock-step schedule, NW waves x 32 rows (attn_body): every wave runs QK^T -> softmax -> PV per tile.  The reference schedule of
// the test-suite (variant 1, 4 waves: two workgroups per CU) and the body of the 128-row variable-block kernel and the profiler.
template <typename T, int D, int NW>
__global__ __launch_bounds__(NW * 64, 2) void band_attn_kernel(typename BandPolicy<T, D, NW, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body<T, D, NW, BandPolicy<T, D, NW, false>>(prm, smem, nullptr);
}

// two-phase ping-pong schedule, 8 waves x 32 rows (attn_body_pp2): variant 2
// head_dim 64: FOUR waves per SIMD — 128 registers (the LEAN form of the body) and 64 KiB of LDS per workgroup put two workgroups on a
// CU.  At head_dim 64 a tile is 512 cycles of matrix work beside the same 141 vector instructions per wave as at head_dim 128, and one
// wave gets a third of what the vector pipe can take (tools/probe_exp.hip): with two waves per SIMD the matrix pipe is 40 % busy, with
// four 49 % — 18 % fewer cycles per launch; the chip then meets its power limit at head_dim 64 too and gives back part of it: −10.7 % in
// time on CogVideoX-v1.5, bit-identical output (profiles/r04zr_ab_d64_four_waves.txt, r04zs_*).
template <typename T, int D>
__global__ __launch_bounds__(512, (D == 64 ? 4 : 2)) void band_attn_pp2_kernel(typename BandPolicy<T, D, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pp2<T, D, BandPolicy<T, D, 8, false>, false, 0, false, D == 64>(prm, smem, nullptr);
}
// svg_band_attention_switch at head_dim 64: band_attn_pp2_kernel<T, 64> — four waves per SIMD, the LEAN form of the body — with the
// device-side choice between two parameter blocks in front of it (`flag[0] != 0` selects prm_alt, the dense warm-up mask without the
// layout transformation — the dense / sparse decision of attention_core_logic, hyvideo/attention.py:491-496, without reading the timestep
// back to the host; a one-wave-per-SIMD switch kernel served this head size until the end of round 4)
template <typename T>
__global__ __launch_bounds__(512, 4) void band_attn_pp2_switch64_kernel(typename BandPolicy<T, 64, 8, false>::Params prm,
                                                                        typename BandPolicy<T, 64, 8, false>::Params prm_alt,
                                                                        const int32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (flag[0] != 0) attn_body_pp2<T, 64, BandPolicy<T, 64, 8, false>, false, 0, false, true>(prm_alt, smem, nullptr);
    else attn_body_pp2<T, 64, BandPolicy<T, 64, 8, false>, false, 0, false, true>(prm, smem, nullptr);
}
// Device-side switch between two masks on the pre-scaled two-phase body (svg_band_attention_switch_prescaled): `flag[0] != 0`
// selects prm_alt — the dense warm-up mask without the layout transformation — otherwise prm (see band_attn_pp2_switch64_kernel)
template <typename T, int D>
__global__ __launch_bounds__(512, 2) void band_attn_pp2q_switch_kernel(typename BandPolicy<T, D, 8, false>::Params prm,
                                                                       typename BandPolicy<T, D, 8, false>::Params prm_alt,
                                                                       const int32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (flag[0] != 0) attn_body_pp2<T, D, BandPolicy<T, D, 8, false>, false, 0, true>(prm_alt, smem, nullptr);
    else attn_body_pp2<T, D, BandPolicy<T, D, 8, false>, false, 0, true>(prm, smem, nullptr);
}

// Frozen reference schedule (variant 6; bf16 / D = 128 only): the two-phase body as it stood at the end of round 1 — running row
// maximum with a deferred rescale, one probability step in the shadow of the PV MFMAs, operands fetched at the start of the matrix
// phase.  Kept so that ONE bench run can time it beside the default on the same box (bench.py `same_box_ab`): box-to-box clock
// spread (+-4 %) is as large as a typical schedule gain.
__global__ __launch_bounds__(512, 2) void band_attn_pp2_frozen_kernel(typename BandPolicy<__bf16, 128, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pp2<__bf16, 128, BandPolicy<__bf16, 128, 8, false>, false, 8>(prm, smem, nullptr);
}
// the two-phase schedule on v_mfma_f32_16x16x32 (attn_m16.h): head_dim 128; variant 8
// (issue priority in the matrix phase, ONE barrier per tile: 32.55 ms against 33.1 with two — profiles/r04g_ab_m16_cfg.txt; the 32x32x16
//  body gained nothing from the single barrier because the clock took it back, this one runs ~300 MHz further from the power limit)
#ifndef SVG_M16_PRIO
#define SVG_M16_PRIO 1
#endif
template <typename T, int PRIO = SVG_M16_PRIO, int ONEBAR = 1>
__global__ __launch_bounds__(512, 2) void band_attn_m16_kernel(typename BandPolicy<T, 128, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_m16<T, BandPolicy<T, 128, 8, false>, false, PRIO, ONEBAR>(prm, smem, nullptr);
}
// pre-scaled q on the 16x16x32 body (PRE form of attn_body_m16).  QKF16 = true (q and k as fp16 carriers with S^T on the f16 MFMA) was
// built and measured in round 4 and is not instantiated: see the note at attn_body_m16.
template <typename T, bool QKF16>
__global__ __launch_bounds__(512, 2) void band_attn_m16q_kernel(typename BandPolicy<T, 128, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_m16<T, BandPolicy<T, 128, 8, false>, false, 1, 1, true, QKF16>(prm, smem, nullptr);
}
// device-side switch between two masks on the 16x16x32 body (svg_band_attention_switch[_prescaled] at head_dim 128): `flag[0] != 0` selects prm_alt
template <typename T, bool PRE = false, bool QKF16 = false>
__global__ __launch_bounds__(512, 2) void band_attn_m16_switch_kernel(typename BandPolicy<T, 128, 8, false>::Params prm,
                                                                      typename BandPolicy<T, 128, 8, false>::Params prm_alt,
                                                                      const int32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (flag[0] != 0) attn_body_m16<T, BandPolicy<T, 128, 8, false>, false, 1, 1, PRE, QKF16>(prm_alt, smem, nullptr);
    else attn_body_m16<T, BandPolicy<T, 128, 8, false>, false, 1, 1, PRE, QKF16>(prm, smem, nullptr);
}
// the same for q that carries sm_scale * log2(e) (svg_band_attention_prescaled): no scale-and-shift per score
template <typename T, int D>
__global__ __launch_bounds__(512, 2) void band_attn_pp2q_kernel(typename BandPolicy<T, D, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pp2<T, D, BandPolicy<T, D, 8, false>, false, 0, true>(prm, smem, nullptr);
}
#ifdef SVG_ABLATIONS
// Diagnostics build only (python sparse-videogen_amd/build.py --ablations): the two-phase kernel with the per-phase cycle trace
// and the launch timeline, and its timing ablations (ABL > 0: results are wrong by construction).  Not in the product library.
template <typename T, int D, int ABL>
__global__ __launch_bounds__(512, 2) void band_attn_pp2_trace_kernel(typename BandPolicy<T, D, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pp2<T, D, BandPolicy<T, D, 8, false>, true, ABL>(prm, smem, nullptr);
}
template <typename T>   // trace code 9: the 16x16x32 body (attn_m16.h) with the cycle trace
__global__ __launch_bounds__(512, 2) void band_attn_m16_trace_kernel(typename BandPolicy<T, 128, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_m16<T, BandPolicy<T, 128, 8, false>, true>(prm, smem, nullptr);
}
template <typename T, int D>   // trace code 3: the pre-scaled-q body (svg_band_attention_prescaled) with the cycle trace
__global__ __launch_bounds__(512, 2) void band_attn_pp2q_trace_kernel(typename BandPolicy<T, D, 8, false>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pp2<T, D, BandPolicy<T, D, 8, false>, true, 0, true>(prm, smem, nullptr);
}
#endif

// =====================================================================================================
// Variable-block policy (SVG2): q rows of block-row i attend the kv rows of the active block-cols.
// The active, non-empty column blocks of the workgroup's block-row are compacted into a run list in LDS
// (start, inclusive prefix of lengths); KV tiles are cut from the *concatenation* of the runs, so tiles are
// always full except the last one — no per-cluster padding waste on the key side.
// =====================================================================================================
constexpr int kVbMaxKB = 4032;   // the run list of a block-row ((KB rounded up to 64) + 2 pairs of ints) has to fit beside the four 32 KB stages in 160 KB of LDS
constexpr int kVbFull = 256;   // mixed tiling: full 256-row tiles go to the 8-wave kernel, the rest of a block-row to 128-row tiles

template <typename T, int D, int NW>
struct VarblockPolicy {
    static constexpr bool kFixup = false;
    static constexpr bool kPartialOut = false;
    static constexpr bool kIntervalMask = true;
    static constexpr bool kFastPartial = false;
    static constexpr int kShadow128 = 2;   // the vector phase also resolves rows through the run list and the index arrays
    static constexpr bool kOneBarrier = true;   // two-phase body: one barrier per tile (attn_core.h kOneBar: -1.7 % at Wan 720p)
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
        const int32_t* order;       // launch order (or nullptr): [0] = #workgroups, then triples (hq, block-row << 16 | sub-tile, partner):
                                    // partner >= 0: the ragged last tile of that block-row also carries the ragged last tile of block-row
                                    // `partner` (same kv head) — "remainder packing", see varblock_pair_kernel
        const int32_t* q_row_idx;   // [Hq, Sq] or null
        const int32_t* kv_row_idx;  // [Hkv, Skv] or null
        AttnLayout lay;             // strides of q, k, v, o (contiguous [H, S, D] unless the call came through svg_varblock_attention_strided)
    };
    struct Ctx {
        int hq, hkv, q0, q_end, nT, total;  // q rows [q0, q_end) in permuted coordinates; total = active keys
        // Remainder packing: tile rows [0, ra) are the rows [q0, q_end) of the block-row ("member A"), tile rows [ra, ra + rb) the last rb
        // rows of the partner block-row ("member B", permuted positions jb0 ...).  The run list holds the key blocks both members
        // attend first (kC keys), then those only A attends (up to kCA), then those only B attends (up to total): a row of A may see
        // [0, kCA), a row of B [0, kC) u [kCA, total) — two intervals per row, which is what the bodies' masks take.  Without a
        // partner rb = 0 and kC = kCA = total.
        int ra, rb, jb0, kC, kCA;
        // per WAVE (like BandPolicy::fk_lo): key ranges on which every row of the wave may see every key (FULL tiles) and on which some
        // row may see some key (anything else is SKIP)
        int f1_lo, f1_hi, f2_lo, f2_hi, any1_hi, any2_lo;
        // LDS run list: .x = inclusive prefix of the run lengths (the end of run j in compact coordinates), .y = permuted start
        // position of run j minus the compact position it starts at — one 8-byte read resolves a key: perm = pos + .y
        const int2* run;
        const int32_t* qidx;
        const int32_t* kidx;
        int nruns;
    };
    struct KvCursor {
        int j;
        int2 r, rn;   // run[j] and run[j + 1], kept across tiles: a lane crosses into the next run every other tile (mean run: 119
    };            // keys) and then finds the entry in a register; the read that refills rn has until the next crossing to land

    static __device__ __forceinline__ bool init(const Params& p, Ctx& c, char* plds) {
        int i, sub, partner = -1;
        if (p.order) {   // 1-D grid in longest-first order (varblock_scatter_kernel)
            const int b = blockIdx.x;
            if (b >= p.order[0]) return false;
            c.hq = p.order[2 + 3 * b];
            const int e = p.order[3 + 3 * b];
            partner = p.order[4 + 3 * b];
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
        c.ra = max(c.q_end - c.q0, 0), c.rb = 0, c.jb0 = 0;
        if (partner >= 0) {   // the partner's ragged last tile: its last (size % BM) rows
            const int nj = qoff[partner + 1] - qoff[partner];
            c.rb = nj % BM;
            c.jb0 = qoff[partner + 1] - c.rb;
        }
        c.qidx = p.q_row_idx ? p.q_row_idx + (size_t)c.hq * p.Sq : nullptr;
        c.kidx = p.kv_row_idx ? p.kv_row_idx + (size_t)c.hkv * p.Skv : nullptr;

        // ---- compact the active non-empty column blocks into the LDS run list: one pass per class of key blocks
        //      (both members | only A | only B; without a partner everything is the first class) ----
        int2* run = (int2*)plds;
        int32_t* wave_cnt = (int32_t*)(run + p.kb_cap + 2);  // [NW] counts, [NW] lengths
        const uint8_t* mrow = p.block_map + ((size_t)c.hkv * p.QB + i) * p.KB;
        const uint8_t* mrow2 = partner >= 0 ? p.block_map + ((size_t)c.hkv * p.QB + partner) * p.KB : mrow;
        const int32_t* koff = p.k_off + (size_t)c.hkv * (p.KB + 1);
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        constexpr int NT = NW * 64;
        int base_cnt = 0, base_len = 0;
        auto scan_class = [&](int want) {   // want: 3 = both, 1 = only A, 2 = only B
            for (int j0 = 0; j0 < p.KB; j0 += NT) {
                const int j = j0 + tid;
                int len = 0, st = 0;
                if (j < p.KB) {
                    const int cls = (mrow[j] ? 1 : 0) | (mrow2[j] ? 2 : 0);
                    if (cls == want) {
                        st = koff[j];
                        len = koff[j + 1] - st;
                    }
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
                    run[wc + icnt - 1] = make_int2(wl + ilen, st - (wl + ilen - len));
                }
                for (int x = 0; x < NW; ++x) base_cnt += wave_cnt[x], base_len += wave_cnt[NW + x];
            }
        };
        scan_class(3);
        c.kC = base_len;
        if (partner >= 0) {
            scan_class(1);
            c.kCA = base_len;
            scan_class(2);
        } else {
            c.kCA = base_len;
        }
        // two sentinels behind the last run: the cursor (kv_phys_at) reads one entry ahead and stops at them
        if (tid < 2) run[base_cnt + tid] = make_int2(0x7fffffff, 0);
        __syncthreads();
        c.nruns = base_cnt;
        c.total = base_len;
        c.run = run;
        c.nT = (c.total + kBN - 1) / kBN;
        // per-wave tile classes: rows of this wave = tile rows [w0, w1)
        {
            const int w0 = wave_id() * 32, w1 = min(w0 + 32, c.ra + c.rb);
            const bool only_a = w1 <= c.ra, only_b = w0 >= c.ra;
            c.f1_lo = 0, c.f1_hi = only_a ? c.kCA : c.kC;                 // every row of the wave sees all of [f1_lo, f1_hi)
            c.f2_lo = c.kCA, c.f2_hi = only_b ? c.total : c.kCA;          // ... and of [f2_lo, f2_hi)
            c.any1_hi = only_a ? c.kCA : (only_b ? c.kC : c.total);        // some row sees some key of [0, any1_hi) u [any2_lo, total)
            c.any2_lo = only_b ? c.kCA : c.total;
        }
        return true;
    }

    static __device__ __forceinline__ const T* q_base(const Params& p, const Ctx& c) { return p.q + layout_head_off(p.lay.q_bs, p.lay.q_hs, p.lay.hpb_q, c.hq); }
    static __device__ __forceinline__ const T* k_base(const Params& p, const Ctx& c) { return p.k + layout_head_off(p.lay.k_bs, p.lay.k_hs, p.lay.hpb_kv, c.hkv); }
    static __device__ __forceinline__ const T* v_base(const Params& p, const Ctx& c) { return p.v + layout_head_off(p.lay.v_bs, p.lay.v_hs, p.lay.hpb_kv, c.hkv); }
    static __device__ __forceinline__ T* o_base(const Params& p, const Ctx& c) { return p.o + layout_head_off(p.lay.o_bs, p.lay.o_hs, p.lay.hpb_q, c.hq); }
    static __device__ __forceinline__ int q_rs(const Params& p) { return p.lay.q_rs; }   // row strides in elements (attn_m16.h only, see BandPolicy)
    static __device__ __forceinline__ int k_rs(const Params& p) { return p.lay.k_rs; }
    static __device__ __forceinline__ int v_rs(const Params& p) { return p.lay.v_rs; }
    static __device__ __forceinline__ int o_rs(const Params& p) { return p.lay.o_rs; }

    // (the "logical" index of a query row is its row inside the tile here: all the mask needs is which member it belongs to)
    static __device__ __forceinline__ int q_logical(const Ctx&, int row) { return row; }
    static __device__ __forceinline__ bool wave_active(const Ctx& c, int wrow0) { return wrow0 < c.ra + c.rb; }
    static __device__ __forceinline__ int q_phys(const Params&, const Ctx& c, int row) {
        if (row >= c.ra + c.rb) return -1;
        const int l = row < c.ra ? c.q0 + row : c.jb0 + (row - c.ra);
        return c.qidx ? c.qidx[l] : l;
    }
    static __device__ __forceinline__ int tile_key0(const Ctx&, int t) { return t * kBN; }
    struct TileCur {
        int k0;
    };
    static __device__ __forceinline__ void tile_cur_init(const Ctx&, TileCur& tc) { tc.k0 = 0; }
    static __device__ __forceinline__ void tile_cur_next(const Ctx&, TileCur& tc) { tc.k0 += kBN; }
    static __device__ __forceinline__ void tile_cur_step(TileCur& tc) { tc.k0 += kBN; }
    static __device__ __forceinline__ bool tile_cur_ended(const TileCur&) { return false; }
    static __device__ __forceinline__ void tile_cur_fix(const Ctx&, TileCur&) {}
    static constexpr bool kRowStep = false;   // rows come from the run list (kv_phys_at), resolved between the phases
    static __device__ __forceinline__ bool fast_full(const Ctx& c, int k0) {
        return (k0 >= c.f1_lo && k0 + kBN <= c.f1_hi) || (k0 >= c.f2_lo && k0 + kBN <= c.f2_hi);
    }
    static __device__ __forceinline__ void kv_cursor_init(const Params&, const Ctx& c, KvCursor& cu, int) {
        cu.j = 0;
        cu.r = c.run[0], cu.rn = c.run[1];   // (entries behind the last run are never used: a key behind the last run is clamped)
    }
    static __device__ __forceinline__ int kv_phys(const Params& p, const Ctx& c, KvCursor& cu, int t, int row) {
        return kv_phys_at(p, c, cu, t * kBN, row);
    }
    static __device__ __forceinline__ int kv_phys_at(const Params&, const Ctx& c, KvCursor& cu, int k0, int row) {
        // compact coordinate; keys behind the last one (ragged last tile; masked by allowed()) read the last key: no branch
        const int pos = min(k0 + row, c.total - 1);
        int j = cu.j;
        int2 r = cu.r, rn = cu.rn;
        while (r.x <= pos) {   // tiles advance monotonically: amortised O(1)
            r = rn;
            ++j;
            rn = c.run[j + 1];
        }
        cu.j = j, cu.r = r, cu.rn = rn;
        const int perm = pos + r.y;
        return c.kidx ? c.kidx[perm] : perm;
    }
    static __device__ __forceinline__ int classify(const Params&, const Ctx& c, int k0, int wrow0) {
        if (wrow0 >= c.ra + c.rb) return TILE_SKIP;
        if (fast_full(c, k0)) return TILE_FULL;
        // no row of the wave sees any key of the tile (a tile of the other member's own key blocks): nothing to compute
        const bool any = (k0 < c.any1_hi) || (k0 + kBN > c.any2_lo && k0 < c.total);
        return any ? TILE_PARTIAL : TILE_SKIP;
    }
    static __device__ __forceinline__ bool allowed(const Params&, const Ctx& c, int row, int k) {
        return row < c.ra ? (k < c.kCA) : ((k < c.kC) | ((k >= c.kCA) & (k < c.total)));
    }
    static __device__ __forceinline__ void row_intervals(const Params&, const Ctx& c, int row, int& a0, unsigned& alen, int& b0,
                                                         unsigned& blen) {
        const bool a = row < c.ra;
        a0 = 0, alen = (unsigned)(a ? c.kCA : c.kC), b0 = c.kCA, blen = a ? 0u : (unsigned)(c.total - c.kCA);
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

// the two-phase body on 16x16x32 MFMAs (attn_m16.h) for the variable-block policy: head_dim 128; svg_varblock_attention variant 8
template <typename T>
__global__ __launch_bounds__(512, 2) void varblock_attn_m16_kernel(typename VarblockPolicy<T, 128, 8>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_m16<T, VarblockPolicy<T, 128, 8>>(prm, smem, smem + attn_m16_lds_bytes());
}

#ifdef SVG_ABLATIONS
// the same kernel with the launch timeline of svg_debug_wg_trace (variant 5, diagnostics build only)
template <typename T, int D>
__global__ __launch_bounds__(512, 2) void varblock_attn_pp2_trace_kernel(typename VarblockPolicy<T, D, 8>::Params prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_pp2<T, D, VarblockPolicy<T, D, 8>, true>(prm, smem, smem + attn_pp2_lds_bytes<D>());
}
#endif

// fp8 (e4m3) form: gathering fp8 body of attn_f8.h, NW x 32-row q tiles, two waves per SIMD (8 / NW workgroups per CU).  The waves
// of this lock-step body are independent between barriers, so a tile's time follows its ACTIVE waves and smaller tiles only cost
// more K / V staging per row: the ragged q-clusters of SVG2 (252 +- 160 rows) fill 69 % of 256-row tiles, 80 % of 128-row tiles,
// 89 % of 64-row tiles (tools/vb_stats.py).
#ifndef SVG_VB_F8_WAVES
#define SVG_VB_F8_WAVES 4
#endif
constexpr int kVbF8Waves = SVG_VB_F8_WAVES;
template <typename T>
__global__ __launch_bounds__(kVbF8Waves * 64, 2) void varblock_attn_f8_kernel(typename VarblockPolicy<T, 128, kVbF8Waves>::Params prm,
                                                                              F8GArgs fa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body_f8g<T, VarblockPolicy<T, 128, kVbF8Waves>, kVbF8Waves>(prm, fa, smem, smem + attn_f8_lds_bytes<128, kVbF8Waves>());
}

static inline int vb_policy_lds(int kb_cap) { return (2 * (kb_cap + 2) + 32) * (int)sizeof(int32_t); }

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

// Remainder packing (round 3).  k-means clusters are ragged (Wan 720p: 252 +- 160 rows), so the last q-tile of a block-row is mostly
// padding: only 69 % of the rows of the 256-row tiles are real, and a tile costs its key-tile iterations whatever its row count.
// Two block-rows i, j of a kv head whose ragged last tiles fit into ONE tile (r_i + r_j <= BM) share that tile: it walks the key
// blocks both attend once instead of twice (see VarblockPolicy::Ctx).  q-clusters of the same neighbourhood of the data select
// nearly the same key blocks (median Jaccard of best partners 0.94 on the bench data), so the shared part is most of the list.
// One workgroup per kv head: bitmap rows of the map in LDS (key blocks without rows count as inactive), every block-row keeps its
// own row in registers and looks for the unmatched partner with the most common key blocks among those its remainder fits with;
// mutual choices are matched ("handshake"), a few rounds.  partner[h][i] = j >= 0: i's last tile carries j's too (i is the primary);
// -2: carried by its partner; -1: alone.  Exactness: the mask inside a shared tile is exact (two key intervals per row), so the
// result does not depend on which rows are paired.
constexpr int kVbPairThreads = 512;
constexpr int kVbPairRounds = 3;
constexpr int kVbPairMinCommon = 8;   // common key blocks (~ 8 x 76 keys = 10 key tiles) that pay for the three-pass run-list build
constexpr int kVbPairRows = 64;       // block-rows scored by one workgroup (8 threads each, an eighth of the candidates per thread)
static inline int vb_pair_ws(int KB) { return (((KB + 31) / 32) + 3) & ~3; }   // bitmap row stride in words (16-byte rows)
static inline size_t vb_pair_lds(int QB, int KB) { return ((size_t)QB * vb_pair_ws(KB) + (size_t)QB + kVbPairRows) * sizeof(int32_t); }

// bitmap rows of the map: bit j of word w of row (h, i) = block (i, 32 w + j) active and key block 32 w + j not empty
__global__ __launch_bounds__(256) void varblock_bitmap_kernel(const uint8_t* __restrict__ block_map, const int32_t* __restrict__ k_sizes,
                                                              uint32_t* __restrict__ bits, int Hkv, int QB, int KB, int WS) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)Hkv * QB * WS) return;
    const int w = (int)(idx % WS);
    const long long row = idx / WS;
    const int h = (int)(row / QB);
    uint32_t b = 0;
    if (w * 32 < KB) {
        const uint8_t* m = block_map + row * KB + w * 32;
        const int32_t* ks = k_sizes + (size_t)h * KB + w * 32;
        const int n = min(32, KB - w * 32);
        for (int j = 0; j < n; ++j) b |= (m[j] && ks[j] > 0) ? (1u << j) : 0u;
    }
    bits[idx] = b;
}
// one round, scoring half: grid = (ceil(QB / 64), Hkv).  Thread (il, part): block-row i = 64 blockIdx.x + il looks at an eighth of
// the candidates j for the unmatched one with the most common key blocks whose remainder fits beside its own; the eight partial
// results meet in an LDS arg-max (key = common << 12 | inverted index: ties go to the lowest index).  Branch-free scan, 16-byte
// broadcast loads of the candidates' rows; the thread's own row lives in registers.
__global__ __launch_bounds__(kVbPairThreads) void varblock_pair_score_kernel(const uint32_t* __restrict__ bits_g,
                                                                             const int32_t* __restrict__ q_sizes,
                                                                             const int32_t* __restrict__ rem_g, int32_t* __restrict__ best_g,
                                                                             int QB, int WS, int BM, int round) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* bits = (uint32_t*)smem;                     // [QB][WS]
    int32_t* rem = (int32_t*)(bits + (size_t)QB * WS);    // [QB]
    int32_t* sbest = rem + QB;                            // [kVbPairRows]
    const int h = blockIdx.y, tid = threadIdx.x, il = tid & (kVbPairRows - 1), part = tid / kVbPairRows;
    constexpr int kParts = kVbPairThreads / kVbPairRows;
    {
        const u32x4* src = (const u32x4*)(bits_g + (size_t)h * QB * WS);
        u32x4* dst = (u32x4*)bits;
        for (int x = tid; x < QB * WS / 4; x += kVbPairThreads) dst[x] = src[x];
    }
    for (int x = tid; x < QB; x += kVbPairThreads) rem[x] = round == 0 ? q_sizes[(size_t)h * QB + x] % BM : rem_g[(size_t)h * QB + x];
    if (tid < kVbPairRows) sbest[tid] = (kVbPairMinCommon << 12) - 1;
    __syncthreads();
    const int i = blockIdx.x * kVbPairRows + il;
    constexpr int kRegW = 32;
    u32x4 mine[kRegW / 4];
#pragma unroll
    for (int w4 = 0; w4 < kRegW / 4; ++w4)
        mine[w4] = (i < QB && 4 * w4 < WS) ? *(const u32x4*)(bits + (size_t)i * WS + 4 * w4) : u32x4{0u, 0u, 0u, 0u};
    const int ri = i < QB ? rem[i] : 0;
    const int chunk = (QB + kParts - 1) / kParts, j_lo = part * chunk, j_hi = min(QB, j_lo + chunk);
    int bkey = -1;
    for (int j = j_lo; j < j_hi; ++j) {
        const int rj = rem[j];
        const u32x4* row = (const u32x4*)(bits + (size_t)j * WS);
        int common = 0;
#pragma unroll
        for (int w4 = 0; w4 < kRegW / 4; ++w4) {
            if (4 * w4 < WS) {
                const u32x4 r = row[w4];
                common += __popc(mine[w4][0] & r[0]) + __popc(mine[w4][1] & r[1]) + __popc(mine[w4][2] & r[2]) + __popc(mine[w4][3] & r[3]);
            }
        }
        const bool ok = (j != i) & (rj > 0) & (ri > 0) & (ri + rj <= BM);
        const int key = ok ? ((common << 12) | (0xFFF - j)) : -1;
        bkey = key > bkey ? key : bkey;
    }
    if (bkey >= (kVbPairMinCommon << 12)) atomicMax(&sbest[il], bkey);
    __syncthreads();
    if (part == 0 && i < QB) {
        const int k2 = sbest[il];
        best_g[(size_t)h * QB + i] = k2 >= (kVbPairMinCommon << 12) ? 0xFFF - (k2 & 0xFFF) : -1;
    }
}
// one round, matching half ("handshake"): mutual choices become pairs.  partner[h][i] = j >= 0: i's last tile carries j's too (the
// lower index is the primary); -2: carried by its partner; -1: alone.
__global__ __launch_bounds__(256) void varblock_pair_match_kernel(const int32_t* __restrict__ q_sizes, const int32_t* __restrict__ best,
                                                                  int32_t* __restrict__ rem, int32_t* __restrict__ partner, int n_rows,
                                                                  int QB, int BM, int round) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n_rows) return;
    const int h = row / QB, i = row - h * QB;
    const int bj = best[row];
    const bool matched = bj >= 0 && best[(size_t)h * QB + bj] == i;
    const int r = round == 0 ? q_sizes[row] % BM : rem[row];
    rem[row] = matched ? 0 : r;
    if (matched) partner[row] = i < bj ? bj : -2;
    else if (round == 0) partner[row] = -1;
}

// Launch order: counting sort on (head, descending work class).  A block-row contributes its full tiles (work = its active keys)
// and, unless its partner carries it, its ragged last tile (work = the keys of the union with the partner's list).
__global__ __launch_bounds__(256) void varblock_work_kernel(const uint8_t* __restrict__ block_map, const int32_t* __restrict__ q_sizes,
                                                            const int32_t* __restrict__ k_sizes, const int32_t* __restrict__ partner,
                                                            int32_t* __restrict__ work, int32_t* __restrict__ hist, int Hkv, int QB,
                                                            int KB, int group, int BM) {
    // (bucket = head-major key: h * kVbBuckets + descending work class)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= Hkv * QB) return;
    const int h = row / QB, i = row - h * QB;
    const int pj = partner ? partner[row] : -1;
    const uint8_t* m = block_map + (size_t)row * KB;
    const uint8_t* m2 = pj >= 0 ? block_map + ((size_t)h * QB + pj) * KB : m;
    const int32_t* ks = k_sizes + (size_t)h * KB;
    int keys = 0, ukeys = 0;
    for (int j = lane; j < KB; j += 64) {
        keys += m[j] ? ks[j] : 0;
        ukeys += (m[j] | m2[j]) ? ks[j] : 0;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) keys += __shfl_xor(keys, o), ukeys += __shfl_xor(ukeys, o);
    if (lane == 0) {
        auto bucket_of = [&](int k) {
            const int tiles = (k + kBN - 1) / kBN;
            return h * kVbBuckets + (kVbBuckets - 1 - min(tiles / 16, kVbBuckets - 1));   // descending work inside the head
        };
        const int n = q_sizes[row];
        const int nfull = n / BM, has_rem = (n % BM) > 0 && pj != -2;
        const int b_full = bucket_of(keys), b_rem = bucket_of(ukeys);
        work[2 * row] = b_full;
        work[2 * row + 1] = has_rem ? b_rem : -1;
        if (nfull > 0) atomicAdd(hist + b_full, nfull * group);
        if (has_rem) atomicAdd(hist + b_rem, group);
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
__global__ __launch_bounds__(256) void varblock_scatter_kernel(const int32_t* __restrict__ q_sizes, const int32_t* __restrict__ partner,
                                                               const int32_t* __restrict__ work, int32_t* __restrict__ cursor,
                                                               int32_t* __restrict__ order, int Hkv, int QB, int group, int BM) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= Hkv * QB) return;
    const int h = row / QB, i = row - h * QB;
    const int n = q_sizes[row], nfull = n / BM;
    const int b_full = work[2 * row], b_rem = work[2 * row + 1];
    if (nfull > 0) {
        int pos = atomicAdd(cursor + b_full, nfull * group);
        for (int g = 0; g < group; ++g)
            for (int sub = 0; sub < nfull; ++sub, ++pos) {
                order[2 + 3 * pos] = h * group + g;
                order[3 + 3 * pos] = (i << 16) | sub;
                order[4 + 3 * pos] = -1;
            }
    }
    if (b_rem >= 0) {
        int pos = atomicAdd(cursor + b_rem, group);
        const int pj = partner ? partner[row] : -1;
        for (int g = 0; g < group; ++g, ++pos) {
            order[2 + 3 * pos] = h * group + g;
            order[3 + 3 * pos] = (i << 16) | nfull;
            order[4 + 3 * pos] = pj >= 0 ? pj : -1;
        }
    }
}

// Similarity order of the 256-row variable-block kernel (variant 7; measured in round 3, NOT the default: see svg_varblock_attention).  The longest-first order above hands an XCD 32
// unrelated block-rows of a head at a time: every workgroup streams its own quarter of the head's K / V through that XCD's 4 MiB
// L2 (PMC, Wan 720p: hit rate 31 %, 117 GB per launch between L2 and the fabric for 3.1 GB of tensors).  Block-rows whose key
// lists are (nearly) the same — q-clusters of the same neighbourhood of the data select the same k-clusters — read the same K / V
// rows in the same order, so this kernel puts them next to each other and hands CONSECUTIVE workgroups to the SAME XCD:
//   * one workgroup per kv head builds a nearest-neighbour chain over the block-rows (bitmap rows of the map in LDS, Jaccard
//     similarity of the active key-block sets, start at the block-row with the most active key blocks; QB steps of one block-wide
//     arg-max each);
//   * the sub-tiles of a block-row and the q heads of a GQA group (same key list by construction) stay adjacent;
//   * position p of the head-major chain order is mapped to dispatch id b so that, inside every window of 256 consecutive
//     positions, XCD x (= b % 8, the hardware's round-robin) receives positions [32 x, 32 x + 32) — the remap of the band kernel.
constexpr int kVbChainThreads = 512;
static inline size_t vb_chain_lds(int QB, int KB) {
    const int W = (KB + 31) / 32;
    return ((size_t)QB * (W + 1) + 3 * (size_t)QB + 64) * sizeof(int32_t);
}
__global__ __launch_bounds__(kVbChainThreads) void varblock_chain_kernel(const uint8_t* __restrict__ block_map,
                                                                         const int32_t* __restrict__ k_sizes,
                                                                         const int32_t* __restrict__ tile_off, int32_t* __restrict__ order,
                                                                         int Hkv, int QB, int KB, int group) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = (KB + 31) / 32, WS = W + 1;   // (row stride W + 1 words: thread c reads word w of row c — conflict-free)
    uint32_t* bits = (uint32_t*)smem;            // [QB][WS]
    int32_t* pc = (int32_t*)(bits + (size_t)QB * WS);   // [QB] active key blocks of a block-row; -1 once it is in the chain
    int32_t* chain = pc + QB;                    // [QB] block-row at chain position
    int32_t* cnt = chain + QB;                   // [QB] workgroups of the block-row at chain position (then their exclusive prefix)
    unsigned long long* red = (unsigned long long*)(cnt + QB);   // [8] per-wave arg-max
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int32_t* toff = tile_off + (size_t)h * (QB + 1);
    const int32_t* ks = k_sizes + (size_t)h * KB;
    // ---- bitmap rows: bit j of row i = block (i, j) active and key block j not empty ----
    for (int idx = tid; idx < QB * W; idx += kVbChainThreads) {
        const int i = idx / W, w = idx - i * W;
        const uint8_t* m = block_map + ((size_t)h * QB + i) * KB + w * 32;
        uint32_t b = 0;
        const int n = min(32, KB - w * 32);
        for (int j = 0; j < n; ++j) b |= (m[j] && ks[w * 32 + j] > 0) ? (1u << j) : 0u;
        bits[(size_t)i * WS + w] = b;
    }
    __syncthreads();
    int n_live = 0;
    for (int i = tid; i < QB; i += kVbChainThreads) {
        int c = 0;
        for (int w = 0; w < W; ++w) c += __popc(bits[(size_t)i * WS + w]);
        const bool live = toff[i + 1] > toff[i];   // block-rows without query rows launch nothing
        pc[i] = live ? c : -1;
        n_live += live;
    }
    __syncthreads();
    // ---- nearest-neighbour chain ----
    // One step = one block-wide arg-max of the Jaccard similarity to the current block-row.  A thread keeps ITS candidate's bitmap
    // row in registers for the whole chain (QB <= 512: one candidate per thread; larger maps loop over LDS), the current row is
    // read from LDS with broadcast 16-byte loads, and the arg-max is one LDS atomic per thread on a 32-bit key (similarity in 20
    // bits, inverted index in 12: ties go to the lowest index) — one barrier per step.
    constexpr int kRegW = 32;   // bitmap words a thread can hold (KB <= 1024)
    const bool in_regs = (W <= kRegW) && (QB <= kVbChainThreads);
    uint32_t mine[kRegW];
#pragma unroll
    for (int w = 0; w < kRegW; ++w) mine[w] = (in_regs && tid < QB && w < W) ? bits[(size_t)tid * WS + w] : 0u;
    uint32_t* slot = (uint32_t*)red;   // [3] arg-max slots, used round-robin so that the reset needs no barrier of its own
    if (tid < 3) slot[tid] = 0u;
    // start: the live block-row with the most active key blocks (ties: lowest index)
    __syncthreads();
    for (int i = tid; i < QB; i += kVbChainThreads)
        if (pc[i] >= 0) atomicMax(&slot[0], ((uint32_t)min(pc[i] + 1, 0xFFFFF) << 12) | (uint32_t)(0xFFF - (i & 0xFFF)));
    __syncthreads();
    uint32_t best = slot[0];
    int npos = 0, step = 0;
    const bool wide_idx = QB > 4096;   // (never: KB and QB are limited by the LDS budget of this kernel; kept as a guard)
    while (best != 0u && !wide_idx) {
        const int cur = 0xFFF - (int)(best & 0xFFFu);
        const int pcur = pc[cur];
        ++step;
        __syncthreads();   // everybody has read slot[(step - 1) % 3] and pc[cur]
        if (tid == 0) {
            chain[npos] = cur;
            pc[cur] = -1;
            slot[(step + 1) % 3] = 0u;   // the slot of the NEXT step (last read two steps ago)
        }
        ++npos;
        const uint32_t* crow = bits + (size_t)cur * WS;
        uint32_t key = 0u;
        if (in_regs) {
            if (tid < QB && tid != cur && pc[tid] >= 0) {   // (pc[cur] is being cleared by thread 0: tid != cur covers the race)
                int inter = 0;
#pragma unroll
                for (int w = 0; w < kRegW; ++w)
                    if (w < W) inter += __popc(mine[w] & crow[w]);
                const int uni = pcur + pc[tid] - inter;
                const float jac = uni > 0 ? (float)inter / (float)uni : 1.f;
                key = ((uint32_t)(jac * 1048574.f + 1.f) << 12) | (uint32_t)(0xFFF - tid);
            }
            if (key) atomicMax(&slot[step % 3], key);
        } else {
            for (int i = tid; i < QB; i += kVbChainThreads) {
                if (i == cur || pc[i] < 0) continue;
                int inter = 0;
                for (int w = 0; w < W; ++w) inter += __popc(bits[(size_t)i * WS + w] & crow[w]);
                const int uni = pcur + pc[i] - inter;
                const float jac = uni > 0 ? (float)inter / (float)uni : 1.f;
                atomicMax(&slot[step % 3], ((uint32_t)(jac * 1048574.f + 1.f) << 12) | (uint32_t)(0xFFF - i));
            }
        }
        __syncthreads();
        best = slot[step % 3];
    }
    // ---- workgroups per chain position, exclusive prefix, scatter with the XCD remap ----
    for (int p = tid; p < npos; p += kVbChainThreads) {
        const int i = chain[p];
        cnt[p] = (toff[i + 1] - toff[i]) * group;
    }
    __syncthreads();
    __shared__ int32_t s_base, s_total, s_head_total;
    if (tid == 0) {
        int run = 0;
        for (int p = 0; p < npos; ++p) {   // (npos <= QB <= a few hundred: a serial scan is ~1 us)
            const int t = cnt[p];
            cnt[p] = run;
            run += t;
        }
        int base = 0, total = 0;
        for (int hh = 0; hh < Hkv; ++hh) {
            const int t = tile_off[(size_t)hh * (QB + 1) + QB] * group;
            if (hh < h) base += t;
            total += t;
        }
        s_base = base, s_total = total, s_head_total = run;
        if (h == 0) order[0] = total;
    }
    __syncthreads();
    const int base = s_base, full = (s_total / (kNumXCD * 32)) * (kNumXCD * 32);
    for (int p = tid; p < npos; p += kVbChainThreads) {
        const int i = chain[p];
        const int nsub = toff[i + 1] - toff[i];
        int pos = base + cnt[p];
        for (int g = 0; g < group; ++g)
            for (int sub = 0; sub < nsub; ++sub, ++pos) {
                int b = pos;
                if (pos < full) {
                    const int win = pos / (kNumXCD * 32), r = pos - win * (kNumXCD * 32);
                    b = win * (kNumXCD * 32) + (r % 32) * kNumXCD + r / 32;
                }
                order[2 + 3 * b] = h * group + g;
                order[3 + 3 * b] = (i << 16) | sub;
                order[4 + 3 * b] = -1;
            }
    }
}

thread_local int g_last_hip_error = 0;

// Schedules of svg_band_attention (`variant`, include/svg_attn.h).
enum BandSchedule : int { kBandAuto = 0, kBandLockstep4 = 1, kBandPingPong = 2, kBandW4 = 3, kBandFrozen = 6, kBandM16 = 8 };
// default (variant 0): head_dim 128 -> the two-phase schedule on 16x16x32 MFMAs (attn_m16.h, variant 8; profiles/r04f_ab_m16.txt);
// head_dim 64 -> the two-phase 32x32x16 body in its four-waves-per-SIMD (LEAN) form (band_attn_pp2_kernel, variant 2; profiles/r04zr_*).
// Measured alternatives at head_dim 64: the one-wave-per-SIMD body (variant 3, 13.7 - 14.3 against 12.7 - 12.9 ms on CogVideoX-v1.5) and the
// 16x16x32 body at four waves per SIMD (round 5, no gain: profiles/r05b_ab_d64_m16.txt, removed).  Launches that count completions
// (svg_band_attention_notify*) take the same defaults.
static inline int band_default(int D) { return D == 128 ? kBandM16 : kBandPingPong; }

int band_waves_per_tile(int variant) {
    const int v = variant == kBandAuto ? kBandPingPong : variant;
    return v == kBandW4 ? 4 : ((v == kBandPingPong || v == kBandM16) ? 8 : -1);   // waves that report per 256-row q-tile; -1: no counters
}

template <typename T, int D>
static int run_band_lockstep4(const void* q, const void* k, const void* v, void* o, int BH, int S, float sm_scale,
                              const svg_band_mask_t* mask, const svg_perm_desc_t* perm, hipStream_t st) {
    using Pol = BandPolicy<T, D, 4, false>;
    const typename Pol::Params p = make_band_params<Pol, T>(q, k, v, o, BH, S, sm_scale, mask, perm);
    return launch_attn(band_attn_kernel<T, D, 4>, p, dim3(p.nqt * BH), 256, attn_lds_bytes<D, 4, 2>(), st);
}

template <typename T, int D>
static int run_band_pp2(const void* q, const void* k, const void* v, void* o, int BH, int S, float sm_scale,
                        const svg_band_mask_t* mask, const svg_perm_desc_t* perm, const BandOpts& opts, int trace_abl,
                        hipStream_t st) {
    using Pol = BandPolicy<T, D, 8, false>;
    const typename Pol::Params p = make_band_params<Pol, T>(q, k, v, o, BH, S, sm_scale, mask, perm, opts);
#ifdef SVG_ABLATIONS
    if constexpr (D == 128 && std::is_same<T, __bf16>::value) {
        if (trace_abl >= 0) {
#define SVG_PP_TRACE(A) case A: return launch_attn(band_attn_pp2_trace_kernel<T, D, A>, p, dim3(p.nqt * BH), 512, attn_pp2_lds_bytes<D>(), st);
            switch (trace_abl) {
                SVG_PP_TRACE(0) SVG_PP_TRACE(1) SVG_PP_TRACE(2) SVG_PP_TRACE(4) SVG_PP_TRACE(5) SVG_PP_TRACE(6) SVG_PP_TRACE(7) SVG_PP_TRACE(8)
                case 3: return launch_attn(band_attn_pp2q_trace_kernel<T, D>, p, dim3(p.nqt * BH), 512, attn_pp2_lds_bytes<D>(), st);
                case 9: return launch_attn(band_attn_m16_trace_kernel<T>, p, dim3(p.nqt * BH), 512, attn_m16_lds_bytes(), st);
                default: return SVG_ERR_UNSUPPORTED;
            }
#undef SVG_PP_TRACE
        }
    }
#endif
    if (trace_abl >= 0) return SVG_ERR_UNSUPPORTED;   // the trace / ablation kernels exist in -DSVG_ABLATIONS builds only
    if (opts.prescaled) return launch_attn(band_attn_pp2q_kernel<T, D>, p, dim3(p.nqt * BH), 512, attn_pp2_lds_bytes<D>(), st);
    return launch_attn(band_attn_pp2_kernel<T, D>, p, dim3(p.nqt * BH), 512, attn_pp2_lds_bytes<D>(), st);
}

}  // namespace svg

using namespace svg;

// one wave: spin until every counter has reached `target` (see BandPolicy::Params::done)
__global__ __launch_bounds__(64) void wait_counters_kernel(const int32_t* __restrict__ counters, int n, int target) {
    for (int i = threadIdx.x; i < n; i += 64) {
        while (__hip_atomic_load(counters + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(32);
    }
}

// the same with a deadline (wall_clock64: the constant 100 MHz counter): a waiter can never hang a stream — when the deadline
// passes it sets *timed_out and returns, and whatever was queued behind it runs on incomplete data, which the caller detects by
// reading the flag (bench.py: after warm-up, then falls back to chunk launches)
__global__ __launch_bounds__(64) void wait_counters_deadline_kernel(const int32_t* __restrict__ counters, int n, int target,
                                                                    long long ticks, int32_t* __restrict__ timed_out) {
    const long long t0 = wall_clock64();
    for (int i = threadIdx.x; i < n; i += 64) {
        while (__hip_atomic_load(counters + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (wall_clock64() - t0 > ticks) {
                __hip_atomic_store(timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            __builtin_amdgcn_s_sleep(32);
        }
    }
}

static bool g_trace_is_w4 = false;   // diagnostics only: which translation unit holds the last cycle trace

static int band_check_args(const void* q, const void* k, const void* v, const void* o, int32_t BH, int32_t S, int32_t D,
                           const svg_band_mask_t* mask, const svg_perm_desc_t* perm) {
    if (!q || !k || !v || !o || !mask || BH <= 0 || S <= 0) return SVG_ERR_BAD_ARG;
    if (mask->real_len < 0 || mask->real_len > S || mask->band < 0 || mask->band > S + 1) return SVG_ERR_BAD_ARG;
    if (mask->colfull_lo > mask->colfull_hi || mask->rowfull_lo > mask->rowfull_hi) return SVG_ERR_BAD_ARG;
    if (perm && perm->head_perm_flag) {
        if (perm->num_frame <= 0 || perm->frame_size <= 0 || perm->vid0 < 0 ||
            (int64_t)perm->vid0 + (int64_t)perm->num_frame * perm->frame_size > S)
            return SVG_ERR_BAD_ARG;
    }
    if ((int64_t)BH * S * D >= (1ll << 40)) return SVG_ERR_UNSUPPORTED;
    if ((int64_t)S * D * 2 >= (1ll << 32)) return SVG_ERR_UNSUPPORTED;   // the LDS-DMA requests carry 32-bit byte offsets per head
    return SVG_OK;
}

static int band_dispatch(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D, int32_t dtype,
                         float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm, int32_t variant,
                         const BandOpts& opts, hipStream_t st) {
    int trace_abl = -1;
    if ((variant & 0xFF) == 32) {   // diagnostics builds: traced one-wave-per-SIMD kernel, bits 8..11 = its timing ablation
        BandOpts o2 = opts;
        o2.trace = true;
        o2.trace_abl = (variant >> 8) & 15;
        g_trace_is_w4 = true;
        return run_band_w4(q, k, v, o, BH, S, D, dtype, sm_scale, mask, perm, o2, st);
    }
    if (variant & 64) {   // diagnostics builds: bit 6 = traced two-phase kernel, bits 8..11 = its timing ablation
        trace_abl = (variant >> 8) & 15;
        variant = kBandPingPong;
        g_trace_is_w4 = false;
    }
    if (variant == kBandAuto) variant = band_default(D);
    if (opts.done && band_waves_per_tile(variant) < 0) return SVG_ERR_UNSUPPORTED;
    if (opts.strided && variant != kBandM16 && variant != kBandPingPong) return SVG_ERR_UNSUPPORTED;   // strided tensors: the two-phase bodies only (see svg_attn_layout_t)
#define SVG_BAND_TD(FN, ...)                                                                    \
    if (dtype == SVG_DTYPE_BF16 && D == 128) return FN<__bf16, 128>(__VA_ARGS__);               \
    if (dtype == SVG_DTYPE_BF16 && D == 64) return FN<__bf16, 64>(__VA_ARGS__);                 \
    if (dtype == SVG_DTYPE_F16 && D == 128) return FN<_Float16, 128>(__VA_ARGS__);              \
    if (dtype == SVG_DTYPE_F16 && D == 64) return FN<_Float16, 64>(__VA_ARGS__);                \
    return SVG_ERR_UNSUPPORTED;
    switch (variant) {
        case kBandW4: return run_band_w4(q, k, v, o, BH, S, D, dtype, sm_scale, mask, perm, opts, st);
        case kBandPingPong: { SVG_BAND_TD(run_band_pp2, q, k, v, o, BH, S, sm_scale, mask, perm, opts, trace_abl, st) }
        case kBandLockstep4: { SVG_BAND_TD(run_band_lockstep4, q, k, v, o, BH, S, sm_scale, mask, perm, st) }
        case kBandFrozen: {
            if (dtype != SVG_DTYPE_BF16 || D != 128 || opts.done || opts.prescaled) return SVG_ERR_UNSUPPORTED;
            using Pol = BandPolicy<__bf16, 128, 8, false>;
            const typename Pol::Params p = make_band_params<Pol, __bf16>(q, k, v, o, BH, S, sm_scale, mask, perm, opts);
            return launch_attn(band_attn_pp2_frozen_kernel, p, dim3(p.nqt * BH), 512, attn_pp2_lds_bytes<128>(), st);
        }
        case kBandM16: {   // two-phase body on 16x16x32 MFMAs (attn_m16.h): head_dim 128
            if (D != 128) return SVG_ERR_UNSUPPORTED;
            if (opts.prescaled) {   // PRE form: q carries the softmax scale
                if (dtype == SVG_DTYPE_BF16) {
                    using Pol = BandPolicy<__bf16, 128, 8, false>;
                    const typename Pol::Params p = make_band_params<Pol, __bf16>(q, k, v, o, BH, S, sm_scale, mask, perm, opts);
                    return launch_attn(band_attn_m16q_kernel<__bf16, false>, p, dim3(p.nqt * BH), 512, attn_m16_lds_bytes(), st);
                }
                if (dtype == SVG_DTYPE_F16) {
                    using Pol = BandPolicy<_Float16, 128, 8, false>;
                    const typename Pol::Params p = make_band_params<Pol, _Float16>(q, k, v, o, BH, S, sm_scale, mask, perm, opts);
                    return launch_attn(band_attn_m16q_kernel<_Float16, false>, p, dim3(p.nqt * BH), 512, attn_m16_lds_bytes(), st);
                }
                return SVG_ERR_UNSUPPORTED;
            }
            if (dtype == SVG_DTYPE_BF16) {
                using Pol = BandPolicy<__bf16, 128, 8, false>;
                const typename Pol::Params p = make_band_params<Pol, __bf16>(q, k, v, o, BH, S, sm_scale, mask, perm, opts);
                return launch_attn(band_attn_m16_kernel<__bf16>, p, dim3(p.nqt * BH), 512, attn_m16_lds_bytes(), st);
            }
            if (dtype == SVG_DTYPE_F16) {
                using Pol = BandPolicy<_Float16, 128, 8, false>;
                const typename Pol::Params p = make_band_params<Pol, _Float16>(q, k, v, o, BH, S, sm_scale, mask, perm, opts);
                return launch_attn(band_attn_m16_kernel<_Float16>, p, dim3(p.nqt * BH), 512, attn_m16_lds_bytes(), st);
            }
            return SVG_ERR_UNSUPPORTED;
        }
        default: return SVG_ERR_BAD_ARG;
    }
#undef SVG_BAND_TD
}

extern "C" int svg_band_attention(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                  int32_t dtype, float sm_scale, const svg_band_mask_t* mask,
                                  const svg_perm_desc_t* perm, int32_t variant, void* stream) {
    const int rc = band_check_args(q, k, v, o, BH, S, D, mask, perm);
    if (rc != SVG_OK) return rc;
    return band_dispatch(q, k, v, o, BH, S, D, dtype, sm_scale, mask, perm, variant, BandOpts(), (hipStream_t)stream);
}

extern "C" int svg_band_attention_strided(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                          int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                          const svg_attn_layout_t* layout, void* stream) {
    int rc = band_check_args(q, k, v, o, BH, S, D, mask, perm);
    if (rc != SVG_OK) return rc;
    BandOpts opts;
    opts.strided = true;
    if (rc = layout_from_abi(layout, BH, BH, S, S, D, q, k, v, o, opts.lay); rc != SVG_OK) return rc;
    return band_dispatch(q, k, v, o, BH, S, D, dtype, sm_scale, mask, perm, kBandAuto, opts, (hipStream_t)stream);
}

extern "C" int svg_band_attention_prescaled(const void* q_scaled, const void* k, const void* v, void* o, int32_t BH, int32_t S,
                                            int32_t D, int32_t dtype, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                            void* stream) {
    const int rc = band_check_args(q_scaled, k, v, o, BH, S, D, mask, perm);
    if (rc != SVG_OK) return rc;
    BandOpts opts;
    opts.prescaled = true;
    // (head_dim 128: the PRE form of the 16x16x32 body — 33.0 - 33.3 ms against 33.6 - 34.2 for the 32x32x16 one, same box, profiles/r04k_ab_m16_prescaled.txt)
    return band_dispatch(q_scaled, k, v, o, BH, S, D, dtype, 1.f, mask, perm, kBandAuto, opts, (hipStream_t)stream);
}

extern "C" int32_t svg_band_attention_notify_target(int32_t S, const svg_band_mask_t* mask) {
    if (!mask || S <= 0) return -1;
    using Pol = svg::BandPolicy<__bf16, 128, 8, false>;
    const auto p = svg::make_band_params<Pol, __bf16>(nullptr, nullptr, nullptr, nullptr, 1, S, 1.f, mask, nullptr);
    return p.nqt * band_waves_per_tile(kBandAuto);   // every wave of every q-tile of a head reports once
}

extern "C" int32_t svg_band_attention_notify_layout(int32_t S, const svg_band_mask_t* mask, int32_t nseg, int32_t* row_bounds,
                                                    int32_t* targets) {
    if (!mask || S <= 0 || nseg <= 0 || !row_bounds || !targets) return -1;
    using Pol = svg::BandPolicy<__bf16, 128, 8, false>;   // (q-tiles are 256 rows in every schedule that counts)
    BandOpts opts;
    opts.done_nseg = nseg;
    const auto p = svg::make_band_params<Pol, __bf16>(nullptr, nullptr, nullptr, nullptr, 1, S, 1.f, mask, nullptr, opts);
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
        targets[sgm] = (t_hi - t_lo) * band_waves_per_tile(kBandAuto);
    }
    row_bounds[p.done_nseg] = S;
    return p.done_nseg;   // segments actually used (<= nseg)
}

extern "C" int svg_band_attention_notify(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                         int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                         int32_t* done_per_head, int32_t done_words, void* stream) {
    return svg_band_attention_notify_seg(q, k, v, o, BH, S, D, dtype, sm_scale, mask, perm, done_per_head, done_words, 1, stream);
}

extern "C" int svg_band_attention_notify_seg(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                             int32_t dtype, float sm_scale, const svg_band_mask_t* mask,
                                             const svg_perm_desc_t* perm, int32_t* done, int32_t done_words, int32_t nseg,
                                             void* stream) {
    if (!done || nseg <= 0) return SVG_ERR_BAD_ARG;
    const int rc = band_check_args(q, k, v, o, BH, S, D, mask, perm);
    if (rc != SVG_OK) return rc;
    if ((int64_t)done_words < (int64_t)BH * (nseg + 1)) return SVG_ERR_WORKSPACE;   // segment counters + one hidden counter per head
    BandOpts opts;
    opts.done = done, opts.done_nseg = nseg;
    return band_dispatch(q, k, v, o, BH, S, D, dtype, sm_scale, mask, perm, kBandAuto, opts, (hipStream_t)stream);
}

extern "C" int svg_band_attention_prescaled_notify_seg(const void* q_scaled, const void* k, const void* v, void* o, int32_t BH,
                                                       int32_t S, int32_t D, int32_t dtype, const svg_band_mask_t* mask,
                                                       const svg_perm_desc_t* perm, int32_t* done, int32_t done_words, int32_t nseg,
                                                       void* stream) {
    if (!done || nseg <= 0) return SVG_ERR_BAD_ARG;
    const int rc = band_check_args(q_scaled, k, v, o, BH, S, D, mask, perm);
    if (rc != SVG_OK) return rc;
    if ((int64_t)done_words < (int64_t)BH * (nseg + 1)) return SVG_ERR_WORKSPACE;
    BandOpts opts;
    opts.done = done, opts.done_nseg = nseg, opts.prescaled = true;
    return band_dispatch(q_scaled, k, v, o, BH, S, D, dtype, 1.f, mask, perm, kBandAuto, opts, (hipStream_t)stream);
}

extern "C" int svg_wait_counters(const int32_t* counters, int32_t n, int32_t target, void* stream) {
    if (!counters || n <= 0) return SVG_ERR_BAD_ARG;
    hipLaunchKernelGGL(wait_counters_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counters, n, target);
    return launch_status();
}

extern "C" int svg_wait_counters_deadline(const int32_t* counters, int32_t n, int32_t target, int32_t timeout_ms,
                                          int32_t* timed_out, void* stream) {
    if (!counters || n <= 0 || timeout_ms <= 0 || !timed_out) return SVG_ERR_BAD_ARG;
    hipLaunchKernelGGL(wait_counters_deadline_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counters, n, target,
                       (long long)timeout_ms * 100000LL, timed_out);
    return launch_status();
}

// svg_band_attention_switch (layout == nullptr: contiguous [BH, S, D] tensors) and svg_band_attention_switch_strided
static int band_switch_entry(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D, int32_t dtype,
                             float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm, const svg_band_mask_t* alt_mask,
                             const int32_t* use_alt_flag, const svg_attn_layout_t* layout, void* stream) {
    if (!alt_mask || !use_alt_flag) return SVG_ERR_BAD_ARG;
    int rc = band_check_args(q, k, v, o, BH, S, D, mask, perm);
    if (rc == SVG_OK) rc = band_check_args(q, k, v, o, BH, S, D, alt_mask, nullptr);
    if (rc != SVG_OK) return rc;
    BandOpts opts;
    if (layout) {   // (both switch kernels run a two-phase body: strides as in svg_band_attention_strided)
        opts.strided = true;
        if (rc = layout_from_abi(layout, BH, BH, S, S, D, q, k, v, o, opts.lay); rc != SVG_OK) return rc;
    }
    if (D == 128 && (dtype == SVG_DTYPE_BF16 || dtype == SVG_DTYPE_F16)) {   // the default schedule of this head size (attn_m16.h)
        auto go = [&](auto t_c) -> int {
            using T = decltype(t_c);
            using Pol = BandPolicy<T, 128, 8, false>;
            const typename Pol::Params a = make_band_params<Pol, T>(q, k, v, o, BH, S, sm_scale, mask, perm, opts);
            const typename Pol::Params b = make_band_params<Pol, T>(q, k, v, o, BH, S, sm_scale, alt_mask, nullptr, opts);
            auto kern = band_attn_m16_switch_kernel<T>;
            if (const int r2 = configure_lds((const void*)kern, attn_m16_lds_bytes()); r2 != SVG_OK) return r2;
            hipLaunchKernelGGL(kern, dim3(std::max(a.nqt, b.nqt) * BH), dim3(512), attn_m16_lds_bytes(), (hipStream_t)stream, a, b, use_alt_flag);
            return launch_status();
        };
        return dtype == SVG_DTYPE_BF16 ? go(__bf16{}) : go(_Float16{});
    }
    if (D == 64 && (dtype == SVG_DTYPE_BF16 || dtype == SVG_DTYPE_F16)) {    // ... and of this one (band_attn_pp2_kernel<T, 64>)
        auto go = [&](auto t_c) -> int {
            using T = decltype(t_c);
            using Pol = BandPolicy<T, 64, 8, false>;
            const typename Pol::Params a = make_band_params<Pol, T>(q, k, v, o, BH, S, sm_scale, mask, perm, opts);
            const typename Pol::Params b = make_band_params<Pol, T>(q, k, v, o, BH, S, sm_scale, alt_mask, nullptr, opts);
            auto kern = band_attn_pp2_switch64_kernel<T>;
            if (const int r2 = configure_lds((const void*)kern, attn_pp2_lds_bytes<64>()); r2 != SVG_OK) return r2;
            hipLaunchKernelGGL(kern, dim3(std::max(a.nqt, b.nqt) * BH), dim3(512), attn_pp2_lds_bytes<64>(), (hipStream_t)stream, a, b, use_alt_flag);
            return launch_status();
        };
        return dtype == SVG_DTYPE_BF16 ? go(__bf16{}) : go(_Float16{});
    }
    return SVG_ERR_UNSUPPORTED;
}

extern "C" int svg_band_attention_switch(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                         int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                         const svg_band_mask_t* alt_mask, const int32_t* use_alt_flag, void* stream) {
    return band_switch_entry(q, k, v, o, BH, S, D, dtype, sm_scale, mask, perm, alt_mask, use_alt_flag, nullptr, stream);
}

extern "C" int svg_band_attention_switch_strided(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                                 int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                                 const svg_band_mask_t* alt_mask, const int32_t* use_alt_flag,
                                                 const svg_attn_layout_t* layout, void* stream) {
    if (!layout) return SVG_ERR_BAD_ARG;
    return band_switch_entry(q, k, v, o, BH, S, D, dtype, sm_scale, mask, perm, alt_mask, use_alt_flag, layout, stream);
}

extern "C" int svg_band_attention_switch_prescaled(const void* q_scaled, const void* k, const void* v, void* o, int32_t BH, int32_t S,
                                                   int32_t D, int32_t dtype, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                                   const svg_band_mask_t* alt_mask, const int32_t* use_alt_flag, void* stream) {
    if (!alt_mask || !use_alt_flag) return SVG_ERR_BAD_ARG;
    int rc = band_check_args(q_scaled, k, v, o, BH, S, D, mask, perm);
    if (rc == SVG_OK) rc = band_check_args(q_scaled, k, v, o, BH, S, D, alt_mask, nullptr);
    if (rc != SVG_OK) return rc;
    auto go = [&](auto t_c, auto d_c) -> int {
        using T = decltype(t_c);
        constexpr int DD = decltype(d_c)::value;
        using Pol = BandPolicy<T, DD, 8, false>;
        const typename Pol::Params a = make_band_params<Pol, T>(q_scaled, k, v, o, BH, S, 1.f, mask, perm);
        const typename Pol::Params b = make_band_params<Pol, T>(q_scaled, k, v, o, BH, S, 1.f, alt_mask, nullptr);
        if constexpr (DD == 128) {   // the PRE form of the 16x16x32 body, like svg_band_attention_prescaled at this head size
            auto kern16 = band_attn_m16_switch_kernel<T, true, false>;
            if (const int r2 = configure_lds((const void*)kern16, attn_m16_lds_bytes()); r2 != SVG_OK) return r2;
            hipLaunchKernelGGL(kern16, dim3(std::max(a.nqt, b.nqt) * BH), dim3(512), attn_m16_lds_bytes(), (hipStream_t)stream, a, b, use_alt_flag);
            return launch_status();
        }
        auto kern = band_attn_pp2q_switch_kernel<T, DD>;
        if (const int r2 = configure_lds((const void*)kern, attn_pp2_lds_bytes<DD>()); r2 != SVG_OK) return r2;
        hipLaunchKernelGGL(kern, dim3(std::max(a.nqt, b.nqt) * BH), dim3(512), attn_pp2_lds_bytes<DD>(), (hipStream_t)stream, a, b,
                           use_alt_flag);
        return launch_status();
    };
    if (dtype == SVG_DTYPE_BF16 && D == 128) return go(__bf16{}, std::integral_constant<int, 128>{});
    if (dtype == SVG_DTYPE_BF16 && D == 64) return go(__bf16{}, std::integral_constant<int, 64>{});
    if (dtype == SVG_DTYPE_F16 && D == 128) return go(_Float16{}, std::integral_constant<int, 128>{});
    if (dtype == SVG_DTYPE_F16 && D == 64) return go(_Float16{}, std::integral_constant<int, 64>{});
    return SVG_ERR_UNSUPPORTED;
}

extern "C" int svg_debug_wg_trace(uint64_t* out, int n_workgroups) {
#ifdef SVG_ABLATIONS
    if (!out || n_workgroups < 0 || n_workgroups > kWgTraceMax) return SVG_ERR_BAD_ARG;
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_trace), (size_t)n_workgroups * 6 * sizeof(uint64_t));
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return SVG_ERR_LAUNCH;
    }
    return SVG_OK;
#else
    (void)out, (void)n_workgroups;
    return SVG_ERR_UNSUPPORTED;   // diagnostics builds only (-DSVG_ABLATIONS)
#endif
}

extern "C" int svg_debug_pp_trace(uint64_t* out104) {
#ifdef SVG_ABLATIONS
    if (!out104) return SVG_ERR_BAD_ARG;
    if (g_trace_is_w4) return w4_read_trace(out104);
    hipError_t e = hipMemcpyFromSymbol(out104, HIP_SYMBOL(g_pp_trace), 104 * sizeof(uint64_t));
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return SVG_ERR_LAUNCH;
    }
    return SVG_OK;
#else
    (void)out104;
    return SVG_ERR_UNSUPPORTED;
#endif
}

extern "C" size_t svg_varblock_workspace_bytes(int32_t Hq, int32_t Hkv, int32_t QB, int32_t KB, int32_t Sq) {
    if (Hq <= 0 || Hkv <= 0 || QB <= 0 || KB <= 0 || Sq <= 0) return 0;
    // plan (prefix sums) + launch order: two buckets and the packing partner per block-row, histogram / cursors,
    // (count, pad, entries[3 * max workgroups])
    const size_t plan = (size_t)Hkv * (3 * (size_t)(QB + 1) + (size_t)(KB + 1));
    const size_t order = 3 * (size_t)Hkv * QB + (size_t)Hkv * kVbBuckets + 2 + 3 * ((size_t)Sq / 64 + QB) * Hq;   // (q tiles of >= 64 rows)
    const size_t bitmap = KB <= 1024 ? (size_t)Hkv * QB * vb_pair_ws(KB) + 4 : 0;   // remainder packing: bitmap rows of the map (16-byte aligned)
    return (plan + order + bitmap) * sizeof(int32_t);
}

namespace svg {
// NW = 4 / 8: uniform tiling (128- / 256-row q tiles).  NW = 0: mixed tiling — the full 256-row tiles of every block-row run
// on the 8-wave kernel, its remaining rows on 128-row tiles of the 4-wave kernel (two workgroups per CU).  k-means clusters are
// ragged (Wan 720p bench: mean 252 rows, sigma 161): uniform 256-row tiles keep 68 % of the processed rows real, uniform
// 128-row tiles 79 % but run the slower 4-wave schedule everywhere; mixed keeps 79 % with most rows on the 8-wave kernel.
template <typename T, int D, int NW>
static int run_varblock(const void* q, const void* k, const void* v, void* o, int Hq, int Hkv, int Sq, int Skv,
                        float sm_scale, const uint8_t* block_map, const int32_t* q_sizes, const int32_t* k_sizes, int QB,
                        int KB, const int32_t* q_row_idx, const int32_t* kv_row_idx, void* ws, bool block_row_order, bool trace,
                        hipStream_t st, const F8GArgs* f8 = nullptr, int order_mode = 0, bool body_m16 = false,
                        const AttnLayout* lay = nullptr) {
    // body_m16 (NW == -8, head_dim 128): the two-phase body on 16x16x32 MFMAs (attn_m16.h) instead of the 32x32x16 one
    int32_t* q_off = (int32_t*)ws;
    int32_t* tile_off = q_off + (size_t)Hkv * (QB + 1);
    int32_t* k_off = tile_off + (size_t)Hkv * (QB + 1);
    int32_t* tile_off2 = k_off + (size_t)Hkv * (KB + 1);
    hipLaunchKernelGGL(varblock_plan_kernel, dim3(Hkv), dim3(256), 0, st, q_sizes, k_sizes, q_off, k_off, tile_off, tile_off2, QB,
                       KB, (NW == -9 ? kVbF8Waves : NW < 0 ? 8 : NW) * 32);
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
        p.lay = lay ? *lay : contiguous_layout(Hq, Hkv, Sq, Skv, D);
        p.order = nullptr;
        if constexpr (NW == -8 || NW == -9) {
            const int group = Hq / Hkv;
            if (!block_row_order && QB < 32768 && Sq / 256 + 1 < 65536) {   // packing of (block-row, sub-tile) in one word
                int32_t* work = tile_off2 + (size_t)Hkv * (QB + 1);          // [2 * Hkv * QB]
                int32_t* partner = work + 2 * (size_t)Hkv * QB;              // [Hkv * QB]
                int32_t* hist = partner + (size_t)Hkv * QB;
                const int nb = Hkv * kVbBuckets;
                int32_t* order = hist + nb;
                constexpr int BMo = W * 32;
                const size_t chain_lds = vb_chain_lds(QB, KB);
                if (order_mode == 2 && chain_lds <= 64 * 1024 && QB <= 4096) {   // similarity order, consecutive workgroups on one XCD (variant 7)
                    hipLaunchKernelGGL(varblock_chain_kernel, dim3(Hkv), dim3(kVbChainThreads), chain_lds, st, block_map, k_sizes, toff,
                                       order, Hkv, QB, KB, group);
                } else {   // longest-first inside every kv head, ragged last tiles packed in pairs (order_mode 0) or not (1)
                    const size_t pair_lds = vb_pair_lds(QB, KB);
                    const bool pack = order_mode == 0 && pair_lds <= 64 * 1024 && QB <= 4095 && KB <= 1024;   // (bitmap row in registers; 12-bit index in the arg-max key)
                    if (pack) {   // bitmap rows once, then kVbPairRounds x (score, match); scratch: the bitmap area behind the order, and
                                  // the bucket array `work` (free until varblock_work_kernel runs) for the remainders and choices
                        const int WSp = vb_pair_ws(KB);
                        uint32_t* bits = (uint32_t*)(((uintptr_t)(order + 2 + 3 * ((size_t)Sq / 64 + QB) * Hq) + 15) & ~(uintptr_t)15);
                        int32_t* rem = work;
                        int32_t* best = work + (size_t)Hkv * QB;
                        const long long nw = (long long)Hkv * QB * WSp;
                        hipLaunchKernelGGL(varblock_bitmap_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, block_map, k_sizes,
                                           bits, Hkv, QB, KB, WSp);
                        for (int round = 0; round < kVbPairRounds; ++round) {
                            hipLaunchKernelGGL(varblock_pair_score_kernel, dim3((QB + kVbPairRows - 1) / kVbPairRows, Hkv),
                                               dim3(kVbPairThreads), pair_lds, st, bits, q_sizes, rem, best, QB, WSp, BMo, round);
                            hipLaunchKernelGGL(varblock_pair_match_kernel, dim3((Hkv * QB + 255) / 256), dim3(256), 0, st, q_sizes, best, rem,
                                               partner, Hkv * QB, QB, BMo, round);
                        }
                    }
                    if (hipMemsetAsync(hist, 0, (size_t)nb * sizeof(int32_t), st) != hipSuccess) return SVG_ERR_LAUNCH;
                    hipLaunchKernelGGL(varblock_work_kernel, dim3((Hkv * QB + 3) / 4), dim3(256), 0, st, block_map, q_sizes, k_sizes,
                                       pack ? partner : nullptr, work, hist, Hkv, QB, KB, group, BMo);
                    hipLaunchKernelGGL(varblock_scan_kernel, dim3(1), dim3(256), 0, st, hist, order, nb);
                    hipLaunchKernelGGL(varblock_scatter_kernel, dim3((Hkv * QB + 255) / 256), dim3(256), 0, st, q_sizes,
                                       pack ? partner : nullptr, work, hist, order, Hkv, QB, group, BMo);
                }
                p.order = order;
                if constexpr (NW == -9) {
                    if constexpr (D == 128) {
                        auto kern = varblock_attn_f8_kernel<T>;
                        const int lds = attn_f8_lds_bytes<128, kVbF8Waves>() + vb_policy_lds(p.kb_cap);
                        if (const int rc = configure_lds((const void*)kern, lds); rc != SVG_OK) return rc;
                        hipLaunchKernelGGL(kern, dim3(p.max_tiles * Hq), dim3(kVbF8Waves * 64), lds, st, p, *f8);
                        return launch_status();
                    }
                    return SVG_ERR_UNSUPPORTED;
                } else {
#ifdef SVG_ABLATIONS
                    if constexpr (D == 128 && std::is_same<T, __bf16>::value) {
                        if (trace)
                            return launch_attn(varblock_attn_pp2_trace_kernel<T, D>, p, dim3(p.max_tiles * Hq), 512,
                                               attn_pp2_lds_bytes<D>() + vb_policy_lds(p.kb_cap), st);
                    }
#endif
                    if (trace) return SVG_ERR_UNSUPPORTED;   // diagnostics builds only (-DSVG_ABLATIONS)
                    if constexpr (D == 128) {
                        if (body_m16)
                            return launch_attn(varblock_attn_m16_kernel<T>, p, dim3(p.max_tiles * Hq), 512,
                                               attn_m16_lds_bytes() + vb_policy_lds(p.kb_cap), st);
                    }
                    return launch_attn(varblock_attn_pp2_kernel<T, D>, p, dim3(p.max_tiles * Hq), 512,
                                       attn_pp2_lds_bytes<D>() + vb_policy_lds(p.kb_cap), st);
                }
            }
            if constexpr (NW == -9)
                return SVG_ERR_UNSUPPORTED;   // (the fp8 kernel takes the ordered 1-D launch only)
            else {
                if constexpr (D == 128) {
                    if (body_m16)
                        return launch_attn(varblock_attn_m16_kernel<T>, p, dim3(p.max_tiles, Hq), 512,
                                           attn_m16_lds_bytes() + vb_policy_lds(p.kb_cap), st);
                }
                return launch_attn(varblock_attn_pp2_kernel<T, D>, p, dim3(p.max_tiles, Hq), 512,
                                   attn_pp2_lds_bytes<D>() + vb_policy_lds(p.kb_cap), st);
            }
        } else
            return launch_attn(varblock_attn_kernel<T, D, W>, p, dim3(p.max_tiles, Hq), W * 64,
                               attn_lds_bytes<D, W>() + vb_policy_lds(p.kb_cap), st);
    };
    if constexpr (NW == 0) {
        int rc = SVG_OK;
        if (Sq >= kVbFull) rc = launch(std::integral_constant<int, 8>{}, 1, tile_off, Sq / kVbFull);
        if (rc != SVG_OK) return rc;
        return launch(std::integral_constant<int, 4>{}, 2, tile_off2, 2 * QB);
    } else if constexpr (NW == -9) {   // fp8 body
        return launch(std::integral_constant<int, kVbF8Waves>{}, 0, tile_off, Sq / (kVbF8Waves * 32) + QB);
    } else if constexpr (NW == -8) {   // two-phase ping-pong body, 256-row q tiles
        return launch(std::integral_constant<int, 8>{}, 0, tile_off, Sq / 256 + QB);
    } else {
        return launch(std::integral_constant<int, NW>{}, 0, tile_off, Sq / (NW * 32) + QB);
    }
}
}  // namespace svg

// svg_varblock_attention (abi_layout == nullptr: contiguous [H, S, D] tensors) and svg_varblock_attention_strided
static int varblock_entry(const void* q, const void* k, const void* v, void* o, int32_t Hq, int32_t Hkv, int32_t Sq, int32_t Skv, int32_t D,
                          int32_t dtype, float sm_scale, const uint8_t* block_map, const int32_t* q_sizes, const int32_t* k_sizes,
                          int32_t QB, int32_t KB, const int32_t* q_row_idx, const int32_t* kv_row_idx, void* workspace,
                          size_t workspace_bytes, int32_t variant, const svg_attn_layout_t* abi_layout, void* stream) {
    if (!q || !k || !v || !o || !block_map || !q_sizes || !k_sizes || !workspace) return SVG_ERR_BAD_ARG;
    if (Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 || Sq <= 0 || Skv <= 0 || QB <= 0 || KB <= 0) return SVG_ERR_BAD_ARG;
    if (KB > kVbMaxKB) return SVG_ERR_UNSUPPORTED;
    if ((int64_t)Skv * D * 2 >= (1ll << 32) || (int64_t)Sq * D * 2 >= (1ll << 32)) return SVG_ERR_UNSUPPORTED;
    if (workspace_bytes < svg_varblock_workspace_bytes(Hq, Hkv, QB, KB, Sq)) return SVG_ERR_WORKSPACE;
    AttnLayout lay_storage;
    const AttnLayout* lay = nullptr;
    if (abi_layout) {
        if (const int rc = layout_from_abi(abi_layout, Hq, Hkv, Sq, Skv, D, q, k, v, o, lay_storage); rc != SVG_OK) return rc;
        lay = &lay_storage;
    }
    hipStream_t st = (hipStream_t)stream;
    // variant 0: 4 waves, 128-row q tiles; 1: 8 waves, 256-row q tiles; 2: mixed (full 256-row tiles on 8 waves, rest on 4)
    // (6 = 3: the longest-first order is the default again — the similarity order, variant 7, raised the L2 hit rate from 31 % to 48 %
    //  and cut the L2 <-> fabric traffic by a quarter but not the kernel time, and its chain kernel costs 0.7 - 1.0 ms per call)
    // two-phase body (variant >= 3): on 16x16x32 MFMAs at head_dim 128 (attn_m16.h: 28.3 vs 29.1 ms at Wan 720p, profiles/r04d_ab_svg2_m16_first.txt);
    // 8 = 3 with that body named explicitly, 9 = 3 on the 32x32x16 body (A/B)
    const bool force_pp2 = (variant == 9);
    if (variant == 8 || variant == 9) variant = 3;
    const bool block_row_order = (variant == 4), trace = (variant == 5);
    const int order_mode = variant == 7 ? 2 : (variant == 6 ? 1 : 0);   // 0: longest-first + remainder packing, 1: longest-first, 2: similarity order
#define SVG_VB_ARGS q, k, v, o, Hq, Hkv, Sq, Skv, sm_scale, block_map, q_sizes, k_sizes, QB, KB, q_row_idx, kv_row_idx, workspace, block_row_order, trace, st, nullptr, order_mode, body_m16, lay
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
    if (variant < -1 || variant > 7) return SVG_ERR_BAD_ARG;
    // -1 (auto): 256-row q tiles with the two-phase ping-pong body once the average block-row is large enough to fill them
    // (Wan 720p, 252-row clusters: 40.4 ms; lock-step 8 waves 45.5, 4 waves 47.7, mixed 46.9), 128-row tiles otherwise
    if (variant == -1) variant = ((int64_t)Sq >= (int64_t)160 * QB) ? 3 : 0;
    const bool body_m16 = (variant >= 3 && D == 128 && !force_pp2);
    if (lay && !(variant >= 3 && !trace)) return SVG_ERR_UNSUPPORTED;   // strided tensors: the two-phase bodies only (see svg_attn_layout_t)
    if (dtype == SVG_DTYPE_BF16) {
        SVG_VB_DISPATCH(__bf16)
    } else if (dtype == SVG_DTYPE_F16) {
        SVG_VB_DISPATCH(_Float16)
    }
#undef SVG_VB_ARGS
#undef SVG_VB_DISPATCH
    return SVG_ERR_UNSUPPORTED;
}

extern "C" int svg_varblock_attention(const void* q, const void* k, const void* v, void* o, int32_t Hq, int32_t Hkv,
                                      int32_t Sq, int32_t Skv, int32_t D, int32_t dtype, float sm_scale,
                                      const uint8_t* block_map, const int32_t* q_sizes, const int32_t* k_sizes, int32_t QB,
                                      int32_t KB, const int32_t* q_row_idx, const int32_t* kv_row_idx, void* workspace,
                                      size_t workspace_bytes, int32_t variant, void* stream) {
    return varblock_entry(q, k, v, o, Hq, Hkv, Sq, Skv, D, dtype, sm_scale, block_map, q_sizes, k_sizes, QB, KB, q_row_idx, kv_row_idx,
                          workspace, workspace_bytes, variant, nullptr, stream);
}

extern "C" int svg_varblock_attention_strided(const void* q, const void* k, const void* v, void* o, int32_t Hq, int32_t Hkv,
                                              int32_t Sq, int32_t Skv, int32_t D, int32_t dtype, float sm_scale,
                                              const uint8_t* block_map, const int32_t* q_sizes, const int32_t* k_sizes, int32_t QB,
                                              int32_t KB, const int32_t* q_row_idx, const int32_t* kv_row_idx, void* workspace,
                                              size_t workspace_bytes, const svg_attn_layout_t* layout, void* stream) {
    if (!layout) return SVG_ERR_BAD_ARG;
    return varblock_entry(q, k, v, o, Hq, Hkv, Sq, Skv, D, dtype, sm_scale, block_map, q_sizes, k_sizes, QB, KB, q_row_idx, kv_row_idx,
                          workspace, workspace_bytes, -1, layout, stream);
}

extern "C" size_t svg_varblock_attention_fp8_workspace_bytes(int32_t Hq, int32_t Hkv, int32_t QB, int32_t KB, int32_t Sq, int32_t Skv,
                                                             int32_t D) {
    if (D != 128) return 0;
    const size_t plan = svg_varblock_workspace_bytes(Hq, Hkv, QB, KB, Sq);
    if (plan == 0 || Skv <= 0) return 0;
    return ((plan + 255) & ~(size_t)255) + f8g_ws_bytes(Hq, Hkv, Sq, Skv);
}

extern "C" int svg_varblock_attention_fp8(const void* q, const void* k, const void* v, void* o, int32_t Hq, int32_t Hkv, int32_t Sq,
                                          int32_t Skv, int32_t D, int32_t dtype, float sm_scale, const uint8_t* block_map,
                                          const int32_t* q_sizes, const int32_t* k_sizes, int32_t QB, int32_t KB,
                                          const int32_t* q_row_idx, const int32_t* kv_row_idx, void* workspace, size_t workspace_bytes,
                                          void* stream) {
    if (!q || !k || !v || !o || !block_map || !q_sizes || !k_sizes || !workspace) return SVG_ERR_BAD_ARG;
    if (Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 || Sq <= 0 || Skv <= 0 || QB <= 0 || KB <= 0) return SVG_ERR_BAD_ARG;
    if (D != 128 || KB > kVbMaxKB || QB >= 32768 || Sq / 256 + 1 >= 65536) return SVG_ERR_UNSUPPORTED;
    if ((int64_t)Skv * D >= (1ll << 31) || (int64_t)Sq * D * 2 >= (1ll << 32)) return SVG_ERR_UNSUPPORTED;
    if (workspace_bytes < svg_varblock_attention_fp8_workspace_bytes(Hq, Hkv, QB, KB, Sq, Skv, D)) return SVG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t plan = (svg_varblock_workspace_bytes(Hq, Hkv, QB, KB, Sq) + 255) & ~(size_t)255;
    F8GArgs fa;
    int rc = f8g_quantize(q, k, v, Hq, Hkv, Sq, Skv, dtype, sm_scale, (char*)workspace + plan, &fa, st);
    if (rc != SVG_OK) return rc;
    if (dtype == SVG_DTYPE_BF16)
        return run_varblock<__bf16, 128, -9>(q, k, v, o, Hq, Hkv, Sq, Skv, sm_scale, block_map, q_sizes, k_sizes, QB, KB, q_row_idx,
                                            kv_row_idx, workspace, false, false, st, &fa);
    if (dtype == SVG_DTYPE_F16)
        return run_varblock<_Float16, 128, -9>(q, k, v, o, Hq, Hkv, Sq, Skv, sm_scale, block_map, q_sizes, k_sizes, QB, KB, q_row_idx,
                                              kv_row_idx, workspace, false, false, st, &fa);
    return SVG_ERR_UNSUPPORTED;
}

