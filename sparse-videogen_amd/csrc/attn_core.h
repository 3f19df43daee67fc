// Flash-attention forward core for gfx950 shared by the SVG1 band kernel, the SVG2 variable-block kernel and the
// online profiler.  One workgroup = NW waves x 32 query rows; KV tiles of 64 keys are staged through LDS.
//
// Both GEMMs are computed "swapped" so that every lane owns ONE query row for the whole kernel
// (row max / row sum / rescale never cross lanes except one lane<->lane+32 exchange per tile):
//   S^T[key][q] = sum_d K[key][d] Q[q][d]     A = K tile (LDS, ds_read_b128, XOR-swizzled), B = Q (registers)
//   O^T[d][q]   = sum_key V[key][d] P[q][key] A = V^T (LDS, ds_read_b64_tr_b16 hardware transpose), B = P (registers)
// MFMA 32x32x16: a lane holds 8 consecutive-k values of row/col (lane & 31), k-group (lane >> 5); the f32x16
// accumulator holds column (lane & 31) and rows (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), r = 0..15.
// Hence the S^T accumulator of a lane *is* the P operand of the second GEMM (keys 16h + 8u + 4g + [0..3] in
// registers 8h + 4u + [0..3]) — no LDS round trip, no shuffles.
//
// LDS images per stage:  K: [64 keys][D] row-major, 16-B chunk c stored at c ^ swz(row) (conflict-free b128 reads)
//                        V: [D/32][64 keys][32 cols] sub-tiled so that 4 consecutive keys x 32 cols are 256
//                           contiguous bytes = one conflict-free ds_read_b64_tr_b16 pass.
// Staging is register-staged and software-pipelined: global loads for tile t+2 are issued right after the
// ds_writes of tile t+1, one barrier per tile, two LDS stages.
#pragma once
#include <type_traits>

#include "svg_common.h"

namespace svg {

constexpr int kBN = 64;  // keys per tile

// TILE_PARTIAL_FAST: lock-step body only, policies with kFastPartial — a cheaper element predicate that holds on tiles the policy
// recognises (allowed_fast(prm, ctx, key offset inside the tile without the lane part))
enum TileClass : int { TILE_SKIP = 0, TILE_FULL = 1, TILE_PARTIAL = 2, TILE_PARTIAL_FAST = 3 };

template <int D>
struct LdsLayout {
    static constexpr int kRowBytes = D * 2;
    static constexpr int kCPR = D / 8;                 // 16-B chunks per row
    static constexpr int kKBytes = kBN * kRowBytes;    // K tile bytes
    static constexpr int kVBytes = kBN * kRowBytes;    // V tile bytes
    static constexpr int kStageBytes = kKBytes + kVBytes;
    // byte offset of 16-B chunk c of key row `row` inside the K image
    static __device__ __forceinline__ int k_off(int row, int c) {
        const int sw = (D == 128) ? (row & 15) : ((row >> 1) & 7);
        return row * kRowBytes + ((c ^ sw) << 4);
    }
    // byte offset of 16-B chunk c (cols 8c..8c+7) of key row `row` inside the V image
    static __device__ __forceinline__ int v_off(int row, int c) { return (c >> 2) * (kBN * 64) + row * 64 + ((c & 3) << 4); }
};

// number of LDS stages: 3 when the policy asks for the skewed two-group schedule (8 waves, one workgroup per CU)
template <int NW, typename P>
constexpr int attn_stages() {
    return (P::kSkew && NW == 8) ? 3 : 2;
}

template <int D, int NW, int NS = 2, int RB = 1, int SUBS = 1>
constexpr int attn_lds_bytes() {
    // NS stages of SUBS tiles, or the epilogue staging of NW*32*RB rows with an 8-byte row pad, whichever is larger
    constexpr int stages = NS * SUBS * LdsLayout<D>::kStageBytes;
    constexpr int epi = NW * 32 * RB * (D * 2 + 8);
    return stages > epi ? stages : epi;
}

__device__ __forceinline__ i16x4 lds_read_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p));
}

// The attention kernel body.  Policy P supplies (all functions static __device__):
//   struct Params; struct Ctx (workgroup-uniform);
//   bool init(const Params&, Ctx&, char* policy_lds)            -> false: nothing to do (uniform exit)
//   const T* q_base/k_base/v_base(const Params&, const Ctx&); T* o_base(...)      head base pointers
//   int  q_logical(ctx, row_in_wg)  / int q_phys(prm, ctx, row_in_wg)            (-1 = row does not exist)
//   struct KvCursor; kv_cursor_init(prm, ctx, cur, row_in_tile); int kv_phys(prm, ctx, cur, t, row_in_tile)
//        (physical row >= 0; rows that do not exist return 0 and MUST be masked by classify/allowed)
//   int  tile_key0(ctx, t)                                       logical index of the first key of tile t
//   int  classify(prm, ctx, tile_key0, wave_row0)                wave-uniform TileClass
//   bool allowed(prm, ctx, q_logical, k_logical)                 element predicate for PARTIAL tiles
//   void row_intervals(prm, ctx, q_logical, a0, alen, b0, blen)  the same predicate as two key intervals of a row (two-phase body)
//   void notify(prm, ctx)                                        called by every wave of the two-phase body after its last store
//   float score_fixup(float raw)                                 (profiler: dtype rounding emulation)
//   epilogue: store(prm, ctx, ...) handled here through P::kPartialOut
template <typename T, int D, int NW, typename P>
__device__ __forceinline__ void attn_body(const typename P::Params& prm, char* smem, char* policy_lds) {
    using E = Elt<T>;
    using V8 = typename E::v8;
    using L = LdsLayout<D>;
    constexpr int RB = P::kRowBlocks;   // 32-row query blocks per wave (2 => 64 rows per wave: every K / V fragment read
                                        // from LDS feeds two MFMAs, halving LDS operand traffic per FLOP)
    constexpr int WR = 32 * RB;         // query rows per wave
    constexpr int NT = NW * 64;
    constexpr int KS = D / 16;          // k-steps of the S^T GEMM
    constexpr int DB = D / 32;          // 32-wide d blocks of O^T
    constexpr int SUBS = P::kSubTiles;  // 64-key tiles per LDS stage (2 => one barrier / one staging round per 128 keys)
    constexpr int NCH = (SUBS * kBN * L::kCPR) / NT;  // 16-B chunks per thread per tensor per stage
    constexpr int kStage = SUBS * L::kStageBytes;     // a stage is SUBS x [K image | V image]
    static_assert((SUBS * kBN * L::kCPR) % NT == 0, "stage chunks must divide evenly");

    typename P::Ctx ctx;  // workgroup-uniform, except fields a policy documents as per-wave / per-lane
    if (!P::init(prm, ctx, policy_lds)) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_id();
    const int g = lane >> 5;
    const int ql = lane & 31;

    const T* __restrict__ qb = P::q_base(prm, ctx);
    const T* __restrict__ kb = P::k_base(prm, ctx);
    const T* __restrict__ vb = P::v_base(prm, ctx);

    // ---- Q fragments (B operand of S^T): 8 consecutive d of this lane's query rows per k-step ----
    // (rows that do not exist read row 0 instead: a lane only ever feeds its own query rows, and such rows are
    //  never stored, so no predication is needed — predicated loads cost exec-masked blocks)
    int q_log[RB];
    V8 qf[RB][KS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int row_in_wg = wave * WR + rb * 32 + ql;
        const int q_phys = P::q_phys(prm, ctx, row_in_wg);
        q_log[rb] = P::q_logical(ctx, row_in_wg);
        const T* qrow = qb + (size_t)(q_phys >= 0 ? q_phys : 0) * D + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[rb][ks] = *(const V8*)(qrow + ks * 16);
    }

    // ---- staging bookkeeping: chunk i of this thread covers (row srow[i], 16-B chunk scol[i]) of the tile ----
    int srow[NCH], k_dst[NCH], v_dst[NCH], scol[NCH], ssub[NCH];
    typename P::KvCursor cur[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = tid + i * NT;
        const int r = id / L::kCPR;      // row inside the stage: sub-tile r / 64, row r % 64
        ssub[i] = r / kBN;
        srow[i] = r - ssub[i] * kBN;
        scol[i] = id - r * L::kCPR;
        k_dst[i] = ssub[i] * L::kStageBytes + L::k_off(srow[i], scol[i]);
        v_dst[i] = ssub[i] * L::kStageBytes + L::kKBytes + L::v_off(srow[i], scol[i]);
        P::kv_cursor_init(prm, ctx, cur[i], srow[i]);
    }
    u32x4 kreg[NCH], vreg[NCH];

    // Loads are issued unconditionally.  A tile row that does not exist (beyond the sequence / beyond the last
    // active key) reads row 0 instead: its scores are masked to -inf element-wise (such a tile is never FULL) and
    // p = 0 times the finite V values of row 0 adds nothing, so neither predication nor zero-fill is needed
    // (predicated loads would cost exec-masked blocks and vmcnt(0) fall-backs in hipcc's waitcnt pass).
    // Physical rows are resolved one tile ahead of the data loads (nphys): for the variable-block policy the
    // resolve is itself a global index load, and this keeps its latency off the critical path.
    int nphys[NCH];
    auto stage_resolve = [&](int st) {  // st = stage index; its tiles are st * SUBS + sub
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int t = st * SUBS + ssub[i];
            nphys[i] = (t < ctx.nT) ? P::kv_phys(prm, ctx, cur[i], t, srow[i]) : 0;
        }
    };
    auto stage_issue = [&](int t) {  // data loads of stage t (rows resolved earlier), then resolve stage t+1
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const size_t off = (size_t)nphys[i] * D + scol[i] * 8;
            kreg[i] = *(const u32x4*)(kb + off);
            vreg[i] = *(const u32x4*)(vb + off);
        }
        stage_resolve(t + 1);
    };
    auto stage_write = [&](int buf) {
        char* base = smem + buf * kStage;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            *(u32x4*)(base + k_dst[i]) = kreg[i];
            *(u32x4*)(base + v_dst[i]) = vreg[i];
        }
    };

    // ---- per-lane LDS read offsets ----
    // K A-fragment (block b, k-step ks): row 32b + ql, chunk 2ks + g
    // the swizzle is an XOR on the chunk index, so the k-step cannot be a plain byte offset: keep the XOR term
    const int ksw0 = (D == 128) ? (ql & 15) : ((ql >> 1) & 7);  // same for row ql and 32+ql (32 % 16 == 0, (32>>1)%8==0)
    // V^T tr-read base: rows 4g + (i>>2), col bytes (16*dhalf + 4*(i&3))*2, i = lane & 15, dhalf = (lane>>4)&1
    const int vi = lane & 15;
    const int v_lane_off = L::kKBytes + (4 * g + (vi >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (vi & 3)) * 2;

    float m_run[RB], l_run[RB];  // running max (log2 domain) / this lane's partial row sum, per owned query row
    f32x16 acc_o[RB][DB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        m_run[rb] = -INFINITY;
        l_run[rb] = 0.f;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[rb][db][r] = 0.f;
    }

    const float c_log2 = prm.scale_log2;
    const int nT64 = ctx.nT;                         // 64-key tiles
    const int nT = (nT64 + SUBS - 1) / SUBS;         // stages

    stage_resolve(0);
    if (nT > 0) stage_issue(0);
    // Pin the Q fragments: the empty asm makes them "defined here", so hipcc waits for the Q loads at this point
    // and knows at the loop header that they have landed.  Without it the waitcnt pass emits vmcnt(0) in front of
    // the first MFMA of EVERY iteration, serialising the tile t+1 prefetch behind the compute of tile t.
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[rb][ks]));
    if (nT > 0) {
        stage_write(0);
        if (nT > 1) stage_issue(1);
    }
    __syncthreads();

    // The two GEMM phases as lambdas so that the "skewed" schedule below can place them differently per wave group.
    V8 pf[RB][2][2];  // P^T operand of the current tile (kept across the barrier by the lagging wave group)
    auto qk_softmax = [&](const char* kbuf, int tk0, int cls) {
        // ---------------- S^T = K Q^T ----------------
        if constexpr (P::kSetPrio) __builtin_amdgcn_s_setprio(1);
        f32x16 s[RB][2];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[rb][b][r] = 0.f;
        if constexpr (P::kAbl == 10) __builtin_amdgcn_iglp_opt(0);
        if constexpr (P::kAbl == 11) __builtin_amdgcn_iglp_opt(1);
        if constexpr (P::kAbl == 6 || P::kAbl == 8 || P::kAbl == 9) {  // ablation: MFMAs without the K-fragment LDS reads
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) s[rb][b] = E::mfma(qf[rb][(ks + b) % KS], qf[rb][ks], s[rb][b]);
        } else if constexpr (P::kAbl != 3) {
            // K fragments through a register ring, kPF k-steps ahead of the MFMAs that consume them
            constexpr int kPF = P::kPrefetch;
            auto kfrag = [&](int b, int ks) -> V8 {
                return *(const V8*)(kbuf + (32 * b + ql) * L::kRowBytes + (((2 * ks + g) ^ ksw0) << 4));
            };
            V8 ring[kPF + 1][2];
#pragma unroll
            for (int i = 0; i < kPF; ++i) {
                ring[i][0] = kfrag(0, i);
                ring[i][1] = kfrag(1, i);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + kPF < KS) {
                    ring[(ks + kPF) % (kPF + 1)][0] = kfrag(0, ks + kPF);
                    ring[(ks + kPF) % (kPF + 1)][1] = kfrag(1, ks + kPF);
                }
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) s[rb][b] = E::mfma(ring[ks % (kPF + 1)][b], qf[rb][ks], s[rb][b]);
            }
        } else {  // ablation: no QK^T (keep the values opaque so that the softmax is not folded away)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) asm volatile("" : "+v"(s[rb][0]), "+v"(s[rb][1]));
        }
        if constexpr (P::kSetPrio) __builtin_amdgcn_s_setprio(0);
        if constexpr (P::kAbl == 1 || P::kAbl == 9) {  // ablation: no softmax VALU
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pf[rb][b][r >> 3][r & 7] = E::from_float(s[rb][b][r]);
            return;
        }
        // ---------------- mask + online softmax (lane-local rows) ----------------
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if constexpr (P::kFixup) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[rb][b][r] = P::score_fixup(prm, s[rb][b][r]);
            }
            if (cls == TILE_PARTIAL) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * g;
                        s[rb][b][r] = P::allowed(prm, ctx, q_log[rb], tk0 + key) ? s[rb][b][r] : -INFINITY;
                    }
            }
            if constexpr (P::kFastPartial) {
                if (cls == TILE_PARTIAL_FAST) {
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            s[rb][b][r] = P::allowed_fast(prm, ctx, 32 * b + (r & 3) + 8 * (r >> 2)) ? s[rb][b][r] : -INFINITY;
                }
            }
            float mx = s[rb][0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[rb][0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[rb][1][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[rb], mx * c_log2);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[rb] - m_use);
            m_run[rb] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[rb][b][r], c_log2, -m_use));
                    psum += p;
                    pf[rb][b][r >> 3][r & 7] = E::from_float(p);
                }
            l_run[rb] = l_run[rb] * alpha + psum;
            if (__any(alpha != 1.f)) {
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_o[rb][db][r] *= alpha;
            }
        }
    };
    auto pv = [&](const char* kbuf) {
        if constexpr (P::kAbl == 2) {  // ablation: no PV (keep P alive)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                asm volatile("" ::"v"(pf[rb][0][0]), "v"(pf[rb][0][1]), "v"(pf[rb][1][0]), "v"(pf[rb][1][1]));
            return;
        }
        // ---------------- O^T += V^T P^T ----------------
        if constexpr (P::kSetPrio) __builtin_amdgcn_s_setprio(1);
        if constexpr (P::kAbl == 10) __builtin_amdgcn_iglp_opt(0);
        if constexpr (P::kAbl == 11) __builtin_amdgcn_iglp_opt(1);
        const char* vbase = kbuf + v_lane_off;
        constexpr int NPV = DB * 4;  // MFMA steps: idx = db * 4 + b * 2 + h
        if constexpr (P::kAbl == 7 || P::kAbl == 8 || P::kAbl == 9) {  // ablation: MFMAs without the V transpose reads
#pragma unroll
            for (int idx = 0; idx < NPV; ++idx)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc_o[rb][idx >> 2] = E::mfma(qf[rb][idx % KS], pf[rb][(idx >> 1) & 1][idx & 1], acc_o[rb][idx >> 2]);
        } else {
            constexpr int kPF = P::kPrefetch;
            auto vfrag = [&](int idx) -> V8 {
                const int db = idx >> 2, kb0 = 32 * ((idx >> 1) & 1) + 16 * (idx & 1);
                const i16x4 lo = lds_read_tr16(vbase + db * (kBN * 64) + kb0 * 64);
                const i16x4 hi = lds_read_tr16(vbase + db * (kBN * 64) + (kb0 + 8) * 64);
                i16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                return __builtin_bit_cast(V8, both);
            };
            V8 ring[kPF + 1];
#pragma unroll
            for (int i = 0; i < kPF; ++i) ring[i] = vfrag(i);
#pragma unroll
            for (int idx = 0; idx < NPV; ++idx) {
                if (idx + kPF < NPV) ring[(idx + kPF) % (kPF + 1)] = vfrag(idx + kPF);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc_o[rb][idx >> 2] = E::mfma(ring[idx % (kPF + 1)], pf[rb][(idx >> 1) & 1][idx & 1], acc_o[rb][idx >> 2]);
            }
        }
        if constexpr (P::kSetPrio) __builtin_amdgcn_s_setprio(0);
    };

    // Schedule.  Lock-step (kStages == 2): every wave runs QK -> softmax -> PV per tile; the two waves that share a
    // SIMD then fight for the matrix pipe in the GEMM phases and for the VALU in the softmax phase (time ~ 4M + 2V).
    // Skewed (kStages == 3): the second half of the waves (one per SIMD) runs PV one tile late, at the top of the
    // next iteration — while group A is in its softmax (VALU) group B is in QK (MFMA) and vice versa (time ~ 4M).
    // The third LDS stage keeps V(t-1) alive for the lagging group while tile t+1 is being written.
    constexpr int NS = attn_stages<NW, P>();
    const bool lag = (NS == 3) && (wave >= NW / 2);
    bool pending = false;
    int buf = 0, pend_buf = 0;
    for (int t = 0; t < nT; ++t) {
        const char* sbuf = smem + buf * kStage;
        if (NS == 3 && lag && pending) {
            pv(smem + pend_buf * kStage);
            pending = false;
        }
#pragma unroll
        for (int sub = 0; sub < SUBS; ++sub) {
            const int t64 = t * SUBS + sub;
            if (SUBS > 1 && t64 >= nT64) break;
            const char* kbuf = sbuf + sub * L::kStageBytes;
            const int tk0 = P::tile_key0(ctx, t64);
            const int cls = P::classify(prm, ctx, tk0, wave * WR);
            if (cls != TILE_SKIP) {
                qk_softmax(kbuf, tk0, cls);
                if (NS == 3 && lag) {
                    pending = true;
                    pend_buf = buf;
                } else {
                    pv(kbuf);
                }
            }
        }
        const int nbuf = (buf + 1 == NS) ? 0 : buf + 1;
        if constexpr (P::kAbl != 4 && P::kAbl != 8 && P::kAbl != 9) {
            if (t + 1 < nT) stage_write(nbuf);
            if (t + 2 < nT) stage_issue(t + 2);
        }
        if constexpr (P::kAbl != 5 && P::kAbl != 8 && P::kAbl != 9) __syncthreads();
        buf = nbuf;
    }
    if (NS == 3) {
        if (lag && pending) pv(smem + pend_buf * kStage);
        __syncthreads();  // the epilogue below reuses the stage buffers
    }

    // ---------------- epilogue ----------------
    float l_tot[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) l_tot[rb] = l_run[rb] + __shfl_xor(l_run[rb], 32);
    if constexpr (P::kPartialOut) {
        // un-normalised fp32 partial: O^T accumulators, running max and row sum (profiler split-KV)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            P::store_partial(prm, ctx, wave * WR + rb * 32 + ql, g, acc_o[rb], m_run[rb], l_tot[rb]);
        return;
    } else {
    // transpose through LDS (all waves are past the last barrier, stage buffers are free): row stride D*2+8 bytes
    constexpr int kEpiStride = D * 2 + 8;
    char* erow = smem + (size_t)(wave * WR) * kEpiStride;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const float inv = l_tot[rb] > 0.f ? 1.f / l_tot[rb] : 0.f;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                typename E::v4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = E::from_float(acc_o[rb][db][rq * 4 + j] * inv);
                const int d0 = 32 * db + 8 * rq + 4 * g;
                *(typename E::v4*)(erow + (rb * 32 + ql) * kEpiStride + d0 * 2) = o4;
            }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // own-wave LDS writes visible to own-wave reads below
    __builtin_amdgcn_wave_barrier();
    T* __restrict__ ob = P::o_base(prm, ctx);
    constexpr int kLanesPerRow = D * 2 / 8;       // 8 B per lane
    constexpr int kRowsPerPass = 64 / kLanesPerRow;
    constexpr int kPasses = WR / kRowsPerPass;
    const int sub = lane / kLanesPerRow;
    const int colb = (lane - sub * kLanesPerRow) * 8;
    int ephys[kPasses];  // resolve all output rows first: keeps the (possibly global) index loads in flight together
#pragma unroll
    for (int i = 0; i < kPasses; ++i) ephys[i] = P::q_phys(prm, ctx, wave * WR + i * kRowsPerPass + sub);
#pragma unroll
    for (int i = 0; i < kPasses; ++i) {
        const int rr = i * kRowsPerPass + sub;
        const u32x2 val = *(const u32x2*)(erow + rr * kEpiStride + colb);
        if (ephys[i] >= 0) *(u32x2*)((char*)(ob + (size_t)ephys[i] * D) + colb) = val;
    }
    }
}


// =====================================================================================================================
// Helpers of the LDS-DMA bodies (attn_body_pp2 below, attn_body_w4 in attn_w4.h).
// K / V tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write).  The DMAs are inline asm: hipcc's
// waitcnt pass would drain a builtin LDS-DMA (vmcnt(0)) in front of the next ds_read; compiler-inserted vmcnt(N) waits stay
// safe (loads retire in order, extra outstanding requests only make vmcnt(N) stricter).
// Images (16-B slots): both are sub-tiled [D/32][64 keys][4 slots] so that one DMA piece = 16 keys x 64 B = 1 KiB lands
// lane-linear; K stores chunk c at slot (c & 3) ^ ((key >> 2) & 3) (conflict-free ds_read_b128 for the 32x32x16 A operand),
// V stores it at slot c & 3 (ds_read_b64_tr_b16 wants 4 keys x 32 columns contiguous).  The swizzle is applied to the
// per-lane SOURCE address.
// (The earlier schedules of this core — intra-wave pipelining on 8 waves, the four-cluster ping-pong — are in the history of
//  this file; profiles/r01_ablation.md has their measurements.)
// =====================================================================================================================
// v_max3_f32 / v_max_f32 without the input canonicalisation hipcc attaches to fmaxf()
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// pp_barrier for the waves whose (wave-uniform) `flag` is non-zero; the others fall through.  The branch lives INSIDE the asm
// block: no control-flow merge at IR level (a C-level `if` around a barrier in the tile loop costs a second register set)
__device__ __forceinline__ void pp_barrier_if(int flag) {
    __builtin_amdgcn_sched_barrier(0);
    const int f = __builtin_amdgcn_readfirstlane(flag);
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n1:" ::"s"(f) : "memory", "scc");
    __builtin_amdgcn_sched_barrier(0);
}
// one LDS-DMA piece: lane l's 16 bytes at (base + voff_l) land at LDS byte address lds + 16 * l (probe: tools/probe_dma.hip)
__device__ __forceinline__ void lds_dma16(unsigned lds, unsigned voff, const void* base) {
    // (s_nop: an SALU write of M0 needs one wait state before an LDS-DMA reads it; hipcc does not look inside inline asm)
    const unsigned lds_u = __builtin_amdgcn_readfirstlane(lds);   // wave-uniform by construction; make it so for the compiler
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_u), "v"(voff), "s"(base) : "memory");
}


// cycle trace of the two-phase body (diagnostics, -DSVG_ABLATIONS builds): per wave sums of s_memtime ticks per phase of
// workgroup blockIdx.x == kPpTraceBlock (static: one copy per translation unit, read with hipMemcpyFromSymbol in attention.hip)
static __device__ unsigned long long g_pp_trace[8 * 8 + 8 + 8 * 4];
constexpr int kPpTraceBlock = 1000;
// Launch timeline of the traced two-phase kernel: per workgroup [s_memtime at entry, at loop start, at loop end, at exit, HW_ID, XCC_ID]
// (wave 0), read with svg_debug_wg_trace; tools/wg_timeline.py turns it into per-CU occupancy and launch gaps.
constexpr int kWgTraceMax = 16384;
static __device__ unsigned long long g_wg_trace[kWgTraceMax * 6];


// =====================================================================================================================
// attn_body_pp2 — two-phase ping-pong: per tile every wave runs ONE matrix phase and ONE vector phase, and the two waves
// of a SIMD are always in opposite phases (waves 4..7 run one phase behind waves 0..3; two barriers per tile).
//     M(t+1)  O^T += V(t)^T P(t)^T  (16 MFMAs), S(t+1)^T = K(t+1) Q^T (16 MFMAs); operands stream from LDS just ahead of the
//             MFMAs that consume them
//     N(t+1)  mask, all 64 probabilities of tile t+1 against the row's reference, the check of their sum and — rarely — the exact
//             path (row maximum, new reference, rescale of O and l, probabilities again): kMaxFree, the shipped softmax;
//             LDS-DMA requests; DMA wait.  (The earlier softmax — row maximum of every tile with a deferred rescale, probabilities
//             of keys 16..63 in the shadow of the PV MFMAs — is kept behind SVG_PP2_MAXFREE=0 and for the timing ablations.)
// In M the wave has the matrix pipe to itself (its partner is in N and issues no MFMA), in N it has the VALU to itself.
// profiles/r01_ablation.md: in the lock-step body both waves of a SIMD sit in the same phase and the phases add up; the
// four-cluster attn_body_pp separates them but pays four barriers per tile and serialises the LDS operand reads.
//     slot         2t      2t+1    2t+2     2t+3
//     waves 0-3    M(t)    N(t)    M(t+1)   N(t+1)          M(t) reads K(t) and V(t-1)
//     waves 4-7    N(t-1)  M(t)    N(t)     M(t+1)
// LDS: four stages; tile w is in use during slots 2w .. 2w+3 (K in 2w, 2w+1; V in 2w+2, 2w+3).  Its stage becomes free at
// slot 2w-4: the leading waves request it in N(w-2) (slot 2w-3), the lagging waves in N(w-3) (slot 2w-4); every wave waits,
// at the end of a vector phase, for the pieces it requested in the previous one.
// =====================================================================================================================
template <int D>
constexpr int attn_pp2_lds_bytes() {
    constexpr int stages = 4 * 2 * kBN * D * 2;
    constexpr int epi = 8 * 32 * (D * 2 + 8);
    return stages > epi ? stages : epi;
}

template <typename T, int D, typename P, bool TRACE = false, int ABL = 0, bool PRE = false, bool LEAN = false>
__device__ __forceinline__ void attn_body_pp2(const typename P::Params& prm, char* smem, char* policy_lds) {
    using E = Elt<T>;
    using V8 = typename E::v8;
    constexpr int NW = 8;
    constexpr int KS = D / 16;
    constexpr int DB = D / 32;
    constexpr bool kDma = true;             // true: LDS-DMA staging (4 stages); false: register staging (3 stages), measured 15 % slower
#ifndef SVG_PP2_MAXFREE
#define SVG_PP2_MAXFREE 1
#endif
    // Softmax without a running maximum (the scheme of attn_body_w4 / attn_body_f8): probabilities are taken relative to a per-row
    // reference that only changes on the exact path; the common path checks that a lane's 32 probabilities of the tile sum to
    // <= 2^11 (one compare instead of the 16-instruction maximum chain + lane exchange) and otherwise takes the exact path — row
    // maximum, new reference, O and l rescaled, probabilities recomputed — BEFORE any PV MFMA has consumed them, which is why all
    // four 16-key steps are computed in the vector phase then (kShadow = 0).
    // (ABL == 8 is not an ablation: the frozen reference schedule of svg_band_attention variant 6 — correct results)
    constexpr bool kMaxFree = (SVG_PP2_MAXFREE != 0 && ABL == 0) || PRE;
    // PRE: q arrives multiplied by sm_scale * log2(e) (svg_band_attention_prescaled; the prologue folds the factor into its single
    // rounding of q), and the S^T accumulators start at minus the row's reference instead of zero — a 16-register tuple that only
    // the exact path rewrites — so what the MFMAs deliver IS the exponent argument: no scale-and-shift FMA per score (the scheme
    // of the fp8 bodies, attn_f8.h).
    static_assert(!PRE || (SVG_PP2_MAXFREE != 0 && ABL == 0), "pre-scaled q: max-free softmax only");
#ifndef SVG_PP2_CARRY
#define SVG_PP2_CARRY 8
#endif
    // V operands of the first kCarry MFMAs of a matrix phase are read in the tail of the PREVIOUS matrix phase (tile t + 1 has been
    // in LDS since the barrier in front of M(t)) and carried through the vector phase in registers: the phase opens with MFMAs
    // instead of with an LDS round trip.
    // LEAN: the register diet of the instance that runs FOUR waves per SIMD (two workgroups per CU; band_attn_pp2_kernel at head_dim 64,
    // attention.hip): no carried operands, operand ring 4 instead of 8 fragments ahead — with a second workgroup on the CU the latency the
    // deep ring hides is hidden by the other workgroup's waves, and 128 registers hold the body without a spill.
    constexpr int kCarry = (ABL == 0 && !LEAN) ? SVG_PP2_CARRY : 0;
#ifndef SVG_PP2_ONEBAR
#define SVG_PP2_ONEBAR -1   // -1: as the policy says (P::kOneBarrier); 0 / 1: force (A/B builds)
#endif
    // ONE workgroup barrier per tile instead of two.  Of the two barriers of a tile only the one in front of the leading waves'
    // matrix phase (= in front of the lagging waves' vector phase) carries data: it publishes the DMA pieces the leading waves have
    // just waited for (tile t + 1, read by their M(t) right behind it) and orders every request of a stage behind the last read of
    // its previous tenant.  The other one (in front of the leading waves' vector phase) only re-aligns the phases; without it a wave
    // runs N and M back to back and the two waves of a SIMD meet once per tile: a slot no longer costs max(M, N) + barrier but the
    // tile costs M + N + one barrier.  Leading waves: [N(t), barrier, M(t)]; lagging waves: [barrier, N(t), M(t)].
    // Measured (round 3, same box): band kernel 32.06 vs 32.13 ms — nothing: at the power limit the clock takes back what a schedule
    // saves without saving energy —, variable-block kernel (whose vector phase, with the gather, is the longer one) 28.07 vs 28.62 ms:
    // on for the variable-block policy, off for the band policy.
    constexpr bool kOneBar = (SVG_PP2_ONEBAR < 0 ? P::kOneBarrier : SVG_PP2_ONEBAR != 0) && ABL == 0;
    constexpr int kShadow = kMaxFree ? 0 : (D == 64) ? 2 : P::kShadow128;   // 16-key probability steps computed in the shadow of the PV MFMAs (0..3); the rest in the vector phase
    constexpr int NS = kDma ? 4 : 3;
    constexpr int kImg = kBN * D * 2;       // bytes of a K or V image
    constexpr int kStage = 2 * kImg;
    constexpr int NP = DB / 2;              // DMA pieces per wave per tensor per tile
#ifndef SVG_PP2_PRIO
#define SVG_PP2_PRIO 1
#endif
    // 1: priority 1 in the matrix phase (shipped); 0: none (+1.1 %); 2: priority 1 in the vector phase (+0.4 %) — max-free body, same box
    constexpr bool kPrioM = SVG_PP2_PRIO == 1;
    constexpr bool kPrioV = SVG_PP2_PRIO == 2;
    constexpr bool kPrioStatic = false;  // true: static priority 1 for the lagging half instead of a flip around every matrix phase — measured 4 % slower (38.6 vs 37.0 ms)
    constexpr float kDefer = 8.f;
    static_assert(D == 64 || D == 128, "head dim");
    static_assert(P::kRowBlocks == 1 && P::kSubTiles == 1, "ping-pong body: 32 rows per wave, one tile per stage");

    unsigned long long wg_t0 = 0;
    if constexpr (TRACE) wg_t0 = __builtin_amdgcn_s_memtime();
    typename P::Ctx ctx;
    if (!P::init(prm, ctx, policy_lds)) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_id();
    const int g = lane >> 5;
    const int ql = lane & 31;
    const int row_in_wg = wave * 32 + ql;
    const bool lagging = wave >= NW / 2;
    const int nT = ctx.nT;

    const T* __restrict__ qb = P::q_base(prm, ctx);
    const T* __restrict__ kb = P::k_base(prm, ctx);
    const T* __restrict__ vb = P::v_base(prm, ctx);
    const unsigned lds0 = (unsigned)(size_t)smem;

    // ---- DMA bookkeeping (same images as attn_body_pp).  A wave's NP pieces per tensor are the d-blocks dma_db0 .. + NP - 1 of
    //      ONE 16-key group: every lane resolves a single key row per tile (one cursor, one index load) ----
    const int dma_kg = wave / 2;
    const int dma_db0 = (wave % 2) * NP;
    const int krow = 16 * dma_kg + (lane >> 2);
    typename P::KvCursor cur;
    P::kv_cursor_init(prm, ctx, cur, krow);
    const unsigned col_v = (unsigned)(dma_db0 * 64 + (lane & 3) * 16);
    const unsigned k_xor = (unsigned)(((lane >> 4) & 3) << 4);
    const unsigned col_k = col_v ^ k_xor;
    const unsigned k_rsb = (unsigned)P::k_rs(prm) * 2u, v_rsb = (unsigned)P::v_rs(prm) * 2u;
    const unsigned lds_piece = lds0 + (unsigned)(dma_db0 * (kBN * 64) + dma_kg * 1024);
    // Physical rows are resolved one vector phase before they are requested (nnext -> nphys): for the variable-block policy
    // the resolve is itself a global index load, and this keeps its latency off the critical path.
    int nphys = 0, nnext = 0;
    // (`guard` = false in the steady-state loop, where every tile index touched by a phase is known to be < nT: the "is there
    //  such a tile" tests and their scalar mask bookkeeping cost ~20 instructions per tile, each an issue slot of the wave)
    auto resolve = [&](int t, auto guard_c) {  // row of tile t into nnext (tiles are resolved in increasing order: the cursor only moves forward)
        if constexpr (decltype(guard_c)::value) nnext = (t < nT) ? P::kv_phys(prm, ctx, cur, t, krow) : 0;
        else nnext = P::kv_phys(prm, ctx, cur, t, krow);
    };
    constexpr std::true_type kGuarded{};
    auto take = [&]() { nphys = nnext; };
    // One piece = the K and the V request of d-block dma_db0 + j.  (One asm block per pair: the swizzle XOR of the K source
    // address doubles as the wait state M0 needs after an SALU write and the V request reuses M0 + kImg: 6 issue slots per
    // tile less than two independent lds_dma16 calls.)
    auto dma_piece = [&](int t, auto j_c) {
        constexpr int j = decltype(j_c)::value;
        static_assert(2 * D >= 256 || NP == 1, "row offset | column offset");
        const unsigned st = __builtin_amdgcn_readfirstlane(lds_piece + (unsigned)((t % NS) * kStage) + j * (kBN * 64));
        // byte offset of the key row inside its head = row * (row stride in bytes, a kernel argument: svg_attn_layout_t — 2 D for contiguous
        // heads) + the lane's 16-byte column (K: swizzled): one v_mad_u32_u24 per tensor (rows, strides < 2^24, products < 2^32: layout_from_abi)
        const unsigned ko = __umul24((unsigned)nphys, k_rsb) + col_k;
        const unsigned vo = __umul24((unsigned)nphys, v_rsb) + col_v;
        // (the d-block offset goes into the scalar base, loop-invariant — an instruction offset would also move the LDS address)
        const char* const kbp = (const char*)kb + j * 64;
        const char* const vbp = (const char*)vb + j * 64;
        asm volatile("s_mov_b32 m0, %0\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\t"
                     "s_add_u32 m0, m0, %5\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, %4"
                     :
                     : "s"(st), "v"(ko), "v"(vo), "s"(kbp), "s"(vbp), "n"(kImg)
                     : "memory", "scc");
    };
    auto dma_issue = [&](int t) {  // request this wave's pieces of tile t (t < nT, rows in nphys) into stage t % NS
        dma_piece(t, std::integral_constant<int, 0>{});
        if constexpr (NP > 1) dma_piece(t, std::integral_constant<int, 1>{});
    };
    // Register staging (kDma = false, kept for comparison): loads for tile w are issued in N(w - dist) and written in the
    // following vector phase: slot 2w-1 (leading) / 2w-2 (lagging), after the last read of the stage's previous tenant
    // (slot 2(w-3)+3) and before the first read of tile w (slot 2w).  Measured 51.0 ms vs 44.7 ms with LDS-DMA: the
    // ds_write_b128 traffic of the vector phase collides with the operand streaming of the partner's matrix phase.
    u32x4 kreg[NP], vreg[NP];
    auto stage_load = [&](int t) {
        resolve(t, kGuarded);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            kreg[j] = *(const u32x4*)((const char*)kb + (size_t)nnext * k_rsb + (col_k + (unsigned)(j * 64)));
            vreg[j] = *(const u32x4*)((const char*)vb + (size_t)nnext * v_rsb + (col_v + (unsigned)(j * 64)));
        }
    };
    char* const piece_ptr = smem + dma_db0 * (kBN * 64) + dma_kg * 1024 + lane * 16;
    auto stage_store = [&](int t) {
        char* st = piece_ptr + (t % NS) * kStage;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            *(u32x4*)(st + j * (kBN * 64)) = kreg[j];
            *(u32x4*)(st + j * (kBN * 64) + kImg) = vreg[j];
        }
    };
    const int dist = kDma ? (lagging ? 3 : 2) : (lagging ? 3 : 2);   // tile u + dist is requested in N(u)
    if constexpr (kDma) {
        // (ablation 6 — no DMA inside the loop — fills all four stages here so that the loop computes on finite stale data)
        const int npro = (ABL == 6) ? NS : dist;
        for (int t = 0; t < npro; ++t) {
            resolve(t, kGuarded);
            take();
            if (t < nT) dma_issue(t);
        }
        resolve(npro, kGuarded);   // requested in N(0)
    } else {
        // tiles 0 .. dist-2 go to LDS here, tile dist-1 stays in the staging registers (written in N(0))
        if (nT > 0) { stage_load(0); stage_store(0); }
        if (nT > 1) stage_load(1);
        if (lagging) {
            if (nT > 1) stage_store(1);
            if (nT > 2) stage_load(2);
        }
    }

    const int q_phys = P::q_phys(prm, ctx, row_in_wg);
    const int q_log = P::q_logical(ctx, row_in_wg);
    V8 qf[KS];
    {
        const T* qrow = qb + (size_t)(q_phys >= 0 ? q_phys : 0) * P::q_rs(prm) + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const V8*)(qrow + ks * 16);
    }

    const int k_lane0 = ql * 64 + (((g) ^ ((ql >> 2) & 3)) << 4);   // even k-steps
    const int k_lane1 = k_lane0 ^ 32;                               // odd k-steps
    const int vi = lane & 15;
    const int v_lane_off = kImg + (4 * g + (vi >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (vi & 3)) * 2;

    int m_a0 = 0, m_b0 = 0;
    unsigned m_alen = 0, m_blen = 0;
    if constexpr (P::kIntervalMask) P::row_intervals(prm, ctx, q_log, m_a0, m_alen, m_b0, m_blen);

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 acc_o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;
    const float c_log2 = prm.scale_log2;

#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pp_barrier();
    if (!kOneBar && lagging) pp_barrier();  // waves 4..7 run one phase behind

    // A wave without query rows (ragged q tiles of the variable-block policy, the last q tile of a sequence) keeps the barrier
    // and staging protocol but computes nothing: its partner then has the SIMD to itself.  (A separate loop, not a branch
    // inside the tile loop — see the note on control-flow merges below.)
    const bool idle = !P::wave_active(ctx, wave * 32);

    unsigned tr_acc[4] = {0, 0, 0, 0};
    unsigned long long tr_last = 0, tr_first = 0;
    if constexpr (TRACE) tr_first = tr_last = __builtin_amdgcn_s_memtime();
    auto tick = [&](auto slot_c) {
        if constexpr (TRACE) {
            constexpr int slot = decltype(slot_c)::value;
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tr_acc[slot] += (unsigned)(now - tr_last);
            tr_last = now;
        }
    };

    f32x16 neg_ref = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // PRE: minus the row's reference
    f32x16 sc[2];          // scores: S(t) until the PV steps of the matrix phase have consumed it, then S(t+1) accumulates here
    V8 pf[2][2];           // probabilities: [32-key block][16-key half]
    float m_use = 0.f, psum = 0.f;
    float psum_thr = -1.f;        // (wave-uniform) max-free softmax: 2048 once every row of the wave has a finite reference; until
                                  // then no sum passes the check and every tile takes the exact path

    auto kfrag = [&](const char* st, int b, int ks) -> V8 {
        return *(const V8*)(st + ((ks & 1) ? k_lane1 : k_lane0) + (ks >> 1) * (kBN * 64) + b * (32 * 64));
    };
    auto vfrag = [&](const char* st, int kk, int db) -> V8 {
        const char* vbase = st + v_lane_off;
        const i16x4 lo = lds_read_tr16(vbase + db * (kBN * 64) + (16 * kk) * 64);
        const i16x4 hi = lds_read_tr16(vbase + db * (kBN * 64) + (16 * kk + 8) * 64);
        i16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(V8, both);
    };
    // probabilities of keys 16 kk + [lo, hi) of the tile in sc
    // (scalar f32 on purpose: the packed forms v_pk_fma_f32 / v_pk_add_f32 measured 48.2 ms vs 42.3 ms here)
    float pre_shift = 0.f;   // PRE, exact path only: old reference minus new reference (the scores in sc are relative to the old one)
    auto probs_impl = [&](int kk, int lo, int hi, auto shifted_c) {
        constexpr bool shifted = decltype(shifted_c)::value;
#pragma unroll
        for (int r = lo; r < hi; ++r) {
            if constexpr (ABL == 4) {
                pf[kk >> 1][kk & 1][r] = E::from_float(sc[kk >> 1][8 * (kk & 1) + r]);
            } else {
                float p;
                if constexpr (PRE && shifted) p = __builtin_amdgcn_exp2f(sc[kk >> 1][8 * (kk & 1) + r] + pre_shift);
                else if constexpr (PRE) p = __builtin_amdgcn_exp2f(sc[kk >> 1][8 * (kk & 1) + r]);   // the MFMAs delivered the exponent argument
                else p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kk >> 1][8 * (kk & 1) + r], c_log2, -m_use));
                psum += p;
                pf[kk >> 1][kk & 1][r] = E::from_float(p);
            }
        }
    };
    auto probs = [&](int kk, int lo, int hi) { probs_impl(kk, lo, hi, std::false_type{}); };
    // staging half of a vector phase: resolve the rows of tile t+dist+1 (index loads first: they are older than the DMA
    // requests below, so the counted wait at the end retires them too), request tile t+dist, wait for tile t+dist-1
    auto stage_resolve_next = [&](int t, auto guard_c) {
        if constexpr (kDma) {
            take();
            resolve(t + dist + 1, guard_c);
        }
    };
    auto stage_request = [&](int t) {
        const bool more = t + dist < nT;
        if constexpr (kDma) {
            if (ABL != 6 && more) dma_issue(t + dist);
            // pieces requested in the previous vector phase have to be in LDS; the ones just requested may stay in flight
            if (ABL != 6 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (ABL != 6) {
            if (t + dist - 1 < nT) stage_store(t + dist - 1);   // loaded in the previous vector phase (or the prologue)
            if (more) stage_load(t + dist);
        }
    };
    // vector phase of tile t on sc: mask, maximum, (rare) rescale, probabilities of keys 0..15, DMA requests, DMA wait
    auto vector_phase = [&](int t, auto guard_c) {
        constexpr bool guard = decltype(guard_c)::value;
        stage_resolve_next(t, guard_c);
        constexpr bool kSpread = kDma && ABL != 6;   // requests spread over the phase instead of back to back at its end
        const bool more = !guard || t + dist < nT;
        if (kSpread && more) dma_piece(t + dist, std::integral_constant<int, 0>{});
        const int tk0 = P::tile_key0(ctx, t);
        const int cls = P::classify(prm, ctx, tk0, wave * 32);
        if constexpr (P::kFixup) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[b][r] = P::score_fixup(prm, sc[b][r]);
        }
        if (ABL != 7 && cls != TILE_FULL) {
            const bool part = (cls == TILE_PARTIAL);  // a tile this wave does not need at all is processed fully masked
            if constexpr (P::kIntervalMask) {
                // the row's allowed keys are [m_a0, +m_alen) u [m_b0, +m_blen) (per lane, loop-invariant)
                int ka = tk0 + 4 * g - m_a0, kb_ = tk0 + 4 * g - m_b0;
                asm volatile("" : "+v"(ka), "+v"(kb_));   // opaque: keeps LICM from hoisting 64 per-element terms out of the loop
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * b + (r & 3) + 8 * (r >> 2);
                        const bool ok = ((unsigned)(ka + key) < m_alen) | ((unsigned)(kb_ + key) < m_blen);
                        sc[b][r] = (part & ok) ? sc[b][r] : -INFINITY;
                    }
            } else {   // general element predicate (profiler masks)
                int qv = q_log, kv0 = tk0 + 4 * g;
                asm volatile("" : "+v"(qv), "+v"(kv0));
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * b + (r & 3) + 8 * (r >> 2);
                        sc[b][r] = (part & P::allowed(prm, ctx, qv, kv0 + key)) ? sc[b][r] : -INFINITY;
                    }
            }
        }
        if constexpr (kMaxFree) {
            // (m_use persists: the reference the accumulators are scaled to; 0 until a row has seen a finite score)
            if constexpr (kSpread && NP > 1) {
                if (more) dma_piece(t + dist, std::integral_constant<int, 1>{});
            }
            psum = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                probs(kk, 0, 8);
                asm volatile("" : "+v"(pf[kk >> 1][kk & 1]), "+v"(psum));   // stays in this phase
            }
            // (a wave takes the exact path until every one of its rows has a reference taken from a finite score — a row maximum
            //  instead of the pseudo-reference 0, under which scores below -126 in log2 units would underflow to nothing)
            float mx = sc[0][0];
            const bool exact = !__all(psum <= psum_thr);
            if (exact) {      // exact path (rare; also a non-finite sum)
#pragma unroll
                for (int e = 1; e < 31; e += 2) mx = vmax3(mx, sc[e >> 4][e & 15], sc[(e + 1) >> 4][(e + 1) & 15]);
                mx = vmax2(mx, sc[1][15]);
                const unsigned u = __builtin_bit_cast(unsigned, mx);
                const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // the other half of the row: lane ^ 32
                mx = vmax2(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
                const float m_prev = m_use;
                if constexpr (PRE) mx += m_prev;   // scores are relative to the reference they were accumulated under
                else mx *= c_log2;
                const float m_new = fmaxf(m_run, mx);
                m_use = (m_new == -INFINITY) ? m_prev : m_new;
                float alpha = __builtin_amdgcn_exp2f(fminf(m_prev - m_use, 126.f));
                asm volatile("s_nop 1" : "+v"(alpha));  // v_exp_f32 -> inline-asm consumer: hipcc does not insert the wait state
                m_run = m_new;
                psum_thr = __all(m_new != -INFINITY) ? 2048.f : -1.f;
                l_run *= alpha;
                if constexpr (PRE) {
                    pre_shift = m_prev - m_use;
                    // what the next S^T accumulators start from.  Rewritten IN PLACE (tied asm operands): as plain assignments hipcc
                    // keeps the old and the new value in two tuples and copies one into the other on the FAST path of every tile
                    const float nm = -m_use;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float c = neg_ref[r];
                        asm volatile("v_mov_b32 %0, %1" : "+v"(c) : "v"(nm));
                        neg_ref[r] = c;
                    }
                }
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {   // in place (tied operands)
                        float x = acc_o[db][r];
                        asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "v"(alpha));
                        acc_o[db][r] = x;
                    }
                psum = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    probs_impl(kk, 0, 8, std::true_type{});
                    asm volatile("" : "+v"(pf[kk >> 1][kk & 1]), "+v"(psum));
                }
            }
        } else {
            float mx = sc[0][0];
            if constexpr (ABL != 2) {
                // (asm: fmaxf() makes hipcc canonicalise every input with a v_max_f32 x, x — 8 extra issue slots per tile)
    #pragma unroll
                for (int e = 1; e < 31; e += 2) mx = vmax3(mx, sc[e >> 4][e & 15], sc[(e + 1) >> 4][(e + 1) & 15]);
                const unsigned u = __builtin_bit_cast(unsigned, vmax2(mx, sc[1][15]));
                const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // the other half of the row: lane ^ 32
                mx = vmax2(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1])) * c_log2;
            }
            m_use = (m_run == -INFINITY) ? 0.f : m_run;
            if (ABL != 2 && !__all(mx <= m_run + kDefer)) {     // some row's maximum moved by more than 2^kDefer: exact update
                const float m_new = fmaxf(m_run, mx);
                m_use = (m_new == -INFINITY) ? 0.f : m_new;
                float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
                asm volatile("s_nop 1" : "+v"(alpha));  // v_exp_f32 -> inline-asm consumer: hipcc does not insert the wait state
                m_run = m_new;
                l_run *= alpha;
    #pragma unroll
                for (int db = 0; db < DB; ++db)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {   // in place (tied operands), see attn_body_pp
                        float x = acc_o[db][r];
                        asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "v"(alpha));
                        acc_o[db][r] = x;
                    }
            }
            if constexpr (kSpread && NP > 1) {
                if (more) dma_piece(t + dist, std::integral_constant<int, 1>{});
            }
            psum = 0.f;
    #pragma unroll
            for (int kk = 0; kk < 4 - kShadow; ++kk) {
                probs(kk, 0, 8);
                asm volatile("" : "+v"(pf[kk >> 1][kk & 1]), "+v"(psum));   // stays in this phase
            }
        }
        if constexpr (kSpread) {
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            stage_request(t);
        }
    };
    // Matrix phase: O^T += V(t)^T P(t)^T (4 DB MFMAs), then S(t+1)^T = K(t+1) Q^T into sn (2 KS MFMAs).
    // One step = { LDS read of the operand kPF steps ahead; one MFMA; a 7-instruction slice of the probabilities of the
    // next 16-key step }; steps are fenced with sched_barrier so the reads stay kPF MFMAs (~kPF x 32 cycles) ahead of their
    // use — left alone, hipcc puts every read directly in front of its MFMA and the phase runs at LDS latency.
    // (Requesting the first kPF operands — all V(t), visible since the barrier in front of M(t-1) — at the end of the vector phase,
    //  in front of the barrier, so that the matrix phase opens with an MFMA, measured 1.1 % slower: the 16 reads lengthen the
    //  vector phase by more than the matrix phase gains.  Only the first 2 or 4 operands there, with the barrier waiting for all LDS
    //  reads but those — s_waitcnt lgkmcnt(4 / 8) — measured 0.8 % / 1.5 % slower, same box.)
    constexpr int kPF = LEAN ? 4 : 8;
    constexpr int NPV = 4 * DB;
    static_assert(kCarry <= kPF && kCarry <= NPV, "carried operands are V operands of the first MFMAs");
    V8 ring[kPF + 1];
    V8 carry[kCarry > 0 ? kCarry : 1];
    auto matrix_prefetch = [&](int t) {
        const char* stv = smem + (t % NS) * kStage;
#pragma unroll
        for (int i = kCarry; i < kPF; ++i) {
            if constexpr (ABL == 1) ring[i % (kPF + 1)] = qf[i % KS];
            else ring[i % (kPF + 1)] = vfrag(stv, i / DB, i % DB);
        }
    };
    auto carry_load = [&](int t, int i) {   // V operand i of tile t (its stage holds the tile by now, see kCarry)
        if constexpr (kCarry > 0) carry[i] = vfrag(smem + (t % NS) * kStage, i / DB, i % DB);
    };
    auto matrix_phase = [&](int t, auto has_next_c) {
        constexpr bool has_next = decltype(has_next_c)::value;
        constexpr int NALL = has_next ? NPV + 2 * KS : NPV;
        const char* stv = smem + (t % NS) * kStage;
        const char* stk = smem + ((t + 1) % NS) * kStage;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto fetch = [&](int i) {  // operand of step i
            if (i >= NALL) return;
            if (i < NPV) {
                if constexpr (ABL == 1) ring[i % (kPF + 1)] = qf[i % KS];
                else ring[i % (kPF + 1)] = vfrag(stv, i / DB, i % DB);
            } else {
                const int j = i - NPV;
                if constexpr (ABL == 5) ring[i % (kPF + 1)] = qf[j % KS];
                else ring[i % (kPF + 1)] = kfrag(stk, j & 1, j >> 1);
            }
        };
        matrix_prefetch(t);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NALL; ++i) {
            fetch(i + kPF);
            if constexpr (has_next && kCarry > 0) {   // the last kPF steps have no operand of this phase left to fetch: the next tile's first
                if (i + kPF >= NALL && i + kPF - NALL < kCarry) carry_load(t + 1, i + kPF - NALL);
            }
            if (i < NPV) {
                const int kk = i / DB, db = i % DB;
                acc_o[db] = E::mfma(i < kCarry ? carry[i < kCarry ? i : 0] : ring[i % (kPF + 1)], pf[kk >> 1][kk & 1], acc_o[db]);
                // probabilities of a later 16-key step in the shadow of this MFMA (one slice per d-block)
                if (kk + 1 < 4 && kk + 1 >= 4 - kShadow) probs(kk + 1, db * (8 / DB), (db + 1) * (8 / DB));
                if (i == NPV - 1) l_run += psum;
            } else {
                const int j = i - NPV, ks = j >> 1, b = j & 1;
                // (PRE, first step: D = A B + neg_ref with neg_ref left where it is — E::mfma_keep_c.  Its result is read by the next
                //  step's MFMA of the same key block only, as C, same tuple: legal back to back; neg_ref is written on the exact path
                //  of a vector phase, a barrier away from any matrix phase.)
                if constexpr (PRE) sc[b] = (ks == 0) ? E::mfma_keep_c(ring[i % (kPF + 1)], qf[ks], neg_ref) : E::mfma(ring[i % (kPF + 1)], qf[ks], sc[b]);
                else sc[b] = E::mfma(ring[i % (kPF + 1)], qf[ks], ks == 0 ? zero : sc[b]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (has_next) asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
    };

    if (idle) {
        for (int t = 0; t < nT; ++t) {
            if (!kOneBar || lagging) pp_barrier();
            stage_resolve_next(t, kGuarded);
            stage_request(t);
            if (!kOneBar || !lagging) pp_barrier();
        }
        pp_barrier();
        if (!kOneBar && !lagging) pp_barrier();
        P::notify(prm, ctx);
        return;
    }

    // ---- M(0): only S(0) ----
    if (nT > 0) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int b = 0; b < 2; ++b) sc[b] = E::mfma(kfrag(smem, b, ks), qf[ks], ks == 0 ? zero : sc[b]);   // (reference 0 so far)
        asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
#pragma unroll
        for (int i = 0; i < kCarry; ++i) carry_load(0, i);
    }
    // (the last tile is peeled: a run-time "has next" test inside the loop makes hipcc hoist the common VALU work above
    //  the branch and keep two register sets for O with 32 copies per tile)
    // (kOneBar: which of the two barriers of a tile a wave keeps is a run-time property of the wave — the skip is a branch inside
    //  the asm block of pp_barrier_if, one copy of the loop)
    const int bar_n = lagging ? 1 : 0, bar_m = lagging ? 0 : 1;
    auto tile = [&](int t, auto has_next_c, auto guard_c) {
        tick(std::integral_constant<int, 0>{});
        if constexpr (kOneBar) pp_barrier_if(bar_n);
        else pp_barrier();
        tick(std::integral_constant<int, 1>{});
        if (kPrioV) __builtin_amdgcn_s_setprio(1);
        vector_phase(t, guard_c);
        if (kPrioV) __builtin_amdgcn_s_setprio(0);
        tick(std::integral_constant<int, 2>{});
        if constexpr (kOneBar) pp_barrier_if(bar_m);
        else pp_barrier();
        tick(std::integral_constant<int, 3>{});
        if (kPrioM && !kPrioStatic) __builtin_amdgcn_s_setprio(1);   // the matrix phase wins the VALU / MFMA issue arbitration against the partner's vector phase
        matrix_phase(t, has_next_c);
        if (kPrioM && !kPrioStatic) __builtin_amdgcn_s_setprio(0);
    };
    if (kPrioM && kPrioStatic && lagging) __builtin_amdgcn_s_setprio(1);
    // steady state: every tile a phase of tile t touches (t + dist + 1 at most) exists; then the guarded tail; then the peeled last tile
    int t = 0;
    for (const int n_main = nT - dist - 1; t < n_main; ++t) tile(t, std::true_type{}, std::false_type{});
    for (; t + 1 < nT; ++t) tile(t, std::true_type{}, kGuarded);
    if (nT > 0) tile(nT - 1, std::false_type{}, kGuarded);
    // the leading waves wait until the lagging waves have read V of the last tile: the epilogue reuses the stages
    pp_barrier();
    if (!kOneBar && !lagging) pp_barrier();
    if constexpr (TRACE) {
        if (blockIdx.x == kPpTraceBlock && lane == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g_pp_trace[wave * 8 + j] = tr_acc[j];
            if (wave == 0) g_pp_trace[64] = (unsigned long long)nT, g_pp_trace[65] = tr_last - tr_first;
        }
    }

    unsigned long long wg_t2 = 0;
    if constexpr (TRACE) wg_t2 = __builtin_amdgcn_s_memtime();
    // ---------------- epilogue (same as attn_body) ----------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if constexpr (P::kPartialOut) {
        // (max-free softmax: O and l are scaled to the reference m_use, not to the running maximum)
        P::store_partial(prm, ctx, row_in_wg, g, acc_o, kMaxFree ? ((m_run == -INFINITY) ? -INFINITY : m_use) : m_run, l_tot);
        return;
    } else {
        const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
        constexpr int kEpiStride = D * 2 + 8;
        char* erow = smem + (size_t)(wave * 32) * kEpiStride;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                typename E::v4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = E::from_float(acc_o[db][rq * 4 + j] * inv);
                const int d0 = 32 * db + 8 * rq + 4 * g;
                *(typename E::v4*)(erow + ql * kEpiStride + d0 * 2) = o4;
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        T* __restrict__ ob = P::o_base(prm, ctx);
        constexpr int kLanesPerRow = D * 2 / 8;
        constexpr int kRowsPerPass = 64 / kLanesPerRow;
        const int sub = lane / kLanesPerRow;
        const int colb = (lane - sub * kLanesPerRow) * 8;
        int ephys[32 / kRowsPerPass];
#pragma unroll
        for (int i = 0; i < 32 / kRowsPerPass; ++i) ephys[i] = P::q_phys(prm, ctx, wave * 32 + i * kRowsPerPass + sub);
#pragma unroll
        for (int i = 0; i < 32 / kRowsPerPass; ++i) {
            const int rr = i * kRowsPerPass + sub;
            const u32x2 val = *(const u32x2*)(erow + rr * kEpiStride + colb);
            if (ephys[i] >= 0) *(u32x2*)((char*)(ob + (size_t)ephys[i] * P::o_rs(prm)) + colb) = val;
        }
        P::notify(prm, ctx);   // completion counter of the policy (band attention: per head, for an exchange that overlaps the launch)
        if constexpr (TRACE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (wave == 0 && lane == 0 && blockIdx.x < (unsigned)kWgTraceMax) {
                unsigned long long* w = g_wg_trace + (size_t)blockIdx.x * 6;
                w[0] = wg_t0, w[1] = tr_first, w[2] = wg_t2, w[3] = __builtin_amdgcn_s_memtime();
                w[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
                w[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
            }
        }
    }
}

}  // namespace svg
