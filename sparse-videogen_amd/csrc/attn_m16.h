// attn_body_m16 — the two-phase ping-pong schedule of attn_body_pp2 (attn_core.h) on v_mfma_f32_16x16x32_{bf16,f16}, head_dim 128.
//
// Why a second matrix shape (round 4, profiles/r04b_energy_table.txt, tools/energy_table.hip): the 16-bit attention kernels run at
// the chip's power limit, and with operand data that changes from one MFMA to the next — real K, V, P — the matrix pipe ALONE is
// power-limited: back-to-back v_mfma_f32_32x32x16_bf16 on all 1024 SIMDs is granted 1.75 GHz (0.72 of the 2.5 PFLOP/s the peak is
// quoted at; 2.39 GHz with constant operands), the same FLOPs issued as v_mfma_f32_16x16x32_bf16 2.15 GHz (0.88).  A 16x16x32 MFMA
// contracts 32 products into each fp32 accumulator per instruction, a 32x32x16 one 16: half the accumulator read-modify-writes per
// FLOP.  Everything else of the schedule is energy-neutral between the shapes: the same LDS operand bytes per tile (every fragment
// read feeds TWO MFMAs here: the two 16-row blocks of a wave's 32 query rows), the same registers, the same vector phase.
//
// Shapes (both GEMMs "swapped" as in attn_core.h, so that a lane owns query rows, not keys):
//   S^T[key][q] = K Q^T      A = K fragment: lane (g4, n): key 16 kb + n, d 32 ks + 8 g4 + [0, 8)       one ds_read_b128
//                            B = Q fragment: lane (g4, n): query row 16 rb + n of the wave, the same d     registers (32 VGPRs)
//                            D: lane (g4, n): query row 16 rb + n, keys 16 kb + 4 g4 + [0, 4)
//   O^T[d][q]   = V^T P^T    A = V^T fragment: lane (g4, n): d 16 dblk + n, k-index 8 g4 + [0, 8) of a 32-key chunk kc
//                            B = P fragment: lane (g4, n): query row 16 rb + n, k-index 8 g4 + [0, 8)
//                            D: lane (g4, n): query row 16 rb + n, d 16 dblk + 4 g4 + [0, 4)
//   with g4 = lane >> 4, n = lane & 15.  The contraction order of the PV GEMM is free, so k-index 8 g4 + j of chunk kc means key
//   32 kc + 4 g4 + j (j < 4) and key 32 kc + 16 + 4 g4 + (j - 4) (j >= 4): the P fragment of a lane is exactly its S^T accumulators of
//   key blocks 2 kc and 2 kc + 1 — no LDS round trip, no shuffles, as in the 32x32x16 bodies — and the V^T fragment is two
//   ds_read_b64_tr_b16 (4 keys x 16 columns each per 16-lane group).
// A lane holds TWO query rows (rb = 0, 1), each spread over the four lane groups g4: the softmax state (reference, partial row sum,
// mask intervals) is per (lane, rb); the row sum is completed across g4 once, in the epilogue; the row maximum crosses lanes on the
// exact path only (max-free softmax, see attn_body_pp2).
//
// LDS images: both tensors sub-tiled [D/32][64 keys][4 x 16 B] as in attn_body_pp2, both with ONE swizzle: 16-byte chunk c of key k sits
// in slot c ^ (((k >> 2) & 1) << 1), i.e. the two 32-byte halves of a row are swapped for keys 4 - 7, 12 - 15, ...  (MI355X_MICROARCH.md
// §LDS gives the lane groups that share an LDS cycle):
//   * K, ds_read_b128 (four groups of 16 lanes: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...): lane (g4, n) reads key n, chunk g4 —
//     a group is 8 keys of one g4 plus 4 + 4 keys of another; with this swizzle their 16 slots x 4 banks are disjoint.  (The swizzle of
//     the 32x32x16 bodies, slot = c ^ ((k >> 2) & 3), is two-way here: SQ_LDS_BANK_CONFLICT 2.6e9 cycles per launch in the first
//     version, profiles/r04e_pmc_lds_m16_first.txt.)
//   * V, ds_read_b64_tr_b16 (two groups of 32 lanes): a wave's transposing read covers 16 keys x 16 columns = 16 rows x 32 B at a 64 B
//     stride; the swap makes every 8 consecutive rows (one lane group pair) cover all 64 banks.
// The swizzle is applied to the per-lane SOURCE address of the LDS-DMA (the same address for K and V), so one lane still resolves one
// key row per tile for both tensors.
#pragma once
#include "attn_core.h"

namespace svg {

template <typename T>
struct Mfma16;
// (mfma_keep_c: D = A B + C with D and C in DIFFERENT registers, as inline asm — for a C that stays live hipcc selects the tied form of
//  the builtin and copies C first; see Elt::mfma_keep_c in svg_common.h.  The caller owns the hazards: tools/asm_hazards.py audits the
//  kept listing, tests/test_w4_asm_audit.py.)
template <>
struct Mfma16<__bf16> {
    static __device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma_keep_c(bf16x8 a, bf16x8 b, const f32x4& c) {
        f32x4 d;
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
    }
};
template <>
struct Mfma16<_Float16> {
    static __device__ __forceinline__ f32x4 mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma_keep_c(f16x8 a, f16x8 b, const f32x4& c) {
        f32x4 d;
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
    }
};

constexpr int attn_m16_lds_bytes() { return attn_pp2_lds_bytes<128>(); }

// lanes l, l ^ 16, l ^ 32, l ^ 48 hold one query row: maximum / sum over them (rare paths and the epilogue only)
__device__ __forceinline__ float quad_group_max(float x) {
    x = vmax2(x, __shfl_xor(x, 16));
    return vmax2(x, __shfl_xor(x, 32));
}
__device__ __forceinline__ float quad_group_sum(float x) {
    x += __shfl_xor(x, 16);
    return x + __shfl_xor(x, 32);
}

// PRIO: which phase raises the wave's issue priority (1: matrix phase, as attn_body_pp2; 0: none; 2: vector phase).  ONEBAR: -1 as the
// policy says, 0 / 1 forced.
// PRE: q arrives multiplied by sm_scale * log2(e) and the S^T accumulators start at minus the row's reference, so the MFMAs deliver the
// exponent argument (no scale-and-shift FMA per score; the scheme of attn_body_pp2's PRE form).  QKF16 (with PRE): the q and k pointers
// hold IEEE fp16 bit patterns whatever T is — q' = fp16(c * q), k16 = fp16(k), written by svg_qk_to_f16 — and S^T runs on the f16 MFMA
// while P, V and O stay T: for T = bf16 the scale then rides in 11 mantissa bits instead of 8 (2^-12 instead of 2^-9 relative rounding
// of c * q; a bf16 k converts to fp16 exactly inside fp16's range), which keeps the pre-scaled form inside the plain kernel's distance
// to the reference's formulation.  MEASURED in round 4 and NOT shipped (an entry point svg_band_attention_f16qk with a conversion pass and a
// device-side overflow fallback to the plain body was built, tested and removed within the session; what is left are the numbers): parity as intended — 1.95 - 2.33e-3 rel. L2 to the reference's formulation on
// |score| up to 80, the plain kernel's 1.85 - 2.32e-3, where the bf16 pre-scaled form reaches 6.4e-3 (profiles/r04l_pytest_f16qk.txt) — but
// no time: the kernel took 33.8 ms where the bf16 PRE form takes 32.8 and the plain body 34.5 under the same profiler run, plus 0.58 ms
// for the conversion pass (profiles/r04m_f16qk_kernel_trace.txt, r04l_ab_m16_f16qk.txt: 32.9 vs 33.4 ms end to end, -1.4 %).  The f16
// MFMA's wider multipliers take back in clock what the missing FMAs save — the power limit once more.  The template flag stays (it is
// four lines of the body); nothing instantiates it.
template <typename T, typename P, bool TRACE = false, int PRIO = 1, int ONEBAR = -1, bool PRE = false, bool QKF16 = false>
__device__ __forceinline__ void attn_body_m16(const typename P::Params& prm, char* smem, char* policy_lds) {
    using E = Elt<T>;
    using M = Mfma16<T>;                                                  // PV
    using TQ = std::conditional_t<QKF16, _Float16, T>;                    // element type of the q / k bits
    using MQ = Mfma16<TQ>;                                                // QK^T
    using V8 = typename E::v8;
    using Q8 = typename Elt<TQ>::v8;
    static_assert(!QKF16 || PRE, "fp16 q / k carriers exist for the pre-scaled form only");
    constexpr int D = 128;
    constexpr int NW = 8;
    constexpr int KS = D / 32;              // 32-wide contraction steps of S^T
    constexpr int NDB = D / 16;             // 16-wide d blocks of O^T
    constexpr int NS = 4;                   // LDS stages
    constexpr int kImg = kBN * D * 2;       // bytes of a K or V image
    constexpr int kStage = 2 * kImg;
    constexpr int NP = 2;                   // DMA pieces (16 keys x 64 B) per wave per tensor per tile
    constexpr int kCarry = 8;               // V fragments of the next matrix phase read in the tail of this one (attn_body_pp2: SVG_PP2_CARRY)
    constexpr int kPF = 8;                  // operand fragments in flight ahead of their MFMAs
    constexpr bool kOneBar = ONEBAR < 0 ? P::kOneBarrier : (ONEBAR != 0);   // one workgroup barrier per tile instead of two (attn_body_pp2: on for the variable-block policy)
    static_assert(P::kRowBlocks == 1 && P::kSubTiles == 1 && !P::kPartialOut && !P::kFixup && P::kIntervalMask, "band / variable-block policy");
    // MSUM (SVG_M16_MFMASUM, bf16, plain form): the row sum on the matrix pipe.  Four extra MFMAs per tile (A = a fragment of ones, B = the
    // P fragment: every accumulator of a lane receives the COMPLETE sum of its query row over the chunk's 32 keys) replace the 32 v_add of
    // the vector phase; the overflow test of the max-free softmax — a probability above the reference by more than 2^kBias — becomes a bit
    // test on the packed bf16 probabilities: with the exponent argument lowered by kBias a probability reaches 2.0 (exponent field >= 128,
    // bit 14 / 30 of the packed word: the OR of all sixteen words shows it, 8 v_or3_b32) exactly when the row sum test of the plain
    // form would be near its 2048.  O and l carry the common factor 2^-kBias, which the final division removes.
    // Measured (profiles/r05a_ab_m16_msum_prio3.txt, same box, HunyuanVideo 720p): 64.1 instead of 65.1 Mcycles per launch, of which the power
    // management returns half as clock (1985 vs 2000 MHz): 32.3 against 32.55 ms; with the sum MFMAs after the chunk's LAST PV step instead of
    // its first the gain is gone (65.4 Mcycles).  fp16 keeps the vector-phase sum: 2^-10 would push small probabilities into fp16's subnormals.
#ifdef SVG_M16_NO_MFMASUM
    constexpr bool MSUM = false;
#else
    constexpr bool MSUM = std::is_same_v<T, __bf16> && !PRE;
#endif
    constexpr float kBias = MSUM ? 10.f : 0.f;

    typename P::Ctx ctx;
    if (!P::init(prm, ctx, policy_lds)) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_id();
    const int g4 = lane >> 4;
    const int n16 = lane & 15;
    const bool lagging = wave >= NW / 2;
    const int nT = ctx.nT;

    const T* __restrict__ qb = P::q_base(prm, ctx);
    const T* __restrict__ kb_ = P::k_base(prm, ctx);
    const T* __restrict__ vb = P::v_base(prm, ctx);
    const unsigned lds0 = (unsigned)(size_t)smem;

    // ---- DMA bookkeeping: as attn_body_pp2 (a wave's NP pieces per tensor are d-blocks dma_db0 .. of ONE 16-key group), with the V
    //      source chunks of both tensors swizzled: LDS slot s of key k holds chunk s ^ (((k >> 2) & 1) << 1) ----
    const int dma_kg = wave / 2;
    const int dma_db0 = (wave % 2) * NP;
    const int krow = 16 * dma_kg + (lane >> 2);
    typename P::KvCursor cur;
    P::kv_cursor_init(prm, ctx, cur, krow);
    const unsigned vsw = (unsigned)((lane >> 4) & 1) << 1;                       // ((key >> 2) & 1) << 1 of this lane's key row: the chunk XOR of BOTH images
    const unsigned col_v = (unsigned)(dma_db0 * 64 + (((lane & 3) ^ vsw) * 16));
    const unsigned lds_piece = lds0 + (unsigned)(dma_db0 * (kBN * 64) + dma_kg * 1024);
    int nphys = 0, nnext = 0;
    auto resolve = [&](int t, auto guard_c) {
        if constexpr (decltype(guard_c)::value) nnext = (t < nT) ? P::kv_phys(prm, ctx, cur, t, krow) : 0;
        else nnext = P::kv_phys(prm, ctx, cur, t, krow);
    };
    constexpr std::true_type kGuarded{};
    auto take = [&]() { nphys = nnext; };
    // byte offsets of a key row inside its head: row * (row stride in bytes) + the lane's 16-byte column.  The row strides are kernel
    // arguments (svg_attn_layout_t: 2 D for contiguous heads, H * D or 3 * H * D elements for k / v read in place from a projection's
    // output) — one v_mad_u32_u24 per tensor and tile (rows and byte strides are below 2^24, the products below 2^32: layout_from_abi)
    const unsigned k_rsb = (unsigned)P::k_rs(prm) * 2u, v_rsb = (unsigned)P::v_rs(prm) * 2u;
    auto dma_piece = [&](int t, auto j_c) {
        constexpr int j = decltype(j_c)::value;
        const unsigned st = __builtin_amdgcn_readfirstlane(lds_piece + (unsigned)((t % NS) * kStage) + j * (kBN * 64));
        const unsigned ko = __umul24((unsigned)nphys, k_rsb) + col_v;
        const unsigned vo = __umul24((unsigned)nphys, v_rsb) + col_v;
        const char* const kbp = (const char*)kb_ + j * 64;
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
    auto dma_issue = [&](int t) {
        dma_piece(t, std::integral_constant<int, 0>{});
        dma_piece(t, std::integral_constant<int, 1>{});
    };
    const int dist = lagging ? 3 : 2;   // tile u + dist is requested in N(u)
    for (int t = 0; t < dist; ++t) {
        resolve(t, kGuarded);
        take();
        if (t < nT) dma_issue(t);
    }
    resolve(dist, kGuarded);   // requested in N(0)

    // ---- Q fragments, mask intervals and softmax state of the lane's two query rows ----
    int q_log[2];
    Q8 qf[2][KS];
    int m_a0[2], m_b0[2];
    unsigned m_alen[2], m_blen[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int row_in_wg = wave * 32 + rb * 16 + n16;
        const int qp = P::q_phys(prm, ctx, row_in_wg);
        q_log[rb] = P::q_logical(ctx, row_in_wg);
        const T* qrow = qb + (size_t)(qp >= 0 ? qp : 0) * P::q_rs(prm) + g4 * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[rb][ks] = *(const Q8*)(qrow + ks * 32);
        P::row_intervals(prm, ctx, q_log[rb], m_a0[rb], m_alen[rb], m_b0[rb], m_blen[rb]);
    }

    const int k_lane = n16 * 64 + ((g4 ^ (((n16 >> 2) & 1) << 1)) << 4);
    const int v_lane0 = kImg + (4 * g4 + (n16 >> 2)) * 64 + (((g4 & 1) * 16) + 4 * (n16 & 3)) * 2;   // even 16-wide d blocks (the row's swizzled half)
    const int v_lane1 = v_lane0 ^ 32;                                                                // odd ones

    float m_run[2] = {-INFINITY, -INFINITY}, m_use[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f}, psum[2] = {0.f, 0.f};
    float m_sub[2] = {kBias, kBias};                                   // what the exponent argument subtracts: m_use + kBias
    f32x4 acc_l[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // MSUM: row sums (every element the same complete sum)
    V8 ones8;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones8[j] = E::from_float(1.f);
    bool force_exact = true;                                           // MSUM: psum_thr < 0
    f32x4 acc_o[NDB][2];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_o[db][rb][r] = 0.f;
    const float c_log2 = prm.scale_log2;
    if constexpr (MSUM) asm volatile("" : "+v"(ones8));

#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[rb][ks]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pp_barrier();
    if (!kOneBar && lagging) pp_barrier();  // waves 4..7 run one phase behind

    const bool idle = !P::wave_active(ctx, wave * 32);

    // per-phase cycle trace (diagnostics builds, svg_debug_pp_trace: the slots of attn_body_pp2 — M, barrier, N, barrier)
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

    f32x4 neg_ref[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // PRE: minus the reference of the lane's two rows (what the S^T accumulators start from)
    float pre_shift[2] = {0.f, 0.f};                                   // PRE, exact path only: old reference minus new reference
    f32x4 sc[4][2];        // scores [16-key block][row block]: S(t) until the PV steps have consumed it, then S(t + 1) accumulates here
    V8 pf[2][2];           // probabilities [32-key chunk][row block]
    float psum_thr = -1.f; // (wave-uniform) 2048 once every row of the wave has a finite reference; until then every tile takes the exact path

    // operand fragments travel as raw bits (the ring holds V fragments of type T and K fragments of type TQ)
    auto kfrag = [&](const char* st, int kblk, int ks) -> i16x8 { return *(const i16x8*)(st + k_lane + ks * (kBN * 64) + kblk * 1024); };
    auto vfrag = [&](const char* st, int kc, int db) -> i16x8 {
        const char* vbase = st + ((db & 1) ? v_lane1 : v_lane0) + (db >> 1) * (kBN * 64) + (32 * kc) * 64;
        const i16x4 lo = lds_read_tr16(vbase);
        const i16x4 hi = lds_read_tr16(vbase + 16 * 64);
        const i16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return both;
    };
    // probabilities of the 32-key chunk kc: the lane's 8 scores per row block (key blocks 2 kc, 2 kc + 1) against the row's reference
    auto probs_impl = [&](int kc, auto shifted_c) {
        constexpr bool shifted = decltype(shifted_c)::value;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p;
                    if constexpr (PRE && shifted) p = __builtin_amdgcn_exp2f(sc[2 * kc + h][rb][r] + pre_shift[rb]);
                    else if constexpr (PRE) p = __builtin_amdgcn_exp2f(sc[2 * kc + h][rb][r]);    // the MFMAs delivered the exponent argument
                    else p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * kc + h][rb][r], c_log2, -m_sub[rb]));
                    if constexpr (!MSUM) psum[rb] += p;
                    pf[kc][rb][4 * h + r] = E::from_float(p);
                }
    };
    auto probs = [&](int kc) { probs_impl(kc, std::false_type{}); };
    auto stage_resolve_next = [&](int t, auto guard_c) {
        take();
        resolve(t + dist + 1, guard_c);
    };
    auto stage_request = [&](int t) {
        const bool more = t + dist < nT;
        if (more) dma_issue(t + dist);
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // vector phase of tile t on sc: mask, probabilities, sum check, (rare) exact path, DMA requests, DMA wait
    auto vector_phase = [&](int t, auto guard_c) {
        constexpr bool guard = decltype(guard_c)::value;
        stage_resolve_next(t, guard_c);
        const bool more = !guard || t + dist < nT;
        if (more) dma_piece(t + dist, std::integral_constant<int, 0>{});
        const int tk0 = P::tile_key0(ctx, t);
        const int cls = P::classify(prm, ctx, tk0, wave * 32);
        if (cls != TILE_FULL) {
            const bool part = (cls == TILE_PARTIAL);  // a tile this wave does not need at all is processed fully masked
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                int ka = tk0 + 4 * g4 - m_a0[rb], kb2 = tk0 + 4 * g4 - m_b0[rb];
                asm volatile("" : "+v"(ka), "+v"(kb2));   // opaque: keeps LICM from hoisting the per-element terms out of the loop
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = 16 * b + r;
                        const bool ok = ((unsigned)(ka + key) < m_alen[rb]) | ((unsigned)(kb2 + key) < m_blen[rb]);
                        sc[b][rb][r] = (part & ok) ? sc[b][rb][r] : -INFINITY;
                    }
            }
        }
        if (more) dma_piece(t + dist, std::integral_constant<int, 1>{});
        psum[0] = 0.f, psum[1] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            probs(kc);
            asm volatile("" : "+v"(pf[kc][0]), "+v"(pf[kc][1]), "+v"(psum[0]), "+v"(psum[1]));   // stays in this phase
        }
        bool exact;
        if constexpr (MSUM) {
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t w0 = __builtin_bit_cast(u32x4_t, pf[0][0]), w1 = __builtin_bit_cast(u32x4_t, pf[0][1]);
            const u32x4_t w2 = __builtin_bit_cast(u32x4_t, pf[1][0]), w3 = __builtin_bit_cast(u32x4_t, pf[1][1]);
            const unsigned o0 = w0[0] | w0[1] | w0[2], o1 = w0[3] | w1[0] | w1[1], o2 = w1[2] | w1[3] | w2[0];   // (v_or3_b32, depth 3)
            const unsigned o3 = w2[1] | w2[2] | w2[3], o4 = w3[0] | w3[1] | w3[2];
            const unsigned bits = (o0 | o1 | o2) | (o3 | o4 | w3[3]);
            exact = force_exact || __any((bits & 0x40004000u) != 0u);
        } else {
            exact = !__all(psum[0] + psum[1] <= psum_thr);
        }
        if (exact) {      // exact path (rare; always until every row has a finite reference; also a non-finite sum)
            bool all_finite = true;
            float alpha[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                float mx = vmax3(sc[0][rb][0], sc[0][rb][1], sc[0][rb][2]);
                mx = vmax3(mx, sc[0][rb][3], sc[1][rb][0]);
                mx = vmax3(mx, sc[1][rb][1], sc[1][rb][2]);
                mx = vmax3(mx, sc[1][rb][3], sc[2][rb][0]);
                mx = vmax3(mx, sc[2][rb][1], sc[2][rb][2]);
                mx = vmax3(mx, sc[2][rb][3], sc[3][rb][0]);
                mx = vmax3(mx, sc[3][rb][1], sc[3][rb][2]);
                mx = vmax2(mx, sc[3][rb][3]);
                const float m_prev = m_use[rb];
                mx = quad_group_max(mx);
                if constexpr (PRE) mx += m_prev;   // scores are relative to the reference they were accumulated under
                else mx *= c_log2;
                const float m_new = fmaxf(m_run[rb], mx);
                m_use[rb] = (m_new == -INFINITY) ? m_prev : m_new;
                m_sub[rb] = m_use[rb] + kBias;
                float a = __builtin_amdgcn_exp2f(fminf(m_prev - m_use[rb], 126.f));
                asm volatile("s_nop 1" : "+v"(a));  // v_exp_f32 -> inline-asm consumer: hipcc does not insert the wait state
                alpha[rb] = a;
                m_run[rb] = m_new;
                all_finite = all_finite && (m_new != -INFINITY);
                l_run[rb] *= a;
                if constexpr (MSUM) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = acc_l[rb][r];
                        asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "v"(a));
                        acc_l[rb][r] = x;
                    }
                }
                if constexpr (PRE) {
                    pre_shift[rb] = m_prev - m_use[rb];
                    // what the next S^T accumulators start from.  Rewritten IN PLACE (tied asm operands): as plain assignments hipcc keeps
                    // the old and the new value in two tuples and copies one into the other on the FAST path of every tile
                    const float nm = -m_use[rb];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float c = neg_ref[rb][r];
                        asm volatile("v_mov_b32 %0, %1" : "+v"(c) : "v"(nm));
                        neg_ref[rb][r] = c;
                    }
                }
            }
            psum_thr = __all(all_finite) ? 2048.f : -1.f;
            force_exact = !__all(all_finite);
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {   // in place (tied operands): plain assignments make hipcc keep two register sets for O
                        float x = acc_o[db][rb][r];
                        asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "v"(alpha[rb]));
                        acc_o[db][rb][r] = x;
                    }
            psum[0] = 0.f, psum[1] = 0.f;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                probs_impl(kc, std::true_type{});
                asm volatile("" : "+v"(pf[kc][0]), "+v"(pf[kc][1]), "+v"(psum[0]), "+v"(psum[1]));
            }
        }
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // Matrix phase: O^T += V(t)^T P(t)^T (16 fragments, 32 MFMAs), then S(t+1)^T = K(t+1) Q^T (16 fragments, 32 MFMAs).  One step =
    // { LDS read of the fragment kPF steps ahead; the fragment's two MFMAs (row blocks 0 and 1) }, fenced with sched_barrier.
    constexpr int NPV = 2 * NDB;
#ifdef SVG_M16_MSUM_AT
    constexpr int kMsumAt = SVG_M16_MSUM_AT;
#else
    constexpr int kMsumAt = 0;         // the d block after whose MFMAs the chunk's two row-sum MFMAs are issued
#endif
    i16x8 ring[kPF + 1];
    i16x8 carry[kCarry];
    auto carry_load = [&](int t, int i) { carry[i] = vfrag(smem + (t % NS) * kStage, i / NDB, i % NDB); };
    auto matrix_phase = [&](int t, auto has_next_c) {
        constexpr bool has_next = decltype(has_next_c)::value;
        constexpr int NALL = has_next ? NPV + 4 * KS : NPV;
        const char* stv = smem + (t % NS) * kStage;
        const char* stk = smem + ((t + 1) % NS) * kStage;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        auto fetch = [&](int i) {  // operand fragment of step i
            if (i >= NALL) return;
            if (i < NPV) {
                ring[i % (kPF + 1)] = vfrag(stv, i / NDB, i % NDB);
            } else {
                const int j = i - NPV;
                ring[i % (kPF + 1)] = kfrag(stk, j & 3, j >> 2);
            }
        };
#pragma unroll
        for (int i = kCarry; i < kPF; ++i) fetch(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NALL; ++i) {
            fetch(i + kPF);
            if constexpr (has_next) {   // the last kPF steps have no operand of this phase left to fetch: the next tile's first V fragments
                if (i + kPF >= NALL && i + kPF - NALL < kCarry) carry_load(t + 1, i + kPF - NALL);
            }
            if (i < NPV) {
                const int kc = i / NDB, db = i % NDB;
                const V8 a = __builtin_bit_cast(V8, i < kCarry ? carry[i < kCarry ? i : 0] : ring[i % (kPF + 1)]);
                acc_o[db][0] = M::mfma(a, pf[kc][0], acc_o[db][0]);
                acc_o[db][1] = M::mfma(a, pf[kc][1], acc_o[db][1]);
                if constexpr (MSUM) {
                    if (db == kMsumAt) {
                        acc_l[0] = M::mfma(ones8, pf[kc][0], acc_l[0]);
                        acc_l[1] = M::mfma(ones8, pf[kc][1], acc_l[1]);
                    }
                } else {
                    if (i == NPV - 1) l_run[0] += psum[0], l_run[1] += psum[1];
                }
            } else {
                const int j = i - NPV, ks = j >> 2, b = j & 3;
                const Q8 a = __builtin_bit_cast(Q8, ring[i % (kPF + 1)]);
                // (PRE, first step: D = A B + neg_ref with neg_ref left where it is.  Its result is read by the next step's MFMA of the same
                //  key block only, as C, same tuple; neg_ref is written on the exact path of a vector phase, a barrier away.)
                if constexpr (PRE) {
                    sc[b][0] = ks == 0 ? MQ::mfma_keep_c(a, qf[0][ks], neg_ref[0]) : MQ::mfma(a, qf[0][ks], sc[b][0]);
                    sc[b][1] = ks == 0 ? MQ::mfma_keep_c(a, qf[1][ks], neg_ref[1]) : MQ::mfma(a, qf[1][ks], sc[b][1]);
                } else {
                    sc[b][0] = MQ::mfma(a, qf[0][ks], ks == 0 ? zero : sc[b][0]);
                    sc[b][1] = MQ::mfma(a, qf[1][ks], ks == 0 ? zero : sc[b][1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (has_next) {
#pragma unroll
            for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(sc[b][0]), "+v"(sc[b][1]));
        }
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
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const Q8 a = __builtin_bit_cast(Q8, kfrag(smem, b, ks));
                sc[b][0] = MQ::mfma(a, qf[0][ks], ks == 0 ? zero : sc[b][0]);   // (reference 0 so far, also for PRE)
                sc[b][1] = MQ::mfma(a, qf[1][ks], ks == 0 ? zero : sc[b][1]);
            }
#pragma unroll
        for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(sc[b][0]), "+v"(sc[b][1]));
#pragma unroll
        for (int i = 0; i < kCarry; ++i) carry_load(0, i);
    }
    // (kOneBar: which of the two barriers of a tile a wave keeps is a run-time property of the wave — the skip is a branch inside the asm
    //  block of pp_barrier_if, one copy of the loop.  Leading waves: [N(t), barrier, M(t)]; lagging waves: [barrier, N(t), M(t)].)
    const int bar_n = lagging ? 1 : 0, bar_m = lagging ? 0 : 1;
    auto tile = [&](int t, auto has_next_c, auto guard_c) {
        tick(std::integral_constant<int, 0>{});
        if constexpr (kOneBar) pp_barrier_if(bar_n);
        else pp_barrier();
        tick(std::integral_constant<int, 1>{});
        if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(1);
        vector_phase(t, guard_c);
        if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        tick(std::integral_constant<int, 2>{});
        if constexpr (kOneBar) pp_barrier_if(bar_m);
        else pp_barrier();
        tick(std::integral_constant<int, 3>{});
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);   // the matrix phase wins the issue arbitration against the partner's vector phase
        matrix_phase(t, has_next_c);
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    };
    if constexpr (PRIO == 3) {   // static priority for the younger half (cdna_hip_programming.md T5, static form): no per-phase flips
        if (lagging) __builtin_amdgcn_s_setprio(1);
    }
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

    // ---------------- epilogue: O^T -> LDS -> whole rows ----------------
    constexpr int kEpiStride = D * 2 + 8;
    char* erow = smem + (size_t)(wave * 32) * kEpiStride;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const float l_tot = MSUM ? acc_l[rb][0] : quad_group_sum(l_run[rb]);
        const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            typename E::v4 o4;
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = E::from_float(acc_o[db][rb][j] * inv);
            *(typename E::v4*)(erow + (rb * 16 + n16) * kEpiStride + (16 * db + 4 * g4) * 2) = o4;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    T* __restrict__ ob = P::o_base(prm, ctx);
    const int o_rs = P::o_rs(prm);
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
        if (ephys[i] >= 0) *(u32x2*)((char*)(ob + (size_t)ephys[i] * o_rs) + colb) = val;
    }
    P::notify(prm, ctx);
}

}  // namespace svg
