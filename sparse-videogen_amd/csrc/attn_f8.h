// fp8 (OCP e4m3) flash-attention body for gfx950: S^T = K Q^T and O^T = V^T P^T on v_mfma_scale_f32_32x32x64_f8f6f4 (64-deep
// contraction per instruction, twice the bf16 rate: 16 instead of 64 MFMA issues per 256 x 64 tile), fp32 softmax, bf16 / fp16
// output.  BASELINE.json configs[4] ("fp8 QK^T / PV"); the reference has no fp8 path (README.md:117 "[ ] Support FP8 attention"),
// so accuracy is stated against the 16-bit path and the fp32 oracle (tests/test_gpu_fp8.py).
//
// Inputs come from the quantise pre-pass (attention_f8.hip), which also applies the head placement, so this body sees every head
// in LOGICAL token order and never gathers:
//   q8, k8 : [BH, S_pad, D] e4m3, row-major, S_pad = 64-row multiple (rows >= S are zero), x * 448 / amax_head(x)
//   vt8    : [BH, S_pad / 64, D, 64] e4m3 — V^T per 64-key tile; inside a tile the key of byte position 32 g + 16 b + 4 j + i of a
//            row is 32 b + 8 j + 4 g + i, which is exactly the order in which the S^T accumulators of lane half g hold their
//            probabilities (see below): the P operand needs no data movement at all.
//   scales : [BH, 4] float: 2^e, 1 / sv, the E8M0 scale word (127 + e) * 0x01010101, 0 — q8 carries the softmax scale:
//            sum(k8 q8) * 2^e = scale_log2 * q.k, the exponent argument itself; 1 / sv is folded into the final 1 / l
//
// Operand layouts.  For the 32x32x64 f8f6f4 MFMA a lane supplies row / column (lane & 31) and 32 of the 64 contraction slots
// (lane >> 5 selects which half) as 32 bytes; which d (or key) sits in which slot is irrelevant as long as A and B agree:
//   S^T[key][q]: A = K row (32 b + ql) bytes [64 ks + 32 g, +32), B = Q row bytes [64 ks + 32 g, +32)          (ks = 0, 1: D = 128)
//   O^T[d][q]  : A = vt8 row (32 db + ql) bytes [32 g, +32), B = the lane's 32 probabilities in the order above
// The f32x16 result holds column (lane & 31) and rows (r & 3) + 8 (r >> 2) + 4 g like every 32x32 MFMA, so the epilogue is the
// 16-bit kernels'.  Probabilities are scaled by 2^8 before the conversion (e4m3 tops out at 448; the scale cancels in O / l): a
// probability of 2^-14 relative to the row maximum is still a normal number.
//
// Schedule: 8 waves x 32 rows (256-row q-tiles), two waves per SIMD, lock-step over 64-key tiles with two register-staged LDS
// stages and one barrier per tile.  With a quarter of the MFMA issues the matrix pipe is
// no longer what a tile waits for; the softmax VALU and the LDS operand reads are.
#pragma once
#include "attn_core.h"

namespace svg {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int v2i32 __attribute__((ext_vector_type(2)));

struct F8Args {
    const uint8_t* q8;
    const uint8_t* k8;
    const uint8_t* vt8;
    const float* scales;   // [BH, 4]
    int S_pad;             // rows of q8 / k8 per head (multiple of 64)
};

constexpr int kF8Scale127 = 0x7f7f7f7f;   // E8M0 block scales of the scaled MFMA: 2^0 for every block

__device__ __forceinline__ f32x16 mfma_f8(i32x8 a, i32x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, kF8Scale127, 0, kF8Scale127);
}

template <int D, int NW = 8, int NS = 2>
constexpr int attn_f8_lds_bytes() {
    constexpr int stages = NS * (2 * kBN * D);           // NS stages of [K image | V^T image], one byte per element
    constexpr int epi = NW * 32 * (D * 2 + 8);           // epilogue staging of the output rows (16-bit)
    return stages > epi ? stages : epi;
}

// byte offset of 16-B chunk c of key row `row` inside the K image ([64][128] bytes): the XOR makes the b128 reads of 16
// consecutive rows hit 16 different 16-byte bank groups
__device__ __forceinline__ int f8_k_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
// byte offset of 16-B chunk c of row d inside the V^T image ([128][64] bytes)
__device__ __forceinline__ int f8_v_off(int d, int c) { return d * 64 + ((c ^ ((d >> 2) & 3)) << 4); }

template <typename T, typename P>
__device__ __forceinline__ void attn_body_f8(const typename P::Params& prm, const F8Args& fa, char* smem) {
    using E = Elt<T>;
    constexpr int D = 128, DB = D / 32, KS = D / 64, NT = 512;
    constexpr int kKBytes = kBN * D, kStage = 2 * kBN * D;
    static_assert(P::kRowBlocks == 1 && P::BM == 256, "fp8 body: 8 waves x 32 rows");

    typename P::Ctx ctx;
    if (!P::init(prm, ctx, nullptr)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), g = lane >> 5, ql = lane & 31;
    const uint8_t* __restrict__ q8 = fa.q8 + (size_t)ctx.head * fa.S_pad * D;
    const uint8_t* __restrict__ k8 = fa.k8 + (size_t)ctx.head * fa.S_pad * D;
    const uint8_t* __restrict__ vt8 = fa.vt8 + (size_t)ctx.head * fa.S_pad * D;   // (S_pad / 64) tiles of 64 * D bytes
    const float two_e = fa.scales[4 * ctx.head], inv_v = fa.scales[4 * ctx.head + 1];

    // ---- Q fragments (B operand of S^T): rows are in logical order in q8; a row that does not exist reads row 0 (never stored) ----
    const int row_in_wg = wave * 32 + ql;
    const int q_log = P::q_logical(ctx, row_in_wg);
    const bool q_exists = P::q_phys(prm, ctx, row_in_wg) >= 0;
    i32x8 qf[KS];
    {
        const uint8_t* qrow = q8 + (size_t)(q_exists ? q_log : 0) * D + g * 32;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const i32x8*)(qrow + ks * 64);
    }

    // ---- staging: one 16-B chunk of the K tile and one of the V^T tile per thread ----
    const int kr = tid >> 3, kc = tid & 7;          // K: row, chunk
    const int vd = tid >> 2, vc = tid & 3;          // V^T: row (= d), chunk
    const int k_dst = f8_k_off(kr, kc), v_dst = kKBytes + f8_v_off(vd, vc);
    u32x4 kreg, vreg;
    auto issue = [&](int k0) {   // k0: first key of the tile (a multiple of 64; rows behind S are zero in the padded images)
        kreg = *(const u32x4*)(k8 + (size_t)(k0 + kr) * D + kc * 16);
        vreg = *(const u32x4*)(vt8 + (size_t)(k0 >> 6) * (kBN * D) + tid * 16);
    };
    auto stage_write = [&](int buf) {
        char* base = smem + buf * kStage;
        *(u32x4*)(base + k_dst) = kreg;
        *(u32x4*)(base + v_dst) = vreg;
    };

    // per-lane operand offsets: K fragment (block b, step ks) = chunks 4 ks + 2 g, +1 of row 32 b + ql (32 % 16 == 0: same swizzle
    // for both blocks); V^T fragment (d block db) = chunks 2 g, 2 g + 1 of row 32 db + ql
    const int ksw = (ql >> 1) & 7;
    const int vsw = (ql >> 2) & 3;                  // (32 db + ql) >> 2 & 3 == ql >> 2 & 3
    const int k_lane = ql * 128, v_lane = kKBytes + ql * 64;

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 acc_o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;

    const float c_log2 = two_e;      // (q8 carries the softmax scale up to this power of two)
    const int nT = ctx.nT;

    // S^T of one tile: 4 MFMAs (2 key blocks x 2 contraction steps of 64)
    auto qk = [&](const char* kbuf, f32x16 (&s)[2]) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[b][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const char* rowp = kbuf + k_lane + b * (32 * 128);
                const u32x4 lo = *(const u32x4*)(rowp + (((4 * ks + 2 * g) ^ ksw) << 4));
                const u32x4 hi = *(const u32x4*)(rowp + (((4 * ks + 2 * g + 1) ^ ksw) << 4));
                const i32x8 kf = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                s[b] = mfma_f8(kf, qf[ks], s[b]);
            }
    };

    // Two LDS stages, register-staged: iteration t computes tile t from stage t & 1, then writes tile t + 1 (registers loaded
    // during t - 1) into the other stage and requests tile t + 2; one barrier per tile.
    typename P::TileCur tc;          // walks two tiles ahead of the tile being processed
    P::tile_cur_init(ctx, tc);
    int f0 = tc.k0;                  // first key of tile t, t + 1 (f1 is what `issue` reads next)
    P::tile_cur_next(ctx, tc);
    int f1 = tc.k0;
    P::tile_cur_next(ctx, tc);
    if (nT > 0) {
        issue(f0);
        stage_write(0);
        if (nT > 1) issue(f1);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));   // Q has landed before the loop (see attn_body)
    __syncthreads();

    // Softmax without a running maximum (the scheme of attn_body_w4): probabilities are taken relative to a per-row reference
    // m_ref that only changes on the exact path, p = 2^(s c - m_ref + 4); as long as a lane's 32 probabilities of a tile sum to
    // <= 448 every one of them fits e4m3 and nothing else has to be checked.  A violation (normally also the first tile, whose
    // reference is the pseudo-reference 0) sends the WAVE through the exact path: row maximum, new reference, O and l rescaled, probabilities
    // recomputed.  On random data that is the first tile of a q-tile and a handful of later ones.
    constexpr float kPShift = 4.f, kPSumMax = 448.f;
    float m_ref = -INFINITY, m_off = -kPShift;    // m_off = (m_ref finite ? m_ref : 0) - kPShift
    float psum_thr = -1.f;   // kPSumMax once every row of the wave has a finite reference (see attn_body_pp2)
    // Packed probabilities.  Lives outside the loop on purpose: v_cvt_pk_fp8_f32 writes one half of its destination and keeps the
    // other, so the builtin takes the destination's old value — a literal 0 there costs a v_mov per word and tile (8 of the ~100
    // VALU instructions of a tile), the word's own stale contents cost nothing, and both halves are rewritten anyway.
    i32x8 pf = {0, 0, 0, 0, 0, 0, 0, 0};
    int buf = 0;
    for (int t = 0; t < nT; ++t) {
        const char* kbuf = smem + buf * kStage;
        const int tk0 = f0;
        f0 = f1, f1 = tc.k0;
        P::tile_cur_next(ctx, tc);
        const int cls = P::fast_full(ctx, tk0) ? (int)TILE_FULL : P::classify(prm, ctx, tk0, wave * 32);
        if (cls != TILE_SKIP) {
            f32x16 s_cur[2];
            qk(kbuf, s_cur);
            if (cls == TILE_PARTIAL) {
                asm volatile("; element-wise mask of a partial tile" ::: "memory");   // (keeps hipcc from if-converting the 32 predicates
                                                                                      //  into code that every tile executes)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * g;
                        s_cur[b][r] = P::allowed(prm, ctx, q_log, tk0 + key) ? s_cur[b][r] : -INFINITY;
                    }
            }
            // probabilities of this lane's 32 keys at reference offset `off`: packed e4m3 operand + their fp32 sum
            float psum;
            auto probs = [&](float off) {
                psum = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) {
                    float p4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = 4 * w8 + i;
                        p4[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[e >> 4][e & 15], c_log2, -off));
                        psum += p4[i];
                    }
                    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(p4[0], p4[1], pf[w8], false);   // (old = the word's stale contents: see pf)
                    pf[w8] = __builtin_amdgcn_cvt_pk_fp8_f32(p4[2], p4[3], w, true);
                }
            };
            probs(m_off);
            if (__any(!(psum <= psum_thr))) {      // exact path (rare; always until every row has a finite reference: see attn_body_pp2)
                float mx = s_cur[0][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s_cur[0][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_cur[1][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                // (until a row has seen a finite score its reference is the pseudo-reference 0 that m_off starts with: whatever was
                //  accumulated under it is rescaled like under any other reference; the clamp keeps alpha finite when nothing was)
                const float m_prev = m_off + kPShift;
                const float m_new = fmaxf(m_ref, mx * c_log2);
                const float m_use = (m_new == -INFINITY) ? m_prev : m_new;
                const float alpha = __builtin_amdgcn_exp2f(fminf(m_prev - m_use, 126.f));
                m_ref = m_new;
                psum_thr = __any(m_new == -INFINITY) ? -1.f : kPSumMax;
                m_off = m_use - kPShift;
                probs(m_off);
                l_run *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;
            }
            l_run += psum;
            // ---------------- O^T += V^T P^T: 4 MFMAs ----------------
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const char* rowp = kbuf + v_lane + db * (32 * 64);
                const u32x4 lo = *(const u32x4*)(rowp + (((2 * g) ^ vsw) << 4));
                const u32x4 hi = *(const u32x4*)(rowp + (((2 * g + 1) ^ vsw) << 4));
                const i32x8 vf = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                acc_o[db] = mfma_f8(vf, pf, acc_o[db]);
            }
        }
        if (t + 1 < nT) stage_write(buf ^ 1);
        if (t + 2 < nT) issue(f1);
        __syncthreads();
        buf ^= 1;
    }

    // ---------------- epilogue: O^T -> LDS -> whole rows (inverse placement through q_phys), as attn_body ----------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    constexpr int kEpiStride = D * 2 + 8;
    char* erow = smem + (size_t)(wave * 32) * kEpiStride;
    {
        const float inv = l_tot > 0.f ? inv_v / l_tot : 0.f;
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
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    T* __restrict__ ob = P::o_base(prm, ctx);
    constexpr int kLanesPerRow = D * 2 / 8, kRowsPerPass = 64 / kLanesPerRow, kPasses = 32 / kRowsPerPass;
    const int sub = lane / kLanesPerRow, colb = (lane - sub * kLanesPerRow) * 8;
    int ephys[kPasses];
#pragma unroll
    for (int i = 0; i < kPasses; ++i) ephys[i] = P::q_phys(prm, ctx, wave * 32 + i * kRowsPerPass + sub);
#pragma unroll
    for (int i = 0; i < kPasses; ++i) {
        const int rr = i * kRowsPerPass + sub;
        const u32x2 val = *(const u32x2*)(erow + rr * kEpiStride + colb);
        if (ephys[i] >= 0) *(u32x2*)((char*)(ob + (size_t)ephys[i] * D) + colb) = val;
    }
    P::notify(prm, ctx);
}


// =====================================================================================================================
// Two-phase ping-pong form of the fp8 band body: the schedule of attn_body_pp2 (attn_core.h) on e4m3 operands.  Per tile every wave
// runs ONE matrix phase — O^T += V(t)^T P(t)^T (4 MFMAs), S(t+1)^T = K(t+1) Q^T (4 MFMAs), operands streamed from LDS ahead of
// their MFMA — and ONE vector phase — mask, softmax numerators without a running maximum, LDS-DMA request of a later tile, DMA
// wait; waves 4..7 run one phase behind waves 0..3 (two barriers per tile), so a SIMD always pairs the matrix phase of one wave
// with the vector phase of the other.  In the lock-step body
// (attn_body_f8) both waves of a SIMD are in the same part of the tile at the same time: they stall on their MFMAs together, then
// share the VALU.
//   slot         2t      2t+1    2t+2     2t+3
//   waves 0-3    M(t)    N(t)    M(t+1)   N(t+1)          M(t) reads V(t-1) and K(t)
//   waves 4-7    N(t-1)  M(t)    N(t)     M(t+1)
// LDS: four stages of [K image 8 KiB | V^T image 8 KiB]; K / V^T arrive by LDS-DMA (one 1-KiB piece of each per wave and tile:
// 8 key rows / 16 d rows, the XOR swizzle of the images applied to the per-lane SOURCE address); tile w is requested in N(w - 2)
// (leading waves) / N(w - 3) (lagging waves), and every wave waits at the end of a vector phase for what it requested in the
// previous one.  Inputs are those of attn_body_f8 (pre-pass images in logical token order: nothing is gathered).
// =====================================================================================================================
template <typename T, typename P>
__device__ __forceinline__ void attn_body_f8pp(const typename P::Params& prm, const F8Args& fa, char* smem) {
    using E = Elt<T>;
    constexpr int D = 128, DB = D / 32, KS = D / 64, NW = 8, NS = 4;
    constexpr int kKBytes = kBN * D, kStage = 2 * kBN * D;
    static_assert(P::kRowBlocks == 1 && P::BM == 256 && P::kIntervalMask, "fp8 two-phase body: 8 waves x 32 rows, interval masks");

    typename P::Ctx ctx;
    if (!P::init(prm, ctx, nullptr)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), g = lane >> 5, ql = lane & 31;
    const bool lagging = wave >= NW / 2;
    const int nT = ctx.nT;
    const uint8_t* __restrict__ q8 = fa.q8 + (size_t)ctx.head * fa.S_pad * D;
    const uint8_t* __restrict__ k8 = fa.k8 + (size_t)ctx.head * fa.S_pad * D;
    const uint8_t* __restrict__ vt8 = fa.vt8 + (size_t)ctx.head * fa.S_pad * D;
    const float inv_v = fa.scales[4 * ctx.head + 1];
    const int q_scale = __float_as_int(fa.scales[4 * ctx.head + 2]);     // E8M0 block scale of the Q operand: 2^e in every byte
    const unsigned lds0 = (unsigned)(size_t)smem;

    // ---- DMA: wave w brings K rows 8 w .. 8 w + 7 and V^T rows 16 w .. 16 w + 15 of every tile ----
    const int kr = 8 * wave + (lane >> 3);                                         // key row of this lane's K chunk
    const unsigned k_src = (unsigned)(kr * D + (((lane & 7) ^ ((kr >> 1) & 7)) << 4));   // chunk that belongs at slot lane & 7
    const int vd = 16 * wave + (lane >> 2);                                        // d row of this lane's V^T chunk
    const unsigned v_src = (unsigned)(vd * 64 + (((lane & 3) ^ ((vd >> 2) & 3)) << 4));
    const unsigned lds_k = lds0 + (unsigned)(wave * 1024), lds_v = lds0 + (unsigned)(kKBytes + wave * 1024);
    typename P::TileCur rc;          // tile whose images are requested next
    P::tile_cur_init(ctx, rc);
    auto dma_issue = [&](int t) {    // request tile t (t < nT; its first key is in the cursor) into stage t % NS
        const unsigned k0 = (unsigned)rc.k0;
        const unsigned st = (unsigned)((t % NS) * kStage);
        lds_dma16(lds_k + st, k0 * (unsigned)D + k_src, k8);
        lds_dma16(lds_v + st, (k0 >> 6) * (unsigned)(kBN * D) + v_src, vt8);
    };
    const int dist = lagging ? 3 : 2;     // tile u + dist is requested in N(u)
    for (int t = 0; t < dist; ++t) {
        if (t < nT) dma_issue(t);
        P::tile_cur_next(ctx, rc);
    }

    // ---- Q fragments, masks, accumulators ----
    const int row_in_wg = wave * 32 + ql;
    const int q_log = P::q_logical(ctx, row_in_wg);
    const bool q_exists = P::q_phys(prm, ctx, row_in_wg) >= 0;
    i32x8 qf[KS];
    {
        const uint8_t* qrow = q8 + (size_t)(q_exists ? q_log : 0) * D + g * 32;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const i32x8*)(qrow + ks * 64);
    }
    int m_a0 = 0, m_b0 = 0;
    unsigned m_alen = 0, m_blen = 0;
    P::row_intervals(prm, ctx, q_log, m_a0, m_alen, m_b0, m_blen);
    const int ksw = (ql >> 1) & 7, vsw = (ql >> 2) & 3;
    const int k_lane = ql * 128, v_lane = kKBytes + ql * 64;
    float l_run = 0.f;
    f32x16 acc_o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;
    // S^T accumulators start at -m_off (16 registers, rewritten on the exact path only) and the Q operand's block scale is 2^e, so
    // what the MFMAs deliver is the exponent argument itself: x = scale_log2 q.k - m_off.  No per-element scale-and-shift.
    auto mfma_qk = [&](i32x8 a, i32x8 b, f32x16 c) -> f32x16 {
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, kF8Scale127, 0, q_scale);
    };

#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef SVG_F8PP_ONEBAR
#define SVG_F8PP_ONEBAR 0
#endif
    // one barrier per tile (attn_core.h kOneBar: only the barrier in front of the leading waves' matrix phase carries data) — measured
    // for this body and NOT shipped: 22.55 vs 21.98 ms (same box, round 3).  Its phases are unbalanced (matrix ~550, vector ~1000
    // cycles), so without the second barrier the two vector phases of a SIMD overlap for half a tile, and unlike the 16-bit kernel
    // this one is not at the power limit (2.33 GHz sustained): the strict opposition of the phases is worth more than the barrier.
    constexpr bool kOneBar = SVG_F8PP_ONEBAR != 0;
    const int bar_n = lagging ? 1 : 0, bar_m = lagging ? 0 : 1;
    pp_barrier();
    if (!kOneBar && lagging) pp_barrier();      // waves 4..7 run one phase behind

    const bool idle = !P::wave_active(ctx, wave * 32);
    // request half of a vector phase: tile t + dist, then wait for the pieces of the previous request
    auto stage_request = [&](int t) {
        const bool more = t + dist < nT;
        if (more) dma_issue(t + dist);
        P::tile_cur_next(ctx, rc);
        if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (idle) {      // a wave without query rows keeps the barrier / staging protocol and computes nothing
        for (int t = 0; t < nT; ++t) {
            if (!kOneBar || lagging) pp_barrier();
            stage_request(t);
            if (!kOneBar || !lagging) pp_barrier();
        }
        pp_barrier();
        if (!kOneBar && !lagging) pp_barrier();
        P::notify(prm, ctx);
        return;
    }

    f32x16 sc[2];          // S(t) until the vector phase has turned it into pf, then S(t + 1) accumulates here
    i32x8 pf = {0, 0, 0, 0, 0, 0, 0, 0};   // probabilities of tile t (e4m3, slot order of the file header); every word is rewritten
                                           // per tile with its own stale contents as the conversions' "old" operand (see attn_body_f8)
    constexpr float kPShift = 4.f, kPSumMax = 448.f;     // softmax without a running maximum: see attn_body_f8
    float m_ref = -INFINITY, m_off = -kPShift, psum = 0.f;
    float psum_thr = -1.f;   // kPSumMax once every row of the wave has a finite reference (see attn_body_pp2)
    f32x16 cneg;           // -m_off in every register: the C operand of the first QK MFMA of a tile
#pragma unroll
    for (int r = 0; r < 16; ++r) cneg[r] = kPShift;
    // cneg is rewritten IN PLACE (tied asm operands) on the exact path: written as plain assignments hipcc keeps the old and the new
    // value in two register tuples and copies one into the other on the FAST path of every tile (8 v_mov_b64 per tile)
    auto set_cneg = [&](float x) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float c = cneg[r];
            asm volatile("v_mov_b32 %0, %1" : "+v"(c) : "v"(x));
            cneg[r] = c;
        }
    };
    // First contraction step of a tile: D = A B + cneg with D != C.  Inline asm, because hipcc selects the tied form (D == C) of the
    // builtin for one of the two key blocks and pays for it with a 16-register copy of cneg per tile (8 v_mov_b64 on the vector
    // pipe, which is what bounds the fp8 kernels).  Hazards the compiler cannot see (it does not look inside): the only reader of
    // D is the next step's MFMA of the same key block, as its C operand, same tuple — legal back to back on gfx950; A / B come from
    // LDS reads the compiler waits for (it sees the operands); cneg is written on the exact path only, a barrier or a whole
    // softmax away.  tools/asm_hazards.py audits the listing (tests/test_w4_asm_audit.py).
    // (the two block-scale words live in VGPRs of their own for the whole loop — opaque to the compiler, so it cannot re-materialise
    //  one with a v_mov directly in front of the asm MFMA, where nobody would pad the VALU-write -> MFMA-read wait states)
    int scale_a_v = kF8Scale127, scale_q_v = q_scale;
    asm volatile("" : "+v"(scale_a_v), "+v"(scale_q_v));
    auto mfma_qk_first = [&](i32x8 a, i32x8 b) -> f32x16 {
        f32x16 d;
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0]"
                     : "=&v"(d)
                     : "v"(a), "v"(b), "v"(cneg), "v"(scale_a_v), "v"(scale_q_v));
        return d;
    };
    typename P::TileCur vc;          // tile of the next vector phase
    P::tile_cur_init(ctx, vc);

    auto kfrag = [&](const char* st, int i) -> i32x8 {     // QK operand i: step i >> 1, key block i & 1
        const char* rowp = st + k_lane + (i & 1) * (32 * 128);
        const int c0 = 4 * (i >> 1) + 2 * g;
        const u32x4 lo = *(const u32x4*)(rowp + ((c0 ^ ksw) << 4));
        const u32x4 hi = *(const u32x4*)(rowp + (((c0 + 1) ^ ksw) << 4));
        return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    };
    auto vfrag = [&](const char* st, int db) -> i32x8 {
        const char* rowp = st + v_lane + db * (32 * 64);
        const u32x4 lo = *(const u32x4*)(rowp + (((2 * g) ^ vsw) << 4));
        const u32x4 hi = *(const u32x4*)(rowp + (((2 * g + 1) ^ vsw) << 4));
        return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    };
    // probabilities from the exponent arguments in sc (+ delta on the exact path, where the reference has just moved)
    auto probs = [&](auto shifted_c, float delta) {
        psum = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) {
            float p4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * w8 + i;
                if constexpr (decltype(shifted_c)::value) p4[i] = __builtin_amdgcn_exp2f(sc[e >> 4][e & 15] + delta);
                else p4[i] = __builtin_amdgcn_exp2f(sc[e >> 4][e & 15]);
                psum += p4[i];
            }
            const int w = __builtin_amdgcn_cvt_pk_fp8_f32(p4[0], p4[1], pf[w8], false);   // (old = the word's stale contents: see pf)
            pf[w8] = __builtin_amdgcn_cvt_pk_fp8_f32(p4[2], p4[3], w, true);
        }
    };
    // vector phase of tile t on sc
    auto vector_phase = [&](int t) {
        const int tk0 = vc.k0;
        P::tile_cur_next(ctx, vc);
        const int cls = P::fast_full(ctx, tk0) ? (int)TILE_FULL : P::classify(prm, ctx, tk0, wave * 32);
        if (cls != TILE_FULL) {
            const bool part = (cls == TILE_PARTIAL);      // a tile this wave does not need at all is processed fully masked
            int ka = tk0 + 4 * g - m_a0, kb_ = tk0 + 4 * g - m_b0;
            asm volatile("" : "+v"(ka), "+v"(kb_));       // opaque: keeps LICM from hoisting the per-element terms out of the loop
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * b + (r & 3) + 8 * (r >> 2);
                    const bool ok = ((unsigned)(ka + key) < m_alen) | ((unsigned)(kb_ + key) < m_blen);
                    sc[b][r] = (part & ok) ? sc[b][r] : -INFINITY;
                }
        }
        probs(std::false_type{}, 0.f);
        const bool exact = __any(!(psum <= psum_thr));
        if (exact) {      // exact path (rare; always until every row has a finite reference: see attn_body_pp2)
            float mx = sc[0][0];
#pragma unroll
            for (int e = 1; e < 31; e += 2) mx = vmax3(mx, sc[e >> 4][e & 15], sc[(e + 1) >> 4][(e + 1) & 15]);
            mx = vmax2(mx, sc[1][15]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // (sc holds x = scaled score - m_off: the row maximum of the scaled scores is mx + m_off)
            const float m_prev = m_off + kPShift;
            const float m_new = fmaxf(m_ref, mx + m_off);
            const float m_use = (m_new == -INFINITY) ? m_prev : m_new;
            const float alpha = __builtin_amdgcn_exp2f(fminf(m_prev - m_use, 126.f));
            const float delta = m_prev - m_use;       // new exponent argument = x + (m_off_old - m_off_new)
            m_ref = m_new;
            psum_thr = __any(m_new == -INFINITY) ? -1.f : kPSumMax;
            m_off = m_use - kPShift;
            set_cneg(-m_off);
            probs(std::true_type{}, delta);
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;
        }
        l_run += psum;
        asm volatile("" : "+v"(pf), "+v"(l_run));      // stays in this phase
        stage_request(t);
    };
    // matrix phase of tile t: PV(t), then S(t + 1); operand i + 2 is read from LDS in front of MFMA i (sched_barrier pins the order)
    auto matrix_phase = [&](int t, auto has_next_c) {
        constexpr bool has_next = decltype(has_next_c)::value;
        constexpr int NALL = has_next ? 8 : 4, kPF = 1;     // (operands one MFMA ahead: 22.6 ms; two: 23.0; three: 23.5 — same box)
        const char* stv = smem + (t % NS) * kStage;
        const char* stk = smem + ((t + 1) % NS) * kStage;
        auto fetch = [&](int i) -> i32x8 { return i < 4 ? vfrag(stv, i) : kfrag(stk, i - 4); };
        i32x8 ring[kPF + 1];
#pragma unroll
        for (int i = 0; i < kPF; ++i) ring[i] = fetch(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NALL; ++i) {
            if (i + kPF < NALL) ring[(i + kPF) % (kPF + 1)] = fetch(i + kPF);
            if (i < 4) {
                acc_o[i] = mfma_f8(ring[i % (kPF + 1)], pf, acc_o[i]);
            } else {
                const int j = i - 4, ks = j >> 1, b = j & 1;
                if (ks == 0) sc[b] = mfma_qk_first(ring[i % (kPF + 1)], qf[ks]);
                else sc[b] = mfma_qk(ring[i % (kPF + 1)], qf[ks], sc[b]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (has_next) asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
    };

    // ---- M(0): only S(0) ----
    if (nT > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[i & 1] = mfma_qk(kfrag(smem, i), qf[i >> 1], i < 2 ? cneg : sc[i & 1]);
        asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
    }
    auto tile = [&](int t, auto has_next_c) {
        if constexpr (kOneBar) pp_barrier_if(bar_n);
        else pp_barrier();
        vector_phase(t);
        if constexpr (kOneBar) pp_barrier_if(bar_m);
        else pp_barrier();
        matrix_phase(t, has_next_c);       // (no s_setprio around it, unlike attn_body_pp2: with 8 MFMAs per phase it costs 1.5 %)
    };
    {   // (the last tile is peeled: a run-time "has next" test inside the loop makes hipcc keep two register sets for O)
        int t = 0;
        for (; t + 1 < nT; ++t) tile(t, std::true_type{});
        if (nT > 0) tile(nT - 1, std::false_type{});
    }
    // the leading waves wait until the lagging waves have read V of the last tile: the epilogue reuses the stages
    pp_barrier();
    if (!kOneBar && !lagging) pp_barrier();

    // ---------------- epilogue: as attn_body_f8 ----------------
    float l_tot = l_run + __shfl_xor(l_run, 32);
    constexpr int kEpiStride = D * 2 + 8;
    char* erow = smem + (size_t)(wave * 32) * kEpiStride;
    {
        const float inv = l_tot > 0.f ? inv_v / l_tot : 0.f;
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
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    T* __restrict__ ob = P::o_base(prm, ctx);
    constexpr int kLanesPerRow = D * 2 / 8, kRowsPerPass = 64 / kLanesPerRow, kPasses = 32 / kRowsPerPass;
    const int sub = lane / kLanesPerRow, colb = (lane - sub * kLanesPerRow) * 8;
    int ephys[kPasses];
#pragma unroll
    for (int i = 0; i < kPasses; ++i) ephys[i] = P::q_phys(prm, ctx, wave * 32 + i * kRowsPerPass + sub);
#pragma unroll
    for (int i = 0; i < kPasses; ++i) {
        const int rr = i * kRowsPerPass + sub;
        const u32x2 val = *(const u32x2*)(erow + rr * kEpiStride + colb);
        if (ephys[i] >= 0) *(u32x2*)((char*)(ob + (size_t)ephys[i] * D) + colb) = val;
    }
    P::notify(prm, ctx);
}

// =====================================================================================================================
// Gathering form of the fp8 body (SVG2 variable-block attention, BASELINE.json configs[4] as named): q, k, v stay in their
// original row order as plain row-major e4m3 tensors [H, S, D] (per-head scales) and rows are gathered through the policy
// (q_phys / kv_phys: the run list of the block-row and the sorted-index arrays), exactly like the 16-bit lock-step body.
// V is therefore row-major in LDS and V^T fragments come from the hardware transpose read ds_read_b64_tr_b8: the 16 lanes of a
// group pool their 16 x 8 bytes as an [8 rows][16 columns] byte matrix (row r = lanes 2r, 2r+1) and lane i receives column i
// (tools/probe_tr8.hip).  Lane i of a group points at key row 16 m + 8 (i >> 3) + 4 g + ((i >> 1) & 3), so the 8 bytes a lane
// gets in read m are its column's values at exactly the keys of its probability slots 8 m .. 8 m + 7.
// =====================================================================================================================
struct F8GArgs {
    const uint8_t* q8;     // [Hq, Sq, D]
    const uint8_t* k8;     // [Hkv, Skv, D]
    const uint8_t* v8;     // [Hkv, Skv, D]
    const float* q_inv;    // [Hq]   E8M0 scale word of the q head, (127 + e) * 0x01010101, as float bits (q carries the softmax scale)
    const float* kv_inv;   // [Hkv, 2]: amax / 448 of k, v
};

// chunk c (16 B) of V row `row` inside the row-major V image ([64][128] bytes): the XOR keeps the transpose reads conflict-free
__device__ __forceinline__ int f8_vrow_off(int row, int c) { return row * 128 + ((c ^ ((row & 2) | ((row >> 1) & 4))) << 4); }

template <typename T, typename P, int NW = 8>
__device__ __forceinline__ void attn_body_f8g(const typename P::Params& prm, const F8GArgs& fa, char* smem, char* policy_lds) {
    using E = Elt<T>;
    constexpr int D = 128, DB = D / 32, KS = D / 64, NT = NW * 64;
    constexpr int kKBytes = kBN * D, kStage = 2 * kBN * D;
    constexpr int NCH = 512 / NT;      // (row, 16-B chunk) pairs per thread
    static_assert(P::kRowBlocks == 1 && P::BM == NW * 32, "fp8 body: NW waves x 32 rows");

    typename P::Ctx ctx;
    if (!P::init(prm, ctx, policy_lds)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), g = lane >> 5, ql = lane & 31;
    const uint8_t* __restrict__ q8 = fa.q8 + (size_t)ctx.hq * prm.Sq * D;
    const uint8_t* __restrict__ k8 = fa.k8 + (size_t)ctx.hkv * prm.Skv * D;
    const uint8_t* __restrict__ v8 = fa.v8 + (size_t)ctx.hkv * prm.Skv * D;
    const float inv_v = fa.kv_inv[2 * ctx.hkv + 1];
    const int q_scale = __float_as_int(fa.q_inv[ctx.hq]);     // E8M0 block scale of the Q operand (q carries the softmax scale)

    const int row_in_wg = wave * 32 + ql;
    i32x8 qf[KS];
    {
        const int qp = P::q_phys(prm, ctx, row_in_wg);
        const uint8_t* qrow = q8 + (size_t)(qp >= 0 ? qp : 0) * D + g * 32;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const i32x8*)(qrow + ks * 64);
    }
    const int q_log = P::q_logical(ctx, row_in_wg);

    // staging: thread = (row kr, chunk kc) of the tile, for K and V alike (the same gathered row)
    // a thread stages NCH chunks of ONE key row (chunks c0, c0 + 8 / NCH, ...): one cursor, one index load per thread and tile
    constexpr int kTPR = 8 / NCH;                  // threads per row
    const int srow = tid / kTPR, c0 = tid % kTPR;
    int scol[NCH], k_dst[NCH], v_dst[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = c0 + i * kTPR;
        scol[i] = c * 16;
        k_dst[i] = f8_k_off(srow, c), v_dst[i] = kKBytes + f8_vrow_off(srow, c);
    }
    typename P::KvCursor cur;
    P::kv_cursor_init(prm, ctx, cur, srow);
    int nphys = 0;
    u32x4 kreg[NCH], vreg[NCH];
    const int nT = ctx.nT;
    auto resolve = [&](int t) { nphys = (t < nT) ? P::kv_phys(prm, ctx, cur, t, srow) : 0; };   // a global index load: one tile ahead
    auto issue = [&](int t) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const size_t off = (size_t)nphys * D + scol[i];
            kreg[i] = *(const u32x4*)(k8 + off);
            vreg[i] = *(const u32x4*)(v8 + off);
        }
        resolve(t + 1);
    };
    auto stage_write = [&](int buf) {
        char* base = smem + buf * kStage;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            *(u32x4*)(base + k_dst[i]) = kreg[i];
            *(u32x4*)(base + v_dst[i]) = vreg[i];
        }
    };

    const int ksw = (ql >> 1) & 7;
    const int k_lane = ql * 128;
    // transpose-read address of read m, d block db: row 16 m + vrow, byte column 32 db + vcol (+ the row's swizzle)
    const int li = lane & 15;
    const int vrow = 8 * (li >> 3) + 4 * g + ((li >> 1) & 3);
    const int vcol = 16 * ((lane >> 4) & 1) + 8 * (li & 1);
    const int vsw = (vrow & 2) | ((vrow >> 1) & 4);       // (16 m does not touch the swizzle bits)
    const int v_lane = kKBytes + vrow * 128 + (vcol & 8);

    float l_run = 0.f;
    f32x16 acc_o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;
    // (as in attn_body_f8pp: the S^T accumulators start at -m_off and the Q operand's block scale is 2^e, so the MFMAs deliver the
    //  exponent argument x = scale_log2 q.k - m_off itself)
    auto mfma_qk = [&](i32x8 a, i32x8 b, f32x16 c) -> f32x16 {
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, kF8Scale127, 0, q_scale);
    };

    resolve(0);
    if (nT > 0) {
        issue(0);
        stage_write(0);
        if (nT > 1) issue(1);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    __syncthreads();

    constexpr float kPShift = 4.f, kPSumMax = 448.f;     // softmax without a running maximum: see attn_body_f8
    float m_ref = -INFINITY, m_off = -kPShift;
    float psum_thr = -1.f;   // kPSumMax once every row of the wave has a finite reference (see attn_body_pp2)
    f32x16 cneg;           // -m_off in every register
#pragma unroll
    for (int r = 0; r < 16; ++r) cneg[r] = kPShift;
    // cneg is rewritten IN PLACE (tied asm operands) on the exact path: written as plain assignments hipcc keeps the old and the new
    // value in two register tuples and copies one into the other on the FAST path of every tile (8 v_mov_b64 per tile)
    auto set_cneg = [&](float x) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float c = cneg[r];
            asm volatile("v_mov_b32 %0, %1" : "+v"(c) : "v"(x));
            cneg[r] = c;
        }
    };
    // First contraction step of a tile: D = A B + cneg with D != C.  Inline asm, because hipcc selects the tied form (D == C) of the
    // builtin for one of the two key blocks and pays for it with a 16-register copy of cneg per tile (8 v_mov_b64 on the vector
    // pipe, which is what bounds the fp8 kernels).  Hazards the compiler cannot see (it does not look inside): the only reader of
    // D is the next step's MFMA of the same key block, as its C operand, same tuple — legal back to back on gfx950; A / B come from
    // LDS reads the compiler waits for (it sees the operands); cneg is written on the exact path only, a barrier or a whole
    // softmax away.  tools/asm_hazards.py audits the listing (tests/test_w4_asm_audit.py).
    // (the two block-scale words live in VGPRs of their own for the whole loop — opaque to the compiler, so it cannot re-materialise
    //  one with a v_mov directly in front of the asm MFMA, where nobody would pad the VALU-write -> MFMA-read wait states)
    int scale_a_v = kF8Scale127, scale_q_v = q_scale;
    asm volatile("" : "+v"(scale_a_v), "+v"(scale_q_v));
    auto mfma_qk_first = [&](i32x8 a, i32x8 b) -> f32x16 {
        f32x16 d;
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0]"
                     : "=&v"(d)
                     : "v"(a), "v"(b), "v"(cneg), "v"(scale_a_v), "v"(scale_q_v));
        return d;
    };
    i32x8 pf = {0, 0, 0, 0, 0, 0, 0, 0};   // (outside the loop: its stale words are the conversions' "old" operand, see attn_body_f8)
    int buf = 0;
    for (int t = 0; t < nT; ++t) {
        const char* kbuf = smem + buf * kStage;
        const int tk0 = P::tile_key0(ctx, t);
        const int cls = P::classify(prm, ctx, tk0, wave * 32);
        if (cls != TILE_SKIP) {
            f32x16 s_cur[2];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const char* rowp = kbuf + k_lane + b * (32 * 128);
                    const u32x4 lo = *(const u32x4*)(rowp + (((4 * ks + 2 * g) ^ ksw) << 4));
                    const u32x4 hi = *(const u32x4*)(rowp + (((4 * ks + 2 * g + 1) ^ ksw) << 4));
                    const i32x8 kf = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                    if (ks == 0) s_cur[b] = mfma_qk_first(kf, qf[ks]);
                    else s_cur[b] = mfma_qk(kf, qf[ks], s_cur[b]);
                }
            if (cls == TILE_PARTIAL) {
                asm volatile("; element-wise mask of a partial tile" ::: "memory");
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * g;
                        s_cur[b][r] = P::allowed(prm, ctx, q_log, tk0 + key) ? s_cur[b][r] : -INFINITY;
                    }
            }
            float psum;
            auto probs = [&](auto shifted_c, float delta) {
                psum = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) {
                    float p4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = 4 * w8 + i;
                        if constexpr (decltype(shifted_c)::value) p4[i] = __builtin_amdgcn_exp2f(s_cur[e >> 4][e & 15] + delta);
                        else p4[i] = __builtin_amdgcn_exp2f(s_cur[e >> 4][e & 15]);
                        psum += p4[i];
                    }
                    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(p4[0], p4[1], pf[w8], false);   // (old = the word's stale contents: see pf)
                    pf[w8] = __builtin_amdgcn_cvt_pk_fp8_f32(p4[2], p4[3], w, true);
                }
            };
            probs(std::false_type{}, 0.f);
            float mx = s_cur[0][0];
            const bool exact = __any(!(psum <= psum_thr));
            if (exact) {      // exact path (rare; always until every row has a finite reference: see attn_body_pp2)
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s_cur[0][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_cur[1][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float m_prev = m_off + kPShift;
                const float m_new = fmaxf(m_ref, mx + m_off);      // (s_cur holds x = scaled score - m_off)
                const float m_use = (m_new == -INFINITY) ? m_prev : m_new;
                const float alpha = __builtin_amdgcn_exp2f(fminf(m_prev - m_use, 126.f));
                const float delta = m_prev - m_use;
                m_ref = m_new;
                psum_thr = __any(m_new == -INFINITY) ? -1.f : kPSumMax;
                m_off = m_use - kPShift;
                set_cneg(-m_off);
                probs(std::true_type{}, delta);
                l_run *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;
            }
            l_run += psum;
            // ---------------- O^T += V^T P^T: 4 MFMAs, V^T through 4 transpose reads each ----------------
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                i32x8 vf;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int c16 = 2 * db + ((lane >> 4) & 1);     // 16-byte chunk of the row that holds this group's columns
                    const char* ap = kbuf + v_lane + m * (16 * 128) + ((c16 ^ vsw) << 4);
                    const v2i32 t2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i32*)(ap));
                    vf[2 * m] = t2[0], vf[2 * m + 1] = t2[1];
                }
                acc_o[db] = mfma_f8(vf, pf, acc_o[db]);
            }
        }
        if (t + 1 < nT) stage_write(buf ^ 1);
        if (t + 2 < nT) issue(t + 2);
        __syncthreads();
        buf ^= 1;
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    constexpr int kEpiStride = D * 2 + 8;
    char* erow = smem + (size_t)(wave * 32) * kEpiStride;
    {
        const float inv = l_tot > 0.f ? inv_v / l_tot : 0.f;
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
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    T* __restrict__ ob = P::o_base(prm, ctx);
    constexpr int kLanesPerRow = D * 2 / 8, kRowsPerPass = 64 / kLanesPerRow, kPasses = 32 / kRowsPerPass;
    const int sub = lane / kLanesPerRow, colb = (lane - sub * kLanesPerRow) * 8;
    int ephys[kPasses];
#pragma unroll
    for (int i = 0; i < kPasses; ++i) ephys[i] = P::q_phys(prm, ctx, wave * 32 + i * kRowsPerPass + sub);
#pragma unroll
    for (int i = 0; i < kPasses; ++i) {
        const int rr = i * kRowsPerPass + sub;
        const u32x2 val = *(const u32x2*)(erow + rr * kEpiStride + colb);
        if (ephys[i] >= 0) *(u32x2*)((char*)(ob + (size_t)ephys[i] * D) + colb) = val;
    }
}

}  // namespace svg
