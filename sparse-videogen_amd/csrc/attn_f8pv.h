// Mixed-precision gathering body for SVG2 (EXPERIMENTAL — written at the end of round 3, compiled, NOT yet run on a GPU; reachable only
// through svg_varblock_attention_fp8pv, which nothing in the package calls by default):
//     S^T = K Q^T on the 16-bit MFMA (bf16 / fp16 K and Q: the scores keep their accuracy),   O^T += V^T P^T on the e4m3 MFMA.
// Why: on the clustered SVG2 data the error of the all-e4m3 kernel is the SCORES' (tools/fp8_precision_study.py, DESIGN.md §3.1.2: all
// e4m3 8.6 %, 16-bit QK^T + e4m3 PV 3.6 %), and in cycles the fp8 kernels are bound by the vector pipe, not by the matrix pipe
// (tools/clock_by_variant.py), so 16 bf16 MFMAs instead of 4 e4m3 ones for QK^T cost far less than their ratio.
//
// It is attn_body_f8g (attn_f8.h) with the K / Q side taken from the 16-bit lock-step body:
//   * q16  [Hq, Sq, D]  T : q * sm_scale * log2(e), rounded once more to T by the pre-pass (the S^T accumulators start at minus the
//                           row's reference, so the MFMAs deliver the exponent argument: no per-score scale-and-shift) — rows gathered
//                           through the policy like every other q;
//   * k    [Hkv, Skv, D] T : the caller's ORIGINAL tensor, rows gathered through the policy, staged as the 16-bit K image
//                           ([64 keys][256 B], chunk XOR (row & 15): LdsLayout<128>::k_off);
//   * v8   [Hkv, Skv, D] e4m3 (x 448 / amax_head) from the pre-pass, rows gathered, staged row-major, V^T fragments by
//                           ds_read_b64_tr_b8 — exactly attn_body_f8g's V side;
//   * probabilities: e4m3 in the slot order of attn_f8.h, range test and exact path as in attn_body_f8g.
// Stage = 16 KiB K + 8 KiB V; two register-staged stages, one barrier per tile; NW waves x 32 rows.
#pragma once
#include "attn_f8.h"

namespace svg {

struct F8PVArgs {
    const void* q16;       // [Hq, Sq, D] T, pre-scaled
    const uint8_t* v8;     // [Hkv, Skv, D] e4m3
    const float* v_inv;    // [Hkv] amax / 448 of v
};

template <int NW>
constexpr int attn_f8pv_lds_bytes() {
    constexpr int stages = 2 * (kBN * 128 * 2 + kBN * 128);      // 2 x (K image 16 KiB + V image 8 KiB)
    constexpr int epi = NW * 32 * (128 * 2 + 8);
    return stages > epi ? stages : epi;
}

template <typename T, typename P, int NW = 4>
__device__ __forceinline__ void attn_body_f8pv(const typename P::Params& prm, const F8PVArgs& fa, char* smem, char* policy_lds) {
    using E = Elt<T>;
    using V8 = typename E::v8;
    using L = LdsLayout<128>;
    constexpr int D = 128, DB = D / 32, KS = D / 16, NT = NW * 64;
    constexpr int kKBytes = kBN * D * 2, kVBytes = kBN * D, kStage = kKBytes + kVBytes;
    constexpr int kTPR = NT / kBN;                 // threads per key row (NW)
    constexpr int NCK = 16 / kTPR, NCV = 8 / kTPR;  // 16-byte chunks of the K row (256 B) / the V row (128 B) per thread
    static_assert(P::kRowBlocks == 1 && P::BM == NW * 32 && (NW == 4 || NW == 8), "mixed body: NW waves x 32 rows");

    typename P::Ctx ctx;
    if (!P::init(prm, ctx, policy_lds)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), g = lane >> 5, ql = lane & 31;
    const T* __restrict__ q16 = (const T*)fa.q16 + (size_t)ctx.hq * prm.Sq * D;
    const T* __restrict__ kb = P::k_base(prm, ctx);
    const uint8_t* __restrict__ v8 = fa.v8 + (size_t)ctx.hkv * prm.Skv * D;
    const float inv_v = fa.v_inv[ctx.hkv];

    const int row_in_wg = wave * 32 + ql;
    V8 qf[KS];
    {
        const int qp = P::q_phys(prm, ctx, row_in_wg);
        const T* qrow = q16 + (size_t)(qp >= 0 ? qp : 0) * D + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const V8*)(qrow + ks * 16);
    }
    const int q_log = P::q_logical(ctx, row_in_wg);

    // staging: kTPR threads share one gathered key row (one cursor, one index load per thread and tile); thread c0 of them takes the
    // chunks c0, c0 + kTPR, ... of the K row and of the V row
    const int srow = tid / kTPR, c0 = tid % kTPR;
    int k_dst[NCK], v_dst[NCV];
#pragma unroll
    for (int i = 0; i < NCK; ++i) k_dst[i] = L::k_off(srow, c0 + i * kTPR);
#pragma unroll
    for (int i = 0; i < NCV; ++i) v_dst[i] = kKBytes + f8_vrow_off(srow, c0 + i * kTPR);
    typename P::KvCursor cur;
    P::kv_cursor_init(prm, ctx, cur, srow);
    int nphys = 0;
    u32x4 kreg[NCK], vreg[NCV];
    const int nT = ctx.nT;
    auto resolve = [&](int t) { nphys = (t < nT) ? P::kv_phys(prm, ctx, cur, t, srow) : 0; };
    auto issue = [&](int t) {
        const char* krow = (const char*)(kb + (size_t)nphys * D);
        const uint8_t* vrow = v8 + (size_t)nphys * D;
#pragma unroll
        for (int i = 0; i < NCK; ++i) kreg[i] = *(const u32x4*)(krow + (c0 + i * kTPR) * 16);
#pragma unroll
        for (int i = 0; i < NCV; ++i) vreg[i] = *(const u32x4*)(vrow + (c0 + i * kTPR) * 16);
        resolve(t + 1);
    };
    auto stage_write = [&](int buf) {
        char* base = smem + buf * kStage;
#pragma unroll
        for (int i = 0; i < NCK; ++i) *(u32x4*)(base + k_dst[i]) = kreg[i];
#pragma unroll
        for (int i = 0; i < NCV; ++i) *(u32x4*)(base + v_dst[i]) = vreg[i];
    };

    const int ksw0 = ql & 15;                      // K image swizzle of row 32 b + ql (32 % 16 == 0: the same for both key blocks)
    // transpose-read addresses of the V image: as attn_body_f8g
    const int li = lane & 15;
    const int vrow_l = 8 * (li >> 3) + 4 * g + ((li >> 1) & 3);
    const int vcol = 16 * ((lane >> 4) & 1) + 8 * (li & 1);
    const int vsw = (vrow_l & 2) | ((vrow_l >> 1) & 4);
    const int v_lane = kKBytes + vrow_l * 128 + (vcol & 8);

    float l_run = 0.f;
    f32x16 acc_o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;

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
    float psum_thr = -1.f;
    f32x16 cneg;           // -m_off in every register: what the S^T accumulators start from
#pragma unroll
    for (int r = 0; r < 16; ++r) cneg[r] = kPShift;
    auto set_cneg = [&](float x) {       // in place (tied operands): see attn_body_f8g
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float c = cneg[r];
            asm volatile("v_mov_b32 %0, %1" : "+v"(c) : "v"(x));
            cneg[r] = c;
        }
    };
    i32x8 pf = {0, 0, 0, 0, 0, 0, 0, 0};
    int buf = 0;
    for (int t = 0; t < nT; ++t) {
        const char* kbuf = smem + buf * kStage;
        const int tk0 = P::tile_key0(ctx, t);
        const int cls = P::classify(prm, ctx, tk0, wave * 32);
        if (cls != TILE_SKIP) {
            // ---------------- S^T = K Q^T: 16 MFMAs on the 16-bit type (2 key blocks x 8 contraction steps of 16) ----------------
            f32x16 s_cur[2];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cch = ((2 * ks + g) ^ ksw0) << 4;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const V8 a = *(const V8*)(kbuf + (32 * b + ql) * L::kRowBytes + cch);
                    if (ks == 0) s_cur[b] = E::mfma_keep_c(a, qf[ks], cneg);      // (hazards: as in attn_body_pp2 — read next by the same block's MFMA as C)
                    else s_cur[b] = E::mfma(a, qf[ks], s_cur[b]);
                }
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
                    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(p4[0], p4[1], pf[w8], false);
                    pf[w8] = __builtin_amdgcn_cvt_pk_fp8_f32(p4[2], p4[3], w, true);
                }
            };
            probs(std::false_type{}, 0.f);
            if (__any(!(psum <= psum_thr))) {      // exact path (rare; always until every row has a finite reference)
                float mx = s_cur[0][0];
#pragma unroll
                for (int e = 1; e < 31; e += 2) mx = vmax3(mx, s_cur[e >> 4][e & 15], s_cur[(e + 1) >> 4][(e + 1) & 15]);
                mx = vmax2(mx, s_cur[1][15]);
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
            // ---------------- O^T += V^T P^T: 4 e4m3 MFMAs, V^T through 4 transpose reads each ----------------
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                i32x8 vf;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int c16 = 2 * db + ((lane >> 4) & 1);
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
