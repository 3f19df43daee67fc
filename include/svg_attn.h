/*
 * svg_attn.h — C ABI of libsvgattn.so, the MI355X (gfx950) sparse-attention engine that sits behind the
 * attention-processor / operator API of svg-project/Sparse-VideoGen.
 *
 * Every entry point is `extern "C"`, takes raw device pointers + sizes + an opaque hipStream_t (as void*),
 * launches asynchronously on that stream, never allocates, never synchronises, never throws.  The return
 * value is 0 on success or a negative SVG_ERR_* code (svg_strerror() gives text).  Tensors are owned by the
 * caller.  "ref:" lines cite the reference interface (path:line under the Sparse-VideoGen tree) that the
 * entry point replaces; INTEGRATION.md shows the Python-side binding a maintainer would add.
 *
 * Layout conventions (same as the reference): Q/K/V/O are contiguous [cfg*H, S, D] ("BH, S, D"), element
 * type bf16 or fp16 (dtype code), D in {64, 128}.  Token order along S is the model's: Hunyuan/Wan/Cosmos
 * video tokens first (frame-major, idx = f*P + p) then text; CogVideoX text first.
 */
#ifndef SVG_ATTN_H_
#define SVG_ATTN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVG_DTYPE_BF16 0
#define SVG_DTYPE_F16 1
#define SVG_DTYPE_F32 2   /* only where an entry point says so (block glue: fp32 LayerNorm outputs, fp32 weights) */

#define SVG_OK 0
#define SVG_ERR_BAD_ARG (-1)       /* null pointer, negative size, inconsistent geometry          */
#define SVG_ERR_UNSUPPORTED (-2)   /* head_dim / dtype / size outside what the kernels implement  */
#define SVG_ERR_WORKSPACE (-3)     /* caller-provided workspace too small                          */
#define SVG_ERR_LAUNCH (-4)        /* hipLaunchKernel / attribute call failed (see svg_last_hip_error) */

/* ABI version of this header: bumped whenever an existing entry point's signature or a struct layout changes (new entry points do
 * not bump it).  A binding compares svg_abi_version() of the library it loaded with the SVG_ABI_VERSION it was written against and
 * refuses to call into a mismatch (svg/_native.py load()).  4: round 4 (svg_band_attention_notify* carry `done_words`). */
#define SVG_ABI_VERSION 4
int svg_abi_version(void);

const char* svg_strerror(int code);
int svg_last_hip_error(void);          /* raw hipError_t of the last failing launch in this thread */
const char* svg_build_info(void);      /* "libsvgattn gfx950 <date> ..."                           */

/* ------------------------------------------------------------------------------------------------
 * Layout transformation ("placement").
 * ref: svg/models/hyvideo/placement.py:124-153 (hunyuan_sparse_head_placement),
 *      svg/models/hyvideo/placement.py:360-387 (hunyuan_hidden_states_placement),
 *      svg/models/cog/placement.py:80-82 (text-first variant), svg/models/wan/placement.py (ctx = 0).
 * For every (cfg, head) with best_mask_idx != 0 ("temporal head") video row f*P+p moves to p*F+f
 * (forward) or p*F+f moves to f*P+p (inverse); text rows and "spatial" heads are copied unchanged.
 * best_mask_idx: device int64 [BH].  text_first = 0: text is the last `context_length` rows
 * (Hunyuan/Wan), 1: the first `context_length` rows (CogVideoX).  n_tensors in {1,2,3}: the same
 * transformation is applied to up to three tensors in one launch (Q,K,V).
 * ---------------------------------------------------------------------------------------------- */
int svg_head_placement(const void* const* src, void* const* dst, int32_t n_tensors, const int64_t* best_mask_idx,
                       int32_t BH, int32_t S, int32_t D, int32_t dtype, int32_t context_length, int32_t num_frame,
                       int32_t frame_size, int32_t text_first, int32_t inverse, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Token permutation (gather) and inverse permutation (scatter) of rows.
 * ref: svg/kernels/triton/permute.py:12-43,82-128 (Y[bh,s,:] = X[bh, idx[bh,s], :])
 *      svg/kernels/triton/permute.py:46-75,131-170 (Y[bh, idx[bh,s], :] = X[bh,s,:])
 * idx: device int32 [BH, S].
 * ---------------------------------------------------------------------------------------------- */
int svg_permute_rows(const void* x, const int32_t* idx, void* y, int32_t BH, int32_t S, int32_t D, int32_t dtype,
                     void* stream);
int svg_inverse_permute_rows(const void* x, const int32_t* idx, void* y, int32_t BH, int32_t S, int32_t D,
                             int32_t dtype, void* stream);

/* Stable argsort of cluster labels (counting sort): sorted_idx[b, :] = argsort(labels[b, :], stable),
 * counts[b, k] = |{n : labels[b,n] == k}|.  This fixes the tie order the reference leaves unspecified
 * (torch.argsort without stable=True, ref: svg/kernels/triton/permute.py:113, svg/kmeans_utils.py:396).
 * labels: device int32 [B, N] with values in [0, K); sorted_idx: int32 [B, N]; counts: int32 [B, K] (may be NULL).
 * workspace: svg_argsort_workspace_bytes(B, N, K) bytes of device scratch. */
size_t svg_argsort_workspace_bytes(int32_t B, int32_t N, int32_t K);
int svg_argsort_labels(const int32_t* labels, int32_t* sorted_idx, int32_t* counts, int32_t B, int32_t N, int32_t K,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SVG1 block-sparse attention (and the dense baseline) — one analytic mask family.
 * ref: flex_attention(q, k, v, block_mask) at svg/models/hyvideo/attention.py:401-403 with the mask_mod of
 *      svg/models/hyvideo/utils.py:20-44, svg/models/wan/utils.py:25-41, svg/models/cog/utils.py:30-46;
 *      dense: flash_attn_varlen_func with cu_seqlens [0, valid, S], svg/models/hyvideo/attention.py:452-470.
 * allowed(q, k) = (q < real_len && k < real_len &&
 *                  (|q - k| < band || colfull_lo <= k < colfull_hi || rowfull_lo <= q < rowfull_hi))
 *              || (q >= real_len && k >= real_len)
 *   Hunyuan : real_len = V + prompt_len, band = floor(w*P/128)*128, colfull = rowfull = [V, real_len)
 *   Wan     : real_len = S, band = ceil(w*P/128)*128 + 1 (the reference uses <=), colfull = [0, P), rowfull = {}
 *   Cog     : real_len = S, band = floor(w*P/128)*128, colfull = [0, Lp (+P if sink)), rowfull = [0, Lp)
 *   dense   : band = S + 1 (two segments when real_len < S, i.e. cu_seqlens [0, real_len, S])
 * Rows with no allowed key produce zeros.  The mask is evaluated element-wise on band-edge tiles, exactly as
 * flex_attention applies mask_mod inside partial blocks.
 *
 * Fused layout transformation: when head_perm_flag != NULL, heads with head_perm_flag[bh] != 0 are processed
 * in token-major order *without* materialising the permuted tensors: logical row i (< perm_V, offset by
 * perm_vid0) is read from / written to physical row vid0 + (i%F)*P + i/F.  The mask is in logical order.
 * This equals placement -> attention -> inverse placement of the reference (attention.py:514-520).
 * ---------------------------------------------------------------------------------------------- */
typedef struct svg_band_mask {
    int32_t real_len;
    int32_t band;
    int32_t colfull_lo, colfull_hi;
    int32_t rowfull_lo, rowfull_hi;
} svg_band_mask_t;

typedef struct svg_perm_desc {
    const int64_t* head_perm_flag; /* device int64 [BH] (best_mask_idx) or NULL                 */
    int32_t vid0;                  /* first video row (0 text-last, context_length text-first)  */
    int32_t num_frame;             /* F                                                          */
    int32_t frame_size;            /* P                                                          */
} svg_perm_desc_t;

/* variant: 0 = default (= 8 at head_dim 128; = 2 at head_dim 64, where the two-phase body runs four waves per SIMD — two workgroups per CU).  All schedules produce the same result up to rounding.
 *   1 = lock-step, 4 waves x 32 query rows, 128-row q-tiles, two workgroups per CU (register-staged K/V) — the plain schedule
 *       the test-suite uses as the in-library reference;
 *   2 = two-phase ping-pong, 8 waves x 32 rows: the two waves that share a SIMD alternate a matrix phase (PV of tile t + QK^T
 *       of tile t+1, operands streamed from LDS) and a vector phase (softmax without a running maximum, LDS-DMA requests), always
 *       in opposite phases;
 *   3 = one wave per SIMD, 4 waves x 64 rows, O and Q in the AGPR half of the register file, softmax / LDS reads / LDS-DMA
 *       requests software-pipelined into the gaps between the wave's own MFMAs, one barrier per tile (csrc/attn_w4.h).
 *   8 = the two-phase schedule of 2 on v_mfma_f32_16x16x32 instead of 32x32x16 (head_dim 128 only; csrc/attn_m16.h): the 16-bit
 *       attention kernels run at the chip's power limit, and the 16x16x32 shape does the same FLOPs with half the accumulator
 *       traffic — the matrix pipe alone is granted 2.15 GHz instead of 1.75 on changing operands (profiles/r04c_energy_table.txt).
 *   6 = frozen reference schedule (bf16, D = 128 only): the two-phase body with the round-1 softmax (running maximum, deferred
 *       rescale) and operand fetch, kept so that a bench run can time it beside the default on the same box.
 * Any other value: SVG_ERR_BAD_ARG.  Builds with -DSVG_ABLATIONS (diagnostics, never the product library) additionally accept
 * 32: variant 3 with the per-phase cycle trace (svg_debug_pp_trace), and
 * 64 | (abl << 8): variant 2 with the per-phase cycle trace / launch timeline (svg_debug_pp_trace, svg_debug_wg_trace) and its
 * timing-only ablations abl = 1..7 (wrong results by construction); the product library returns SVG_ERR_UNSUPPORTED for them. */
int svg_band_attention(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                       int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                       int32_t variant, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Strided tensors: attention straight out of / into the projection layout.
 * ref: the reference's processors build q, k, v as `proj(x).unflatten(2, (heads, -1)).transpose(1, 2)` and hand the result back as
 *      `hidden_states.transpose(1, 2).flatten(2, 3)` (svg/models/wan/attention.py:123-125,168-170, hyvideo/attention.py:83-85,202) —
 *      flex_attention / flash-attn take the strided VIEWS, and what they cannot take is copied: one transpose copy per tensor and
 *      layer (V in, O out: 2 x 2 x S x H x D x 2 bytes that carry no arithmetic).
 * A tensor here is a [B, H, S, D] view whose last dimension is contiguous; strides are in ELEMENTS.  Head index bh = b * H + h with
 * H = heads_per_batch (k / v of svg_varblock_attention_strided: kv_heads_per_batch).  The contiguous [BH, S, D] layout of every other
 * entry point is { batch = H * S * D, head = S * D, row = D }; the projection layout [B, S, H * D] is { S * H * D, D, H * D }; a slice
 * of a fused QKV projection [B, S, 3 * H * D] is { 3 * S * H * D, D, 3 * H * D } behind a base pointer moved to the slice.
 * Requirements: base pointers and row strides multiples of 16 bytes; S * row stride * 2 < 2^32 for k and v (their LDS-DMA requests
 * carry 32-bit byte offsets per head), row strides < 2^23 elements.  The default schedules only — the two-phase bodies: 16x16x32 at
 * head_dim 128, 32x32x16 at head_dim 64 (variant 0 of svg_band_attention, the default of svg_varblock_attention on block-rows large
 * enough for it; svg_sample_mse: its second form, bf16); anything else returns SVG_ERR_UNSUPPORTED and the caller copies, as the
 * reference does.  Results are bit-identical to the contiguous call on the same values (tests/test_gpu_strided.py).
 * ---------------------------------------------------------------------------------------------- */
typedef struct svg_tensor_strides {
    int64_t batch, head, row;          /* element strides of a [B, H, S, D] view, stride(D) == 1 */
} svg_tensor_strides_t;

typedef struct svg_attn_layout {
    int32_t heads_per_batch;           /* H of q / o (and of k / v unless kv_heads_per_batch says otherwise) */
    int32_t kv_heads_per_batch;        /* 0: = heads_per_batch */
    svg_tensor_strides_t q, k, v, o;
} svg_attn_layout_t;

int svg_band_attention_strided(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                               int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                               const svg_attn_layout_t* layout, void* stream);

/* svg_band_attention for a q that already carries the softmax scale: q_scaled = q * sm_scale * log2(e), rounded ONCE to the 16-bit
 * type by whoever produced q (svg_qk_norm_rope* with q_scale — the processors' prologue — so no second rounding happens on that
 * path).  The kernel then starts its score accumulators at minus the row's softmax reference and the MFMAs deliver the exponent
 * argument directly: one FMA per score less on the vector pipe (the PRE form of the default two-phase schedule: on 16x16x32 MFMAs at
 * D = 128, on 32x32x16 at D = 64).  Same mask / perm semantics and the same result as svg_band_attention(q, ..., sm_scale) up to the
 * rounding of q_scaled — which is the catch: the scores come from round(c * q) instead of c * round(q), a second 2^-9 perturbation on
 * bf16 (2.8e-3 ... 6.4e-3 rel. L2 to the reference's formulation where svg_band_attention holds 1.9 - 2.3e-3; fp16: 3.5e-4 ... 8e-4 vs
 * 2.4 - 2.9e-4).  It is what flex_attention calls its PRESCALE_QK kernel option ("about 20% more numerical error, but slightly faster");
 * the reference runs flex_attention with the default, PRESCALE_QK = False (svg/models/hyvideo/attention.py:401-403), and so do the
 * processors of this package unless `prescale_q` is switched on. */
int svg_band_attention_prescaled(const void* q_scaled, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                 int32_t dtype, const svg_band_mask_t* mask, const svg_perm_desc_t* perm, void* stream);

/* svg_band_attention_switch (device-side dense / sparse switch, below) for a pre-scaled q. */
int svg_band_attention_switch_prescaled(const void* q_scaled, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                        int32_t dtype, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                        const svg_band_mask_t* alt_mask, const int32_t* use_alt_flag, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SVG2 variable-block sparse attention.
 * ref: dynamic_block_sparse_fwd_flashinfer, svg/kmeans_utils.py:1319-1392 (VariableBlockSparseAttentionWrapper
 *      plan/run, patch assets/patches/modifications.patch:18-78); semantics = dynamic_block_sparse_fwd_torch,
 *      svg/kmeans_utils.py:902-995: query rows of block-row i attend exactly the kv rows of block-cols j with
 *      block_map[h, i, j]; blocks are consecutive ranges of sizes q_sizes[h, :], k_sizes[h, :]; rows with no
 *      active key (and rows of empty blocks) give zeros.
 * q/o: [Hq, Sq, D], k/v: [Hkv, Skv, D] (Hq % Hkv == 0: GQA group shares map/sizes of its kv head).
 * block_map: device uint8/bool [Hkv, QB, KB]; q_sizes int32 [Hkv, QB]; k_sizes int32 [Hkv, KB].  KB <= 4032 (the compacted key-block list of a
 * block-row lives in LDS beside the K / V stages; more returns SVG_ERR_UNSUPPORTED).
 * Fused token permutation: q_row_idx (int32 [Hq, Sq]) / kv_row_idx (int32 [Hkv, Skv]) map a *permuted* position
 * to the physical row of q,o / k,v (the `sorted_indices` of permute_tensor_by_labels); NULL = tensors are
 * already permuted.  With both given the call equals permute(q,k,v) -> attention -> inverse_permute(o)
 * (ref: svg/models/hyvideo/attention.py:651-653,778-783).
 * variant: -1 = auto (what the Python layer passes), 0 = 128-row q tiles (4 waves, two workgroups per CU), 1 = 256-row q tiles
 * (8 waves, lock-step), 2 = mixed (full 256-row tiles on 8 waves, the rest of each block-row on 128-row tiles; two launches),
 * 3 = 256-row q tiles with the two-phase ping-pong body of svg_band_attention (waves without query rows idle), workgroups
 * launched longest-first inside every kv head (device-side counting sort on the active keys of the block-rows), and REMAINDER
 * PACKING: the ragged last tiles of two block-rows of a kv head that fit into one tile and share many key blocks (a device-side
 * matching on the map's bitmap rows) run as ONE tile that walks the common key blocks once — exact: a row sees its own block-row's
 * keys only (two key intervals per row); 6 = variant 3 without the packing (the default of round 2);
 * 4 = the same kernel in block-row order (A/B measurements); 5 = variant 3 recording the launch timeline (svg_debug_wg_trace);
 * at head_dim 128 the two-phase body of 3 .. 7 runs on 16x16x32 MFMAs (csrc/attn_m16.h) — 8 = 3 naming that body, 9 = 3 forcing the
 * 32x32x16 body (A/B);
 * 7 = similarity order: block-rows with (nearly) the same active key blocks are neighbours of a device-built nearest-neighbour
 * chain and consecutive workgroups go to the same XCD so that they meet in its L2 (maps whose bitmap does not fit the chain
 * kernel's 64 KiB of LDS — QB * (KB / 32 + 4) words — fall back to 3).  Measured at Wan 2.1 720p (profiles/r03b_pmc_svg2_*):
 * L2 hit rate 48 % instead of 31 %, 90 GB instead of 119 GB between L2 and the fabric, kernel time unchanged, chain kernel
 * 0.7 - 1.0 ms — kept for the traffic it saves when the fabric is shared (multi-GPU exchange), not the default.
 * ---------------------------------------------------------------------------------------------- */
size_t svg_varblock_workspace_bytes(int32_t Hq, int32_t Hkv, int32_t QB, int32_t KB, int32_t Sq);
int svg_varblock_attention(const void* q, const void* k, const void* v, void* o, int32_t Hq, int32_t Hkv, int32_t Sq,
                           int32_t Skv, int32_t D, int32_t dtype, float sm_scale, const uint8_t* block_map,
                           const int32_t* q_sizes, const int32_t* k_sizes, int32_t QB, int32_t KB,
                           const int32_t* q_row_idx, const int32_t* kv_row_idx, void* workspace,
                           size_t workspace_bytes, int32_t variant, void* stream);
/* svg_varblock_attention (variant -1) on strided q, k, v, o: svg_attn_layout_t above; heads_per_batch counts q heads,
 * kv_heads_per_batch kv heads (0: heads_per_batch * Hkv / Hq).  Block-rows large enough for the default (two-phase) body
 * (Sq >= 160 * QB), otherwise SVG_ERR_UNSUPPORTED. */
int svg_varblock_attention_strided(const void* q, const void* k, const void* v, void* o, int32_t Hq, int32_t Hkv, int32_t Sq,
                                   int32_t Skv, int32_t D, int32_t dtype, float sm_scale, const uint8_t* block_map,
                                   const int32_t* q_sizes, const int32_t* k_sizes, int32_t QB, int32_t KB,
                                   const int32_t* q_row_idx, const int32_t* kv_row_idx, void* workspace,
                                   size_t workspace_bytes, const svg_attn_layout_t* layout, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Online profiler (SVG1): mean-squared error of the two candidate masks on sampled query rows.
 * ref: sample_mse, svg/models/hyvideo/attention.py:376-399 (wan :211-234, cog :120-145) with the profiling masks
 *      of get_attention_mask, svg/models/hyvideo/utils.py:47-93 (wan/utils.py:63-110, cog/utils.py:61-88),
 *      evaluated analytically instead of from two materialised [10000, S] fp32 masks.
 * rows: device int64 [R] sampled query rows (R <= 64 multiple-of-anything; padded internally).
 * profile mask (block = 128): |floor(x/128) - floor(y/128)| < band_blocks with x, y the row / column index in
 * frame-major order (mask 0, "spatial") or in token-major order (mask 1, "temporal"); the per-model quirks of
 * the reference masks (Wan sink columns, Cog's un-offset spatial band and text-less temporal mask) are expressed
 * through svg_profile_variant_t.  A sampled row whose mask admits no key yields NaN like the reference's softmax
 * over all -inf.  out_mse: device float [2, BH].
 * Numerics (csrc/profiler.hip, second form): one score tile serves the dense rows and both masks — the masked softmaxes use
 * the dense rows' running maximum as their reference (exact in real arithmetic; a masked row loses keys that score more than
 * 129 (bf16; fp16: 27) in logit below the row's overall maximum), and a row's denominator is the sum of the probabilities as
 * the numerator sees them (rounded to the input dtype).  The rows are processed in the order of their coordinate under mask 1;
 * the result does not depend on the order they are passed in (up to fp32 summation order over rows).
 * workspace: svg_sample_mse_workspace_bytes(BH, R, D, S).
 * ---------------------------------------------------------------------------------------------- */
typedef struct svg_profile_variant {
    int32_t coord;        /* 0: indices as stored (frame-major); 1: token-major (p*F + f) inside the video range */
    int32_t origin;       /* subtracted from the coordinate before blocking                                    */
    int32_t span;         /* band domain: 0 <= x - origin < span                                               */
    int32_t band_blocks;  /* |floor(x/128) - floor(y/128)| < band_blocks                                       */
    int32_t sink_cols;    /* keys with y < sink_cols always visible (Wan first-frame sink), in variant coords  */
    int32_t text_lo, text_hi; /* rows / cols in [text_lo, text_hi) are all-ones (empty when lo >= hi)          */
} svg_profile_variant_t;

typedef struct svg_profile_desc {
    int32_t vid0;         /* first video row                                   */
    int32_t num_frame;    /* F                                                 */
    int32_t frame_size;   /* P                                                 */
    int32_t emulate_bf16; /* 1: round scores / outputs to the input dtype like the reference's bf16 torch ops */
    svg_profile_variant_t variant[2]; /* [0] "spatial" mask, [1] "temporal" mask */
} svg_profile_desc_t;

size_t svg_sample_mse_workspace_bytes(int32_t BH, int32_t R, int32_t D, int32_t S);
int svg_sample_mse(const void* q, const void* k, const void* v, const int64_t* rows, int32_t R, int32_t BH, int32_t S,
                   int32_t D, int32_t dtype, float sm_scale, const svg_profile_desc_t* prof, float* out_mse,
                   void* workspace, size_t workspace_bytes, void* stream);
/* svg_sample_mse[_flagged] on strided q, k, v (svg_attn_layout_t above; layout->o is ignored).  skip_flag: NULL, or the device flag
 * of svg_sample_mse_flagged.  bf16 only (the second form of the kernel); fp16 with row strides other than D: SVG_ERR_UNSUPPORTED. */
int svg_sample_mse_strided(const void* q, const void* k, const void* v, const int64_t* rows, int32_t R, int32_t BH, int32_t S,
                           int32_t D, int32_t dtype, float sm_scale, const svg_profile_desc_t* prof, float* out_mse,
                           void* workspace, size_t workspace_bytes, const int32_t* skip_flag, const svg_attn_layout_t* layout,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * flash-kmeans (Euclidean Lloyd iterations, batched over heads).
 * ref: batch_kmeans_Euclid, svg/kmeans_utils.py:684-733; assign kernel :464-554; sorted centroid update
 *      :258-322,375-421.  One call = one Lloyd iteration, entirely on device:
 *   labels[b,n]   = argmin_k max(0, xsq[b,n] + csq[b,k] - 2 <x[b,n], c[b,k]>)   (lowest index wins ties)
 *   c_new[b,k]    = mean of assigned points (fp32 accumulate, deterministic order), old centroid if empty
 *   counts[b,k]   = cluster sizes;  sorted_idx = stable argsort(labels);  shift[b] = max_k ||c_new - c||
 * x: [B, N, D]; centroids_in/out: [B, K, D] (same dtype as x); labels int32 [B, N]; counts int32 [B, K];
 * sorted_idx int32 [B, N]; shift float [B].  xsq float [B, N] is the reference's `x_sq` (svg_kmeans_xsq; constant over
 * iterations, ref :704) and MAY BE NULL: the kernel assigns by argmax_k (<x, c_k> - |c_k|^2 / 2), which has the same arg-min as
 * the distance form and never reads |x|^2 — pass it only if you compute it anyway.
 * ---------------------------------------------------------------------------------------------- */
int svg_kmeans_xsq(const void* x, float* xsq, int32_t B, int32_t N, int32_t D, int32_t dtype, void* stream);
size_t svg_kmeans_workspace_bytes(int32_t B, int32_t N, int32_t K, int32_t D);
int svg_kmeans_iter(const void* x, const float* xsq, const void* centroids_in, void* centroids_out, int32_t* labels,
                    int32_t* counts, int32_t* sorted_idx, float* shift, int32_t B, int32_t N, int32_t K, int32_t D,
                    int32_t dtype, void* workspace, size_t workspace_bytes, void* stream);

/* The whole Lloyd loop of batch_kmeans_Euclid (ref: svg/kmeans_utils.py:684-733) as ONE call, launches only — no host
 * synchronisation: max_iters x (svg_kmeans_iter into scratch + a commit kernel that applies the reference's stopping rule on the
 * device).  Iteration `it` assigns with the current centroids and computes new ones; while the loop has not stopped its labels /
 * sizes / stable sorted indices are the result; if the largest centre shift (over all batches) is < tol the loop stops THERE and the
 * OLD centroids are the result (:723-724), otherwise the new centroids become current — without convergence the returned centroids
 * are one update ahead of the returned labels, exactly like the reference.  Later iterations still run (no read-back decides how
 * many to launch) but cannot change the result.  n_iters: device int32, the iterations the reference's loop would have run.
 * c_init [B, K, D] is not written; c_work_a / c_work_b [B, K, D] are scratch; centroids_out [B, K, D]; labels / sorted_idx int32
 * [B, N]; counts int32 [B, K].  Replaces ~10 framework micro-launches per iteration of the host-driven loop. */
/* The two halves of svg_kmeans_iter as entry points of their own (host plumbing added at the end of round 3; the kernels are the ones
 * svg_kmeans_iter launches): ref euclid_assign_triton svg/kmeans_utils.py:562-627 and triton_centroid_update_sorted_euclid :375-421.
 * workspace: svg_kmeans_workspace_bytes(B, N, K, D).  labels int32 [B, N]; counts int32 [B, K]; sorted_idx int32 [B, N]; shift float [B]. */
int svg_kmeans_assign(const void* x, const void* centroids, int32_t* labels, int32_t B, int32_t N, int32_t K, int32_t D, int32_t dtype,
                      void* workspace, size_t workspace_bytes, void* stream);
int svg_kmeans_update(const void* x, const int32_t* labels, const void* centroids_in, void* centroids_out, int32_t* counts,
                      int32_t* sorted_idx, float* shift, int32_t B, int32_t N, int32_t K, int32_t D, int32_t dtype, void* workspace,
                      size_t workspace_bytes, void* stream);
size_t svg_kmeans_loop_workspace_bytes(int32_t B, int32_t N, int32_t K, int32_t D);
int svg_kmeans_loop(const void* x, const float* xsq, const void* c_init, void* c_work_a, void* c_work_b, int32_t* labels,
                    int32_t* counts, int32_t* sorted_idx, void* centroids_out, int32_t* n_iters, int32_t B, int32_t N, int32_t K,
                    int32_t D, int32_t dtype, int32_t max_iters, float tol, void* workspace, size_t workspace_bytes, void* stream);
/* svg_kmeans_loop on an x whose batches are `x_batch_stride` elements apart (rows of a batch contiguous, stride D): the video tokens of a
 * [H, S, D] tensor with text tokens behind them — `q[:, :V]` — without the copy the reference makes (`query[:, :, :-context_length, :]`
 * materialised by its Triton kernels' `.contiguous()`, svg/models/hyvideo/attention.py:592-599).  x_batch_stride >= N * D, a multiple of
 * 8, x 16-byte aligned; no xsq argument (the assignment does not read |x|^2).  Same result as svg_kmeans_loop on the copy. */
int svg_kmeans_loop_strided(const void* x, int64_t x_batch_stride, const void* c_init, void* c_work_a, void* c_work_b, int32_t* labels,
                            int32_t* counts, int32_t* sorted_idx, void* centroids_out, int32_t* n_iters, int32_t B, int32_t N,
                            int32_t K, int32_t D, int32_t dtype, int32_t max_iters, float tol, void* workspace,
                            size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Top-p block selection.
 * ref: identify_dynamic_map + weighted_softmax, svg/kmeans_utils.py:852-896.
 * qc: [BH, QC, D], kc: [BH, KC, D] (dtype), k_sizes int32 [BH, KC]; out_map uint8 [BH, QC, KC].
 * Ties in the descending sort are broken towards the lower cluster index (stable sort).
 * ---------------------------------------------------------------------------------------------- */
int svg_identify_dynamic_map(const void* qc, const void* kc, const int32_t* k_sizes, uint8_t* out_map, int32_t BH,
                             int32_t QC, int32_t KC, int32_t D, int32_t dtype, float top_p, int32_t preserve_length,
                             void* stream);

/* Uniform-block (BSR) front end of the variable-block kernel — the reference's alternative backend
 * (flashinfer.BlockSparseAttentionWrapper plan/run + text merge, svg/kernels/ops/attention_ops.py:107-197, wan variant
 * attention_ops_wan.py): expands (indptr [MB + 1], indices) over MB x NB blocks of row_block x col_block tokens into the dense
 * map [heads, MB + 1, NB + 1] and the size arrays [heads, MB + 1] / [heads, NB + 1] that svg_varblock_attention takes; block-row 0
 * / block-column 0 hold the len_text text tokens (text first) and are always active: video rows see their BSR blocks and all
 * text, text rows see everything — the merge_state of the reference in one kernel. */
int svg_bsr_to_block_map(const int32_t* indptr, const int32_t* indices, int32_t MB, int32_t NB, int32_t row_block,
                         int32_t col_block, int32_t len_text, int32_t heads, uint8_t* block_map, int32_t* q_sizes,
                         int32_t* k_sizes, void* stream);

/* density of a dynamic map, ref: density_calculation, svg/kmeans_utils.py:13-31.  out: float [BH] */
int svg_map_density(const uint8_t* block_map, const int32_t* q_sizes, const int32_t* k_sizes, float* out, int32_t BH,
                    int32_t QB, int32_t KB, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pre-attention prologue: in-place QK normalisation over head_dim and rotary embedding.
 * ref: the reference's only native extension, svg/kernels/csrc/ops.h:19-260, bound as `_kernels` in
 *      svg/kernels/csrc/ops.cu and called from svg/models/hyvideo/attention.py:162-188, wan/attention.py:45-48,
 *      cog/attention.py:23-34.  Tensors are contiguous [bsz, H, S, D] (norm-only entry points: [m, n] rows), D (n) in
 *      {32, 64, 128, 256}, bf16 or fp16; weights / biases have the tensor dtype; cos / sin tables are fp32.
 * Semantics (pinned by the torch references of the reference's tests, svg/kernels/test/test_*.py):
 *   rms_norm:   y = w * T(x * rsqrt(mean(x^2) + eps))    fp32 statistics, rounded to T before AND after the weight
 *   layer_norm: y = T((x - mean) * rsqrt(var + 1e-5) * w + b)     fp32, biased variance
 *   rope cos/sin ("interleaved pairs"): out = T(x * cos + rotate(x) * sin) in fp32, rotate(x)[2i] = -x[2i+1],
 *               rotate(x)[2i+1] = x[2i]; cos, sin: [S - len_text_prompt, D]
 *   rope complex: (x[2i] + i x[2i+1]) * (real + i imag) in fp64; real, imag: [S - len_text_prompt, D / 2]
 *   the first (`cossin`, `complex`) or the last (`txtlast`) len_text_prompt positions are left untouched.
 * ---------------------------------------------------------------------------------------------- */
int svg_rms_norm_forward(void* x, const void* weight, int64_t m, int32_t n, int32_t dtype, float eps, void* stream);
int svg_layer_norm_forward(void* x, const void* weight, const void* bias, int64_t m, int32_t n, int32_t dtype,
                           void* stream);
int svg_apply_qk_rope_inplace_cossin(void* q, void* k, const float* cos_cache, const float* sin_cache, int32_t bsz,
                                     int32_t Hq, int32_t Hkv, int32_t S, int32_t D, int32_t dtype, int32_t len_text_prompt,
                                     void* stream);
int svg_apply_qk_rope_inplace_cossin_txtlast(void* q, void* k, const float* cos_cache, const float* sin_cache,
                                             int32_t bsz, int32_t Hq, int32_t Hkv, int32_t S, int32_t D, int32_t dtype,
                                             int32_t len_text_prompt, void* stream);
int svg_apply_qk_rope_inplace_cossin_complex(void* q, void* k, const float* freqs_real, const float* freqs_imag,
                                             int32_t bsz, int32_t Hq, int32_t Hkv, int32_t S, int32_t D, int32_t dtype,
                                             int32_t len_text_prompt, void* stream);
/* Fused form: normalisation (norm_kind 0 none / 1 rms / 2 layer) followed by rotary embedding (rope_kind 0 none / 1 cos-sin /
 * 2 complex) of positions [rope_lo, rope_hi) with table row (position - rope_lo), ONE pass over q and k (either may be NULL).
 * Bit-identical to the corresponding sequence of the entry points above (the intermediate rounding is kept). */
int svg_qk_norm_rope(void* q, void* k, int32_t bsz, int32_t Hq, int32_t Hkv, int32_t S, int32_t D, int32_t dtype,
                     int32_t norm_kind, const void* q_weight, const void* q_bias, const void* k_weight, const void* k_bias,
                     float eps, int32_t rope_kind, const float* cos_or_real, const float* sin_or_imag, int32_t rope_lo,
                     int32_t rope_hi, void* stream);

/* The same pass reading the projection output in token-major layout [bsz, S, H, D] and writing head-major [bsz, H, S, D]:
 * replaces `x.unflatten(2, (heads, -1)).transpose(1, 2)` + contiguous copy + norm + norm + rope
 * (ref: svg/models/hyvideo/attention.py:268-286) by one read and one write per element.  norm_kind = rope_kind = 0 is a plain
 * transpose (use it for V: pass it as q_in / q_out with k_in = NULL). */
int svg_qk_norm_rope_transpose(const void* q_in, const void* k_in, void* q_out, void* k_out, int32_t bsz, int32_t Hq,
                               int32_t Hkv, int32_t S, int32_t D, int32_t dtype, int32_t norm_kind, const void* q_weight,
                               const void* q_bias, const void* k_weight, const void* k_bias, float eps, int32_t rope_kind,
                               const float* cos_or_real, const float* sin_or_imag, int32_t rope_lo, int32_t rope_hi,
                               void* stream);

/* The two fused passes with a factor folded into the LAST rounding of q: q_out = round(q_scale * (norm + RoPE result in fp32)); k is
 * not touched by it.  q_scale = sm_scale * log2(e) produces the q that svg_band_attention_prescaled expects without a second
 * rounding of the rotated positions (positions outside [rope_lo, rope_hi) — the 256 text tokens of Hunyuan — were rounded by the
 * norm already and are rounded again).  q_scale = 1 is bit-identical to the plain entry points; q_scale <= 0: SVG_ERR_BAD_ARG. */
int svg_qk_norm_rope_qscale(void* q, void* k, int32_t bsz, int32_t Hq, int32_t Hkv, int32_t S, int32_t D, int32_t dtype,
                            int32_t norm_kind, const void* q_weight, const void* q_bias, const void* k_weight, const void* k_bias,
                            float eps, int32_t rope_kind, const float* cos_or_real, const float* sin_or_imag, int32_t rope_lo,
                            int32_t rope_hi, float q_scale, void* stream);
int svg_qk_norm_rope_transpose_qscale(const void* q_in, const void* k_in, void* q_out, void* k_out, int32_t bsz, int32_t Hq,
                                      int32_t Hkv, int32_t S, int32_t D, int32_t dtype, int32_t norm_kind, const void* q_weight,
                                      const void* q_bias, const void* k_weight, const void* k_bias, float eps, int32_t rope_kind,
                                      const float* cos_or_real, const float* sin_or_imag, int32_t rope_lo, int32_t rope_hi,
                                      float q_scale, void* stream);

/* The Wan 2.1 prologue in ONE pass (round 6): RMSNorm across ALL heads in the reference's Triton form (svg_rmsnorm_forward's arithmetic over
 * the H * D row) -> rotary embedding of positions [rope_lo, rope_hi) -> head-major transpose for q and k, and the plain transpose of v, in one
 * launch.  Replaces, bit for bit, svg_rmsnorm_forward(q), svg_rmsnorm_forward(k), svg_qk_norm_rope_transpose(q, k, norm 0, rope), the transpose
 * of v — i.e. the reference's get_qk_norm (triton_rmsnorm_forward) + get_transpose_qkv (three `.transpose(1, 2).contiguous()`) +
 * get_rotary_emb (_kernels.apply_qk_rope_inplace_cossin_complex): svg/models/wan/attention.py:99-148.  q_in / k_in / v_in: token-major
 * [bsz, S, H * D] (any of them may be NULL); outputs head-major [bsz, H, S, D]; weights [H * D] of w_dtype or NULL; H * D <= 8192;
 * q_scale as in svg_qk_norm_rope_qscale. */
int svg_rmsnorm_rope_transpose(const void* q_in, const void* k_in, const void* v_in, void* q_out, void* k_out, void* v_out, int32_t bsz,
                               int32_t H, int32_t S, int32_t D, int32_t dtype, const void* q_weight, const void* k_weight,
                               int32_t w_dtype, float eps, int32_t rope_kind, const float* cos_or_real, const float* sin_or_imag,
                               int32_t rope_lo, int32_t rope_hi, float q_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Transformer-block glue of the Wan blocks, rows of [M, N], N % 8 == 0, N <= 8192; dtypes per tensor
 * (SVG_DTYPE_BF16 / F16 / F32); scale, shift, gate are fp32 [M / rows_per_batch, N] (one row per batch element).
 * ref: triton_layernorm_forward svg/kernels/triton/layernorm.py:204-216 (fp32 statistics; the reference writes fp32),
 *      triton_modulate_shift_forward svg/kernels/triton/modulate.py:45-85      y = x * (1 + scale) + shift,
 *      triton_modulate_gate_residual_forward modulate.py:125-164               y = residual + x * gate,
 *      call sites svg/models/wan/custom_models.py:37-111.
 * svg_layernorm_modulate_forward fuses the first two (one read, one write of the 16-bit hidden states instead of
 * read 2 + write 4 + read 4 + write 2 bytes per element); weight / bias NULL = no affine, scale / shift NULL = no modulate.
 * ---------------------------------------------------------------------------------------------- */
int svg_layernorm_forward(const void* x, void* y, const void* weight, const void* bias, int64_t M, int32_t N, int32_t x_dtype,
                          int32_t y_dtype, int32_t w_dtype, float eps, void* stream);
int svg_modulate_shift_forward(const void* x, void* y, const float* scale, const float* shift, int64_t M, int32_t N,
                               int64_t rows_per_batch, int32_t x_dtype, int32_t y_dtype, void* stream);
int svg_modulate_gate_residual_forward(const void* residual, const void* x, const float* gate, void* y, int64_t M, int32_t N,
                                       int64_t rows_per_batch, int32_t r_dtype, int32_t x_dtype, int32_t y_dtype,
                                       void* stream);
int svg_layernorm_modulate_forward(const void* x, void* y, const void* weight, const void* bias, const float* scale,
                                   const float* shift, int64_t M, int32_t N, int64_t rows_per_batch, int32_t x_dtype,
                                   int32_t y_dtype, int32_t w_dtype, float eps, void* stream);
/* RMSNorm of the rows as the reference's TRITON kernel computes it — y = T(x * rsqrt(mean(x^2) + eps) * w), fp32, ONE rounding —:
 * ref: triton_rmsnorm_forward svg/kernels/triton/rmsnorm.py:51-105, the q / k normalisation across all heads of the reference's Wan
 * processors (svg/models/wan/attention.py:105-120).  (svg_rms_norm_forward above is the CUDA extension's form, rounded before the
 * weight like diffusers' RMSNorm; the two differ by at most one ulp of T.)  weight may be NULL. */
int svg_rmsnorm_forward(const void* x, void* y, const void* weight, int64_t M, int32_t N, int32_t x_dtype, int32_t y_dtype,
                        int32_t w_dtype, float eps, void* stream);
/* The same two with `reference_padding`.  0: as above — diffusers' FP32LayerNorm, the branch the reference's kernels replace
 * (svg/models/wan/custom_models.py:44-47).  != 0: the variance of the reference's Triton kernels AS THEY ARE: they load a row
 * zero-padded to N2 = next_power_of_2(N) and the padding takes part in the variance, var' = var + (N2 - N) / N * mean^2
 * (svg/kernels/triton/layernorm.py:35-41, :134-140; found by executing the kernels with Triton's interpreter,
 * tests/test_triton_golden.py).  For a maintainer who needs the reference's ENABLE_FAST_KERNEL numbers rather than LayerNorm's;
 * no difference when N is a power of two or the row mean is 0. */
int svg_layernorm_forward_ex(const void* x, void* y, const void* weight, const void* bias, int64_t M, int32_t N, int32_t x_dtype,
                             int32_t y_dtype, int32_t w_dtype, float eps, int32_t reference_padding, void* stream);
int svg_layernorm_modulate_forward_ex(const void* x, void* y, const void* weight, const void* bias, const float* scale,
                                      const float* shift, int64_t M, int32_t N, int64_t rows_per_batch, int32_t x_dtype,
                                      int32_t y_dtype, int32_t w_dtype, float eps, int32_t reference_padding, void* stream);

/* Device-side dense / sparse switch.  The reference decides per layer on the host,
 * `timestep[0] > first_times_fp` (svg/models/hyvideo/attention.py:491-496), which reads the GPU tensor back in every layer-call.
 * Here the caller turns that comparison into a device flag (one torch op, no synchronisation) and both kernels read it:
 *   svg_band_attention_switch: use_alt_flag[0] != 0 -> attention under alt_mask (the dense warm-up mask) without the layout
 *       transformation of `perm`; otherwise exactly svg_band_attention(mask, perm).  Two-phase kernel only (variant 0).
 *   svg_sample_mse_flagged: skip_flag[0] != 0 -> every kernel of the call returns at once (out_mse is left untouched: the
 *       profiler's result is not used on a dense step); skip_flag == NULL -> svg_sample_mse. */
int svg_band_attention_switch(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                              int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                              const svg_band_mask_t* alt_mask, const int32_t* use_alt_flag, void* stream);
/* svg_band_attention_switch on strided tensors (svg_attn_layout_t above). */
int svg_band_attention_switch_strided(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                      int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                      const svg_band_mask_t* alt_mask, const int32_t* use_alt_flag,
                                      const svg_attn_layout_t* layout, void* stream);
int svg_sample_mse_flagged(const void* q, const void* k, const void* v, const int64_t* rows, int32_t R, int32_t BH, int32_t S,
                           int32_t D, int32_t dtype, float sm_scale, const svg_profile_desc_t* prof, float* out_mse,
                           void* workspace, size_t workspace_bytes, const int32_t* skip_flag, void* stream);

/* fp8 (OCP e4m3) QK^T and PV for the same mask family (BASELINE.json configs[4]; the reference has NO fp8 attention —
 * /root/reference/README.md:117 "[ ] Support FP8 attention" — so there is no interface to cite: this entry point is
 * svg_band_attention with a workspace).  q, k, v, o are the 16-bit tensors of svg_band_attention (D = 128 only); the call runs
 * (1) a per-head absolute-maximum pass, (2) a quantise pass that writes q, k (row-major) and V^T (per 64-key tile) as e4m3 in
 * LOGICAL token order — the head placement of `perm` is applied there — scaled by 448 / amax of the head, and (3) the attention
 * kernel on v_mfma_scale_f32_32x32x64_f8f6f4 with fp32 softmax; probabilities are e4m3 (x 2^8), the output is written in the
 * input dtype at the physical rows (inverse placement fused).  Accuracy is a property of fp8, not of this kernel: see
 * tests/test_gpu_fp8.py for the measured distance to the fp32 oracle and to the 16-bit path.
 * workspace: svg_band_attention_fp8_workspace_bytes(BH, S, D) bytes of device memory (3 * BH * ceil64(S) * D + a few words). */
size_t svg_band_attention_fp8_workspace_bytes(int32_t BH, int32_t S, int32_t D);
int svg_band_attention_fp8(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D, int32_t dtype,
                           float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm, void* workspace,
                           size_t workspace_bytes, void* stream);
/* The two halves of svg_band_attention_fp8 on their own (same arguments): stage 1 = passes (1) + (2) into the workspace, stage 2 =
 * the attention kernel on a workspace that stage 1 has filled for the same BH, S and perm — for callers that time or overlap the
 * pre-pass separately. */
int svg_band_attention_fp8_stage(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D, int32_t dtype,
                                 float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm, void* workspace,
                                 size_t workspace_bytes, int32_t stage, void* stream);

/* svg_varblock_attention with e4m3 QK^T / PV — BASELINE.json configs[4] as named (SVG2 with fp8 on the CDNA4 fp8 MFMA; no
 * reference implementation, see svg_band_attention_fp8).  Same arguments as svg_varblock_attention (D = 128; the default schedule:
 * 256-row q tiles in longest-first order); q, k, v are quantised per head (x 448 / amax) in their original row order inside the call
 * and the kernel gathers rows through the block map's run list and q_row_idx / kv_row_idx exactly like the 16-bit kernel; V^T
 * fragments come from ds_read_b64_tr_b8.  workspace: svg_varblock_attention_fp8_workspace_bytes(...) bytes. */
size_t svg_varblock_attention_fp8_workspace_bytes(int32_t Hq, int32_t Hkv, int32_t QB, int32_t KB, int32_t Sq, int32_t Skv, int32_t D);
int svg_varblock_attention_fp8(const void* q, const void* k, const void* v, void* o, int32_t Hq, int32_t Hkv, int32_t Sq, int32_t Skv,
                               int32_t D, int32_t dtype, float sm_scale, const uint8_t* block_map, const int32_t* q_sizes,
                               const int32_t* k_sizes, int32_t QB, int32_t KB, const int32_t* q_row_idx, const int32_t* kv_row_idx,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Exchange overlapped with ONE launch (multi-GPU, SURVEY §8 e).  svg_band_attention_notify = svg_band_attention (variant 0) that
 * also counts completions: every wave adds 1 to done_per_head[h] (int32 [2 * BH], zeroed by the caller; the second half is
 * scratch of the library) after its last store of head h, so done_per_head[h] == svg_band_attention_notify_target(S, mask)
 * means head h of `o` is complete and visible.  The launch is
 * head-major, so heads complete in order; svg_wait_counters enqueues a one-wave kernel on another stream that returns once
 * `n` counters have reached `target` — the all-gather of those heads goes behind it and runs while the launch is still working
 * on the next heads (no chunked launches: at N = 8 three launches of one head cost 5.7 ms, one launch of three heads 4.9 ms). */
int32_t svg_band_attention_notify_target(int32_t S, const svg_band_mask_t* mask);
/* done_words: the number of int32 words the caller allocated at `done_per_head` / `done` — the entry points cannot see the size
 * of a device buffer, and the hidden per-head counters behind the segment counters made the requirement grow once (round 2): a
 * buffer smaller than BH * (nseg + 1) words returns SVG_ERR_WORKSPACE instead of being written out of bounds. */
int svg_band_attention_notify(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                              int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                              int32_t* done_per_head, int32_t done_words, void* stream);
int svg_wait_counters(const int32_t* counters, int32_t n, int32_t target, void* stream);
/* svg_wait_counters with a deadline: the one-wave kernel gives up `timeout_ms` after it started, stores 1 to timed_out[0] (int32
 * in device or pinned host memory, zeroed by the caller) and returns — a waiter of this kind cannot hang its stream whatever
 * happens to the launch it watches; what was queued behind it then runs on incomplete rows, so the caller must read the flag
 * before trusting the exchange (bench.py does after warm-up and falls back to one launch per chunk of heads). */
int svg_wait_counters_deadline(const int32_t* counters, int32_t n, int32_t target, int32_t timeout_ms, int32_t* timed_out,
                               void* stream);
/* The same with several counters per head.  svg_band_attention_notify_layout fills row_bounds[0 .. n] and targets[0 .. n) for this
 * S / mask and returns n <= nseg, the number of segments used; pass that n as `nseg` of svg_band_attention_notify_seg together
 * with done = int32 [BH * (n + 1)], zeroed by the caller (counter (h, s) at done[h * n + s]; the last BH words are scratch of
 * the library).  Contract: done[h * n + s] >= targets[s]  =>  the PHYSICAL rows [row_bounds[s], row_bounds[s + 1]) of head h of
 * `o` are complete and visible.  Segments are cut in q-tile (logical row) order; a head that runs with the fused layout
 * permutation (perm->head_perm_flag[h] != 0) writes the rows of a logical segment to every frame, so its segments are all
 * released together when the head is complete — the contract holds for both kinds of head without the host knowing which
 * is which (best_mask_idx is device data). */
int32_t svg_band_attention_notify_layout(int32_t S, const svg_band_mask_t* mask, int32_t nseg, int32_t* row_bounds, int32_t* targets);
int svg_band_attention_notify_seg(const void* q, const void* k, const void* v, void* o, int32_t BH, int32_t S, int32_t D,
                                  int32_t dtype, float sm_scale, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                  int32_t* done, int32_t done_words, int32_t nseg, void* stream);
/* svg_band_attention_notify_seg for a pre-scaled q (see svg_band_attention_prescaled): same counters, layout and targets. */
int svg_band_attention_prescaled_notify_seg(const void* q_scaled, const void* k, const void* v, void* o, int32_t BH, int32_t S,
                                            int32_t D, int32_t dtype, const svg_band_mask_t* mask, const svg_perm_desc_t* perm,
                                            int32_t* done, int32_t done_words, int32_t nseg, void* stream);

/* Diagnostics (not part of the reference's interface; -DSVG_ABLATIONS builds, otherwise SVG_ERR_UNSUPPORTED): cycle trace of
 * the two-phase attention schedule.  After a svg_band_attention call with variant 64 (bf16, D = 128) and a synchronised
 * stream, copies 104 counters to the host: out[8 * wave + i] = s_memtime ticks wave `wave` of one workgroup spent in
 *   variant 64: i = 0..3  [matrix phase, barrier, vector phase, barrier]
 *   variant 32: i = 0..3  [phase A up to the barrier, wait + barrier, rest of A, phase B]   (waves 0..3)
 * out[64] = KV tiles of that workgroup, out[65] = ticks of its tile loop. */
int svg_debug_pp_trace(uint64_t* out104);

/* Shader-clock probe (measurement aid of bench.py, product library): a one-wave kernel on `stream` that sleeps until
 * stop_flag[0] != 0 (int32 in device or pinned host memory, written by a later memset / copy on ANOTHER stream) or max_ms have
 * passed, then stores { shader-clock ticks (s_memtime), 100 MHz ticks } elapsed to out2[0..1]: the sustained shader clock of
 * the span is 100 MHz * out2[0] / out2[1]. */
int svg_debug_clock_probe(const int32_t* stop_flag, uint64_t* out2, int32_t max_ms, void* stream);

/* Launch timeline of the same traced kernel (variant 64): for each of the first n_workgroups (<= 16384) workgroups
 * out[6 * b + ..] = [s_memtime at entry, at the start of the tile loop, at its end, after the last store of O, HW_ID, XCC_ID]. */
int svg_debug_wg_trace(uint64_t* out, int32_t n_workgroups);

#ifdef __cplusplus
}
#endif
#endif /* SVG_ATTN_H_ */
