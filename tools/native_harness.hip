// native_harness — a torch-free driver of libsvgattn.so's SVG1 entry points through the C ABI (include/svg_attn.h): device-generated
// inputs, HIP-event timing on the launch stream, an fp32 spot-row check of the result.  A process of this binary starts in
// milliseconds, so a rocprofv3 counter pass over it costs seconds of GPU-box time instead of the minute a `python bench.py` pass
// pays for importing torch — it is the tool for PMC passes and for A/B timing of tagged library builds.  Diagnostics only: nothing
// of the product links or loads it, and its numbers are labelled with this file's name wherever they are quoted.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/native_harness.hip -o tools/native_harness -ldl
//   tools/native_harness [--lib PATH] [--geom hy720p|wan720p|hy480p|small|cog480p|cog15|small64] [--dtype bf16|f16] [--variant N] [--prescaled]
//                        [--flags half|zero|one] [--heads H] [--warm W] [--reps R] [--check ROWS] [--seed N]
//                        [--fill normal|zero|const] [--band KEYS] [--no-clock] [--profiler]
// --profiler: time svg_sample_mse (the online profiler, 64 sampled rows below min(10000, V), the model's profiling masks with the bf16
// emulation on — what bench.py's timed step runs in front of the attention launch) instead of the attention; prints the 2 x H mse values'
// sum and a checksum of their bits (A/B of library builds).
// --fill zero / const: q, k, v all zero / all 1.0 — the same instruction stream on operands that never toggle (the schedule-only ceiling:
// what the launch takes when the power management has nothing to limit; MI355X_MICROARCH.md "DVFS give-back").  --band overrides the
// band half-width of the geometry's mask (tools/band_sweep through the harness).  The shader clock granted to the timed launches is
// read by the library's one-wave probe (svg_debug_clock_probe) on a second stream: "sclk_mhz", and "mcycles" = ms x sclk.
// Output: one JSON line (ms per launch, algorithmic TFLOP/s, spot-row error against the fp32 restatement below).
//
// The fp32 restatement (ref_rows_kernel) follows the predicate documented at svg_band_mask_t in include/svg_attn.h and the fused
// placement rule of svg_perm_desc_t (logical video row i of a temporal head lives at physical row vid0 + (i % F) * P + i / F) —
// the reference's flex_attention mask_mod + placement, /root/reference/svg/models/hyvideo/utils.py:20-44, placement.py:76-78.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/svg_attn.h"

static inline void hip_ok(hipError_t e, const char* what, int line) {
    if (e != hipSuccess) {
        fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, line, what, hipGetErrorString(e));
        exit(2);
    }
}
#define HIP_OK(x) hip_ok((x), #x, __LINE__)

// ---------------------------------------------------------------- 16-bit encodings (bit-level, host and device)
__host__ __device__ inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__device__ inline uint16_t f32_to_f16_dev(float f) {
    _Float16 h = (_Float16)f;
    uint16_t b;
    memcpy(&b, &h, 2);
    return b;
}
__device__ inline float f16_to_f32_dev(uint16_t b) {
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float)h;
}
static float f16_to_f32_host(uint16_t b) {
    const uint32_t s = (b >> 15) & 1, e = (b >> 10) & 31, m = b & 1023;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 1024), (int)e - 25);
    return s ? -v : v;
}

static inline uint32_t mix32_host(uint64_t x) {
    x ^= x >> 33, x *= 0xff51afd7ed558ccdull, x ^= x >> 33, x *= 0xc4ceb9fe1a85ec53ull, x ^= x >> 33;
    return (uint32_t)x;
}
// ---------------------------------------------------------------- input generator: counter-based normal variates
__device__ inline uint32_t mix32(uint64_t x) {
    x ^= x >> 33, x *= 0xff51afd7ed558ccdull, x ^= x >> 33, x *= 0xc4ceb9fe1a85ec53ull, x ^= x >> 33;
    return (uint32_t)x;
}
__global__ void fill_const_kernel(uint16_t* dst, size_t n, uint16_t bits) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = bits;
}
__global__ void set_flag_kernel(int32_t* f, int32_t v) { *f = v; }
__global__ void fill_normal_kernel(uint16_t* dst, size_t n, uint64_t seed, float scale, int f16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t a = mix32(seed * 0x9e3779b97f4a7c15ull + 2 * i), b = mix32(seed * 0x9e3779b97f4a7c15ull + 2 * i + 1);
        const float u1 = ((a >> 8) + 1) * (1.f / 16777216.f), u2 = (b >> 8) * (1.f / 16777216.f);
        const float z = sqrtf(-2.f * logf(u1)) * cosf(6.28318530718f * u2) * scale;
        dst[i] = f16 ? f32_to_f16_dev(z) : f32_to_bf16(z);
    }
}

// ---------------------------------------------------------------- order-independent 64-bit checksum of a buffer (bit-exact A/B of two builds)
__global__ void checksum_kernel(const uint32_t* p, size_t nwords, unsigned long long* out) {
    unsigned long long h = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x)
        h += (unsigned long long)p[i] * (2ull * mix32(i) + 1ull) + mix32(i ^ p[i]);
    atomicAdd(out, h);
}

// ---------------------------------------------------------------- fp32 restatement of the masked attention of selected rows
struct RefGeom {
    int S, D, F, P, V;   // V = F * P video rows
    int vid0;            // first video row (0: text last; context_length: text first, CogVideoX)
    svg_band_mask_t m;
    float scale;
    int f16;
};
__device__ inline bool allowed(const svg_band_mask_t& m, int q, int k) {
    const bool rq = q < m.real_len, rk = k < m.real_len;
    const int d = q > k ? q - k : k - q;
    const bool in = d < m.band || (k >= m.colfull_lo && k < m.colfull_hi) || (q >= m.rowfull_lo && q < m.rowfull_hi);
    return (rq && rk && in) || (!rq && !rk);
}
__device__ inline int phys_row(const RefGeom& g, bool temporal, int i) {
    const int j = i - g.vid0;
    return (temporal && j >= 0 && j < g.V) ? g.vid0 + (j % g.F) * g.P + j / g.F : i;
}
__device__ inline float ld16(const uint16_t* p, int f16) { return f16 ? f16_to_f32_dev(*p) : bf16_to_f32(*p); }

// one workgroup per (checked head, checked logical row); scratch: S floats per workgroup; out: D floats per workgroup
__global__ void __launch_bounds__(256) ref_rows_kernel(const uint16_t* q, const uint16_t* k, const uint16_t* v, const int64_t* flags,
                                                       const int* heads, const int* rows, int nrows, RefGeom g, float* scratch, float* out,
                                                       int q_prescaled) {
    const int h = heads[blockIdx.x / nrows], i = rows[blockIdx.x % nrows];
    const bool temporal = flags && flags[h] != 0;
    const size_t hb = (size_t)h * g.S * g.D;
    __shared__ float qs[128];
    __shared__ float red[256];
    float* sc = scratch + (size_t)blockIdx.x * g.S;
    const int pq = phys_row(g, temporal, i);
    if ((int)threadIdx.x < g.D) qs[threadIdx.x] = ld16(q + hb + (size_t)pq * g.D + threadIdx.x, g.f16);
    __syncthreads();
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < g.S; j += 256) {
        float s = -INFINITY;
        if (allowed(g.m, i, j)) {
            const uint16_t* kr = k + hb + (size_t)phys_row(g, temporal, j) * g.D;
            float acc = 0.f;
            for (int d = 0; d < g.D; ++d) acc += qs[d] * ld16(kr + d, g.f16);
            s = q_prescaled ? acc * 0.6931471805599453f : acc * g.scale;   // a pre-scaled q carries sm_scale * log2(e)
        }
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    // thread t: column d = t % D of keys j = t / D, t / D + 256 / D, ...
    const int d = threadIdx.x % g.D, part = threadIdx.x / g.D, nparts = 256 / g.D;
    float acc = 0.f, l = 0.f;
    if (mx != -INFINITY) {
        for (int j = part; j < g.S; j += nparts) {
            const float s = sc[j];
            if (s == -INFINITY) continue;
            const float p = expf(s - mx);
            l += p;
            acc += p * ld16(v + hb + (size_t)phys_row(g, temporal, j) * g.D + d, g.f16);
        }
    }
    __shared__ float accs[256], ls[256];
    accs[threadIdx.x] = acc, ls[threadIdx.x] = l;
    __syncthreads();
    if (part == 0) {
        for (int p2 = 1; p2 < nparts; ++p2) acc += accs[p2 * g.D + d], l += ls[p2 * g.D + d];
        out[(size_t)blockIdx.x * g.D + d] = l > 0.f ? acc / l : 0.f;
    }
}

// ---------------------------------------------------------------- library binding (dlopen: the harness times whichever build it is given)
typedef int (*band_fn)(const void*, const void*, const void*, void*, int32_t, int32_t, int32_t, int32_t, float, const svg_band_mask_t*,
                       const svg_perm_desc_t*, int32_t, void*);
typedef int (*band_switch_fn)(const void*, const void*, const void*, void*, int32_t, int32_t, int32_t, int32_t, float, const svg_band_mask_t*,
                              const svg_perm_desc_t*, const svg_band_mask_t*, const int32_t*, void*);
typedef int (*band_pre_fn)(const void*, const void*, const void*, void*, int32_t, int32_t, int32_t, int32_t, const svg_band_mask_t*,
                           const svg_perm_desc_t*, void*);

struct Geom {
    const char* name;
    int H, D, F, P, ctx, L;
    double width_frames;   // band half-width in frames (sparsity_to_width of the model's script)
    int kind;              // 0 Hunyuan (text last, prompt L of ctx), 1 Wan (ceil + 1, sink columns, no text), 2 CogVideoX (text first)
};
// widths: sparsity_to_width (svg/models/hyvideo/utils.py:142-151) of each geometry at sparsity 0.25 (Hunyuan) / 0.3 (Wan) — SURVEY §8(d): bands
// 15616 / 12416 + 1 at 720p, 5632 at Hunyuan 480p; tests/test_native_harness_cpu.py holds these constants to the product's mask builders
static const Geom kGeoms[] = {{"hy720p", 24, 128, 33, 3600, 256, 64, 4.348694090495091, 0},
                              {"wan720p", 40, 128, 21, 3600, 0, 0, 3.430139442784413, 1},
                              {"hy480p", 24, 128, 33, 1350, 256, 64, 4.228429541193822, 0},
                              {"small", 4, 128, 5, 160, 256, 64, 1.7, 0},
                              // svg/models/cog/utils.py at sparsity 0.25: bands 2048 / 5760 (tools/svg1_models.py), cfg = 2 x 48 heads, head_dim 64
                              {"cog480p", 96, 64, 13, 1350, 226, 226, 1.5724038952394224, 2},
                              {"cog15", 96, 64, 11, 4080, 226, 226, 1.4173925802088359, 2},
                              {"small64", 6, 64, 5, 160, 226, 226, 1.7, 2}};

// #allowed (q, k) pairs of a mask, counted from the predicate row by row (closed forms: SURVEY §8(d))
static double count_pairs(const svg_band_mask_t& m, int S) {
    double pairs = 0;
    for (int i = 0; i < S; ++i) {
        if (i >= m.real_len) { pairs += S - m.real_len; continue; }
        if (i >= m.rowfull_lo && i < m.rowfull_hi) { pairs += m.real_len; continue; }
        const int lo = std::max(0, i - m.band + 1), hi = std::min(m.real_len, i + m.band);   // band: [lo, hi)
        double c = hi - lo;
        const int clo = std::max(m.colfull_lo, 0), chi = std::min(m.colfull_hi, m.real_len);
        if (chi > clo) c += (chi - clo) - std::max(0, std::min(hi, chi) - std::max(lo, clo));
        pairs += c;
    }
    return pairs;
}

int main(int argc, char** argv) {
    std::string lib = "sparse-videogen_amd/lib/libsvgattn.so", geom = "hy720p", dtype = "bf16", flags = "half";
    int variant = 0, warm = 2, reps = 5, check = 10, heads = 0, prescaled = 0;
    uint64_t seed = 0;
    std::string occ_sym;   // --occupancy <kernel handle symbol> <dynamic LDS bytes>: resident 512-thread workgroups per CU, then exit
    int occ_lds = 0;
    int dry = 0;           // --dry: print the geometry and the mask the run would use (no GPU, no library call), then exit
    std::string fill = "normal";
    int band_override = 0, use_clock = 1, profiler = 0;
    int sw = -1;           // --switch F: svg_band_attention_switch with the device flag F (0: the sparse mask with placement, 1: the dense alternative without)
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); }
            return argv[++i];
        };
        if (a == "--lib") lib = next();
        else if (a == "--geom") geom = next();
        else if (a == "--dtype") dtype = next();
        else if (a == "--flags") flags = next();
        else if (a == "--variant") variant = atoi(next());
        else if (a == "--warm") warm = atoi(next());
        else if (a == "--reps") reps = atoi(next());
        else if (a == "--check") check = atoi(next());
        else if (a == "--heads") heads = atoi(next());
        else if (a == "--seed") seed = strtoull(next(), nullptr, 10);
        else if (a == "--prescaled") prescaled = 1;
        else if (a == "--dry") dry = 1;
        else if (a == "--fill") fill = next();
        else if (a == "--band") band_override = atoi(next());
        else if (a == "--no-clock") use_clock = 0;
        else if (a == "--profiler") profiler = 1;
        else if (a == "--switch") sw = atoi(next());
        else if (a == "--occupancy") occ_sym = next(), occ_lds = atoi(next());
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    const Geom* G = nullptr;
    for (const Geom& g : kGeoms) if (geom == g.name) G = &g;
    if (!G) { fprintf(stderr, "unknown geometry %s\n", geom.c_str()); return 2; }
    const int f16 = dtype == "f16";
    const int H = heads > 0 ? heads : G->H, D = G->D, V = G->F * G->P, S = V + G->ctx;

    void* so = dlopen(lib.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!so) { fprintf(stderr, "dlopen %s: %s\n", lib.c_str(), dlerror()); return 2; }
    auto abi = (int (*)())dlsym(so, "svg_abi_version");
    auto band = (band_fn)dlsym(so, "svg_band_attention");
    auto band_pre = (band_pre_fn)dlsym(so, "svg_band_attention_prescaled");
    auto band_sw = (band_switch_fn)dlsym(so, "svg_band_attention_switch");
    auto strerr = (const char* (*)(int))dlsym(so, "svg_strerror");
    auto info = (const char* (*)())dlsym(so, "svg_build_info");
    if (!abi || !band || !band_pre || !strerr) { fprintf(stderr, "library lacks an entry point of include/svg_attn.h\n"); return 2; }
    if (abi() != SVG_ABI_VERSION) { fprintf(stderr, "ABI %d, header %d\n", abi(), SVG_ABI_VERSION); return 2; }
    if (!occ_sym.empty()) {
        const void* kh = dlsym(so, occ_sym.c_str());
        if (!kh) { fprintf(stderr, "no symbol %s\n", occ_sym.c_str()); return 2; }
        int nb = -1;
        HIP_OK(hipFuncSetAttribute(kh, hipFuncAttributeMaxDynamicSharedMemorySize, occ_lds));
        HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kh, 512, (size_t)occ_lds));
        hipFuncAttributes fa;
        HIP_OK(hipFuncGetAttributes(&fa, kh));
        printf("{\"lib\": \"%s\", \"kernel\": \"%s\", \"dynamic_lds\": %d, \"workgroups_per_cu\": %d, \"num_regs\": %d, \"local_bytes_per_lane\": %zu, \"static_lds\": %zu}\n",
               lib.c_str(), occ_sym.c_str(), occ_lds, nb, fa.numRegs, (size_t)fa.localSizeBytes, (size_t)fa.sharedSizeBytes);
        return 0;
    }

    svg_band_mask_t m;
    const int vid0 = G->kind == 2 ? G->ctx : 0;
    if (G->kind == 1) m = {S, (int)std::ceil(G->width_frames * G->P / 128.0) * 128 + 1, 0, G->P, 0, 0};
    else if (G->kind == 2) m = {S, (int)std::floor(G->width_frames * G->P / 128.0) * 128, 0, G->L, 0, G->L};
    else m = {V + G->L, (int)std::floor(G->width_frames * G->P / 128.0) * 128, V, V + G->L, V, V + G->L};
    if (band_override > 0) m.band = band_override;
    const float sm_scale = 1.f / sqrtf((float)D);
    if (dry) {   // what tests/test_boundary_cpu.py holds against the oracle's mask builders
        printf("{\"geom\": \"%s\", \"H\": %d, \"S\": %d, \"D\": %d, \"vid0\": %d, \"F\": %d, \"P\": %d, \"mask\": [%d, %d, %d, %d, %d, %d], \"pairs\": %.0f}\n", G->name, H, S, D,
               vid0, G->F, G->P, m.real_len, m.band, m.colfull_lo, m.colfull_hi, m.rowfull_lo, m.rowfull_hi, count_pairs(m, S));
        return 0;
    }

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    const size_t n = (size_t)H * S * D;
    uint16_t *q, *k, *v, *o;
    HIP_OK(hipMalloc(&q, n * 2)), HIP_OK(hipMalloc(&k, n * 2)), HIP_OK(hipMalloc(&v, n * 2)), HIP_OK(hipMalloc(&o, n * 2));
    const float qmul = prescaled ? sm_scale * 1.4426950408889634f : 1.f;   // rounded once, like the prologue's q_scale
    if (fill == "normal") {
        fill_normal_kernel<<<4096, 256, 0, st>>>(q, n, 3 * seed + 1, qmul, f16);
        fill_normal_kernel<<<4096, 256, 0, st>>>(k, n, 3 * seed + 2, 1.f, f16);
        fill_normal_kernel<<<4096, 256, 0, st>>>(v, n, 3 * seed + 3, 1.f, f16);
    } else if (fill == "zero" || fill == "const") {
        const uint16_t bits = fill == "zero" ? 0 : (f16 ? 0x3c00 : 0x3f80);   // 1.0
        fill_const_kernel<<<4096, 256, 0, st>>>(q, n, bits);
        fill_const_kernel<<<4096, 256, 0, st>>>(k, n, bits);
        fill_const_kernel<<<4096, 256, 0, st>>>(v, n, bits);
    } else { fprintf(stderr, "unknown --fill %s\n", fill.c_str()); return 2; }
    HIP_OK(hipMemsetAsync(o, 0xff, n * 2, st));

    if (profiler) {
        typedef size_t (*ws_fn)(int32_t, int32_t, int32_t, int32_t);
        typedef int (*mse_fn)(const void*, const void*, const void*, const int64_t*, int32_t, int32_t, int32_t, int32_t, int32_t, float,
                              const svg_profile_desc_t*, float*, void*, size_t, void*);
        auto ws_bytes = (ws_fn)dlsym(so, "svg_sample_mse_workspace_bytes");
        auto mse = (mse_fn)dlsym(so, "svg_sample_mse");
        if (!ws_bytes || !mse) { fprintf(stderr, "library lacks svg_sample_mse\n"); return 2; }
        const int R = 64;
        std::vector<int64_t> hrows(R);
        for (int i = 0; i < R; ++i) hrows[i] = (int64_t)(mix32_host(seed * 1315423911ull + i) % (uint32_t)std::min(10000, V)) + (G->kind == 2 ? 0 : 0);
        int64_t* drows;
        HIP_OK(hipMalloc(&drows, R * 8));
        HIP_OK(hipMemcpyAsync(drows, hrows.data(), R * 8, hipMemcpyHostToDevice, st));
        svg_profile_desc_t pd;
        memset(&pd, 0, sizeof(pd));
        pd.vid0 = vid0, pd.num_frame = G->F, pd.frame_size = G->P, pd.emulate_bf16 = 1;
        if (G->kind == 0) {          // svg/models/hyvideo/utils.py get_attention_mask (bench.py uses text_hi = S)
            const int bb = (int)((G->P * 1.5) / 128);
            pd.variant[0] = {0, 0, V, bb, 0, V, S};
            pd.variant[1] = {1, 0, V, bb, 0, V, S};
        } else if (G->kind == 1) {   // svg/models/wan/utils.py
            const int bb = (int)((G->P * 2) / 128);
            pd.variant[0] = {0, 0, V, bb, G->P, 0, 0};
            pd.variant[1] = {1, 0, V, bb, G->P, 0, 0};
        } else {                     // svg/models/cog/utils.py
            const int bb = (int)((G->P * 1.5) / 128);
            pd.variant[0] = {0, 0, std::min(S, (V + 127) / 128 * 128), bb, 0, 0, G->ctx};
            pd.variant[1] = {1, G->ctx, V, bb, 0, 0, 0};
        }
        const size_t wsb = ws_bytes(H, R, D, S);
        void* ws;
        float* dmse;
        HIP_OK(hipMalloc(&ws, wsb)), HIP_OK(hipMalloc(&dmse, 2 * H * 4));
        auto call = [&]() {
            const int rc = mse(q, k, v, drows, R, H, S, D, f16 ? SVG_DTYPE_F16 : SVG_DTYPE_BF16, sm_scale, &pd, dmse, ws, wsb, st);
            if (rc != 0) { fprintf(stderr, "svg_sample_mse: %s\n", strerr(rc)); exit(3); }
        };
        for (int i = 0; i < warm; ++i) call();
        std::vector<hipEvent_t> e0(reps), e1(reps);
        for (int i = 0; i < reps; ++i) {
            HIP_OK(hipEventCreate(&e0[i])), HIP_OK(hipEventCreate(&e1[i]));
            HIP_OK(hipEventRecord(e0[i], st));
            call();
            HIP_OK(hipEventRecord(e1[i], st));
        }
        HIP_OK(hipStreamSynchronize(st));
        std::vector<float> hm(2 * H);
        HIP_OK(hipMemcpy(hm.data(), dmse, 2 * H * 4, hipMemcpyDeviceToHost));
        double sum = 0, mean = 0;
        unsigned long long bits = 0;
        for (int i = 0; i < 2 * H; ++i) {
            uint32_t u;
            memcpy(&u, &hm[i], 4);
            bits = bits * 1099511628211ull + u;
            if (hm[i] == hm[i]) sum += hm[i];
        }
        printf("{\"tool\": \"tools/native_harness --profiler\", \"lib\": \"%s\", \"geom\": \"%s\", \"H\": %d, \"S\": %d, \"D\": %d, \"R\": %d, \"workspace_bytes\": %zu, \"ms\": [",
               lib.c_str(), G->name, H, S, D, R, wsb);
        for (int i = 0; i < reps; ++i) {
            float t;
            HIP_OK(hipEventElapsedTime(&t, e0[i], e1[i]));
            mean += t / reps;
            printf("%s%.4f", i ? ", " : "", t);
        }
        if (auto ptrace = (int (*)(uint64_t*, int))dlsym(so, "svg_debug_prof_trace")) {   // -DSVG_PROF_TRACE builds: per-workgroup timeline of the last launch
            const int nwg = 2048;
            std::vector<uint64_t> tr((size_t)nwg * 4);
            if (ptrace(tr.data(), nwg) == 0) {
                uint64_t t_min = ~0ull, t_max = 0;
                int n = 0;
                for (int w = 0; w < nwg; ++w) if (tr[4 * w + 1] > tr[4 * w]) { t_min = std::min(t_min, tr[4 * w]); t_max = std::max(t_max, tr[4 * w + 1]); ++n; }
                fprintf(stderr, "prof trace: %d workgroups, span %llu ticks\n", n, (unsigned long long)(t_max - t_min));
                // per chunk: mean start, mean duration, max end (ticks of s_memtime relative to the first start)
                for (int c = 0; c < 64; ++c) {
                    double st = 0, du = 0; uint64_t en = 0, dmax = 0; int m = 0;
                    for (int w = 0; w < nwg; ++w) {
                        if (!(tr[4 * w + 1] > tr[4 * w]) || (int)(tr[4 * w + 3] >> 16) != c) continue;
                        st += (double)(tr[4 * w] - t_min), du += (double)(tr[4 * w + 1] - tr[4 * w]), en = std::max(en, tr[4 * w + 1] - t_min), dmax = std::max(dmax, tr[4 * w + 1] - tr[4 * w]), ++m;
                    }
                    if (m) fprintf(stderr, "  chunk %2d: %2d wgs  mean start %9.0f  mean dur %9.0f  max dur %9llu  last end %9llu\n", c, m, st / m, du / m, (unsigned long long)dmax, (unsigned long long)en);
                }
            }
        }
        if (auto pphase = (int (*)(uint64_t*, int))dlsym(so, "svg_debug_prof_phase")) {   // second form: where wave 0 of the workgroups spent its ticks
            const int nwg = 2048;
            std::vector<uint64_t> ph((size_t)nwg * 8);
            if (pphase(ph.data(), nwg) == 0) {
                double v[8] = {0};
                for (int i = 0; i < nwg; ++i) for (int j = 0; j < 8; ++j) v[j] += (double)ph[8 * i + j];
                const double tiles = v[3];
                if (tiles > 0) fprintf(stderr, "prof phases (wave 0, ticks per tile): wait+barrier %.1f | scores %.1f  rounding+max %.1f  exp+pack %.1f  masks %.1f | PV %.1f  (tiles %.0f)\n",
                                       v[0] / tiles, v[4] / tiles, v[5] / tiles, v[6] / tiles, v[1] / tiles, v[2] / tiles, tiles);
            }
        }
        auto num = [](float x) { char buf[32]; if (x != x) return std::string("null"); snprintf(buf, sizeof buf, "%.6e", x); return std::string(buf); };
        const double kv_bytes = 2.0 * H * (double)S * D * 2;
        printf("], \"ms_mean\": %.4f, \"kv_bytes\": %.0f, \"gbps\": %.1f, \"frac_of_8tbps\": %.4f, \"mse_sum\": %.9e, \"mse0\": [%s, %s], \"mse_bits\": \"%016llx\"}\n",
               mean, kv_bytes, kv_bytes / (mean * 1e-3) / 1e9, kv_bytes / (mean * 1e-3) / 8e12, sum, num(hm[0]).c_str(), num(hm[H]).c_str(), bits);   // (NaN -> null: CogVideoX's text-row quirk)
        return 0;
    }

    std::vector<int64_t> hf(H);
    for (int h = 0; h < H; ++h) hf[h] = flags == "one" ? 1 : flags == "zero" ? 0 : (h & 1);
    int64_t* dflags;
    HIP_OK(hipMalloc(&dflags, H * 8));
    HIP_OK(hipMemcpyAsync(dflags, hf.data(), H * 8, hipMemcpyHostToDevice, st));
    svg_perm_desc_t perm = {dflags, vid0, G->F, G->P};

    // --switch: the alternative is the dense warm-up mask of the processors (keys of the real sequence, no placement); with flag 1 the
    // checks below are made against IT
    const svg_band_mask_t sparse_m = m;
    const svg_band_mask_t dense_m = {m.real_len, S + 1, 0, 0, 0, 0};
    int32_t* dflag = nullptr;
    if (sw >= 0) {
        if (!band_sw) { fprintf(stderr, "library lacks svg_band_attention_switch\n"); return 2; }
        HIP_OK(hipMalloc(&dflag, 4));
        HIP_OK(hipMemcpyAsync(dflag, &sw, 4, hipMemcpyHostToDevice, st));
        if (sw) {
            m = dense_m;
            for (auto& f : hf) f = 0;
            HIP_OK(hipMemsetAsync(dflags, 0, H * 8, st));   // (the reference rows below see no placement; the call still gets the real flags)
        }
    }
    int64_t* dflags_call = dflags;
    if (sw == 1) {   // the call keeps the sparse side's placement flags: a dense step must ignore them
        std::vector<int64_t> hf2(H);
        for (int h = 0; h < H; ++h) hf2[h] = flags == "one" ? 1 : flags == "zero" ? 0 : (h & 1);
        HIP_OK(hipMalloc(&dflags_call, H * 8));
        HIP_OK(hipMemcpyAsync(dflags_call, hf2.data(), H * 8, hipMemcpyHostToDevice, st));
    }
    svg_perm_desc_t perm_call = {dflags_call, vid0, G->F, G->P};
    auto launch = [&]() {
        if (sw >= 0) {
            const int rc = band_sw(q, k, v, o, H, S, D, f16 ? SVG_DTYPE_F16 : SVG_DTYPE_BF16, sm_scale, &sparse_m, &perm_call, &dense_m, dflag, st);
            if (rc != 0) { fprintf(stderr, "band attention switch: %s\n", strerr(rc)); exit(3); }
            return;
        }
        const int rc = prescaled ? band_pre(q, k, v, o, H, S, D, f16 ? SVG_DTYPE_F16 : SVG_DTYPE_BF16, &m, &perm, st)
                                 : band(q, k, v, o, H, S, D, f16 ? SVG_DTYPE_F16 : SVG_DTYPE_BF16, sm_scale, &m, &perm, variant, st);
        if (rc != 0) { fprintf(stderr, "band attention: %s\n", strerr(rc)); exit(3); }
    };
    for (int i = 0; i < warm; ++i) launch();
    // shader-clock probe beside the timed launches: started on its own stream once the warm-up is through, stopped by a flag store that
    // a third stream issues behind the last timed launch
    auto clock_probe = (int (*)(const int32_t*, uint64_t*, int32_t, void*))dlsym(so, "svg_debug_clock_probe");
    hipStream_t st_probe = nullptr, st_flag = nullptr;
    int32_t* stop_flag = nullptr;
    uint64_t* probe_out = nullptr;
    hipEvent_t ev_done;
    if (!clock_probe) use_clock = 0;
    if (use_clock) {
        HIP_OK(hipStreamCreate(&st_probe)), HIP_OK(hipStreamCreate(&st_flag));
        HIP_OK(hipMalloc(&stop_flag, 4)), HIP_OK(hipMalloc(&probe_out, 16));
        HIP_OK(hipMemset(stop_flag, 0, 4)), HIP_OK(hipMemset(probe_out, 0, 16));
        HIP_OK(hipEventCreate(&ev_done));
        HIP_OK(hipStreamSynchronize(st));
        if (clock_probe(stop_flag, probe_out, 20000, st_probe) != 0) use_clock = 0;
    }
    std::vector<hipEvent_t> e0(reps), e1(reps);
    for (int i = 0; i < reps; ++i) {
        HIP_OK(hipEventCreate(&e0[i])), HIP_OK(hipEventCreate(&e1[i]));
        HIP_OK(hipEventRecord(e0[i], st));
        launch();
        HIP_OK(hipEventRecord(e1[i], st));
    }
    double sclk_mhz = 0;
    if (use_clock) {
        HIP_OK(hipEventRecord(ev_done, st));
        HIP_OK(hipStreamWaitEvent(st_flag, ev_done, 0));
        set_flag_kernel<<<1, 1, 0, st_flag>>>(stop_flag, 1);
        HIP_OK(hipStreamSynchronize(st_probe));
        uint64_t po[2] = {0, 0};
        HIP_OK(hipMemcpy(po, probe_out, 16, hipMemcpyDeviceToHost));
        if (po[0] > 0 && po[1] > 0) sclk_mhz = 100.0 * (double)po[0] / (double)po[1];
    }
    HIP_OK(hipStreamSynchronize(st));
    std::vector<float> ms(reps);
    double mean = 0;
    for (int i = 0; i < reps; ++i) HIP_OK(hipEventElapsedTime(&ms[i], e0[i], e1[i])), mean += ms[i] / reps;

    unsigned long long* dsum;
    unsigned long long osum = 0;
    HIP_OK(hipMalloc(&dsum, 8)), HIP_OK(hipMemsetAsync(dsum, 0, 8, st));
    checksum_kernel<<<2048, 256, 0, st>>>((const uint32_t*)o, n / 2, dsum);
    HIP_OK(hipMemcpyAsync(&osum, dsum, 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));

    const double pairs = count_pairs(m, S);
    const double flop = 4.0 * D * H * pairs;

    // spot rows against the fp32 restatement
    double err2 = 0, ref2 = 0, maxabs = 0;
    int nck = 0;
    if (check > 0) {
        std::vector<int> rows, hs;
        const int cand[] = {0, 1, G->P - 1, V / 2 + 17, V - 1, V, V + G->L - 1, V + G->L, S - 1, m.band, m.band + 63, V / 3, vid0, vid0 + 1, vid0 + V - 1};
        for (int c : cand) if (c >= 0 && c < S && (int)rows.size() < check) rows.push_back(c);
        for (int h : {0, 1, H - 1}) if (h < H && (hs.empty() || hs.back() != h)) hs.push_back(h);
        int *drows, *dheads;
        float *scratch, *dout;
        const int nb = (int)(rows.size() * hs.size());
        HIP_OK(hipMalloc(&drows, rows.size() * 4)), HIP_OK(hipMalloc(&dheads, hs.size() * 4));
        HIP_OK(hipMalloc(&scratch, (size_t)nb * S * 4)), HIP_OK(hipMalloc(&dout, (size_t)nb * D * 4));
        HIP_OK(hipMemcpy(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dheads, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
        RefGeom rg = {S, D, G->F, G->P, V, vid0, m, sm_scale, f16};
        ref_rows_kernel<<<nb, 256, 0, st>>>(q, k, v, dflags, dheads, drows, (int)rows.size(), rg, scratch, dout, prescaled);
        HIP_OK(hipStreamSynchronize(st));
        std::vector<float> ref((size_t)nb * D);
        HIP_OK(hipMemcpy(ref.data(), dout, ref.size() * 4, hipMemcpyDeviceToHost));
        std::vector<uint16_t> row(D);
        for (size_t hi = 0; hi < hs.size(); ++hi)
            for (size_t ri = 0; ri < rows.size(); ++ri) {
                const int h = hs[hi], i = rows[ri];
                const int jv = i - vid0;
                const int pr = (hf[h] && jv >= 0 && jv < V) ? vid0 + (jv % G->F) * G->P + jv / G->F : i;
                HIP_OK(hipMemcpy(row.data(), o + ((size_t)h * S + pr) * D, D * 2, hipMemcpyDeviceToHost));
                for (int d = 0; d < D; ++d) {
                    const double got = f16 ? f16_to_f32_host(row[d]) : bf16_to_f32(row[d]), want = ref[(hi * rows.size() + ri) * D + d];
                    err2 += (got - want) * (got - want), ref2 += want * want;
                    maxabs = std::max(maxabs, std::fabs(got - want));
                }
                ++nck;
            }
    }
    const double rel = ref2 > 0 ? std::sqrt(err2 / ref2) : 0.0;
    printf("{\"tool\": \"tools/native_harness\", \"lib\": \"%s\", \"build\": \"%s\", \"geom\": \"%s\", \"H\": %d, \"S\": %d, \"D\": %d, \"dtype\": \"%s\", "
           "\"band\": %d, \"variant\": %d, \"prescaled\": %d, \"switch\": %d, \"head_flags\": \"%s\", \"fill\": \"%s\", \"sclk_mhz\": %.1f, \"mcycles\": %.2f, \"ms\": [",
           lib.c_str(), info ? info() : "?", G->name, H, S, D, dtype.c_str(), m.band, variant, prescaled, sw, flags.c_str(), fill.c_str(), sclk_mhz,
           sclk_mhz * mean * 1e-3);
    for (int i = 0; i < reps; ++i) printf("%s%.3f", i ? ", " : "", ms[i]);
    printf("], \"ms_mean\": %.3f, \"density\": %.4f, \"algorithmic_tflop\": %.3f, \"tflops\": %.1f, \"frac_of_2500\": %.4f, "
           "\"spot_rows\": %d, \"rel_l2\": %.3e, \"max_abs\": %.3e, \"o_checksum\": \"%016llx\"}\n",
           mean, pairs / S / S, flop / 1e12, flop / (mean * 1e-3) / 1e12, flop / (mean * 1e-3) / 2.5e15, nck, rel, maxabs, osum);
    const double tol = f16 ? 1e-3 : (prescaled ? 1e-2 : 3e-3);   // the bounds of tests/test_gpu_fullsize.py / test_gpu_prescaled.py, spot rows
    if (nck > 0 && !(rel <= tol)) { fprintf(stderr, "spot rows: rel. L2 %.3e above %.1e\n", rel, tol); return 4; }
    return 0;
}
