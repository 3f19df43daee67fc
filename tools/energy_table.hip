// Energy table of the headline kernel's instruction mix on gfx950 ("next" item 3a): which clock does the power
// management grant a full chip (256 workgroups x 8 waves, one per CU, two waves per SIMD in complementary phases like
// attn_body_pp2) that runs
//     32 MFMA 32x32x16 bf16 per wave and tile at a FIXED duty cycle of the matrix pipe (default 0.77, the measured figure)
// when one adds, one at a time, what band_attn_pp2q_kernel issues beside them:
//     L  the LDS operand reads (16 ds_read_b128 + 32 ds_read_b64_tr_b16 per 32 MFMAs = 1.5 per MFMA; also half of that),
//     V  the vector phase (112 VALU per tile and wave = 3.5 per MFMA: 32 v_exp_f32, 32 v_add_f32, 16 v_cvt_pk_bf16_f32, 32 v_fma/max; also half),
//     D  the K / V stream by LDS-DMA (32 KiB per tile and workgroup; all L2 hits, or one request in four from a region streamed once).
// The duty cycle is held by padding the vector phase with s_nop (tuned per mix by bisection on short runs), so every row spends the
// same number of MFMA-busy cycles per wall cycle and the granted clock isolates the POWER each ingredient costs:
//     frac of nominal peak = duty x granted MHz / 2400.
// Clock: s_memtime (shader clock) over wall_clock64 (100 MHz) inside the kernel, summed over workgroups.  Power: best effort, the
// hwmon power1_average / power1_input of the card sampled every 2 ms by a host thread while the measured launch runs.
// build: hipcc --offload-arch=gfx950 -O2 tools/energy_table.hip -o tools/energy_table -lpthread
// usage: tools/energy_table [duty x 100, default 77] [tiles, default 40000] [kernel: only the kernel-like rows] [1: the DMA source is the constant 0x3c of rounds 4 / 5b / 5c]
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>
#include <chrono>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4m __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int kStage = 32 * 1024;      // one K + V tile (64 keys x 128 x 2 B x 2)
constexpr int kStages = 4;

// LDSR: LDS operand reads per 32 MFMAs in units of 1/4 of the real kernel's 48 (0, 2 = half, 4 = all).  NV: VALU per tile in units of
// 1/4 of 112 (0, 2, 4).  DMA: 0 none, 1 all requests hit a 1 MiB window (L2), 2 one request in four streams fresh lines.
template <int LDSR, int NV, int DMA, int SHAPE = 0>
__global__ __launch_bounds__(512) void mix_kernel(const char* __restrict__ src, unsigned long long stream_bytes, int tiles, int pad,
                                                   unsigned long long* __restrict__ ticks, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // pseudo-random bf16 in LDS (operand data that toggles like real K / V)
    for (int i = threadIdx.x; i < kStage * kStages / 4; i += blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        x ^= x >> 13;
        const unsigned hi = 0x3f000000u | ((x & 0x7f00u) << 8) | ((x & 0x8000u) << 16);      // +-[0.5, 1)
        const unsigned lo = 0x3f00u | ((x >> 16) & 0x7fu) | ((x >> 8) & 0x8000u);
        ((unsigned*)lds)[i] = hi | lo;
    }
    __syncthreads();
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = 0;
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    f32x4m d16[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
    bf16x8 Bq[4], Areg[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 8; ++e) {
            Bq[j][e] = (__bf16)(0.01f * (float)(((lane * 8 + e) * 37 + j * 11) % 97 - 48));
            Areg[j][e] = (__bf16)(0.02f * (float)(((lane * 8 + e) * 53 + j * 29) % 89 - 44));
        }
    const char* pk = lds + lane * 16;        // ds_read_b128 addresses (K operand)
    const char* pv = lds + lane * 8;         // ds_read_b64_tr_b16 addresses (V operand)
    const bool second = wave >= 4;           // waves 4-7 run one phase behind waves 0-3
    constexpr int PF = 4;
    bf16x8 ring[PF + 1];
    const unsigned dma_cursor = blockIdx.x * 8u + wave;

    auto fetch = [&](int i, int stage, int slot) {
        // i in 0..31: steps 0..15 read K (one b128 each = 16 reads), steps 16..31 read V^T (two b64_tr each = 32 reads): 48 reads.
        const char* base_k = pk + stage * kStage;
        const char* base_v = pv + stage * kStage + kStage / 2;
        if (i < 16) {
            i32x4 x = *((__attribute__((address_space(3))) i32x4*)(base_k + i * 1024));
            ring[slot] = __builtin_bit_cast(bf16x8, x);
        } else {
            const int off = (i - 16) * 1024;
            i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(base_v + off));
            i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(base_v + off + 512));
            i16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            ring[slot] = __builtin_bit_cast(bf16x8, both);
        }
    };
    auto matrix_phase = [&](int stage) {
        __builtin_amdgcn_s_setprio(1);
        if (LDSR == 4) {
#pragma unroll
            for (int i = 0; i < PF; ++i) fetch(i, stage, i % (PF + 1));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LDSR == 2) {     // half the reads: one fragment feeds two MFMAs (what 64 query rows per wave would do); fragment j = step 2j
#pragma unroll
            for (int j = 0; j < PF; ++j) fetch(2 * j, stage, j % (PF + 1));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (LDSR == 4 && SHAPE == 1) {
                // the same 32 operand fragments per tile, each feeding TWO 16x16x32 MFMAs (the two 16-row blocks of a wave's 32 rows)
                if (i + PF < 32) fetch(i + PF, stage, (i + PF) % (PF + 1));
                d16[(2 * i) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[i % (PF + 1)], Bq[i & 3], d16[(2 * i) & 7], 0, 0, 0);
                d16[(2 * i + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[i % (PF + 1)], Bq[(i + 1) & 3], d16[(2 * i + 1) & 7], 0, 0, 0);
            } else if (LDSR == 4) {
                if (i + PF < 32) fetch(i + PF, stage, (i + PF) % (PF + 1));
                c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i % (PF + 1)], Bq[i & 3], c[i & 3], 0, 0, 0);
            } else if (LDSR == 2) {
                const int j = i >> 1;
                if ((i & 1) == 0 && j + PF < 16) fetch(2 * (j + PF), stage, (j + PF) % (PF + 1));
                c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[j % (PF + 1)], Bq[i & 3], c[i & 3], 0, 0, 0);
            } else if (SHAPE == 1) {
                d16[(2 * i) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Areg[(i >> 2) & 3], Bq[i & 3], d16[(2 * i) & 7], 0, 0, 0);
                d16[(2 * i + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Areg[(i >> 1) & 3], Bq[(i + 1) & 3], d16[(2 * i + 1) & 7], 0, 0, 0);
            } else {
                c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Areg[(i >> 2) & 3], Bq[i & 3], c[i & 3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto vector_phase = [&](int t) {
#pragma unroll
        for (int r = 0; r < NV * 4; ++r) {       // 7 VALU per round, 16 rounds at NV = 4: 112
            // bounded pseudo-random walk (values stay finite, so the operands keep toggling): 2 v_exp, 2 v_fract, add, fma, cvt_pk
            asm volatile("v_exp_f32 %0, %1\n\tv_add_f32 %1, %1, %0\n\tv_fract_f32 %1, %1\n\tv_exp_f32 %2, %3\n\t"
                         "v_fma_f32 %3, %3, %6, %2\n\tv_fract_f32 %3, %3\n\tv_cvt_pk_bf16_f32 %4, %0, %2"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=v"(a4) : "v"(0.001f), "v"(0.999f));
        }
        if (DMA) {
            // this wave's share of the next K / V tile: 4 KiB = 4 DMA instructions of 64 lanes x 16 B
            const unsigned stage = (unsigned)((t + 3) % kStages);     // three tiles ahead (the kernel: two / three)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds + stage * kStage + (wave * 4 + j) * 1024u);
                unsigned long long off;
                if (DMA == 2 && j == 3) {
                    off = (((unsigned long long)t * 2048ull + dma_cursor) * 1024ull) % stream_bytes;   // 1 KiB nobody else reads: 2 MiB per tile period chip-wide
                } else {
                    off = (((unsigned)t * 32768u + (wave * 4 + j) * 1024u) & (1048576u - 1));    // a 1 MiB window shared by the whole chip
                }
                const char* base = src + (off & ~1023ull);
                const unsigned voff = lane * 16u;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsb), "v"(voff), "s"(base) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // the requests of two phases may still be in flight
        }
        for (int i = 0; i < pad; ++i) asm volatile("s_nop 15");
    };

    __syncthreads();
    const unsigned long long w0 = wall_clock64(), t0 = __builtin_amdgcn_s_memtime();
    if (second) __syncthreads();      // one phase behind
    for (int t = 0; t < tiles; ++t) {
        matrix_phase(t % kStages);
        __syncthreads();
        vector_phase(t);
        __syncthreads();
    }
    if (!second) __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        ticks[blockIdx.x * 2 + 0] = t1 - t0;
        ticks[blockIdx.x * 2 + 1] = w1 - w0;
    }
    float acc16 = 0.f;
    for (int j = 0; j < 8; ++j) acc16 += d16[j][0] + d16[j][3];
    sink[blockIdx.x * 512 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3] + a0 + a1 + a2 + a3 + a4 + acc16;
}

// "the kernel" row (round 5, VERDICT r04 next #1b): a synthetic stream that should REPRODUCE band_attn_m16_kernel's operating point (matrix pipe
// busy 0.64, 1.96 - 2.04 GHz) — if it does, the table's other rows can be trusted for what-if questions about that kernel.  Beyond mix_kernel<4, 4, 2, 1>:
//   * the tile structure of attn_body_m16 with ONE barrier per tile (leading waves: N, barrier, M; lagging waves: barrier, N, M);
//   * XV extra VALU and XS extra SALU per tile in the vector phase (the mask classification, DMA address arithmetic, reference test: the listing
//     has ~24 vector and ~50 scalar instructions beside the 112);
//   * the K / V stream at the kernel's prefetch distance (2 tiles for the leading, 3 for the lagging waves; the wait leaves one tile's four
//     requests in flight), every `miss_every`-th request of a wave going to a window of `window` bytes that it shares with every other workgroup
//     in a scattered order (64 MiB - 192 MiB: resident in the 256 MiB Infinity Cache, not in a 4 MiB L2: the kernel's 19 % L2 misses are its
//     neighbours' tiles, i.e. fabric / MALL traffic, not first-touch HBM reads), the others to a 1 MiB window (L2 hits).
template <int XV, int XS>
__global__ __launch_bounds__(512) void kernel_like(const char* __restrict__ src, unsigned long long window, int miss_every, int normal, int tiles,
                                                    unsigned long long* __restrict__ ticks, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // LDS operand data: normal = 0: +-[0.5, 1) with random mantissas (ONE exponent, as mix_kernel); 1: approximately N(0, 1) bf16 (sum of four uniform bytes:
    // exponents vary over ~8 binades like real K / V rows — products then need alignment shifts in the MFMA's adder tree)
    for (int i = threadIdx.x; i < kStage * kStages / 4; i += blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        x ^= x >> 13;
        if (normal) {
            unsigned y = x * 2246822519u;
            y ^= y >> 15;
            const float f0 = ((float)((x & 255u) + ((x >> 8) & 255u) + ((x >> 16) & 255u) + (x >> 24)) - 510.f) * (1.f / 148.f);
            const float f1 = ((float)((y & 255u) + ((y >> 8) & 255u) + ((y >> 16) & 255u) + (y >> 24)) - 510.f) * (1.f / 148.f);
            ((unsigned*)lds)[i] = (__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xffff0000u);
        } else {
            const unsigned hi = 0x3f000000u | ((x & 0x7f00u) << 8) | ((x & 0x8000u) << 16);
            const unsigned lo = 0x3f00u | ((x >> 16) & 0x7fu) | ((x >> 8) & 0x8000u);
            ((unsigned*)lds)[i] = hi | lo;
        }
    }
    __syncthreads();
    unsigned x0 = threadIdx.x * 7u + 1u, x1 = threadIdx.x * 13u + 5u;
    f32x4m d16[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
    bf16x8 Bq[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 8; ++e) {
            unsigned z = (unsigned)((lane * 8 + e) * 4 + j) * 2654435761u;
            z ^= z >> 13;
            Bq[j][e] = normal ? (__bf16)(((float)((z & 255u) + ((z >> 8) & 255u) + ((z >> 16) & 255u) + (z >> 24)) - 510.f) * (1.f / 148.f))
                              : (__bf16)(0.01f * (float)(((lane * 8 + e) * 37 + j * 11) % 97 - 48));
        }
    const char* pk = lds + lane * 16;
    const char* pv = lds + lane * 8;
    const bool second = wave >= 4;
    constexpr int PF = 8;
    bf16x8 ring[PF + 1];
    auto fetch = [&](int i, int stage, int slot) {
        const char* base_k = pk + stage * kStage;
        const char* base_v = pv + stage * kStage + kStage / 2;
        if (i >= 16) {   // (the kernel's order: V fragments first, then K)
            i32x4 x = *((__attribute__((address_space(3))) i32x4*)(base_k + (i - 16) * 1024));
            ring[slot] = __builtin_bit_cast(bf16x8, x);
        } else {
            const int off = i * 1024;
            i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(base_v + off));
            i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(base_v + off + 512));
            i16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            ring[slot] = __builtin_bit_cast(bf16x8, both);
        }
    };
    auto matrix_phase = [&](int stage) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < PF; ++i) fetch(i, stage, i % (PF + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i + PF < 32) fetch(i + PF, stage, (i + PF) % (PF + 1));
            d16[(2 * i) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[i % (PF + 1)], Bq[i & 3], d16[(2 * i) & 7], 0, 0, 0);
            d16[(2 * i + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[i % (PF + 1)], Bq[(i + 1) & 3], d16[(2 * i + 1) & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    const int dist = second ? 3 : 2;
    unsigned req = (blockIdx.x * 8u + wave) * 977u;
    unsigned miss_seq = 0;   // request counter of this wave (which requests miss)
    auto dma = [&](int t) {
        const unsigned stage = (unsigned)((t + dist) % kStages);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds + stage * kStage + (wave * 4 + j) * 1024u);
            unsigned long long off;
            ++req;
            if (miss_every > 0 && (req % (unsigned)miss_every) == 0u) {
                // this wave's own slice of the window, walked 1 KiB at a time (a sequential stream like a neighbour's K / V tiles: page-friendly; the
                // slices of an XCD's 256 waves together exceed its 4 MiB L2 from a 64 MiB window on, so every such request goes to the fabric)
                const unsigned long long slice = window / 2048ull;
                const unsigned long long wid = (unsigned long long)blockIdx.x * 8ull + (unsigned long long)wave;
                off = (1ull << 20) + wid * slice + (((unsigned long long)(miss_seq++) << 10) % slice);
            } else {
                off = (((unsigned)t * 32768u + (wave * 4 + j) * 1024u) & (1048576u - 1));
            }
            const char* base = src + off;
            const unsigned voff = lane * 16u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsb), "v"(voff), "s"(base) : "memory");
        }
    };
    // 32 elements per lane and tile like the kernel's probabilities: e = 2^x, y = fma(x, 0.5, e), x = y + k (k in [-1.5, -0.5), a new one per group of
    // eight and tile: the values wander in [-2.3, 0.5) and keep toggling), one cvt_pk per two e — 32 v_exp, 32 v_fma, 32 v_add, 16 v_cvt_pk, eight independent
    // elements per group (the kernel's vector phase has no dependent chains either)
    float xs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xs[e] = -0.25f * (float)e - 0.001f * (float)(threadIdx.x & 63);
    float pk0 = 0.f, pk1 = 0.f, pk2 = 0.f, pk3 = 0.f;
    auto vector_phase = [&](int t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned hsh = ((unsigned)t * 4u + (unsigned)g) * 2654435761u;
            hsh ^= hsh >> 15;
            const float kk = -1.5f + (float)(hsh & 1023u) * (1.f / 1024.f);
            float e0, e1, e2, e3, e4, e5, e6, e7;
            asm volatile("v_exp_f32 %0, %8\n\tv_exp_f32 %1, %9\n\tv_exp_f32 %2, %10\n\tv_exp_f32 %3, %11\n\t"
                         "v_exp_f32 %4, %12\n\tv_exp_f32 %5, %13\n\tv_exp_f32 %6, %14\n\tv_exp_f32 %7, %15\n\t"
                         "v_fma_f32 %8, %8, 0.5, %0\n\tv_fma_f32 %9, %9, 0.5, %1\n\tv_fma_f32 %10, %10, 0.5, %2\n\tv_fma_f32 %11, %11, 0.5, %3\n\t"
                         "v_fma_f32 %12, %12, 0.5, %4\n\tv_fma_f32 %13, %13, 0.5, %5\n\tv_fma_f32 %14, %14, 0.5, %6\n\tv_fma_f32 %15, %15, 0.5, %7\n\t"
                         "v_add_f32 %8, %8, %20\n\tv_add_f32 %9, %9, %20\n\tv_add_f32 %10, %10, %20\n\tv_add_f32 %11, %11, %20\n\t"
                         "v_add_f32 %12, %12, %20\n\tv_add_f32 %13, %13, %20\n\tv_add_f32 %14, %14, %20\n\tv_add_f32 %15, %15, %20\n\t"
                         "v_cvt_pk_bf16_f32 %16, %0, %1\n\tv_cvt_pk_bf16_f32 %17, %2, %3\n\tv_cvt_pk_bf16_f32 %18, %4, %5\n\tv_cvt_pk_bf16_f32 %19, %6, %7"
                         : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "=&v"(e4), "=&v"(e5), "=&v"(e6), "=&v"(e7), "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]),
                           "+v"(xs[3]), "+v"(xs[4]), "+v"(xs[5]), "+v"(xs[6]), "+v"(xs[7]), "=v"(pk0), "=v"(pk1), "=v"(pk2), "=v"(pk3)
                         : "v"(kk));
            if (g == 0) dma(t);
        }
#pragma unroll
        for (int r = 0; r < XV / 2; ++r) asm volatile("v_xor_b32 %0, %0, %1\n\tv_add_u32 %1, %1, %0" : "+v"(x0), "+v"(x1));
        unsigned s0 = (unsigned)t, s1 = 17u;
#pragma unroll
        for (int r = 0; r < XS / 2; ++r) asm volatile("s_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0" : "+s"(s0), "+s"(s1) : : "scc");
        asm volatile("" :: "s"(s0), "s"(s1));
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    };
    for (int t = 0; t < 2; ++t) dma(t - dist);   // (stages 0, 1 requested up front; the addresses do not matter)
    __syncthreads();
    const unsigned long long w0 = wall_clock64(), t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        if (second) __syncthreads();
        vector_phase(t);
        if (!second) __syncthreads();
        matrix_phase(t % kStages);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        ticks[blockIdx.x * 2 + 0] = t1 - t0;
        ticks[blockIdx.x * 2 + 1] = w1 - w0;
    }
    float acc16 = 0.f;
    for (int j = 0; j < 8; ++j) acc16 += d16[j][0] + d16[j][3];
    sink[blockIdx.x * 512 + threadIdx.x] = xs[0] + xs[3] + xs[7] + pk0 + pk1 + pk2 + pk3 + acc16 + (float)(x0 ^ x1);
}

// duty 1.0 reference: both waves of a SIMD issue MFMAs back to back, nothing else.  SHAPE 0: 32x32x16 (32 per tile and wave), 1: 16x16x32
// (64 per tile and wave: the same FLOPs).  TOGGLE: four different A and four different B fragments in rotation (operand data that changes
// from MFMA to MFMA, like real K / V / P); else one constant fragment of small values for everything.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE, bool TOGGLE>
__global__ __launch_bounds__(512) void mfma_only_kernel(int tiles, unsigned long long* __restrict__ ticks, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    f32x4 d[4] = {{0}, {0}, {0}, {0}};
    bf16x8 Bq[4], Areg[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 8; ++e) {
            Bq[j][e] = TOGGLE ? (__bf16)(0.01f * (float)(((lane * 8 + e) * 37 + j * 11) % 97 - 48)) : (__bf16)0.0078125f;
            Areg[j][e] = TOGGLE ? (__bf16)(0.02f * (float)(((lane * 8 + e) * 53 + j * 29) % 89 - 44)) : (__bf16)0.0078125f;
        }
    const unsigned long long w0 = wall_clock64(), t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        if (SHAPE == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Areg[(i >> 2) & 3], Bq[i & 3], c[i & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                d[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Areg[(i >> 2) & 3], Bq[i & 3], d[i & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (lane == 0) {       // every wave reports: the older wave of a SIMD wins the arbitration and finishes first
        ticks[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 0] = t1 - t0;
        ticks[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = w1 - w0;
    }
    sink[blockIdx.x * 512 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3] + d[0][0] + d[1][1] + d[2][2] + d[3][3];
}

__global__ void fill_src_kernel(unsigned* dst, unsigned long long nwords, int constant) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + (unsigned)(i >> 32) * 40503u;
        x ^= x >> 13;
        unsigned y = x * 2246822519u;
        y ^= y >> 15;
        const float f0 = ((float)((x & 255u) + ((x >> 8) & 255u) + ((x >> 16) & 255u) + (x >> 24)) - 510.f) * (1.f / 148.f);
        const float f1 = ((float)((y & 255u) + ((y >> 8) & 255u) + ((y >> 16) & 255u) + (y >> 24)) - 510.f) * (1.f / 148.f);
        dst[i] = constant ? 0x3c3c3c3cu : ((__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xffff0000u));
    }
}

struct Power {
    std::string path;
    std::atomic<bool> run{false};
    std::vector<double> samples;
    std::thread th;
    Power() {
        glob_t g;
        for (const char* pat : {"/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"}) {
            if (glob(pat, 0, nullptr, &g) == 0 && g.gl_pathc > 0) { path = g.gl_pathv[0]; globfree(&g); break; }
        }
    }
    double read_once() const {
        if (path.empty()) return 0;
        FILE* f = fopen(path.c_str(), "r");
        if (!f) return 0;
        double uw = 0;
        if (fscanf(f, "%lf", &uw) != 1) uw = 0;
        fclose(f);
        return uw * 1e-6;
    }
    void start() {
        samples.clear();
        if (path.empty()) return;
        run = true;
        th = std::thread([this] { while (run) { samples.push_back(read_once()); std::this_thread::sleep_for(std::chrono::milliseconds(2)); } });
    }
    double stop() {
        if (path.empty()) return 0;
        run = false;
        th.join();
        // the sensor averages over a window: take the upper half of the samples (the launch at steady state)
        if (samples.empty()) return 0;
        double mx = 0;
        for (double s : samples) mx = s > mx ? s : mx;
        return mx;
    }
};

struct Result { double mhz, duty, ms, watts; };

template <int LDSR, int NV, int DMA, int SHAPE = 0>
Result launch(const char* src, unsigned long long stream_bytes, int tiles, int pad, unsigned long long* ticks, float* sink, Power* pw) {
    CHECK(hipFuncSetAttribute((const void*)mix_kernel<LDSR, NV, DMA, SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, kStage * kStages));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    if (pw) pw->start();
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix_kernel<LDSR, NV, DMA, SHAPE>), dim3(256), dim3(512), kStage * kStages, 0, src, stream_bytes, tiles, pad, ticks, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    const double watts = pw ? pw->stop() : 0;
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(512);
    CHECK(hipMemcpy(h.data(), ticks, 512 * 8, hipMemcpyDeviceToHost));
    double sh = 0, wl = 0;
    for (int b = 0; b < 256; ++b) { sh += (double)h[2 * b]; wl += (double)h[2 * b + 1]; }
    Result r;
    r.ms = ms;
    r.mhz = 100.0 * sh / wl;
    r.duty = (double)tiles * 64.0 * 32.0 / (ms * 1e-3 * r.mhz * 1e6);     // two waves per SIMD x 32 MFMAs x 32 cycles per tile, over the launch's cycles
    r.ms = ms;
    r.watts = watts;
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return r;
}

template <int LDSR, int NV, int DMA, int SHAPE = 0>
void row(const char* name, double target, int tiles, const char* src, unsigned long long stream_bytes, unsigned long long* ticks, float* sink, Power* pw) {
    // bisection on the pad for the target duty (short runs), then one warm launch and the measured one
    int lo = 0, hi = 160, pad = 0;
    Result r0 = launch<LDSR, NV, DMA, SHAPE>(src, stream_bytes, 3000, 0, ticks, sink, nullptr);
    if (r0.duty > target) {
        for (int it = 0; it < 8 && lo < hi; ++it) {
            const int mid = (lo + hi) / 2;
            Result r = launch<LDSR, NV, DMA, SHAPE>(src, stream_bytes, 3000, mid, ticks, sink, nullptr);
            if (r.duty > target) lo = mid + 1; else hi = mid;
        }
        pad = lo;
    }
    launch<LDSR, NV, DMA, SHAPE>(src, stream_bytes, tiles, pad, ticks, sink, nullptr);
    Result r = launch<LDSR, NV, DMA, SHAPE>(src, stream_bytes, tiles, pad, ticks, sink, pw);
    printf("| %-58s | %3d | %.3f | %.3f | %4.0f | %.4f | %6.1f | %5.0f |\n", name, pad, r0.duty, r.duty, r.mhz, r.duty * r.mhz / 2400.0, r.ms, r.watts);
    fflush(stdout);
}

// kernel_like2 (third form): the DATA PATHS of attn_body_m16 as well — 32 O accumulators fed by V fragments (LDS) x P fragments, 8 S accumulators
// started from zero every tile and fed by K fragments (LDS) x 8 constant Q fragments, probabilities p = 2^(s c - m) computed from those S accumulators
// (32 v_fma, 32 v_exp, 32 v_add, 16 v_cvt_pk per tile) and packed into the P fragments the next matrix phase multiplies: every operand and every
// result toggles the way the kernel's do.  The first two forms kept 8 accumulators running and constant B operands and were NOT limited by power on
// the boxes of r05b / r05c (2375 - 2382 MHz in every row) where the kernel itself is granted 2.0 GHz.
template <int XV, int XS>
__global__ __launch_bounds__(512, 2) void kernel_like2(const char* __restrict__ src, unsigned long long window, int miss_every, int zero_data, int tiles,
                                                     unsigned long long* __restrict__ ticks, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < kStage * kStages / 4; i += blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        x ^= x >> 13;
        unsigned y = x * 2246822519u;
        y ^= y >> 15;
        const float f0 = ((float)((x & 255u) + ((x >> 8) & 255u) + ((x >> 16) & 255u) + (x >> 24)) - 510.f) * (1.f / 148.f);
        const float f1 = ((float)((y & 255u) + ((y >> 8) & 255u) + ((y >> 16) & 255u) + (y >> 24)) - 510.f) * (1.f / 148.f);
        ((unsigned*)lds)[i] = zero_data ? 0u : ((__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xffff0000u));
    }
    __syncthreads();
    unsigned x0 = threadIdx.x * 7u + 1u, x1 = threadIdx.x * 13u + 5u;
    f32x4m acc_o[8][2], sc[4][2];
    bf16x8 qf[2][4], pf[2][2];
    for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 2; ++b) acc_o[a][b] = f32x4m{0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 2; ++b) sc[a][b] = f32x4m{0.f, 0.f, 0.f, 0.f};
    for (int rb = 0; rb < 2; ++rb)
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 8; ++e) {
                unsigned z = (unsigned)(((lane * 8 + e) * 4 + j) * 2 + rb) * 2654435761u;
                z ^= z >> 13;
                qf[rb][j][e] = zero_data ? (__bf16)0.f : (__bf16)(((float)((z & 255u) + ((z >> 8) & 255u) + ((z >> 16) & 255u) + (z >> 24)) - 510.f) * (1.f / 148.f));
            }
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int e = 0; e < 8; ++e) pf[a][b][e] = (__bf16)0.f;
    const char* pk = lds + lane * 16;
    const char* pv = lds + lane * 8;
    const bool second = wave >= 4;
    constexpr int PF = 8;
    bf16x8 ring[PF + 1];
    auto fetch = [&](int i, int stage, int slot) {   // steps 0..15: V fragments (two transposing reads each), 16..31: K fragments
        const char* base_k = pk + stage * kStage;
        const char* base_v = pv + stage * kStage + kStage / 2;
        if (i >= 16) {
            i32x4 x = *((__attribute__((address_space(3))) i32x4*)(base_k + (i - 16) * 1024));
            ring[slot] = __builtin_bit_cast(bf16x8, x);
        } else {
            const int off = i * 1024;
            i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(base_v + off));
            i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(base_v + off + 512));
            i16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            ring[slot] = __builtin_bit_cast(bf16x8, both);
        }
    };
    const f32x4m zero4 = {0.f, 0.f, 0.f, 0.f};
    auto matrix_phase = [&](int stage) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < PF; ++i) fetch(i, stage, i % (PF + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i + PF < 32) fetch(i + PF, stage, (i + PF) % (PF + 1));
            const bf16x8 a = ring[i % (PF + 1)];
            if (i < 16) {            // O^T += V^T P^T: chunk kc = i / 8, d block i % 8
                acc_o[i & 7][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pf[i >> 3][0], acc_o[i & 7][0], 0, 0, 0);
                acc_o[i & 7][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pf[i >> 3][1], acc_o[i & 7][1], 0, 0, 0);
            } else {                 // S^T = K Q^T: contraction step ks = (i - 16) / 4, key block (i - 16) % 4
                const int j = i - 16, ks = j >> 2, b = j & 3;
                sc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[0][ks], ks == 0 ? zero4 : sc[b][0], 0, 0, 0);
                sc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[1][ks], ks == 0 ? zero4 : sc[b][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    const int dist = second ? 3 : 2;
    unsigned req = (blockIdx.x * 8u + wave) * 977u;
    unsigned miss_seq = 0;
    auto dma = [&](int t) {
        const unsigned stage = (unsigned)((t + dist) % kStages);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds + stage * kStage + (wave * 4 + j) * 1024u);
            unsigned long long off;
            ++req;
            if (miss_every > 0 && (req % (unsigned)miss_every) == 0u) {
                const unsigned long long slice = window / 2048ull;
                const unsigned long long wid = (unsigned long long)blockIdx.x * 8ull + (unsigned long long)wave;
                off = (1ull << 20) + wid * slice + (((unsigned long long)(miss_seq++) << 10) % slice);
            } else {
                off = (((unsigned)t * 32768u + (wave * 4 + j) * 1024u) & (1048576u - 1));
            }
            const char* base = src + off;
            const unsigned voff = lane * 16u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsb), "v"(voff), "s"(base) : "memory");
        }
    };
    float psum0 = 0.f, psum1 = 0.f, lrun = 0.f;
    const float c_log2 = 0.1275174f, m_ref = 4.0f;     // scores ~ N(0, 128) x c: exponent arguments ~ N(-4, 2)
    auto vector_phase = [&](int t) {
        dma(t);
        psum0 = 0.f, psum1 = 0.f;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * kc + h][rb][r], c_log2, -m_ref));
                        if (rb) psum1 += pr; else psum0 += pr;
                        pf[kc][rb][4 * h + r] = (__bf16)pr;
                    }
        lrun += psum0 + psum1;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) asm volatile("" : "+v"(pf[kc][0]), "+v"(pf[kc][1]));
#pragma unroll
        for (int r = 0; r < XV / 2; ++r) asm volatile("v_xor_b32 %0, %0, %1\n\tv_add_u32 %1, %1, %0" : "+v"(x0), "+v"(x1));
        unsigned s0 = (unsigned)t, s1 = 17u;
#pragma unroll
        for (int r = 0; r < XS / 2; ++r) asm volatile("s_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0" : "+s"(s0), "+s"(s1) : : "scc");
        asm volatile("" :: "s"(s0), "s"(s1));
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    };
    for (int t = 0; t < 2; ++t) dma(t - dist);
    __syncthreads();
    const unsigned long long w0 = wall_clock64(), t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        if (second) __syncthreads();
        vector_phase(t);
        if (!second) __syncthreads();
        matrix_phase(t % kStages);
        if ((t & 255) == 255) {     // keep O bounded (the kernel's exact path does this when the reference moves)
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc_o[a][b] *= 0.00390625f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        ticks[blockIdx.x * 2 + 0] = t1 - t0;
        ticks[blockIdx.x * 2 + 1] = w1 - w0;
    }
    float acc = lrun;
    for (int a = 0; a < 8; ++a) acc += acc_o[a][0][0] + acc_o[a][1][3];
    sink[blockIdx.x * 512 + threadIdx.x] = acc + (float)(x0 ^ x1);
}

template <int XV, int XS>
void kernel_row2(const char* name, const char* src, unsigned long long window, int miss_every, int zero_data, int tiles, unsigned long long* ticks, float* sink, Power* pw) {
    CHECK(hipFuncSetAttribute((const void*)kernel_like2<XV, XS>, hipFuncAttributeMaxDynamicSharedMemorySize, kStage * kStages));
    double mhz = 0, duty = 0, watts = 0;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        if (rep == 2) pw->start();
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((kernel_like2<XV, XS>), dim3(256), dim3(512), kStage * kStages, 0, src, window, miss_every, zero_data, tiles, ticks, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        if (rep == 2) watts = pw->stop();
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(512);
        CHECK(hipMemcpy(h.data(), ticks, 512 * 8, hipMemcpyDeviceToHost));
        double sh = 0, wl = 0;
        for (int b = 0; b < 256; ++b) { sh += (double)h[2 * b]; wl += (double)h[2 * b + 1]; }
        mhz = 100.0 * sh / wl;
        duty = (double)tiles * 64.0 * 32.0 / (ms * 1e-3 * mhz * 1e6);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    printf("| %-58s | %3d | %.3f | %.3f | %4.0f | %.4f | %6.1f | %5.0f |\n", name, 0, duty, duty, mhz, duty * mhz / 2400.0, ms, watts);
    fflush(stdout);
}

template <int XV, int XS>
void kernel_row(const char* name, const char* src, unsigned long long window, int miss_every, int normal, int tiles, unsigned long long* ticks, float* sink, Power* pw) {
    CHECK(hipFuncSetAttribute((const void*)kernel_like<XV, XS>, hipFuncAttributeMaxDynamicSharedMemorySize, kStage * kStages));
    double mhz = 0, duty = 0, watts = 0;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        if (rep == 2) pw->start();
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((kernel_like<XV, XS>), dim3(256), dim3(512), kStage * kStages, 0, src, window, miss_every, normal, tiles, ticks, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        if (rep == 2) watts = pw->stop();
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(512);
        CHECK(hipMemcpy(h.data(), ticks, 512 * 8, hipMemcpyDeviceToHost));
        double sh = 0, wl = 0;
        for (int b = 0; b < 256; ++b) { sh += (double)h[2 * b]; wl += (double)h[2 * b + 1]; }
        mhz = 100.0 * sh / wl;
        duty = (double)tiles * 64.0 * 32.0 / (ms * 1e-3 * mhz * 1e6);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    printf("| %-58s | %3d | %.3f | %.3f | %4.0f | %.4f | %6.1f | %5.0f |\n", name, 0, duty, duty, mhz, duty * mhz / 2400.0, ms, watts);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double target = (argc > 1 ? atof(argv[1]) : 77.0) / 100.0;
    const int tiles = argc > 2 ? atoi(argv[2]) : 40000;
    const unsigned long long stream_bytes = 2ull << 30;
    char* src; unsigned long long* ticks; float* sink;
    CHECK(hipMalloc(&src, stream_bytes + (2 << 20)));
    // what the LDS-DMA streams into the stages: approximately N(0, 1) bf16 like the LDS image's initial content.  (Through round 5's second form this
    // buffer was a memset of 0x3c bytes: after the first tiles every stage held ONE constant, the MFMA operands stopped toggling, and the rows with a
    // DMA stream were granted MORE clock than the rows without — profiles/r04c_energy_table.txt, r05b / r05c: 2375 MHz in every kernel-like row.)
    fill_src_kernel<<<4096, 256>>>((unsigned*)src, (stream_bytes + (2 << 20)) / 4, argc > 4 ? atoi(argv[4]) : 0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMalloc(&ticks, 256 * 8 * 2 * 8));
    CHECK(hipMalloc(&sink, 256 * 512 * 4));
    Power pw;
    const bool only_kernel_rows = argc > 3 && std::string(argv[3]) == "kernel";
    printf("energy table: 256 workgroups x 8 waves, 32 MFMA 32x32x16 bf16 per wave and tile, %d tiles per launch, target duty %.2f; power sensor: %s\n",
           tiles, target, pw.path.empty() ? "none found" : pw.path.c_str());
    printf("| mix (per 32 MFMAs and wave)                                | pad | duty unpadded | duty | MHz | frac of nominal peak | ms | W (max sample) |\n|---|---|---|---|---|---|---|---|\n");
    auto only = [&](auto kern, const char* name) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            if (rep) pw.start();
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, tiles, ticks, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            const double watts = rep ? pw.stop() : 0;
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(256 * 8 * 2);
            CHECK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
            double sh = 0, wl = 0;
            for (int b = 0; b < 256 * 8; ++b) { sh += (double)h[2 * b]; wl += (double)h[2 * b + 1]; }
            const double mhz = 100.0 * sh / wl, duty = (double)tiles * 64.0 * 32.0 / (ms * 1e-3 * mhz * 1e6);
            if (rep) printf("| %-58s | %3d | %.3f | %.3f | %4.0f | %.4f | %6.1f | %5.0f |\n", name, 0, duty, duty, mhz, duty * mhz / 2400.0, ms, watts);
        }
        fflush(stdout);
    };
    if (!only_kernel_rows) {
    only(mfma_only_kernel<0, true>, "MFMA 32x32x16 only, back to back, operands toggling");
    only(mfma_only_kernel<0, false>, "MFMA 32x32x16 only, back to back, constant operands");
    only(mfma_only_kernel<1, true>, "MFMA 16x16x32 only, back to back, operands toggling");
    row<0, 0, 0>("MFMA at the target duty (pad only)", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<2, 0, 0>("+ 24 LDS operand reads (half: 64 rows per wave)", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<4, 0, 0>("+ 48 LDS operand reads (the kernel's 1.5 per MFMA)", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<0, 2, 0>("+ 56 VALU (half)", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<0, 4, 0>("+ 112 VALU (the kernel's 3.5 per MFMA, 32 of them v_exp)", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<0, 0, 1>("+ K/V stream by LDS-DMA, 32 KiB per tile, L2 hits", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<0, 0, 2>("+ K/V stream by LDS-DMA, one request in four misses L2", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<4, 4, 0>("+ 48 LDS + 112 VALU", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<4, 4, 1>("+ 48 LDS + 112 VALU + DMA (L2 hits)", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<4, 4, 2>("+ 48 LDS + 112 VALU + DMA (1/4 misses): the kernel", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<2, 4, 2>("  the kernel with half the LDS reads", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<4, 2, 2>("  the kernel with half the VALU", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<4, 4, 1>("  the kernel with no L2 misses (repeat of the L2-hit row)", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<2, 2, 1>("  half LDS, half VALU, no misses", target, tiles, src, stream_bytes, ticks, sink, &pw);
    printf("| the same with v_mfma_f32_16x16x32_bf16 (64 per tile and wave, every operand fragment feeds two) | | | | | | | |\n");
    row<0, 0, 0, 1>("16x16x32: MFMA at the target duty (pad only)", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<4, 0, 0, 1>("16x16x32: + 48 LDS operand reads", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<4, 4, 0, 1>("16x16x32: + 48 LDS + 112 VALU", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<4, 4, 1, 1>("16x16x32: + 48 LDS + 112 VALU + DMA (L2 hits)", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<4, 4, 2, 1>("16x16x32: + 48 LDS + 112 VALU + DMA (1/4 misses)", target, tiles, src, stream_bytes, ticks, sink, &pw);
    row<4, 4, 1, 0>("32x32x16 again: + 48 LDS + 112 VALU + DMA (L2 hits)", target, tiles, src, stream_bytes, ticks, sink, &pw);
    }
    printf("| kernel-like rows: attn_body_m16's tile structure, one barrier per tile, prefetch distance 2 / 3, no padding (what should reproduce duty 0.64 at 1.96 - 2.04 GHz) | | | | | | | |\n");
    printf("| third form (kernel_like2): the kernel's data paths — P from the S accumulators, 32 + 8 accumulators, S restarted every tile | | | | | | | |\n");
    kernel_row2<0, 0>("like2, N(0,1) data, no misses, no bookkeeping", src, 192ull << 20, 0, 0, tiles, ticks, sink, &pw);
    kernel_row2<24, 48>("like2, N(0,1) data, no misses, 24 VALU + 48 SALU", src, 192ull << 20, 0, 0, tiles, ticks, sink, &pw);
    kernel_row2<0, 0>("like2, N(0,1) data, 1 in 5 to 192 MiB, no bookkeeping", src, 192ull << 20, 5, 0, tiles, ticks, sink, &pw);
    kernel_row2<24, 48>("like2, N(0,1) data, 1 in 5 to 192 MiB, 24 + 48", src, 192ull << 20, 5, 0, tiles, ticks, sink, &pw);
    kernel_row2<0, 0>("like2, zero Q / first LDS image, 1 in 5, no bookkeeping", src, 192ull << 20, 5, 1, tiles, ticks, sink, &pw);
    kernel_row2<24, 48>("like2, zero Q / first LDS image, 1 in 5 to 192 MiB, 24 + 48", src, 192ull << 20, 5, 1, tiles, ticks, sink, &pw);
    kernel_row<0, 0>("second form again: N(0,1) LDS data, 1 in 5, no bookkeeping", src, 192ull << 20, 5, 1, tiles, ticks, sink, &pw);
    return 0;
}
