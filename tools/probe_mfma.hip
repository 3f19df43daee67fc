// How much of a wave's own filler instructions hides behind its MFMAs on gfx950, alone on its SIMD and next to a partner wave
// that issues VALU (the two-phase ping-pong situation).  Ticks of s_memtime per MFMA 32x32x16 bf16.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_mfma.hip -o tools/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short i16x4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMAs only (4 accumulators).  1: + two ds_read_b64_tr_b16 and a counted wait per MFMA.  2: + 4 VALU per MFMA.
// 3: 1 + 2.   PARTNER: waves 4..7 of the workgroup spin on VALU instead of running the MFMA loop.
template <int MODE, bool PRIO>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* ticks, int partner) {
    __shared__ __attribute__((aligned(16))) short lds[16384];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (short)(i * 7);
    __syncthreads();
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    if (wave >= 4) {
        if (!partner) return;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < 4000; ++it) {
            asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %4, %5\n"
                         "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0.999f), "v"(0.001f));
        }
        out[threadIdx.x] = a0 + a1 + a2 + a3;
        if (lane == 0) ticks[wave] = __builtin_amdgcn_s_memtime() - t0;
        return;
    }
    bf16x8 A = {1, 2, 3, 4, 5, 6, 7, 8}, B = {1, 1, 2, 2, 3, 3, 4, 4};
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const char* p = (const char*)lds + lane * 8;
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 256; ++it) {
#define STEP(acc, off)                                                                                                         \
        {                                                                                                                      \
            if (MODE & 1) {                                                                                                    \
                i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p + off));       \
                i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p + off + 512)); \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
                A[0] = (__bf16)(float)lo[0]; A[4] = (__bf16)(float)hi[1];                                                      \
            }                                                                                                                  \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc, 0, 0, 0);                                                 \
            if (MODE & 2) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_exp_f32 %1, %1\n v_fma_f32 %2, %2, %4, %5\n v_add_f32 %3, %3, %5" \
                                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0.999f), "v"(0.001f));                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                 \
        }
        STEP(c0, 0) STEP(c1, 1024) STEP(c2, 2048) STEP(c3, 3072)
#undef STEP
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + a0 + a1 + a2 + a3;
    if (lane == 0) ticks[wave] = t1 - t0;
}

// the matrix phase of attn_body_pp2 in isolation: operands read from LDS kPF steps ahead (ring), counted waits, optional VALU slice
template <int VALU, bool LDSCONTEND>
__global__ __launch_bounds__(512) void k_ring(float* out, unsigned long long* ticks, int partner) {
    __shared__ __attribute__((aligned(16))) short lds[32768];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (short)(i * 7);
    __syncthreads();
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    if (wave >= 4) {
        if (!partner) return;
        for (int it = 0; it < 6000; ++it) {
            asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %4, %5\n"
                         "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0.999f), "v"(0.001f));
            if (LDSCONTEND) { volatile short* q = lds + ((it * 64 + lane) & 16383); a0 += (float)q[0]; }
        }
        out[threadIdx.x] = a0 + a1 + a2 + a3;
        return;
    }
    typedef short i16x8 __attribute__((ext_vector_type(8)));
    bf16x8 B = {1, 1, 2, 2, 3, 3, 4, 4};
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    const char* p = (const char*)lds + lane * 8;
    constexpr int PF = 8;
    bf16x8 ring[PF + 1];
    auto fetch = [&](int i) {
        const int off = (i & 31) * 1024;
        i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p + off));
        i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p + off + 512));
        i16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        ring[i % (PF + 1)] = __builtin_bit_cast(bf16x8, both);
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int i = 0; i < PF; ++i) fetch(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 36; ++i) {
            if (i + PF < 36) fetch(i + PF);
            c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i % (PF + 1)], B, c[i & 3], 0, 0, 0);
            if (VALU) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_exp_f32 %1, %1\n v_fma_f32 %2, %2, %4, %5\n v_add_f32 %3, %3, %5"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0.999f), "v"(0.001f));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3] + a0 + a1 + a2 + a3;
    if (lane == 0) ticks[wave] = t1 - t0;
}

// the intra-wave pipelined regime: BOTH waves of every SIMD run { ring fetch, MFMA, NV VALU (every third one a v_exp) } per step
template <int NV>
__global__ __launch_bounds__(512) void k_both(float* out, unsigned long long* ticks, int nwaves) {
    __shared__ __attribute__((aligned(16))) short lds[32768];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (short)(i * 7);
    __syncthreads();
    if (wave >= nwaves) return;
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    typedef short i16x8 __attribute__((ext_vector_type(8)));
    bf16x8 B = {1, 1, 2, 2, 3, 3, 4, 4};
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    const char* p = (const char*)lds + lane * 8;
    constexpr int PF = 8;
    bf16x8 ring[PF + 1];
    auto fetch = [&](int i) {
        const int off = (i & 31) * 1024;
        i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p + off));
        i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p + off + 512));
        i16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        ring[i % (PF + 1)] = __builtin_bit_cast(bf16x8, both);
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int i = 0; i < PF; ++i) fetch(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 36; ++i) {
            if (i + PF < 36) fetch(i + PF);
            c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i % (PF + 1)], B, c[i & 3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (v % 3 == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a1));
                else if (v % 3 == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(0.999f), "v"(0.001f));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(0.001f));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3] + a0 + a1 + a2 + a3;
    if (lane == 0) ticks[wave] = t1 - t0;
}

template <int NV>
void run_both(float* out, unsigned long long* ticks) {
    for (int nw = 4; nw <= 8; nw += 4) {
        hipLaunchKernelGGL((k_both<NV>), dim3(1), dim3(512), 0, 0, out, ticks, nw);
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        printf("both-waves regime, %d VALU per step, %d waves per SIMD: %.1f ticks per MFMA per wave (wave 0), %.1f (wave %d)\n", NV, nw / 4,
               h[0] / (64.0 * 36), h[nw - 1] / (64.0 * 36), nw - 1);
    }
}

template <int VALU, bool C>
void run_ring(const char* name, float* out, unsigned long long* ticks) {
    for (int partner = 0; partner < 2; ++partner) {
        hipLaunchKernelGGL((k_ring<VALU, C>), dim3(1), dim3(512), 0, 0, out, ticks, partner);
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-44s %s: %.1f ticks per MFMA\n", name, partner ? "partner wave issuing VALU" : "alone on the SIMD        ", h[0] / (64.0 * 36));
    }
}

template <int MODE, bool PRIO>
void run(const char* name, float* out, unsigned long long* ticks) {
    for (int partner = 0; partner < 2; ++partner) {
        hipLaunchKernelGGL((k<MODE, PRIO>), dim3(1), dim3(512), 0, 0, out, ticks, partner);
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-44s %s: %.1f ticks per MFMA\n", name, partner ? "partner wave issuing VALU" : "alone on the SIMD        ", h[0] / 1024.0);
    }
}

int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 512 * 4); hipMalloc(&ticks, 64);
    run<0, false>("MFMA only", out, ticks);
    run<1, false>("MFMA + 2 ds_read_tr + wait", out, ticks);
    run<2, false>("MFMA + 4 VALU (1 exp)", out, ticks);
    run<3, false>("MFMA + 2 ds_read_tr + wait + 4 VALU", out, ticks);
    run<3, true>("same, s_setprio 1", out, ticks);
    run<0, true>("MFMA only, s_setprio 1", out, ticks);
    run_ring<0, false>("ring: MFMA + 2 tr reads 8 ahead", out, ticks);
    run_ring<1, false>("ring: + 4 VALU", out, ticks);
    run_ring<1, true>("ring: + 4 VALU, partner also reads LDS", out, ticks);
    run_both<0>(out, ticks);
    run_both<4>(out, ticks);
    run_both<6>(out, ticks);
    run_both<8>(out, ticks);
    return 0;
}
