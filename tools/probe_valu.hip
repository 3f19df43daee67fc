// Issue cost of the VALU instructions of the softmax on gfx950, in s_memtime ticks per wave-instruction, with one and
// with two waves per SIMD (256 / 512 threads in one workgroup).  Eight independent chains per instruction kind.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_valu.hip -o tools/probe_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void k(float* out, unsigned long long* ticks, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float c = 0.999f, d = 0.001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 16; ++it) {
        if (KIND == 0) {
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 1) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (KIND == 2) {
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));)
        } else if (KIND == 3) {
            REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 4) {
            REP8(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 5) {  // packed f32 fma: two elements per instruction
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3"
                              : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(const double*)&a4), "v"(*(const double*)&a6));)
        } else if (KIND == 6) {  // f16 exp
            REP8(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 7) {  // alternating exp / fma (does the transcendental unit run beside the main VALU?)
            REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %8, %9\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (KIND == 8) {  // exp followed by three plain VALU
            REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_add_f32 %2, %2, %9\n v_fma_f32 %3, %3, %8, %9\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_add_f32 %6, %6, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) ticks[threadIdx.x >> 6] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* ticks) {
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, ticks, 0.5f);
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        const double n = 16.0 * 64.0;
        printf("%-28s %d waves/SIMD: %.2f ticks per instruction per wave (wave 0), %.2f (last wave)\n", name, threads / 256, h[0] / n,
               h[threads / 64 - 1] / n);
    }
}

int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 512 * 4); hipMalloc(&ticks, 64);
    run<0>("v_exp_f32", out, ticks);
    run<6>("v_exp_f16", out, ticks);
    run<1>("v_fma_f32", out, ticks);
    run<2>("v_add_f32", out, ticks);
    run<3>("v_cvt_pk_bf16_f32", out, ticks);
    run<4>("v_max3_f32", out, ticks);
    run<5>("v_pk_fma_f32", out, ticks);
    run<7>("exp,fma alternating", out, ticks);
    run<8>("exp + 3 plain", out, ticks);
    return 0;
}
