// Issue rate of v_exp_f32 against v_fma_f32 / v_add_f32 / v_cvt_pk_bf16_f32 on gfx950 (one wave per SIMD, independent registers):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_exp tools/probe_exp.hip && /tmp/probe_exp
// prints ns per wave instruction and SIMD with 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = -1.0f - 0.01f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
            if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
            if (OP == 2) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[i]));
            if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(r[i]));
            if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double*)&r[i & ~1]));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int OP>
double run(float* out, int iters, int threads) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    k<OP><<<256, threads>>>(out, 10);
    hipEventRecord(a);
    k<OP><<<256, threads>>>(out, iters);   // 256 workgroups x threads / 64 waves: threads / 256 waves per SIMD
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e6 / ((double)iters * 16 * (threads / 256));   // ns per wave instruction and SIMD
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 100000;
    for (int threads = 256; threads <= 1024; threads *= 2) {
        const double e = run<0>(out, iters, threads), f = run<1>(out, iters, threads), ad = run<2>(out, iters, threads),
                     c = run<3>(out, iters, threads), pk = run<4>(out, iters, threads);
        printf("%d wave(s) per SIMD, ns per wave instruction and SIMD: v_exp_f32 %.2f  v_fma_f32 %.2f  v_add_f32 %.2f  v_cvt_pk_bf16_f32 %.2f  "
               "v_pk_mul_f32 %.2f   (exp / fma %.2f)\n", threads / 256, e, f, ad, c, pk, e / f);
    }
    return 0;
}
