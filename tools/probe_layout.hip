// Probe the two hardware layouts attn_core.h relies on and print PASS / FAIL (+ a decoded table on FAIL).
//   1. v_mfma_f32_32x32x16_bf16: A[i=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][n=l&31], C[(r&3)+8*(r>>2)+4*(l>>5)][l&31]
//   2. ds_read_b64_tr_b16: with lane l pointing at elements [4l, 4l+4) of a 16-bit array, lane l gets
//      elements (l&15) + 16*j + 64*(l>>4), j = 0..3
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_layout.hip -o tools/probe_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short i16x4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const float* A, const float* B, float* C) {  // A [32][16], B [16][32] row-major
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[(l & 31) * 16 + 8 * (l >> 5) + j];
        b[j] = (__bf16)B[(8 * (l >> 5) + j) * 32 + (l & 31)];
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k_tr(short* out) {
    __shared__ short lds[256];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    i16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(lds + 4 * l));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = t[j];
}
int main() {
    std::vector<float> A(32 * 16), B(16 * 32), C(32 * 32), R(32 * 32, 0.f);
    srand(1);
    for (auto& x : A) x = (float)(rand() % 7 - 3);
    for (auto& x : B) x = (float)(rand() % 5 - 2);
    for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) R[i * 32 + n] += A[i * 16 + k] * B[k * 32 + n];
    float *dA, *dB, *dC; short* dT;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dT, 256 * 2);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += (C[i] != R[i]);
    printf("PROBE mfma_32x32x16_bf16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    if (bad) for (int i = 0; i < 4; ++i) { for (int n = 0; n < 8; ++n) printf("%6.0f/%-6.0f", C[i * 32 + n], R[i * 32 + n]); printf("\n"); }
    k_tr<<<1, 64>>>(dT);
    std::vector<short> T(256);
    hipMemcpy(T.data(), dT, 512, hipMemcpyDeviceToHost);
    bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) bad += (T[l * 4 + j] != (l & 15) + 16 * j + 64 * (l >> 4));
    printf("PROBE ds_read_b64_tr_b16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    if (bad) for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, T[l * 4], T[l * 4 + 1], T[l * 4 + 2], T[l * 4 + 3]);
    hipError_t e = hipDeviceSynchronize();
    printf("PROBE done (%s)\n", hipGetErrorString(e));
    return 0;
}
