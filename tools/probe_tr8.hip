// Probe ds_read_b64_tr_b8 (gfx950) and the k-slot consistency of v_mfma_scale_f32_32x32x64_f8f6f4, the two hardware facts the fp8
// variable-block body relies on; prints PASS / FAIL.
//   1. ds_read_b64_tr_b8: with lane l pointing at bytes [8l, 8l+8) of a byte array, lane l gets bytes (l&15) + 16 j + 128 (l>>4),
//      j = 0..7 — the 16 lanes of a group pool their 16 x 8 bytes as an [8 rows][16 columns] byte matrix (row r = lanes 2r, 2r+1)
//      and lane i receives column i.
//   2. 32x32x64 f8f6f4 MFMA with e4m3 operands: C[i][n] = sum_k A[i][k] B[k][n] when lane l supplies A row (l&31) / B column (l&31)
//      and the SAME 32 k values in the same byte order for both operands, 32 per lane half — which k is which slot does not matter.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_tr8.hip -o /tmp/probe_tr8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_tr(unsigned char* out) {
    __shared__ unsigned char lds[512];
    const int l = threadIdx.x;
    for (int i = l; i < 512; i += 64) lds[i] = (unsigned char)i;
    __syncthreads();
    v2i t = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lds + 8 * l));
    unsigned char b[8];
    memcpy(b, &t, 8);
    for (int j = 0; j < 8; ++j) out[l * 8 + j] = b[j];
}

// A [32][64], B [64][32] as float (values exactly representable in e4m3); slot s of lane half h holds k = perm[32 h + s]
__global__ void k_mfma(const float* A, const float* B, const int* perm, float* C) {
    const int l = threadIdx.x, h = l >> 5;
    v8i a, b;
    for (int w = 0; w < 8; ++w) {
        int wa = 0, wb = 0;
        float fa[4], fb[4];
        for (int i = 0; i < 4; ++i) {
            const int k = perm[32 * h + 4 * w + i];
            fa[i] = A[(l & 31) * 64 + k];
            fb[i] = B[k * 32 + (l & 31)];
        }
        wa = __builtin_amdgcn_cvt_pk_fp8_f32(fa[0], fa[1], wa, false);
        wa = __builtin_amdgcn_cvt_pk_fp8_f32(fa[2], fa[3], wa, true);
        wb = __builtin_amdgcn_cvt_pk_fp8_f32(fb[0], fb[1], wb, false);
        wb = __builtin_amdgcn_cvt_pk_fp8_f32(fb[2], fb[3], wb, true);
        a[w] = wa, b[w] = wb;
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + (l & 31)] = c[r];
}

int main() {
    int fails = 0;
    {
        unsigned char* d;
        hipMalloc(&d, 512);
        k_tr<<<1, 64>>>(d);
        std::vector<unsigned char> h(512);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) bad += h[l * 8 + j] != (unsigned char)((l & 15) + 16 * j + 128 * (l >> 4));
        printf("ds_read_b64_tr_b8: %s\n", bad ? "FAIL" : "PASS");
        if (bad) {
            for (int l = 0; l < 64; ++l) {
                printf("lane %2d:", l);
                for (int j = 0; j < 8; ++j) printf(" %3d", h[l * 8 + j]);
                printf("\n");
            }
        }
        fails += bad != 0;
    }
    for (int trial = 0; trial < 2; ++trial) {
        std::vector<float> A(32 * 64), B(64 * 32), C(32 * 32), R(32 * 32, 0.f);
        std::vector<int> perm(64);
        srand(7 + trial);
        const float vals[8] = {0.f, 0.5f, 1.f, -1.f, 2.f, -0.25f, 1.5f, -3.f};
        for (auto& x : A) x = vals[rand() % 8];
        for (auto& x : B) x = vals[rand() % 8];
        for (int i = 0; i < 64; ++i) perm[i] = i;
        if (trial == 1)
            for (int i = 63; i > 0; --i) std::swap(perm[i], perm[rand() % (i + 1)]);   // any bijection of the 64 k values onto the slots
        for (int i = 0; i < 32; ++i)
            for (int n = 0; n < 32; ++n)
                for (int k = 0; k < 64; ++k) R[i * 32 + n] += A[i * 64 + k] * B[k * 32 + n];
        float *dA, *dB, *dC;
        int* dP;
        hipMalloc(&dA, A.size() * 4), hipMalloc(&dB, B.size() * 4), hipMalloc(&dC, C.size() * 4), hipMalloc(&dP, 256);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dP, perm.data(), 256, hipMemcpyHostToDevice);
        k_mfma<<<1, 64>>>(dA, dB, dP, dC);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; ++i) bad += C[i] != R[i];
        printf("mfma_scale_f32_32x32x64_f8f6f4 e4m3 (%s slot order): %s\n", trial ? "shuffled" : "natural", bad ? "FAIL" : "PASS");
        fails += bad != 0;
    }
    return fails;
}
