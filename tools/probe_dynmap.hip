// probe_dynmap — isolates the two building blocks of dynmap16_kernel (csrc/dynmap.hip): the in-wave bitonic sort against std::sort, and the operand / result
// layout of v_mfma_f64_16x16x4_f64 against a host matrix product; then both forms of svg_identify_dynamic_map (two library builds) row by row.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include tools/probe_dynmap.hip -o tools/probe_dynmap -ldl
//   tools/probe_dynmap [libA.so libB.so]
#include "../sparse-videogen_amd/csrc/dynmap.hip"
namespace svg { thread_local int g_last_hip_error = 0; }
#include <dlfcn.h>
#include <cstring>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <random>

__global__ void sort_test(unsigned* keys) {
    const int lane = threadIdx.x;
    unsigned k[16];
    for (int e = 0; e < 16; ++e) k[e] = keys[lane * 16 + e];
    svg::wave_bitonic_sort<16>(k, lane);
    for (int e = 0; e < 16; ++e) keys[lane * 16 + e] = k[e];
}
__global__ void mfma_test(const double* A /*16x4*/, const double* B /*4x16*/, double* Dm /*16x16*/) {
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    svg::f64x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[n * 4 + g], B[g * 16 + n], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) Dm[(4 * i + g) * 16 + n] = acc[i];
}
typedef int (*dyn_fn)(const void*, const void*, const int32_t*, uint8_t*, int32_t, int32_t, int32_t, int32_t, int32_t, float, int32_t, void*);
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    std::mt19937 rng(5);
    {   // sort
        std::vector<unsigned> h(1024), ref;
        for (auto& x : h) x = rng();
        for (int i = 1000; i < 1024; ++i) h[i] = 0xffffffffu;
        std::shuffle(h.begin(), h.end(), rng);
        ref = h;
        std::sort(ref.begin(), ref.end());
        unsigned* d;
        hipMalloc(&d, 4096), hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
        sort_test<<<1, 64>>>(d);
        hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; ++i) bad += h[i] != ref[i];
        printf("in-wave bitonic sort of 1024 keys: %d positions differ from std::sort\n", bad);
    }
    {   // MFMA layout
        std::vector<double> A(64), B(64), Dm(256), ref(256, 0.0);
        for (auto& x : A) x = (double)(int)(rng() % 17) - 8;
        for (auto& x : B) x = (double)(int)(rng() % 13) - 6;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) ref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
        double *dA, *dB, *dD;
        hipMalloc(&dA, 512), hipMalloc(&dB, 512), hipMalloc(&dD, 2048);
        hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
        mfma_test<<<1, 64>>>(dA, dB, dD);
        hipMemcpy(Dm.data(), dD, 2048, hipMemcpyDeviceToHost);
        int bad = 0, badT = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { bad += Dm[i * 16 + j] != ref[i * 16 + j]; badT += Dm[j * 16 + i] != ref[i * 16 + j]; }
        printf("v_mfma_f64_16x16x4_f64 with A[n][g], B[g][n], D[4i+g][n]: %d of 256 differ (as the transpose: %d)\n", bad, badT);
    }
    if (argc >= 3) {
        const int BH = 3, QC = 50, KC = 1000, D = 128;
        std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<uint16_t> q((size_t)BH * QC * D), k((size_t)BH * KC * D);
        std::vector<int32_t> ks((size_t)BH * KC);
        for (auto& x : q) x = f2bf(nd(rng));
        for (auto& x : k) x = f2bf(nd(rng));
        for (auto& x : ks) x = 1 + rng() % 200;
        void *dq, *dk; int32_t* dks; uint8_t* dm;
        hipMalloc(&dq, q.size() * 2), hipMalloc(&dk, k.size() * 2), hipMalloc(&dks, ks.size() * 4), hipMalloc(&dm, (size_t)BH * QC * KC);
        hipMemcpy(dq, q.data(), q.size() * 2, hipMemcpyHostToDevice), hipMemcpy(dk, k.data(), k.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dks, ks.data(), ks.size() * 4, hipMemcpyHostToDevice);
        std::vector<uint8_t> m[2];
        for (int v = 0; v < 2; ++v) {
            void* so = dlopen(argv[1 + v], RTLD_NOW | RTLD_LOCAL);
            if (!so) { printf("dlopen %s failed\n", argv[1 + v]); return 2; }
            auto f = (dyn_fn)dlsym(so, "svg_identify_dynamic_map");
            hipMemset(dm, 7, (size_t)BH * QC * KC);
            const int rc = f(dq, dk, dks, dm, BH, QC, KC, D, SVG_DTYPE_BF16, 0.9f, 100, nullptr);
            hipDeviceSynchronize();
            m[v].resize((size_t)BH * QC * KC);
            hipMemcpy(m[v].data(), dm, m[v].size(), hipMemcpyDeviceToHost);
            long ones = 0; for (auto x : m[v]) ones += x == 1;
            printf("%s rc=%d ones=%ld of %zu (bytes not 0/1: %ld)\n", argv[1 + v], rc, ones, m[v].size(), (long)std::count_if(m[v].begin(), m[v].end(), [](uint8_t x) { return x > 1; }));
        }
        int rows_bad = 0;
        for (int r = 0; r < BH * QC; ++r) {
            int diff = 0, a1 = 0, b1 = 0;
            for (int j = 0; j < KC; ++j) { diff += m[0][(size_t)r * KC + j] != m[1][(size_t)r * KC + j]; a1 += m[0][(size_t)r * KC + j]; b1 += m[1][(size_t)r * KC + j]; }
            if (diff && rows_bad++ < 8) printf("row %d: %d entries differ (ones %d vs %d)\n", r, diff, a1, b1);
        }
        printf("%d of %d rows differ\n", rows_bad, BH * QC);
    }
    return 0;
}
