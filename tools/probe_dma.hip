// Probe the LDS-DMA semantics attn_core.h (ping-pong body) relies on and print PASS / FAIL.
//   global_load_lds_dwordx4 voff, s[base:base+1]  with M0 = LDS byte address (wave-uniform):
//     lane l's 16 bytes from (base + voff_l) land at LDS[M0 + 16*l]   — also for M0 >= 64 KiB
//   several DMAs in flight are retired in order and counted by vmcnt.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_dma.hip -o tools/probe_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_dma(const unsigned* __restrict__ src, unsigned* __restrict__ out, unsigned lds_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // every lane picks a scattered source chunk: chunk index = (lane * 7 + wave * 3) % 256
    const unsigned chunk = (unsigned)((lane * 7 + wave * 3) & 255);
    const unsigned voff = chunk * 16u;
    const unsigned ldsb = (unsigned)(size_t)smem + lds_off + (unsigned)wave * 2048u;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsb), "v"(voff), "s"(src) : "memory");
    const unsigned voff2 = ((chunk + 1) & 255) * 16u;
    const unsigned ldsb2 = ldsb + 1024u;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsb2), "v"(voff2), "s"(src) : "memory");
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    const char* p = smem + lds_off + wave * 2048;
    for (int j = 0; j < 2; ++j) {
        const uint4 x = *(const uint4*)(p + j * 1024 + lane * 16);
        unsigned* o = out + ((wave * 2 + j) * 64 + lane) * 4;
        o[0] = x.x, o[1] = x.y, o[2] = x.z, o[3] = x.w;
    }
}

int main() {
    std::vector<unsigned> h(256 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
    unsigned *d, *o;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&o, 4 * 2 * 64 * 4 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    bool all = true;
    for (unsigned off : {0u, 32768u, 98304u, 147456u}) {
        hipFuncSetAttribute((const void*)k_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipMemset(o, 0, 4 * 2 * 64 * 4 * 4);
        hipLaunchKernelGGL(k_dma, dim3(1), dim3(256), off + 8192, 0, d, o, off);
        hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned> r(4 * 2 * 64 * 4);
        hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < 4; ++w)
            for (int j = 0; j < 2; ++j)
                for (int l = 0; l < 64; ++l) {
                    const unsigned chunk = ((l * 7 + w * 3) + j) & 255;
                    for (int q = 0; q < 4; ++q) bad += r[((w * 2 + j) * 64 + l) * 4 + q] != h[chunk * 4 + q];
                }
        printf("lds offset %6u: %s (%d mismatches, hip=%d)\n", off, bad ? "FAIL" : "PASS", bad, (int)e);
        all &= (bad == 0);
    }
    printf("probe_dma: %s\n", all ? "PASS" : "FAIL");
    return all ? 0 : 1;
}
