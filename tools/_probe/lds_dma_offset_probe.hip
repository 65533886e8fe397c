// Does the instruction offset of global_load_lds_dwordx4 move the LDS destination too (LDS = M0 + inst_offset + lane * 16), or only the
// global address?  One wave copies a 1 KiB piece with offset:1024 from src (global byte 1024 + lane * 16 holds its own index) and reports
// where in LDS it landed.   hipcc --offload-arch=gfx950 -O2 lds_dma_offset_probe.hip -o lds_dma_offset_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const uint32_t* src, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds;
    uint32_t voff = threadIdx.x * 16, keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds0 + 2048) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
    uint32_t *src, *out, h[4096], hs[2048];
    for (int i = 0; i < 2048; ++i) hs[i] = i;
    hipMalloc(&src, sizeof(hs)); hipMalloc(&out, sizeof(h));
    hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(src, out);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int first = -1, n = 0;
    for (int i = 0; i < 4096; ++i) if (h[i] != 0xdeadbeefu) { if (first < 0) first = i; ++n; }
    printf("landed: first dword %d (byte %d), %d dwords, first value %u (M0 = byte 2048, inst offset 1024, source dword 256 expected)\n", first, first * 4, n, first >= 0 ? h[first] : 0);
    printf(first * 4 == 2048 ? "=> inst_offset moves the GLOBAL address only\n" : (first * 4 == 3072 ? "=> inst_offset moves BOTH addresses\n" : "=> unexpected\n"));
    return 0;
}
