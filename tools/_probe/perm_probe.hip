// v_perm_b32 semantics probe: prints __builtin_amdgcn_perm(a, b, sel) for a few selectors (which operand supplies selector values 0-3 / 4-7)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned a = 0xA3A2A1A0u, b = 0xB3B2B1B0u;
    out[0] = __builtin_amdgcn_perm(a, b, 0x03020100u);
    out[1] = __builtin_amdgcn_perm(a, b, 0x07060504u);
    out[2] = __builtin_amdgcn_perm(a, b, 0x0c0c0501u);
    out[3] = __builtin_amdgcn_perm(a, b, 0x0d0c0400u);
}
int main() {
    unsigned* d; unsigned h[4];
    hipMalloc(&d, 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("perm(a=A3A2A1A0, b=B3B2B1B0): sel 03020100 -> %08X, 07060504 -> %08X, 0c0c0501 -> %08X, 0d0c0400 -> %08X\n", h[0], h[1], h[2], h[3]);
    return 0;
}
