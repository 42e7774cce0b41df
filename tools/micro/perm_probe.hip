#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out) {
    unsigned s0 = 0xA3A2A1A0u, s1 = 0xB3B2B1B0u;
    unsigned sels[8] = {0x03020100u, 0x07060504u, 0x0C0C0400u, 0x0C0C0500u, 0x0C0C0600u, 0x0C0C0700u, 0x0D0C0100u, 0x00010203u};
    for (int i = 0; i < 8; ++i) out[i] = __builtin_amdgcn_perm(s0, s1, sels[i]);
}
int main() {
    unsigned *d, h[8];
    hipMalloc(&d, 32);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d);
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    const char *n[8] = {"03020100", "07060504", "0C0C0400", "0C0C0500", "0C0C0600", "0C0C0700", "0D0C0100", "00010203"};
    for (int i = 0; i < 8; ++i) printf("perm(s0=A3A2A1A0, s1=B3B2B1B0, sel=%s) = %08X\n", n[i], h[i]);
    return 0;
}
