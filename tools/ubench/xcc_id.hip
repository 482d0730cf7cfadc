#include <hip/hip_runtime.h>
__global__ void k(unsigned *out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    out[blockIdx.x] = x;
}
int main() {
    unsigned *d; hipMalloc(&d, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d);
    unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i++) printf("%u:%x ", i, h[i]);
    printf("\n");
    return 0;
}
