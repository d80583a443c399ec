// Explorer: (X + 0.5) / W for every pixel X < W <= 16384 with a Newton-refined reciprocal of W and one
// residual correction, against IEEE division.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(unsigned long long *bad) {
    const uint32_t W = blockIdx.x + 1;
    const float w = (float)W;
    float r = __builtin_amdgcn_rcpf(w);
    r = __builtin_fmaf(__builtin_fmaf(-w, r, 1.0f), r, r);
    unsigned long long b = 0;
    for (uint32_t x = threadIdx.x; x < W; x += blockDim.x) {
        const float a = (float)x + 0.5f;
        const float q0 = a * r;
        const float q = __builtin_fmaf(__builtin_fmaf(-w, q0, a), r, q0);
        if (__float_as_uint(q) != __float_as_uint(a / w)) b++;
    }
    if (b) atomicAdd(bad, b);
}
int main() {
    unsigned long long *d, h = 0;
    hipMalloc(&d, 8); hipMemcpy(d, &h, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(16384), dim3(256), 0, 0, d);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mismatches %llu\n", h);
    return 0;
}
