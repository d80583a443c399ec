// Microbenchmark: cost of wave64 vector instructions as a function of the EXEC mask.
// Times fma chains with n contiguous lanes on, n scattered lanes on, for one dependent chain
// and for four independent chains per lane.
// Measured on MI355X: the time is the same from 64 down to 17 active lanes, contiguous or
// scattered, so whole 16-lane groups that are switched off are not skipped: compacting the
// active lanes of a wave buys nothing.  (With very few active lanes the whole chip drops to
// about a quarter of its clock, which is power management, not instruction cost.)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ bool lane_on(int lane, int n, int scattered) {
    if (!scattered) return lane < n;
    // bit-reversed lane order: n lanes spread over the whole wave
    int r = ((lane & 1) << 5) | ((lane & 2) << 3) | ((lane & 4) << 1) | ((lane & 8) >> 1) | ((lane & 16) >> 3) | ((lane & 32) >> 5);
    return r < n;
}
__global__ void k1(float *out, int n, int scattered, int iters) {
    const int lane = threadIdx.x & 63;
    float a = (float)threadIdx.x * 1e-3f; const float b = 1.0001f, c = 0.5f;
    if (lane_on(lane, n, scattered)) {
        for (int i = 0; i < iters; i++) {
            a = __builtin_fmaf(a, b, c); a = __builtin_fmaf(a, b, c); a = __builtin_fmaf(a, b, c); a = __builtin_fmaf(a, b, c);
            a = __builtin_fmaf(a, b, c); a = __builtin_fmaf(a, b, c); a = __builtin_fmaf(a, b, c); a = __builtin_fmaf(a, b, c);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k4(float *out, int n, int scattered, int iters) {
    const int lane = threadIdx.x & 63;
    float a0 = (float)threadIdx.x * 1e-3f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f; const float b = 1.0001f, c = 0.5f;
    if (lane_on(lane, n, scattered)) {
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ void k8(float *out, int n, int scattered, int iters) {
    const int lane = threadIdx.x & 63;
    float a[8]; const float b = 1.0001f, c = 0.5f;
    for (int j = 0; j < 8; j++) a[j] = (float)threadIdx.x * 1e-3f + (float)j;
    if (lane_on(lane, n, scattered)) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = __builtin_fmaf(a[j], b, c);
        }
    }
    float r = 0.0f;
    for (int j = 0; j < 8; j++) r += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main(int argc, char **argv) {
    const int blocksPerCU = argc > 1 ? atoi(argv[1]) : 8;
    float *d; if (hipMalloc(&d, 256 * 2048 * 8 * sizeof(float)) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ns[] = {64, 63, 48, 33, 32, 31, 24, 17, 16, 15, 8, 1};
    printf("blocks per CU %d (waves per SIMD %d)\n", blocksPerCU, blocksPerCU);
    for (int kern = 0; kern < 3; kern++)
        for (int scattered = 0; scattered < 2; scattered++)
            for (int n : ns) {
                auto launch = [&](int iters) {
                    if (kern == 0) k1<<<256 * blocksPerCU, 256>>>(d, n, scattered, iters);
                    else if (kern == 1) k4<<<256 * blocksPerCU, 256>>>(d, n, scattered, iters);
                    else k8<<<256 * blocksPerCU, 256>>>(d, n, scattered, iters);
                };
                launch(100);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                launch(20000);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("%s %s n=%2d  %.3f ms\n", kern == 2 ? "8 chains" : (kern ? "4 chains" : "1 chain "), scattered ? "scattered " : "contiguous", n, ms);
            }
    return 0;
}
