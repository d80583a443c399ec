// Explorer: for every fp32 bit pattern, compares short reciprocal / square-root sequences built on
// v_rcp_f32 / v_rsq_f32 with the compiler's IEEE-754 division and sqrt.  Prints mismatch counts and
// the range of mismatching inputs per candidate.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o exact_math tools/micro/exact_math.hip && ./exact_math
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ bool same(float a, float b) {
    return (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
}

__device__ float recip1(float x) { float r = __builtin_amdgcn_rcpf(x); float e = fma_(-x, r, 1.0f); return fma_(e, r, r); }
__device__ float recip2(float x) { float r = recip1(x); float e = fma_(-x, r, 1.0f); return fma_(e, r, r); }
__device__ float sqrt1(float x) {
    float y = __builtin_amdgcn_rsqf(x); float s = x * y; float h = 0.5f * y;
    float d = fma_(-s, s, x); return fma_(d, h, s);
}
__device__ float sqrt2(float x) {
    float y = __builtin_amdgcn_rsqf(x); float s = x * y; float h = 0.5f * y;
    float d = fma_(-s, s, x); s = fma_(d, h, s);
    d = fma_(-s, s, x); return fma_(d, h, s);
}

struct Res { unsigned long long bad; uint32_t lo, hi; unsigned int hist[256]; };

template <int WHICH>
__global__ void k(Res *res) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long bad = 0; uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (unsigned long long b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        float want, got;
        if (WHICH == 0) { want = 1.0f / x; got = recip1(x); }
        else if (WHICH == 1) { want = 1.0f / x; got = recip2(x); }
        else if (WHICH == 2) { want = __builtin_sqrtf(x); got = sqrt1(x); }
        else if (WHICH == 3) { want = __builtin_sqrtf(x); got = sqrt2(x); }
        else if (WHICH == 4) { want = 1.0f / __builtin_sqrtf(x); got = recip1(sqrt1(x)); }
        else { want = 1.0f / __builtin_sqrtf(x); got = recip2(sqrt2(x)); }
        if (!same(want, got)) {
            const uint32_t m = (uint32_t)b & 0x7FFFFFFFu;
            bad++; if (m < lo) lo = m; if (m > hi) hi = m;
            atomicAdd(&res->hist[m >> 23], 1u);
        }
    }
    if (bad) { atomicAdd(&res->bad, bad); atomicMin(&res->lo, lo); atomicMax(&res->hi, hi); }
}

template <int WHICH> void run(const char *name) {
    Res *d; Res h; memset(&h, 0, sizeof h); h.lo = 0xFFFFFFFFu;
    hipMalloc(&d, sizeof h); hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<WHICH>, dim3(4096), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    float flo, fhi; memcpy(&flo, &h.lo, 4); memcpy(&fhi, &h.hi, 4);
    printf("%-28s mismatches %llu  |x| in [%08x %g, %08x %g]\n", name, h.bad, h.lo, flo, h.hi, fhi);
    printf("    biased exponents with mismatches:");
    for (int e = 0; e < 256; e++) if (h.hist[e]) printf(" %d:%u", e, h.hist[e]);
    printf("\n");
    hipFree(d);
}

int main() {
    run<0>("recip: rcp + 1 step");
    run<1>("recip: rcp + 2 steps");
    run<2>("sqrt: rsq + 1 step");
    run<3>("sqrt: rsq + 2 steps");
    run<4>("1/sqrt: sqrt1 then recip1");
    run<5>("1/sqrt: sqrt2 then recip2");
    return 0;
}
