// Microbenchmark: what a divergent gather of small records costs beyond L2.
// Every lane reads whole records of R bytes (R/16 loads of 16 bytes) at random positions of a
// table of N records spaced S bytes apart (S >= R), the table far larger than the Infinity Cache,
// eight independent records in flight per lane.  Prints records per second and requested bytes
// per second for each (R, S); under `rocprofv3 --pmc FETCH_SIZE` (or TCC_EA0_RDREQ_sum
// TCC_EA0_RDREQ_32B_sum) the counters per dispatch give the bytes the L2 asked the fabric for,
// i.e. how many bytes one record of each shape really moves:
//   tools/micro/gather_bytes [table GiB] [million records per launch]
// This is what the traversal does with index nodes (64 B), triangle records (48 B) and shading
// records (80 B) once a scene outgrows the caches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ uint32_t xs(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <int R16>     // 16-byte pieces per record
__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ table, const uint32_t nrec, const uint32_t stride16,
                                                const int rounds, float *__restrict__ out) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int r = 0; r < rounds; r++) {
        float4 v[8][R16];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            // records of a wave are unrelated (a multiply-shift keeps the index below nrec without a division)
            const uint32_t idx = (uint32_t)(((unsigned long long)xs(s) * nrec) >> 32);
            const float4 *p = table + (size_t)idx * stride16;
#pragma unroll
            for (int j = 0; j < R16; j++) v[k][j] = p[j];
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int j = 0; j < R16; j++) acc += v[k][j].x + v[k][j].w;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int R16>
static void run(const char *name, const float4 *table, size_t tableBytes, int strideBytes, long long records, float *out) {
    const uint32_t nrec = (uint32_t)(tableBytes / (size_t)strideBytes);
    const int blocks = 256 * 8, threads = blocks * 256;
    const int rounds = (int)(records / ((long long)threads * 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather<R16>, dim3(blocks), dim3(256), 0, 0, table, nrec, (uint32_t)(strideBytes / 16), rounds, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)threads * 8.0 * rounds;
    printf("%-22s record %3d B stride %3d B: %7.2f G records/s, %6.2f TB/s requested, %6.2f TB/s of whole strides (%.1f ms, %.0fM records)\n", name, R16 * 16,
           strideBytes, n / ms / 1e6, n * R16 * 16 / ms / 1e9, n * strideBytes / ms / 1e9, ms, n / 1e6);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const long long records = (long long)((argc > 2 ? atof(argv[2]) : 256.0) * 1e6);
    const size_t bytes = (size_t)(gib * (double)(1ull << 30));
    float4 *table = nullptr; float *out = nullptr;
    if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&out, 256 * 8 * 256 * sizeof(float)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(table, 0, bytes);
    hipDeviceSynchronize();
    printf("table %.1f GiB, %lld M records per launch\n", gib, records / 1000000);
    run<1>("k_gather<1>", table, bytes, 16, records, out);
    run<2>("k_gather<2>", table, bytes, 32, records, out);
    run<3>("k_gather<3> packed", table, bytes, 48, records, out);
    run<3>("k_gather<3> padded", table, bytes, 64, records, out);
    run<4>("k_gather<4>", table, bytes, 64, records, out);
    run<5>("k_gather<5> packed", table, bytes, 80, records, out);
    run<4>("k_gather<4> in 128", table, bytes, 128, records, out);
    run<8>("k_gather<8>", table, bytes, 128, records, out);
    hipFree(table); hipFree(out);
    return 0;
}
