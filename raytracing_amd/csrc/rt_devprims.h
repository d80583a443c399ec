// Two device-wide primitives of the index build on the GPU (rt_gpubuild.hip), written for wave64:
//   exclusive_sum<T>    out[i] = in[0] + ... + in[i-1]           (T = uint32_t or uint64_t)
//   radix_sort_pairs    stable least-significant-digit radix sort of 64-bit keys with 32-bit values
// Both work on tiles of TILE elements per 256-thread block and need scratch memory the caller
// provides (scan_scratch_bytes / sort_scratch_bytes).  Plain kernels on the given stream, no host
// synchronisation.  The reference has no counterpart: it builds its one tree on the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtp {

constexpr int PBLOCK = 256;
constexpr int ITEMS = 8;
constexpr int TILE = PBLOCK * ITEMS;        // elements per block

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Inclusive scan over the 64 lanes of a wave.
template <typename T>
__device__ __forceinline__ T wave_inclusive(T v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T u = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += u;
    }
    return v;
}

// Exclusive scan of one value per thread over the block; *total receives the block's sum.
// `lds` has PBLOCK / 64 + 1 elements.  Ends with a barrier, so `lds` may be reused at once.
template <typename T>
__device__ __forceinline__ T block_exclusive(T v, T *lds, T *total) {
    const T inc = wave_inclusive(v);
    const uint32_t w = threadIdx.x >> 6;
    if (lane_id() == 63u) lds[w] = inc;
    __syncthreads();
    T base = 0, sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PBLOCK / 64; k++) {
        const T c = lds[k];
        if (k < w) base += c;
        sum += c;
    }
    __syncthreads();
    *total = sum;
    return base + inc - v;
}

// ---- exclusive sum

template <typename T>
__global__ __launch_bounds__(PBLOCK) void k_scan_reduce(const T *__restrict__ in, const uint32_t n, T *__restrict__ partial) {
    __shared__ T lds[PBLOCK / 64 + 1];
    const size_t first = (size_t)blockIdx.x * TILE + (size_t)threadIdx.x * ITEMS;
    T s = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++)
        if (first + k < n) s += in[first + k];
    T total;
    (void)block_exclusive(s, lds, &total);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// One block turns the tile sums into their exclusive prefix, in place, chunk after chunk.
template <typename T>
__global__ __launch_bounds__(PBLOCK) void k_scan_partials(T *__restrict__ partial, const uint32_t nb) {
    __shared__ T lds[PBLOCK / 64 + 1];
    T carry = 0;
    for (uint32_t first = 0; first < nb; first += PBLOCK) {
        const uint32_t i = first + threadIdx.x;
        const T v = (i < nb) ? partial[i] : (T)0;
        T total;
        const T ex = block_exclusive(v, lds, &total);
        if (i < nb) partial[i] = carry + ex;
        carry += total;
    }
}

template <typename T>
__global__ __launch_bounds__(PBLOCK) void k_scan_apply(const T *in, const uint32_t n, const T *__restrict__ partial, T *out) {   // in may be out: a thread writes only where it has read
    __shared__ T lds[PBLOCK / 64 + 1];
    const size_t first = (size_t)blockIdx.x * TILE + (size_t)threadIdx.x * ITEMS;
    T v[ITEMS];
    T s = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) { v[k] = (first + k < n) ? in[first + k] : (T)0; s += v[k]; }
    T total;
    T run = partial[blockIdx.x] + block_exclusive(s, lds, &total);
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        if (first + k < n) out[first + k] = run;
        run += v[k];
    }
}

inline uint32_t tiles_of(uint32_t n) { return (n + TILE - 1) / TILE; }
template <typename T>
inline size_t scan_scratch_bytes(uint32_t n) { return (size_t)(tiles_of(n) + 1) * sizeof(T); }

// out[0..n) = exclusive prefix sums of in[0..n); `in` and `out` may be the same array.
template <typename T>
inline hipError_t exclusive_sum(const T *in, T *out, uint32_t n, void *scratch, hipStream_t st) {
    if (n == 0) return hipSuccess;
    T *partial = (T *)scratch;
    const uint32_t nb = tiles_of(n);
    hipLaunchKernelGGL(k_scan_reduce<T>, dim3(nb), dim3(PBLOCK), 0, st, in, n, partial);
    hipLaunchKernelGGL(k_scan_partials<T>, dim3(1), dim3(PBLOCK), 0, st, partial, nb);
    hipLaunchKernelGGL(k_scan_apply<T>, dim3(nb), dim3(PBLOCK), 0, st, in, n, (const T *)partial, out);
    return hipGetLastError();
}

// ---- radix sort, 8 bits per pass

// hist[d * nb + b]: how many keys of tile b have digit d.
__global__ __launch_bounds__(PBLOCK) void k_radix_hist(const uint64_t *__restrict__ keys, const uint32_t n, const int shift,
                                                       const uint32_t nb, uint32_t *__restrict__ hist) {
    __shared__ uint32_t cnt[256];
    cnt[threadIdx.x] = 0u;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * TILE;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const size_t i = base + (size_t)r * PBLOCK + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nb + blockIdx.x] = cnt[threadIdx.x];
}

// Moves the keys of tile b to offs[d * nb + b] + (rank among the tile's keys with digit d), in
// index order: rounds of 256 consecutive keys, within a round the rank comes from the lanes of
// lower number with the same digit (eight ballots) and the waves of lower number (LDS counts).
__global__ __launch_bounds__(PBLOCK) void k_radix_scatter(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, const uint32_t n,
                                                          const int shift, const uint32_t nb, const uint32_t *__restrict__ offs,
                                                          uint64_t *__restrict__ keysOut, uint32_t *__restrict__ valsOut) {
    __shared__ uint32_t next[256];                  // where the next key with this digit goes
    __shared__ uint32_t waveCnt[PBLOCK / 64][256];  // this round: keys with this digit per wave
    next[threadIdx.x] = offs[(size_t)threadIdx.x * nb + blockIdx.x];
    const uint32_t w = threadIdx.x >> 6, lane = lane_id();
    const unsigned long long below = (1ull << lane) - 1ull;
    const size_t base = (size_t)blockIdx.x * TILE;
    for (int r = 0; r < ITEMS; r++) {
#pragma unroll
        for (uint32_t k = 0; k < PBLOCK / 64; k++) waveCnt[k][threadIdx.x] = 0u;
        __syncthreads();
        const size_t i = base + (size_t)r * PBLOCK + threadIdx.x;
        const bool valid = i < n;
        const uint64_t key = valid ? keys[i] : 0ull;
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const bool set = (d >> bit) & 1u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(set);
            peers &= set ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & below);
        if (valid && rank == 0u) waveCnt[w][d] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t pos = next[d] + rank;
            for (uint32_t k = 0; k < w; k++) pos += waveCnt[k][d];
            keysOut[pos] = key;
            valsOut[pos] = vals[i];
        }
        __syncthreads();
        uint32_t add = 0u;
#pragma unroll
        for (uint32_t k = 0; k < PBLOCK / 64; k++) add += waveCnt[k][threadIdx.x];
        next[threadIdx.x] += add;
        // the zeroing at the top of the next round is separated from these reads by this barrier
        __syncthreads();
    }
}

inline size_t sort_scratch_bytes(uint32_t n) {
    const size_t histWords = (size_t)256 * tiles_of(n);
    return histWords * sizeof(uint32_t) + scan_scratch_bytes<uint32_t>((uint32_t)histWords) + 256;
}

// Sorts (keys, vals) by key, stable.  keysAlt / valsAlt are buffers of the same size; after the
// eight passes the sorted pairs are back in keys / vals.  n * 256 / TILE must fit 32 bits (n < 2^29).
inline hipError_t radix_sort_pairs(uint64_t *keys, uint64_t *keysAlt, uint32_t *vals, uint32_t *valsAlt, uint32_t n, void *scratch, hipStream_t st) {
    if (n <= 1) return hipSuccess;
    const uint32_t nb = tiles_of(n);
    const uint32_t histWords = 256u * nb;
    uint32_t *hist = (uint32_t *)scratch;
    void *scanScratch = (void *)(((uintptr_t)(hist + histWords) + 255u) & ~(uintptr_t)255u);
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 8 * pass;
        hipLaunchKernelGGL(k_radix_hist, dim3(nb), dim3(PBLOCK), 0, st, (const uint64_t *)keys, n, shift, nb, hist);
        const hipError_t e = exclusive_sum<uint32_t>(hist, hist, histWords, scanScratch, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_radix_scatter, dim3(nb), dim3(PBLOCK), 0, st, (const uint64_t *)keys, (const uint32_t *)vals, n, shift, nb,
                           (const uint32_t *)hist, keysAlt, valsAlt);
        uint64_t *tk = keys; keys = keysAlt; keysAlt = tk;
        uint32_t *tv = vals; vals = valsAlt; valsAlt = tv;
    }
    return hipGetLastError();
}

}  // namespace rtp
