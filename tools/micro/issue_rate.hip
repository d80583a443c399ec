// Microbenchmark: what the instructions of the traversal loop cost to issue on gfx950 when a SIMD
// is shared by 8 waves (the path kernel's occupancy).  Every kernel runs a loop of 64 copies of one
// instruction (chains over four registers, so a wave's own dependencies are covered by the other
// waves); the time per instruction is printed relative to v_fma_f32.
//   hipcc -O3 --offload-arch=gfx950 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP4(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP16(...) REP4(__VA_ARGS__) REP4(__VA_ARGS__) REP4(__VA_ARGS__) REP4(__VA_ARGS__)

#define KERNEL(name, ...)                                                                           \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void name(float *out, const int rounds, const float seed) { \
        float a = seed + threadIdx.x, b = seed * 2.0f, c = seed * 3.0f, d = seed * 0.5f;             \
        unsigned int u = threadIdx.x * 2654435761u, w = threadIdx.x + 12345u;                        \
        unsigned long long m = 0, m2 = 0;                                                            \
        for (int r = 0; r < rounds; r++) { REP16(__VA_ARGS__) }                                             \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (float)u + (float)w + (float)(m ^ m2); \
    }

// four instructions per body (so 64 per round)
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %2, %1\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %0, %3\n v_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_add, asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %0\n v_add_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_mul_sgpr, asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed));)
KERNEL(k_cvt_ubyte, asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(u));)
KERNEL(k_max3, asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_min, asm volatile("v_min_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_min_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_cndmask_vcc, asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");)
KERNEL(k_cndmask_sgpr, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %4\n v_cndmask_b32_e64 %1, %1, %2, %4\n v_cndmask_b32_e64 %2, %2, %3, %4\n v_cndmask_b32_e64 %3, %3, %0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(m));)
KERNEL(k_cmp_vcc, asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %0" : : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");)
KERNEL(k_cmp_sgpr, asm volatile("v_cmp_gt_f32_e64 %0, %2, %3\n v_cmp_gt_f32_e64 %1, %3, %4\n v_cmp_gt_f32_e64 %0, %4, %5\n v_cmp_gt_f32_e64 %1, %5, %2" : "=s"(m), "=s"(m2) : "v"(a), "v"(b), "v"(c), "v"(d));)
KERNEL(k_cmp_cnd, asm volatile("v_cmp_gt_f32_e64 %4, %0, %1\n v_cndmask_b32_e64 %0, %0, %2, %4\n v_cmp_gt_f32_e64 %5, %1, %3\n v_cndmask_b32_e64 %1, %1, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=s"(m), "=s"(m2));)
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %0, %2\n v_perm_b32 %0, %0, %1, %3\n v_perm_b32 %1, %1, %0, %3" : "+v"(u), "+v"(w) : "v"(0x05010400u), "v"(0x07030602u));)
KERNEL(k_fma_mix, asm volatile("v_fma_mix_f32 %0, %4, %2, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %4, %3, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(u));)
KERNEL(k_int_alu, asm volatile("v_lshlrev_b32 %0, 3, %0\n v_and_b32 %1, %1, %0\n v_bfe_u32 %0, %0, 8, 8\n v_lshl_add_u32 %1, %1, 2, %0" : "+v"(u), "+v"(w));)
KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0" : "+v"(u), "+v"(w));)
KERNEL(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %0\n v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %0" : "+v"(u), "+v"(w));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_salu, asm volatile("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %0\n s_andn2_b64 %0, %0, %1\n s_xor_b64 %1, %1, %0" : "+s"(m), "+s"(m2) : : "scc");)
KERNEL(k_readlane, { unsigned int s0, s1; asm volatile("v_readfirstlane_b32 %2, %0\n v_readfirstlane_b32 %3, %1\n v_readfirstlane_b32 %2, %1\n v_readfirstlane_b32 %3, %0" : "+v"(u), "+v"(w), "=s"(s0), "=s"(s1)); m += s0 + s1; })
KERNEL(k_valu_salu, asm volatile("v_fma_f32 %0, %0, %2, %1\n s_and_b64 %4, %4, %5\n v_fma_f32 %1, %1, %2, %3\n s_or_b64 %5, %5, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(m), "+s"(m2) : : "scc");)
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %0\n v_mul_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_sub, asm volatile("v_sub_f32 %0, %0, %2\n v_sub_f32 %1, %1, %2\n v_sub_f32 %2, %2, %0\n v_sub_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_fmac, asm volatile("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_fma_sgpr, asm volatile("v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %4, %3\n v_fma_f32 %2, %2, %4, %3\n v_fma_f32 %3, %3, %4, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed));)
KERNEL(k_fma_const, asm volatile("v_fma_f32 %0, %0, 0.5, %1\n v_fma_f32 %1, %1, 0.5, %3\n v_fma_f32 %2, %2, 0.5, %3\n v_fma_f32 %3, %3, 0.5, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_add_const, asm volatile("v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_pk_fma, { typedef float f2 __attribute__((ext_vector_type(2))); f2 x = {a, b}, y = {c, d}; asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0" : "+v"(x), "+v"(y)); a = x.x; b = x.y; c = y.x; d = y.y; })
KERNEL(k_pk_mul, { typedef float f2 __attribute__((ext_vector_type(2))); f2 x = {a, b}, y = {c, d}; asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0" : "+v"(x), "+v"(y)); a = x.x; b = x.y; c = y.x; d = y.y; })
KERNEL(k_pk_add, { typedef float f2 __attribute__((ext_vector_type(2))); f2 x = {a, b}, y = {c, d}; asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %0\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %0" : "+v"(x), "+v"(y)); a = x.x; b = x.y; c = y.x; d = y.y; })
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_and, asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %0\n v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %0" : "+v"(u), "+v"(w));)
KERNEL(k_lshl, asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1" : "+v"(u), "+v"(w));)
KERNEL(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %0\n v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %0" : "+v"(u), "+v"(w));)
KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %1, %1\n v_add3_u32 %1, %1, %0, %0\n v_add3_u32 %0, %0, %1, %1\n v_add3_u32 %1, %1, %0, %0" : "+v"(u), "+v"(w));)
KERNEL(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %1, %1\n v_mad_u32_u24 %1, %1, %0, %0\n v_mad_u32_u24 %0, %0, %1, %1\n v_mad_u32_u24 %1, %1, %0, %0" : "+v"(u), "+v"(w));)
KERNEL(k_sub_u32, asm volatile("v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %0\n v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %0" : "+v"(u), "+v"(w));)
KERNEL(k_min_u32, asm volatile("v_min_u32 %0, %0, %1\n v_max_u32 %1, %1, %0\n v_min_u32 %0, %0, %1\n v_max_u32 %1, %1, %0" : "+v"(u), "+v"(w));)
KERNEL(k_med3, asm volatile("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %0\n v_med3_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_cvt_u32, asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %5\n v_cvt_f32_u32 %2, %4\n v_cvt_f32_u32 %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(u), "v"(w));)
KERNEL(k_add_sdwa, asm volatile("v_add_u32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %1, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %0, %0, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa %1, %1, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(u), "+v"(w) : "v"(0x4B000000u), "v"(threadIdx.x));)
KERNEL(k_cndmask_vcc2, asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");)
KERNEL(k_lds, { extern __shared__ float lds[]; float *p = lds + threadIdx.x; asm volatile("ds_write_b32 %2, %0\n ds_read_b32 %1, %2\n s_waitcnt lgkmcnt(0)\n ds_write_b32 %2, %1 offset:1024\n ds_read_b32 %0, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : "v"((unsigned int)(size_t)p) : "memory"); })
KERNEL(k_cndmask_vcc_e64, asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %3, %3, %0, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");)
KERNEL(k_cndmask_vcc_indep, asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(seed), "v"(1.0f) : "vcc");)
KERNEL(k_cndmask_sgpr_indep, asm volatile("v_cndmask_b32_e64 %0, %4, %5, %6\n v_cndmask_b32_e64 %1, %4, %5, %6\n v_cndmask_b32_e64 %2, %4, %5, %6\n v_cndmask_b32_e64 %3, %4, %5, %6" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(seed), "v"(1.0f), "s"(m));)
KERNEL(k_fma_vcc_mix, asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %0, %3\n v_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");)
KERNEL(k_addc, asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %1, vcc, %1, %0, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %1, vcc, %1, %0, vcc" : "+v"(u), "+v"(w) : : "vcc");)
KERNEL(k_div_chain, { a = a / b; b = b + 1.0f; c = c / d; d = d + 1.0f; })     // two IEEE divisions + two adds as the compiler expands them

typedef void (*kern_t)(float *, int, float);

static double run(kern_t k, float *out, int rounds) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), 4096, 0, out, rounds, 1.25f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main() {
    float *out = nullptr;
    if (hipMalloc(&out, 256 * 8 * 256 * sizeof(float)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    const int rounds = 2000;
    struct { const char *name; kern_t k; double per; } t[] = {
        {"v_fma_f32", k_fma, 64}, {"v_add_f32", k_add, 64}, {"v_mul_f32 (SGPR operand)", k_mul_sgpr, 64}, {"v_cvt_f32_ubyteN", k_cvt_ubyte, 64},
        {"v_max3_f32", k_max3, 64}, {"v_min_f32", k_min, 64}, {"v_cndmask_b32 (vcc)", k_cndmask_vcc, 64}, {"v_cndmask_b32 (SGPR pair)", k_cndmask_sgpr, 64},
        {"v_cmp_gt_f32 -> vcc", k_cmp_vcc, 64}, {"v_cmp_gt_f32 -> SGPR pair", k_cmp_sgpr, 64}, {"v_cmp -> SGPR, v_cndmask on it (pairs)", k_cmp_cnd, 64},
        {"v_perm_b32", k_perm, 64}, {"v_fma_mix_f32", k_fma_mix, 64}, {"shift / and / bfe / lshl_add", k_int_alu, 64}, {"v_add_u32", k_add_u32, 64},
        {"v_mul_lo_u32", k_mul_lo, 64}, {"v_rcp_f32", k_rcp, 64}, {"s_and/or/andn2/xor_b64", k_salu, 64}, {"v_readfirstlane_b32", k_readlane, 64},
        {"v_fma_f32 + s_and_b64 alternating", k_valu_salu, 64},
        {"v_mul_f32", k_mul, 64}, {"v_sub_f32", k_sub, 64}, {"v_fmac_f32", k_fmac, 64}, {"v_fma_f32 (SGPR operand)", k_fma_sgpr, 64}, {"v_fma_f32 (inline constant)", k_fma_const, 64},
        {"v_add_f32 (inline constant)", k_add_const, 64}, {"v_pk_fma_f32 (two results each)", k_pk_fma, 64}, {"v_pk_mul_f32", k_pk_mul, 64}, {"v_pk_add_f32", k_pk_add, 64},
        {"v_mov_b32", k_mov, 64}, {"v_and_b32", k_and, 64}, {"v_lshlrev_b32", k_lshl, 64}, {"v_lshl_add_u32", k_lshl_add, 64}, {"v_add3_u32", k_add3, 64},
        {"v_mad_u32_u24", k_mad24, 64}, {"v_sub_u32", k_sub_u32, 64}, {"v_min_u32 / v_max_u32", k_min_u32, 64}, {"v_med3_f32", k_med3, 64}, {"v_cvt_f32_u32", k_cvt_u32, 64},
        {"v_add_u32_sdwa (byte of a source)", k_add_sdwa, 64}, {"v_cmp -> vcc, three v_cndmask on vcc", k_cndmask_vcc2, 64}, {"ds_write_b32 + ds_read_b32 + wait (per pair)", k_lds, 32},
        {"v_cndmask_b32_e64 with vcc as the mask, chain", k_cndmask_vcc_e64, 64}, {"v_cndmask_b32 (vcc), independent results", k_cndmask_vcc_indep, 64},
        {"v_cndmask_b32_e64 (SGPR pair), independent results", k_cndmask_sgpr_indep, 64}, {"one v_cndmask_b32 (vcc) among three v_fma_f32", k_fma_vcc_mix, 64},
        {"v_addc_co_u32 (vcc in and out)", k_addc, 64}, {"a/b (IEEE) + add, compiler's expansion, per division", k_div_chain, 32},
    };
    const double base = run(k_fma, out, rounds);
    printf("8 waves per SIMD, %d rounds of 64 instructions; v_fma_f32: %.3f ms = 1.00\n", rounds, base);
    for (auto &e : t) {
        const double ms = run(e.k, out, rounds);
        printf("  %-52s %8.3f ms   %5.2f x v_fma_f32 per instruction\n", e.name, ms, ms / base * 64.0 / e.per);
        fflush(stdout);
    }
    hipFree(out);
    return 0;
}
