// Issue-rate micro-benchmark for a LONE wavefront on gfx950 (one wave per workgroup, one workgroup per CU):
// cycles per instruction (s_memtime) of the instruction mixes the workgroup LU is made of.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.hip -o tools/ubench_issue.bin && tools/ubench_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 64
#define T0 asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory")
#define T1 asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory")

__global__ void __launch_bounds__(64) k_bench(uint64_t *out, double seed)
{
    uint64_t t0, t1;
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    double m = 1.0000001, c = 1e-9;
    int n = 0;
    int j0 = threadIdx.x, j1 = j0 * 3, j2 = j0 * 5, j3 = j0 * 7;
    // 1: eight independent v_fma_f64 chains
    T0;
    asm volatile(".rept %c8\n v_fma_f64 %0, %0, %9, %10\n v_fma_f64 %1, %1, %9, %10\n v_fma_f64 %2, %2, %9, %10\n v_fma_f64 %3, %3, %9, %10\n"
                 "v_fma_f64 %4, %4, %9, %10\n v_fma_f64 %5, %5, %9, %10\n v_fma_f64 %6, %6, %9, %10\n v_fma_f64 %7, %7, %9, %10\n .endr"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "n"(REP), "v"(m), "v"(c));
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 2: one dependent v_fma_f64 chain
    T0;
    asm volatile(".rept %c1\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n"
                 "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n .endr"
                 : "+v"(a0) : "n"(REP), "v"(m), "v"(c));
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 3: independent v_readlane_b32 (8 per rept)
    T0;
    asm volatile(".rept %c0\n v_readlane_b32 s20, %1, 3\n v_readlane_b32 s21, %2, 3\n v_readlane_b32 s22, %3, 3\n v_readlane_b32 s23, %4, 3\n"
                 "v_readlane_b32 s24, %1, 5\n v_readlane_b32 s25, %2, 5\n v_readlane_b32 s26, %3, 5\n v_readlane_b32 s27, %4, 5\n .endr"
                 :: "n"(REP), "v"(j0), "v"(j1), "v"(j2), "v"(j3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 4: the update's mix: 4 x (2 readlane) then 4 x (2 fma with the scalar pair), 16 instructions per rept
    T0;
    asm volatile("v_mov_b64 v[100:101], %0\n v_mov_b64 v[102:103], %1\n v_mov_b64 v[104:105], %2\n v_mov_b64 v[106:107], %3\n"
                 "v_mov_b64 v[108:109], %4\n v_mov_b64 v[110:111], %5\n v_mov_b64 v[112:113], %6\n v_mov_b64 v[114:115], %7\n"
                 ".rept %c8\n"
                 "v_readlane_b32 s20, v100, 3\n v_readlane_b32 s21, v101, 3\n v_readlane_b32 s22, v102, 3\n v_readlane_b32 s23, v103, 3\n"
                 "v_readlane_b32 s24, v104, 3\n v_readlane_b32 s25, v105, 3\n v_readlane_b32 s26, v106, 3\n v_readlane_b32 s27, v107, 3\n"
                 "v_fma_f64 v[100:101], -s[20:21], %9, v[100:101]\n v_fma_f64 v[108:109], -s[20:21], %10, v[108:109]\n"
                 "v_fma_f64 v[102:103], -s[22:23], %9, v[102:103]\n v_fma_f64 v[110:111], -s[22:23], %10, v[110:111]\n"
                 "v_fma_f64 v[104:105], -s[24:25], %9, v[104:105]\n v_fma_f64 v[112:113], -s[24:25], %10, v[112:113]\n"
                 "v_fma_f64 v[106:107], -s[26:27], %9, v[106:107]\n v_fma_f64 v[114:115], -s[26:27], %10, v[114:115]\n .endr\n"
                 "v_mov_b64 %0, v[100:101]\n v_mov_b64 %4, v[108:109]"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "n"(REP), "v"(m), "v"(c)
                 : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "v100", "v101", "v102", "v103", "v104", "v105", "v106",
                   "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115");
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 5: independent v_cndmask_b32 (8 per rept)
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    T0;
    asm volatile(".rept %c4\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n"
                 "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n .endr"
                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "n"(REP) : "vcc");
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 6: s_nop 0 (8 per rept)
    T0;
    asm volatile(".rept %c0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n .endr" :: "n"(REP));
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 7: SALU (8 s_add_u32 per rept, dependent)
    T0;
    asm volatile(".rept %c0\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                 "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n .endr" :: "n"(REP)
                 : "s20", "s21", "s22", "s23", "scc");
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 8: v_mul_f64 independent (8 per rept)
    T0;
    asm volatile(".rept %c8\n v_mul_f64 %0, %0, %9\n v_mul_f64 %1, %1, %9\n v_mul_f64 %2, %2, %9\n v_mul_f64 %3, %3, %9\n"
                 "v_mul_f64 %4, %4, %9\n v_mul_f64 %5, %5, %9\n v_mul_f64 %6, %6, %9\n v_mul_f64 %7, %7, %9\n .endr"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "n"(REP), "v"(m));
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 9: readlane -> dependent fma (pair), serial: readlane, readlane, fma on the SAME column (the column-outer chain)
    T0;
    asm volatile("v_mov_b64 v[100:101], %0\n .rept %c1\n"
                 "v_readlane_b32 s20, v100, 3\n v_readlane_b32 s21, v101, 3\n s_nop 1\n v_fma_f64 v[100:101], -s[20:21], %2, v[100:101]\n"
                 "v_readlane_b32 s20, v100, 3\n v_readlane_b32 s21, v101, 3\n s_nop 1\n v_fma_f64 v[100:101], -s[20:21], %2, v[100:101]\n .endr\n"
                 "v_mov_b64 %0, v[100:101]"
                 : "+v"(a0) : "n"(REP), "v"(c) : "s20", "s21", "v100", "v101");
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 10: ds_bpermute_b32 independent (8 per rept) + one wait
    int p0 = threadIdx.x * 4;
    T0;
    asm volatile(".rept %c4\n ds_bpermute_b32 %0, %5, %0\n ds_bpermute_b32 %1, %5, %1\n ds_bpermute_b32 %2, %5, %2\n ds_bpermute_b32 %3, %5, %3\n"
                 "ds_bpermute_b32 %0, %5, %0\n ds_bpermute_b32 %1, %5, %1\n ds_bpermute_b32 %2, %5, %2\n ds_bpermute_b32 %3, %5, %3\n .endr\n s_waitcnt lgkmcnt(0)"
                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "n"(REP), "v"(p0));
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 11: dependent chain of IEEE divisions 1.0 / x (the compiler's expansion: 2 v_div_scale, v_rcp, 6 fma / mul, v_div_fmas, v_div_fixup)
    double x = seed + 2.5;
    T0;
#pragma unroll
    for (int i = 0; i < REP; i++) { x = 1.0 / x; asm volatile("" : "+v"(x)); }
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 12: the same chain through the lean reciprocal (v_rcp + 6 fma: what the expansion computes when nothing is scaled)
    T0;
#pragma unroll
    for (int i = 0; i < REP; i++) {
        double r = __builtin_amdgcn_rcp(x), e = __builtin_fma(-x, r, 1.0);
        r = __builtin_fma(r, e, r); e = __builtin_fma(-x, r, 1.0);
        r = __builtin_fma(r, e, r); e = __builtin_fma(-x, r, 1.0);
        x = __builtin_fma(e, r, r); asm volatile("" : "+v"(x));
    }
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 13: v_rcp_f64 dependent chain
    T0;
#pragma unroll
    for (int i = 0; i < REP; i++) { x = __builtin_amdgcn_rcp(x); asm volatile("" : "+v"(x)); }
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    // 14: v_cmp_gt_f64 -> s_or_b64 chain (VALU writes a scalar pair, the scalar unit consumes it), 2 instructions per link
    T0;
    asm volatile("s_mov_b64 s[20:21], 0\n .rept %c0\n v_cmp_gt_f64 s[22:23], %1, %2\n s_or_b64 s[20:21], s[20:21], s[22:23]\n"
                 "v_cmp_gt_f64 s[22:23], %2, %1\n s_or_b64 s[20:21], s[20:21], s[22:23]\n"
                 "v_cmp_gt_f64 s[22:23], %1, %2\n s_or_b64 s[20:21], s[20:21], s[22:23]\n"
                 "v_cmp_gt_f64 s[22:23], %2, %1\n s_or_b64 s[20:21], s[20:21], s[22:23]\n .endr"
                 :: "n"(REP), "v"(a1), "v"(a2) : "s20", "s21", "s22", "s23", "scc");
    T1; if (threadIdx.x == 0) out[n] = t1 - t0; n++;
    a0 += x;
    if (threadIdx.x == 0) out[15] = (uint64_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) + i0 + i1 + i2 + i3;
}

// the lean reciprocal against the compiler's 1.0 / x, bit for bit, over random x with a biased exponent in [523, 1523]
__global__ void k_check(unsigned long long *bad, int rounds)
{
    uint64_t h = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    unsigned long long nb = 0;
    for (int i = 0; i < rounds; i++) {
        h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
        uint64_t e = 523 + (h >> 12) % 1001, bits = (h & 0x800fffffffffffffull) | (e << 52);
        if ((i & 15) == 0) bits &= 0xfff0000000000000ull | (h >> 40);          /* (short mantissas: exact cases) */
        if ((i & 15) == 1) bits |= 0x000fffffffffff00ull;                      /* (mantissas next to a power of two) */
        double x = __builtin_bit_cast(double, bits);
        double r = __builtin_amdgcn_rcp(x), q = __builtin_fma(-x, r, 1.0);
        r = __builtin_fma(r, q, r); q = __builtin_fma(-x, r, 1.0);
        r = __builtin_fma(r, q, r); q = __builtin_fma(-x, r, 1.0);
        r = __builtin_fma(q, r, r);
        double d = 1.0 / x;
        nb += __builtin_bit_cast(uint64_t, d) != __builtin_bit_cast(uint64_t, r);
    }
    if (nb) atomicAdd(bad, nb);
}

int main()
{
    {
        unsigned long long *db, hb = 0;
        hipMalloc(&db, sizeof hb); hipMemset(db, 0, sizeof hb);
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, db, 1024);
        hipMemcpy(&hb, db, sizeof hb, hipMemcpyDeviceToHost);
        printf("lean reciprocal (v_rcp_f64 + 6 fma) against 1.0 / x over 2^30 random x, exponents 2^-500 .. 2^500: %llu differ\n", hb);
    }
    uint64_t *d, h[16];
    hipMalloc(&d, sizeof h);
    for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, d, 1.0); hipDeviceSynchronize(); }
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[] = {"v_fma_f64 x8 independent", "v_fma_f64 dependent chain", "v_readlane_b32 independent",
                           "update mix (8 readlane + 8 fma)", "v_cndmask_b32 (dependent pairs)", "s_nop 0", "s_add_u32",
                           "v_mul_f64 x8 independent", "readlane,readlane,s_nop 1,fma serial chain (4 instr)", "ds_bpermute_b32 x8 + wait",
                           "1.0 / x dependent chain (per division)", "lean reciprocal chain (per reciprocal)", "v_rcp_f64 dependent chain",
                           "v_cmp_gt_f64 -> s_or_b64 chain"};
    const int per[] = {8, 8, 8, 16, 8, 8, 8, 8, 8, 8, 1, 1, 1, 8};
    for (int i = 0; i < 14; i++)
        printf("%-56s %8.2f s_memtime ticks per instruction (%llu ticks / %d)\n", names[i], (double)h[i] / (REP * per[i]),
               (unsigned long long)h[i], REP * per[i]);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("device clock %d kHz; s_memtime ticks are shader cycles here (whole-function check in the LU: 79.6 k ticks = 33.4 us at 2.4 GHz)\n", clk);
    return 0;
}
