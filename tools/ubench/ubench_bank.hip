// Does the issue rate of v_mad_u64_u32 depend on WHICH registers its operands sit in (VGPR bank conflicts between the two factors and the 64-bit
// addend)?  Eight independent accumulator pairs, all in the same pair of banks (register index mod 4 = 2, 3), factors in explicit registers:
//   f01: factors in banks 0, 1 (no overlap with the accumulator)     f23: factors in banks 2, 3 (the accumulator's)
//   f00: both factors in bank 0                                       f22: both in bank 2
//   s0 : one factor an SGPR, the other in bank 0                      s2 : the other in bank 2
// Not part of the product path.  Build: hipcc --offload-arch=gfx950 -O3 ubench_bank.hip -o ubench_bank
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define ITERS 2000
#define ACCS "v[10:11]", "v[14:15]", "v[18:19]", "v[22:23]", "v[26:27]", "v[30:31]", "v[34:35]", "v[38:39]"
#define MAD(acc, x, y) "v_mad_u64_u32 " acc ", s[10:11], " x ", " y ", " acc "\n\t"
#define BODY(x, y) MAD("v[10:11]", x, y) MAD("v[14:15]", x, y) MAD("v[18:19]", x, y) MAD("v[22:23]", x, y) MAD("v[26:27]", x, y) MAD("v[30:31]", x, y) MAD("v[34:35]", x, y) MAD("v[38:39]", x, y)

#define KERNEL(NAME, X, Y)                                                                                                      \
    __global__ void __launch_bounds__(256) NAME(uint64_t* out, uint32_t sa, uint32_t sb)                                        \
    {                                                                                                                           \
        uint32_t x = sa + threadIdx.x, y = sb ^ threadIdx.x, r;                                                                 \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                                                             \
        asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %1\n\tv_mov_b32 v43, %2\n\tv_mov_b32 v44, %2\n\tv_mov_b32 v46, %2\n\t" \
                     "v_mov_b32 v10, %1\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v14, %2\n\tv_mov_b32 v15, 0\n\tv_mov_b32 v18, %1\n\tv_mov_b32 v19, 0\n\t"     \
                     "v_mov_b32 v22, %2\n\tv_mov_b32 v23, 0\n\tv_mov_b32 v26, %1\n\tv_mov_b32 v27, 0\n\tv_mov_b32 v30, %2\n\tv_mov_b32 v31, 0\n\t"     \
                     "v_mov_b32 v34, %1\n\tv_mov_b32 v35, 0\n\tv_mov_b32 v38, %2\n\tv_mov_b32 v39, 0\n\t"                                              \
                     "s_mov_b32 s12, %3\n\ts_movk_i32 s13, 2000\n"                                                                           \
                     "1:\n\t" BODY(X, Y) BODY(X, Y)                                                                                                       \
                     "s_sub_u32 s13, s13, 1\n\ts_cmp_lg_u32 s13, 0\n\ts_cbranch_scc1 1b\n\t"                                                            \
                     "v_xor_b32 %0, v10, v14\n\tv_xor_b32 %0, %0, v18\n\tv_xor_b32 %0, %0, v22\n\tv_xor_b32 %0, %0, v26\n\tv_xor_b32 %0, %0, v30\n\t"   \
                     "v_xor_b32 %0, %0, v34\n\tv_xor_b32 %0, %0, v38\n\t"                                                                               \
                     : "=&v"(r)                                                                                                                        \
                     : "v"(x), "v"(y), "s"(sb)                                                                                                          \
                     : "v10", "v11", "v14", "v15", "v18", "v19", "v22", "v23", "v26", "v27", "v30", "v31", "v34", "v35", "v38", "v39", "v40", "v41",   \
                       "v42", "v43", "v44", "v46", "s10", "s11", "s12", "s13", "scc");                                                                \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                                         \
        if ((threadIdx.x & 63) == 0) out[gridDim.x * blockDim.x + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    }
KERNEL(k_f01, "v40", "v41")
KERNEL(k_f23, "v42", "v43")
KERNEL(k_f00, "v40", "v44")
KERNEL(k_f22, "v42", "v46")
KERNEL(k_f03, "v40", "v43")
KERNEL(k_s0, "v40", "s12")
KERNEL(k_s2, "v42", "s12")

typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t);
int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    struct { const char* name; kern_t k; } ks[] = { { "factors in banks 0,1 (accumulator 2,3)", k_f01 }, { "factors in banks 2,3 (= accumulator's)", k_f23 }, { "both factors in bank 0", k_f00 },
                                                    { "both factors in bank 2", k_f22 }, { "factors in banks 0,3", k_f03 }, { "SGPR factor + bank 0", k_s0 }, { "SGPR factor + bank 2", k_s2 } };
    for (int wpc : { 4, 8, 12, 16 }) { // waves per CU: 1 .. 4 per SIMD
        const int blocks = cus * wpc / 4;
        uint64_t* d;
        hipMalloc(&d, (size_t)(blocks * 256 + blocks * 4) * 8);
        for (auto& kk : ks) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            kk.k<<<blocks, 256>>>(d, 3, 5);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            kk.k<<<blocks, 256>>>(d, 3, 5);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts = (double)blocks * 4 * ITERS * 16;
            printf("%d waves/SIMD  %-42s %.1f G wave-instr/s\n", wpc / 4, kk.name, insts / (ms * 1e-3) / 1e9);
        }
        hipFree(d);
    }
    return 0;
}
