// Instruction-throughput microbenchmark for gfx950 integer/FP64 VALU ops that a
// 256-bit Montgomery multiplier can be built from. Not part of the product path.
// Build: hipcc --offload-arch=gfx950 -O3 ubench_inst.hip -o ubench_inst
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>

#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

#define ITERS 2000
// 16 instructions per loop body, 8 independent dependency chains
#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

#define KERNEL(NAME, DECL, OPM, FOLD) \
__global__ void __launch_bounds__(256) NAME(uint64_t* out, uint32_t sa, uint32_t sb) { \
  DECL \
  uint64_t t0 = __builtin_amdgcn_s_memtime(); \
  for (int it = 0; it < ITERS; ++it) { BODY8(OPM) BODY8(OPM) } \
  uint64_t t1 = __builtin_amdgcn_s_memtime(); \
  FOLD \
  out[blockIdx.x * blockDim.x + threadIdx.x] = r; \
  if ((threadIdx.x & 63) == 0) out[gridDim.x * blockDim.x + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
}

#define DECL64 uint64_t a[8]; uint32_t x = sa + threadIdx.x, y = sb ^ threadIdx.x; for (int i=0;i<8;i++) a[i] = threadIdx.x * 7 + i;
#define FOLD64 uint64_t r = 0; for (int i=0;i<8;i++) r ^= a[i];
#define DECL32 uint32_t a[8]; uint32_t x = sa + threadIdx.x, y = sb ^ threadIdx.x; for (int i=0;i<8;i++) a[i] = threadIdx.x * 7 + i;
#define FOLD32 uint64_t r = 0; for (int i=0;i<8;i++) r ^= a[i];
#define DECLF64 double a[8]; double x = 1.0 + 1e-9 * sa + threadIdx.x, y = 1e-3 * sb; for (int i=0;i<8;i++) a[i] = threadIdx.x * 7 + i;
#define FOLDF64 double rr = 0; for (int i=0;i<8;i++) rr += a[i]; uint64_t r = (uint64_t)__double_as_longlong(rr);

#define OP_MAD64(i) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "s10", "s11");
#define OP_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define OP_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define OP_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define OP_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(x) : "vcc");
#define OP_ADDC(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
#define OP_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
#define OP_MAD24(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define OP_MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define OP_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
#define OP_FMA64(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define OP_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(a[i]) : "v"(x));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
#define OP_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define OP_MADU32(i) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define OP_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
#define OP_LSHR64(i) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(a[i]));
#define OP_LSHR32(i) asm volatile("v_lshrrev_b32 %0, 29, %0" : "+v"(a[i]));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(x));
#define OP_MAD64S(i) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(sb) : "s10", "s11");
#define OP_ANDLIT(i) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(a[i]));
#define OP_MULLOS(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(sb));
#define OP_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[i]));
#define OP_DOT4(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));

KERNEL(k_mad64, DECL64, OP_MAD64, FOLD64)
KERNEL(k_mullo, DECL32, OP_MULLO, FOLD32)
KERNEL(k_mulhi, DECL32, OP_MULHI, FOLD32)
KERNEL(k_add, DECL32, OP_ADD, FOLD32)
KERNEL(k_addco, DECL32, OP_ADDCO, FOLD32)
KERNEL(k_addc, DECL32, OP_ADDC, FOLD32)
KERNEL(k_add3, DECL32, OP_ADD3, FOLD32)
KERNEL(k_mad24, DECL32, OP_MAD24, FOLD32)
KERNEL(k_mulhi24, DECL32, OP_MULHI24, FOLD32)
KERNEL(k_lshladd64, DECL64, OP_LSHLADD64, FOLD64)
KERNEL(k_fma64, DECLF64, OP_FMA64, FOLDF64)
KERNEL(k_alignbit, DECL32, OP_ALIGNBIT, FOLD32)
KERNEL(k_cndmask, DECL32, OP_CNDMASK, FOLD32)
KERNEL(k_and, DECL32, OP_AND, FOLD32)
KERNEL(k_madu16, DECL32, OP_MADU32, FOLD32)
KERNEL(k_pkfma32, DECL64, OP_PKFMA32, FOLD64)
KERNEL(k_dot4, DECL32, OP_DOT4, FOLD32)
KERNEL(k_lshr64, DECL64, OP_LSHR64, FOLD64)
KERNEL(k_lshr32, DECL32, OP_LSHR32, FOLD32)
KERNEL(k_mov, DECL32, OP_MOV, FOLD32)
KERNEL(k_mad64s, DECL64, OP_MAD64S, FOLD64)
KERNEL(k_andlit, DECL32, OP_ANDLIT, FOLD32)
KERNEL(k_mullos, DECL32, OP_MULLOS, FOLD32)
KERNEL(k_lshl64, DECL64, OP_LSHL64, FOLD64)

typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t);
struct Entry { const char* name; kern_t k; };

int main() {
  Entry es[] = {{"v_mad_u64_u32", k_mad64}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_add_u32", k_add},
                {"v_add_co_u32", k_addco}, {"v_addc_co_u32", k_addc}, {"v_add3_u32", k_add3}, {"v_mad_u32_u24", k_mad24},
                {"v_mul_hi_u32_u24", k_mulhi24}, {"v_lshl_add_u64", k_lshladd64}, {"v_fma_f64", k_fma64},
                {"v_alignbit_b32", k_alignbit}, {"v_cndmask_b32", k_cndmask}, {"v_and_b32", k_and}, {"v_mad_u32_u16", k_madu16},
                {"v_pk_fma_f32", k_pkfma32}, {"v_dot4_u32_u8", k_dot4}, {"v_lshrrev_b64", k_lshr64}, {"v_lshrrev_b32", k_lshr32}, {"v_mov_b32", k_mov}, {"v_mad_u64_u32(sgpr)", k_mad64s}, {"v_and_b32(literal)", k_andlit}, {"v_mul_lo_u32(sgpr)", k_mullos}, {"v_lshlrev_b64", k_lshl64}};
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
  uint64_t* d; size_t maxthreads = (size_t)cus * 8 * 256; CHECK(hipMalloc(&d, (maxthreads + maxthreads / 64) * 8));
  std::vector<uint64_t> h(maxthreads + maxthreads / 64);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-22s %6s %12s %14s %16s\n", "instr", "w/SIMD", "ms", "cyc/wave-inst", "Gwaveinst/s/chip");
  for (auto& e : es) {
    for (int wps : {1, 2, 3, 4, 8}) {
      int blocks = cus * wps;  // 256 threads = 4 waves = 1 wave per SIMD per block
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      size_t nt = (size_t)blocks * 256;
      CHECK(hipMemcpy(h.data(), d, (nt + nt / 64) * 8, hipMemcpyDeviceToHost));
      double cyc = 0; for (size_t w = 0; w < nt / 64; ++w) cyc += (double)h[nt + w]; cyc /= (nt / 64);
      double ninst = (double)ITERS * 16;
      // per-SIMD issue cost: cycles elapsed per wave / instr per wave / waves sharing the SIMD
      printf("%-22s %6d %12.4f %14.3f %16.2f\n", e.name, wps, ms, cyc / ninst / wps, (double)(nt / 64) * ninst / (ms * 1e-3) / 1e9);
    }
  }
  return 0;
}
