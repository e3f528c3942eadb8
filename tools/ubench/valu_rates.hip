// VALU instruction-rate microbenchmark for gfx950 (MI355X).
//
// Purpose: pick the GF(2^255-19) limb representation by measurement, not by guess.
// Each kernel runs ITERS iterations of UNROLL independent chains of one instruction
// per lane and reports cycles per wave-instruction per SIMD, (a) with the chip full
// (8 waves/SIMD) and (b) with one wave per SIMD.
//
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int NCH = 8;  // independent chains per lane

enum Op { MAD_U64_U32, MAD_I64_I32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U32_U24, FMA_F64, ADD_F64,
          ADD_U64, ADD_U32, ADD3_U32, LSHL_ADD_U64, ALIGNBIT, XOR3, FMA_F32, MUL_U64, ADDC_PAIR, ASHR_I64, LSHL_B64, ASHR_I32, AND_B32, BFE_I32, SUB_U32, ADD_LIT, MAD_I32_I24, LSHL_ADD_U32, AND_OR, MUL_U32_U24, CNDMASK, OP_COUNT };

static const char* op_name[] = { "v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24",
  "v_mul_hi_u32_u24", "v_fma_f64", "v_add_f64", "u64 add (C)", "v_add_u32", "v_add3_u32", "v_lshl_add_u64",
  "v_alignbit_b32", "v_xor3/bfi(C xor)", "v_fma_f32", "u64 mul (C)", "v_add_co+v_addc pair", "v_ashrrev_i64", "v_lshlrev_b64", "v_ashrrev_i32", "v_and_b32", "v_bfe_i32", "v_sub_u32", "v_add_u32 literal", "v_mad_i32_i24", "v_lshl_add_u32", "v_and_or_b32", "v_mul_u32_u24", "v_cndmask_b32" };

template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint64_t* out, uint32_t seed) {
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint64_t a64[NCH];
  uint32_t a32[NCH];
  double ad[NCH];
  float af[NCH];
  uint32_t x = seed * 2654435761u + t, y = (seed ^ t) | 1u;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    a64[c] = ((uint64_t)(x + c) << 32) | (y + 3 * c);
    a32[c] = x + 7 * c;
    ad[c] = (double)(x & 0xffff) + c;
    af[c] = (float)(x & 0xff) + c;
  }
  double dy = (double)(y & 1023) * 1e-3 + 0.5;
  float fy = (float)(y & 1023) * 1e-3f + 0.5f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if constexpr (OP == MAD_U64_U32) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a64[c]) : "v"(x), "v"(y) : "vcc");
      } else if constexpr (OP == MAD_I64_I32) {
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a64[c]) : "v"(x), "v"(y) : "vcc");
      } else if constexpr (OP == MUL_LO_U32) {
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a32[c]) : "v"(y));
      } else if constexpr (OP == MUL_HI_U32) {
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a32[c]) : "v"(y));
      } else if constexpr (OP == MAD_U32_U24) {
        asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a32[c]) : "v"(x), "v"(y));
      } else if constexpr (OP == MUL_HI_U32_U24) {
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a32[c]) : "v"(y));
      } else if constexpr (OP == FMA_F64) {
        asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(ad[c]) : "v"(dy));
      } else if constexpr (OP == ADD_F64) {
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(ad[c]) : "v"(dy));
      } else if constexpr (OP == ADD_U64) {
        a64[c] += ((uint64_t)x << 32 | y);
        asm volatile("" : "+v"(a64[c]));
      } else if constexpr (OP == ADD_U32) {
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a32[c]) : "v"(y));
      } else if constexpr (OP == ADD3_U32) {
        asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a32[c]) : "v"(y), "v"(x));
      } else if constexpr (OP == LSHL_ADD_U64) {
        asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(a64[c]) : "v"(a64[(c + 1) % NCH]));
      } else if constexpr (OP == ALIGNBIT) {
        asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a32[c]) : "v"(y));
      } else if constexpr (OP == XOR3) {
        a32[c] = a32[c] ^ x ^ y;
        asm volatile("" : "+v"(a32[c]));
      } else if constexpr (OP == FMA_F32) {
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(af[c]) : "v"(fy));
      } else if constexpr (OP == MUL_U64) {
        a64[c] *= ((uint64_t)x << 32 | y);
        asm volatile("" : "+v"(a64[c]));
      } else if constexpr (OP == ASHR_I64) {
        asm volatile("v_ashrrev_i64 %0, 3, %0" : "+v"(a64[c]));
      } else if constexpr (OP == LSHL_B64) {
        asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a64[c]));
      } else if constexpr (OP == ASHR_I32) {
        asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a32[c]));
      } else if constexpr (OP == AND_B32) {
        asm volatile("v_and_b32 %0, 0x3ffffff, %0" : "+v"(a32[c]));
      } else if constexpr (OP == BFE_I32) {
        asm volatile("v_bfe_i32 %0, %0, 0, 26" : "+v"(a32[c]));
      } else if constexpr (OP == SUB_U32) {
        asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a32[c]) : "v"(y));
      } else if constexpr (OP == ADD_LIT) {
        asm volatile("v_add_u32 %0, 0x40000, %0" : "+v"(a32[c]));
      } else if constexpr (OP == MAD_I32_I24) {
        asm volatile("v_mad_i32_i24 %0, %1, 19, %0" : "+v"(a32[c]) : "v"(y));
      } else if constexpr (OP == LSHL_ADD_U32) {
        asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a32[c]) : "v"(y));
      } else if constexpr (OP == AND_OR) {
        asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a32[c]) : "v"(y), "v"(x));
      } else if constexpr (OP == MUL_U32_U24) {
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a32[c]) : "v"(y));
      } else if constexpr (OP == CNDMASK) {
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a32[c]) : "v"(y));
      } else if constexpr (OP == ADDC_PAIR) {
        uint32_t lo = (uint32_t)a64[c], hi = (uint32_t)(a64[c] >> 32);
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(y) : "vcc");
        a64[c] = ((uint64_t)hi << 32) | lo;
      }
    }
  }
  uint64_t r = 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) r ^= a64[c] ^ a32[c] ^ (uint64_t)ad[c] ^ (uint64_t)af[c];
  if (r == 0x123456789abcdefull) out[t] = r;  // keep live, practically never taken
}

template <int OP>
int run(uint64_t* d_out, int blocks_per_cu, int threads, double clk_ghz, int ncu, int insn_per_op) {
  dim3 grid(ncu * blocks_per_cu), block(threads);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<OP>, grid, block, 0, 0, d_out, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<OP>, grid, block, 0, 0, d_out, (uint32_t)rep + 2);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  // wave-instructions per SIMD
  double waves_per_simd = (double)blocks_per_cu * threads / 64.0 / 4.0;
  double winst_per_simd = waves_per_simd * ITERS * NCH * insn_per_op;
  double cycles = best * 1e-3 * clk_ghz * 1e9;
  printf("  %-22s waves/SIMD=%4.1f  time=%8.3f ms  cyc/wave-inst/SIMD=%6.2f  (lane-ops/s chip = %.2f T)\n", op_name[OP],
         waves_per_simd, best, cycles / winst_per_simd,
         (double)ncu * blocks_per_cu * threads * ITERS * NCH / (best * 1e-3) / 1e12);
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount;
  double clk = p.clockRate / 1e6;  // kHz -> GHz
  printf("device %s  CUs=%d  clock=%.3f GHz (nominal; DVFS may lower it)\n", p.name, ncu, clk);
  uint64_t* d_out; CK(hipMalloc(&d_out, sizeof(uint64_t) * ncu * 8 * 256));
  for (int pass = 0; pass < 3; ++pass) {
    int bpc = pass == 0 ? 8 : (pass == 1 ? 1 : 2);  // 8 / 1 / 2 waves per SIMD (k_msm runs at 2)
    printf("== %d block(s) of 256 threads per CU ==\n", bpc);
    if (run<MAD_U64_U32>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<MAD_I64_I32>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<MUL_LO_U32>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<MUL_HI_U32>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<MAD_U32_U24>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<MUL_HI_U32_U24>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<FMA_F64>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<ADD_F64>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<FMA_F32>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<ADD_U64>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<LSHL_ADD_U64>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<ADDC_PAIR>(d_out, bpc, 256, clk, ncu, 2)) return 1;
    if (run<ADD_U32>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<ADD3_U32>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<ALIGNBIT>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<XOR3>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<MUL_U64>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<ASHR_I64>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<LSHL_B64>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<ASHR_I32>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<AND_B32>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<BFE_I32>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<SUB_U32>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<ADD_LIT>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<MAD_I32_I24>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<LSHL_ADD_U32>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<AND_OR>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<MUL_U32_U24>(d_out, bpc, 256, clk, ncu, 1)) return 1;
    if (run<CNDMASK>(d_out, bpc, 256, clk, ncu, 1)) return 1;
  }
  CK(hipFree(d_out));
  return 0;
}
