// Does the time of ds_bpermute_b32 depend on WHICH lanes are read?  (round 4)  kernels.hip msm_add_positional_secret takes a table
// entry for a secret digit from the lane that holds it; the exchange goes through the LDS crossbar, whose banks are what could
// make its duration a function of the digits.  Its source slots are lanes 0..31 only - one per bank of the 32-bank rule that
// ds_read_b32 follows (MI355X_MICROARCH.md, LDS) - so no pattern of digits should conflict.  This program times 27 exchanges per
// iteration (one lookup) under index patterns that stay inside lanes 0..31:
//   same      every lane reads lane 7                                  (a broadcast)
//   identity  lane l reads lane l & 31
//   random    a fresh random lane 0..31 per lane and iteration          (what recoded secret digits look like)
//   two       half the lanes read lane 3, the others lane 19            (two slots 16 apart)
//   stride    lane l reads (5 l) & 31;   reverse: lane l reads 31 - (l & 31)
//   near1 / near8   two slots 1 / 8 apart, a random half of the lanes each;   four: four slots 8 apart, random lanes each
// and, as a control that the program can SEE a conflict at all, patterns that leave that range:
//   wide      a random lane 0..63: lanes l and l + 32 share a bank
//   clash     even lanes read lane 0, odd lanes lane 32                 (same bank, two addresses, inside each 32-lane half)
// One wave per SIMD and eight waves per SIMD; cycles per exchange from s_memtime around the loop (first lane of block 0), and the
// kernel's duration.  With an argument: one launch per pattern, for rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
// (tools/rocpd_pmc.py db --each k_lookup lists the launches in pattern order).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/bperm_lookup.hip -o variants/bperm_lookup
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

enum { P_SAME, P_IDENTITY, P_RANDOM, P_TWO, P_STRIDE, P_NEAR1, P_NEAR8, P_FOUR, P_REVERSE, P_WIDE, P_CLASH, P_COUNT };
static const char* NAMES[P_COUNT] = { "same", "identity", "random", "two", "stride", "near1", "near8", "four", "reverse", "wide(control)", "clash(control)" };

__global__ void __launch_bounds__(256) k_lookup(uint32_t* out, unsigned long long* probe, int pattern, int iters, uint32_t seed) {
  const uint32_t lane = threadIdx.x & 63u, t = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t held[27], acc[27];
#pragma unroll
  for (int l = 0; l < 27; l++) { held[l] = t * 2654435761u + l * 40503u; acc[l] = 0; }
  uint32_t x = (seed ^ t) * 2654435761u | 1u;
  const bool p = blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long c0 = 0;
  if (p) c0 = clock64();
  for (int it = 0; it < iters; it++) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    uint32_t src;
    switch (pattern) {   // uniform
      case P_SAME: src = 7; break;
      case P_IDENTITY: src = lane & 31u; break;
      case P_RANDOM: src = x & 31u; break;
      case P_TWO: src = (x & 0x100u) ? 3u : 19u; break;
      case P_STRIDE: src = (5u * lane) & 31u; break;
      case P_NEAR1: src = (x & 0x100u) ? 12u : 13u; break;
      case P_NEAR8: src = (x & 0x100u) ? 4u : 12u; break;
      case P_FOUR: src = ((x >> 8) & 3u) * 8u + (uint32_t)(it & 7); break;
      case P_REVERSE: src = 31u - (lane & 31u); break;
      case P_WIDE: src = x & 63u; break;
      default: src = (lane & 1u) ? 32u : 0u; break;
    }
    const int a = (int)(src << 2);
#pragma unroll
    for (int l = 0; l < 27; l++) acc[l] += (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)held[l]);
    held[it % 27] += acc[(it + 5) % 27];   // keeps the values moving
  }
  uint32_t s = 0;
#pragma unroll
  for (int l = 0; l < 27; l++) s ^= acc[l];
  out[t] = s;
  if (p) probe[0] = clock64() - c0;
}

int main(int argc, char** argv) {
  const bool once = argc > 1;   // any argument: one launch per pattern at 2 waves per SIMD (for a rocprofv3 --pmc run: tools/rocpd_pmc.py db --each k_lookup)
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int ncu = pr.multiProcessorCount, iters = 20000;
  uint32_t* out; unsigned long long* probe;
  hipMalloc(&out, sizeof(uint32_t) * ncu * 8 * 256); hipMalloc(&probe, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("# ds_bpermute_b32, 27 exchanges per iteration, %d iterations, %d CUs\n", iters, ncu);
  for (int wps : { 1, 2, 8 }) {
    if (once && wps != 2) continue;
    const int blocks = ncu * wps;   // blocks of 256 = one wave per SIMD each
    printf("== %d wave(s) per SIMD ==\n", wps);
    for (int pat = 0; pat < P_COUNT; pat++) {
      float best = 1e30f; unsigned long long cyc = 0;
      for (int rep = 0; rep < (once ? 1 : 4); rep++) {
        hipMemset(probe, 0, 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_lookup, dim3(blocks), dim3(256), 0, 0, out, probe, pat, iters, 1234u + rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(&cyc, probe, 8, hipMemcpyDeviceToHost); }
      }
      printf("  %-15s kernel %8.3f ms   %6.2f shader cycles per exchange (one wave's view)\n", NAMES[pat], best, (double)cyc / ((double)iters * 27));
    }
  }
  return 0;
}
