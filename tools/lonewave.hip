// Microbenchmark: what ONE wavefront alone on its SIMD pays per instruction, for the kinds of instruction the
// cooperative fill's step is made of: independent and dependent integer operations, the one-lane DPP shift
// across the wavefront, LDS reads whose value is used at once or a few instructions later.
// Build: hipcc -O3 --offload-arch=gfx950 tools/lonewave.hip -o tools/lonewave ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(64) void spin(uint32_t* out, int iters, uint32_t seed) {
  __shared__ uint32_t lds[256];
  lds[threadIdx.x] = seed + threadIdx.x;
  lds[threadIdx.x + 64] = seed;
  uint32_t a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = seed + threadIdx.x * (j + 1);
  uint32_t c = seed | 1;
  const uint32_t addr = threadIdx.x * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) {  // 8 independent adds
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
      } else if (KIND == 1) {  // 8 dependent adds
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
      } else if (KIND == 2) {  // 8 dependent three-way maxima
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c), "v"(a[1]));
      } else if (KIND == 3) {  // 8 independent v_perm / v_alignbit
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c), "v"(a[7]));
          asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[j + 1]) : "v"(c));
        }
      } else if (KIND == 4) {  // 8 dependent DPP shifts across the wavefront
#pragma unroll
        for (int j = 0; j < 8; ++j)
          asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[0]));
      } else if (KIND == 5) {  // the step's chain: shift, max3, add, max3, add - dependent
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[1]) : "v"(a[0]));
          asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(a[0]) : "v"(a[1]), "v"(c), "v"(a[2]));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
          asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c), "v"(a[3]));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
        }
      } else if (KIND == 6) {  // the same chain with 6 independent operations between its links (16 in all)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[1]) : "v"(a[0]));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[4]) : "v"(c));
          asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(a[0]) : "v"(a[1]), "v"(c), "v"(a[2]));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[5]) : "v"(c));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[6]) : "v"(c));
          asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c), "v"(a[3]));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[7]) : "v"(c));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[4]) : "v"(c));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[5]) : "v"(c));
        }
      } else if (KIND == 7) {  // LDS read used at once (4 per unit)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint32_t v;
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(v));
        }
      } else if (KIND == 8) {  // LDS write, nobody waits (8 per unit)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(a[j]) : "memory");
      } else if (KIND == 9) {  // 8 independent adds with an SALU instruction after each
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_add_u32 %0, %0, %1\n\ts_add_u32 s20, s20, 1" : "+v"(a[j]) : "v"(c) : "s20");
      }
    }
  }
  uint32_t r = lds[threadIdx.x];
#pragma unroll
  for (int j = 0; j < 8; ++j) r ^= a[j];
  out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int KIND>
void run(const char* name, int blocks, double instr_per_unit) {
  uint32_t* out;
  (void)hipMalloc(&out, (size_t)blocks * 64 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(64), 0, 0, out, 100, 12345u);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, 12345u);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 8 * instr_per_unit;
  printf("%-62s blocks %5d: %7.2f ns per instruction (%5.1f cycles at 2.4 GHz)\n", name, blocks, ms * 1e6 / n,
         ms * 1e6 / n * 2.4);
  (void)hipFree(out);
}

int main() {
  for (int blocks : {1, 256, 1024, 2048, 4096}) {
    run<0>("independent v_add_u32", blocks, 8);
    run<1>("dependent v_add_u32", blocks, 8);
    run<2>("dependent v_max3_i32", blocks, 8);
    run<3>("independent v_perm_b32 / v_alignbit_b32", blocks, 8);
    run<4>("dependent DPP wave_shr:1 (+ s_nop 1)", blocks, 8);
    run<5>("chain shift, max3, add, max3, add (per link)", blocks, 10);
    run<6>("the chain with 6 independent adds between its links (per instr)", blocks, 22);
    run<7>("ds_read_b32 used at once (per read + add)", blocks, 4);
    run<8>("ds_write_b32", blocks, 8);
    run<9>("independent v_add_u32 + s_add_u32 (per pair)", blocks, 8);
  }
  return 0;
}
