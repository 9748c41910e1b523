// Microbenchmark: issue cost of MIXED streams of "fast" (v_add/v_sub/v_xor/v_lshrrev) and "slow"
// (v_pk_max_u16/v_bfi/v_perm) integer instructions, to see whether fast ops keep their 2.4-cycle
// rate when interleaved with 4.2-cycle ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(256) void spin(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = seed + threadIdx.x * (j + 1);
  uint32_t c = seed | 1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        if (KIND == 0) {  // fast, fast
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
          asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[j + 1]) : "v"(c));
        } else if (KIND == 1) {  // slow, slow
          asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[j]) : "v"(c));
          asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[j + 1]) : "v"(c));
        } else if (KIND == 2) {  // fast, slow alternating
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
          asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[j + 1]) : "v"(c));
        } else if (KIND == 3) {  // dependent chain fast->slow on same register
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
          asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        } else if (KIND == 4) {  // two fast then one slow (3 instr; count separately)
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
          asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[j + 1]) : "v"(c));
          asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[(j + 2) & 7]) : "v"(c));
        } else if (KIND == 5) {  // fast ops with literal constant
          asm volatile("v_xor_b32 %0, 0x80008000, %0" : "+v"(a[j]));
          asm volatile("v_or_b32 %0, 0x3f003f, %0" : "+v"(a[j + 1]));
        } else if (KIND == 6) {  // fast with SGPR operand
          asm volatile("v_add_u32 %0, s20, %0" : "+v"(a[j]));
          asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[j + 1]) : "v"(c));
        } else if (KIND == 7) {  // distinct dst (3-address) fast ops
          uint32_t t0, t1;
          asm volatile("v_add_u32 %0, %1, %2" : "=v"(t0) : "v"(a[j]), "v"(c));
          asm volatile("v_sub_u32 %0, %1, %2" : "=v"(t1) : "v"(a[j + 1]), "v"(c));
          a[j] = t1; a[j + 1] = t0;
        }
      }
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) r ^= a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char* name, int waves_per_simd, double instr_per_inner) {
  int blocks = 256 * waves_per_simd;
  uint32_t* out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  spin<KIND><<<blocks, 256>>>(out, 100, 1);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  spin<KIND><<<blocks, 256>>>(out, iters, 1);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double instr = (double)waves_per_simd * iters * 8.0 * 4.0 * instr_per_inner;  // per SIMD
  printf("%-28s waves/SIMD=%d  %8.3f ms  %.2f cyc/instr/SIMD @2.4GHz\n", name, waves_per_simd, ms,
         (ms * 1e-3 * 2.4e9) / instr);
  (void)hipFree(out);
}

int main() {
  for (int w : {2, 3, 4}) {
    run<0>("fast,fast", w, 2);
    run<1>("slow,slow", w, 2);
    run<2>("fast,slow alternating", w, 2);
    run<3>("fast->slow dependent", w, 2);
    run<4>("fast,fast,slow", w, 3);
    run<5>("fast with literal", w, 2);
    run<6>("fast with sgpr", w, 2);
    run<7>("fast 3-address", w, 2);
  }
  return 0;
}
