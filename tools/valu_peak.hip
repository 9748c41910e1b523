// Microbenchmark: sustained integer VALU issue rate on one GPU, for the roofline denominator.
// Measures lane-operations/s for v_add_u32, v_max_i32, v_pk_add_i16 / v_pk_max_i16 / v_pk_mad_u16 chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short short2_t __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void spin(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = seed + threadIdx.x * (j + 1);
  uint32_t c = seed | 1;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 d2[4]; f2 c2 = {1.0f, 2.0f};
  for (int j = 0; j < 4; ++j) d2[j] = f2{(float)a[j], 1.0f};
  asm volatile("s_mov_b64 s[10:11], 0x5555" ::: "s10", "s11");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 1) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 3) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 5) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a[j]));
        if (KIND == 6) asm volatile("v_lshrrev_b32 %0, 2, %0" : "+v"(a[j]));
        if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 8) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 9) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 10) asm volatile("v_max_u16 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 11) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 12) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 13) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 14) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 15) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 16) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 17) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 18) asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 19) asm volatile("v_bfe_u32 %0, %0, 2, 5" : "+v"(a[j]));
        if (KIND == 20) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 21) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 22) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 23) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 24) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 25) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 26) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 27) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 28) asm volatile("v_pk_lshrrev_b16 %0, %1, %0" : "+v"(a[j]) : "v"(c));
        if (KIND == 29) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 30) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 31) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d2[j&3]) : "v"(c2));
        if (KIND == 32) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(c));
        if (KIND == 33) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[j]) : "v"(c));
        if (KIND == 34) asm volatile("v_cmp_eq_u32 vcc, %0, %1" :: "v"(a[j]), "v"(c) : "vcc");
        if (KIND == 35) asm volatile("v_cmp_gt_i32_e64 s[12:13], %0, %1" :: "v"(a[j]), "v"(c) : "s12","s13");
        if (KIND == 36) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 37) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(a[j]) : "v"(c));
        if (KIND == 38) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(c));
        if (KIND == 39) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(c));
        if (KIND == 40) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(a[j]) : "v"(c));
        if (KIND == 41) asm volatile("v_sad_u16 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 42) asm volatile("v_msad_u8 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 43) asm volatile("v_sub_u32_e64 %0, %0, %1 clamp" : "+v"(a[j]) : "v"(c));
        if (KIND == 44) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a[j]) : "v"(c));
        if (KIND == 45) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
        if (KIND == 46) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
      }
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) r ^= a[j];
  for (int j = 0; j < 4; ++j) r ^= (uint32_t)d2[j].x ^ (uint32_t)d2[j].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char* name, int lanes_per_op, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
  uint32_t* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  spin<KIND><<<blocks, 256>>>(out, 100, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  spin<KIND><<<blocks, 256>>>(out, iters, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr = (double)blocks * 4 /*waves*/ * iters * 16.0 * 8.0;
  double lane_ops = instr * 64 * lanes_per_op;
  printf("%-22s waves/SIMD=%d  %8.3f ms  %7.2f T wave-lane-instr/s  %7.2f T elem-ops/s  (%.2f cyc/instr/SIMD @2.4GHz)\n", name,
         waves_per_simd, ms, instr * 64 / ms / 1e9, lane_ops / ms / 1e9,
         (ms * 1e-3 * 2.4e9) / (instr / (256.0 * 4)));
  hipFree(out);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  for (int w : {2, 4}) {
    run<0>("v_add_u32", 1, w);
    run<1>("v_sub_u32", 1, w);
    run<2>("v_and_b32", 1, w);
    run<3>("v_or_b32", 1, w);
    run<4>("v_xor_b32", 1, w);
    run<5>("v_lshlrev_b32", 1, w);
    run<6>("v_lshrrev_b32", 1, w);
    run<7>("v_mov_b32", 1, w);
    run<8>("v_max_i32", 1, w);
    run<9>("v_min_u32", 1, w);
    run<10>("v_max_u16", 1, w);
    run<11>("v_add_u16", 1, w);
    run<12>("v_max3_i32", 1, w);
    run<13>("v_add3_u32", 1, w);
    run<14>("v_lshl_add_u32", 1, w);
    run<15>("v_lshl_or_b32", 1, w);
    run<16>("v_and_or_b32", 1, w);
    run<17>("v_or3_b32", 1, w);
    run<18>("v_bfi_b32", 1, w);
    run<19>("v_bfe_u32", 1, w);
    run<20>("v_mad_u32_u24", 1, w);
    run<21>("v_mul_u32_u24", 1, w);
    run<22>("v_fma_f32", 1, w);
    run<23>("v_add_f32", 1, w);
    run<24>("v_max_f32", 1, w);
    run<25>("v_pk_add_u16", 2, w);
    run<26>("v_pk_max_i16", 2, w);
    run<27>("v_pk_mad_u16", 2, w);
    run<28>("v_pk_lshrrev_b16", 2, w);
    run<29>("v_pk_fma_f16", 2, w);
    run<30>("v_pk_max_f16", 2, w);
    run<31>("v_pk_add_f32", 2, w);
    run<32>("v_cndmask_vcc", 1, w);
    run<33>("v_cndmask_sgpr", 1, w);
    run<34>("v_cmp_eq_u32", 1, w);
    run<35>("v_cmp_gt_i32_sgpr", 1, w);
    run<36>("v_perm_b32", 1, w);
    run<37>("v_alignbit_b32", 1, w);
    run<38>("v_add_u32_dpp", 1, w);
    run<39>("v_mov_b32_dpp", 1, w);
    run<40>("v_add_u32_sdwa", 1, w);
    run<41>("v_sad_u16", 1, w);
    run<42>("v_msad_u8", 1, w);
    run<43>("v_sub_u32_e64_clamp", 1, w);
    run<44>("v_pk_sub_u16_clamp", 2, w);
    run<45>("v_min3_u32", 1, w);
    run<46>("v_med3_i32", 1, w);
  }
  return 0;
}
