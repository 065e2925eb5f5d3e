// valu_rate.hip -- what single VALU instructions sustain on gfx950 with every SIMD holding 1..8 waves: the
// cost model the median's select loop and the fp64 reprojection are compared with.
// build + run:
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/valu_rate tools/valu_rate.hip && ./tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <initializer_list>

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t *out, uint32_t seed, int iters) {
  uint32_t r[8], w = seed ^ threadIdx.x, m = ~seed;
  double d[8], dw = double(seed) * 1.000001, dm = 0.5;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = seed + i * 77u + threadIdx.x, d[i] = 1.0 + i + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 1) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 2) asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x48" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 3) asm volatile("v_or3_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 4) asm volatile("v_and_or_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 5) asm volatile("v_lshl_or_b32 %0, %0, 1, %2" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 6) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 7) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 8) asm volatile("v_alignbit_b32 %0, %1, %0, 7" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 9) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 10) asm volatile("v_bfe_u32 %0, %0, 1, 30" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 11) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[i]) : "v"(w), "v"(m) : "vcc");
        if (OP == 12) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 13) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 14) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 15) asm volatile("v_add3_u32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 16) asm volatile("v_lshl_add_u32 %0, %0, 1, %2" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 17) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 18) asm volatile("v_max_u32 %0, %1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 19) asm volatile("v_min3_u32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 20) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 21) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 22) asm volatile("v_cmp_gt_u32 vcc, %1, %0" : "+v"(r[i]) : "v"(w), "v"(m) : "vcc");
        if (OP == 23) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 24) asm volatile("v_mad_u32_u24 %0, %1, %0, %2" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 25) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 26) asm volatile("v_sad_u8 %0, %1, %0, %2" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 27) asm volatile("v_sad_u32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 28) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 29) asm volatile("v_dot8_u32_u4 %0, %1, %2, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 30) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 31) asm volatile("v_pk_max_u16 %0, %1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 32) asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 33) asm volatile("v_med3_u32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 34) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(w), "v"(m));
        if (OP == 35) asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(d[i]) : "v"(dw));
        if (OP == 36) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(d[i]) : "v"(dw), "v"(dm));
        if (OP == 37) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[i]) : "v"(dw));
        if (OP == 38) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(dw));
        if (OP == 39) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
        if (OP == 40) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[i]) : "v"(d[i]));
        if (OP == 41) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(r[i]));
        if (OP == 42) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(r[i]));
        if (OP == 43) asm volatile("v_div_scale_f64 %0, vcc, %1, %1, %0" : "+v"(d[i]) : "v"(dw) : "vcc");
        if (OP == 44) asm volatile("v_div_fmas_f64 %0, %1, %0, %2" : "+v"(d[i]) : "v"(dw), "v"(dm) : "vcc");
        if (OP == 45) asm volatile("v_div_fixup_f64 %0, %1, %0, %2" : "+v"(d[i]) : "v"(dw), "v"(dm));
        if (OP == 46) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(d[i]) : "v"(dw));
        if (OP == 100) asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x96" : "+v"(r[0]) : "v"(w), "v"(m));
        if (OP == 101) asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x96" : "+v"(r[i & 1]) : "v"(w), "v"(m));
        if (OP == 102) asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x96" : "+v"(r[i & 3]) : "v"(w), "v"(m));
        if (OP == 103) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[0]) : "v"(w));
        if (OP == 104) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i & 1]) : "v"(w));
      }
    }
  }
  uint32_t s = m;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= r[i] ^ uint32_t(d[i]);
  if (s == 0x12345u) out[0] = s;
}

template <int OP>
void run(const char *name) {
  uint32_t *out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  printf("%-20s", name);
  for (int waves : {1, 2, 3, 4, 8}) {
    const int iters = 2000, blocks = 256 * waves;  // 256-thread blocks: one wave on each SIMD of a CU
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  %d w/SIMD %5.2f", waves, ms * 1e-3 * 2.4e9 / (double(iters) * 64 * waves));
  }
  printf("   (cycles per wave-instruction per SIMD, clock taken as 2.4 GHz)\n");
  hipFree(out);
}

int main() {
  run<0>("v_and_b32");
  run<1>("v_xor_b32");
  run<2>("v_bitop3_b32");
  run<3>("v_or3_b32");
  run<4>("v_and_or_b32");
  run<5>("v_lshl_or_b32");
  run<6>("v_lshlrev_b32");
  run<7>("v_ashrrev_i32");
  run<8>("v_alignbit_b32");
  run<9>("v_perm_b32");
  run<10>("v_bfe_u32");
  run<11>("v_cndmask_b32");
  run<12>("v_mov_b32");
  run<13>("v_add_u32");
  run<14>("v_sub_u32");
  run<15>("v_add3_u32");
  run<16>("v_lshl_add_u32");
  run<17>("v_add_lshl_u32");
  run<18>("v_max_u32");
  run<19>("v_min3_u32");
  run<20>("v_bcnt_u32_b32");
  run<21>("v_mbcnt_lo_u32_b32");
  run<22>("v_cmp_gt_u32");
  run<23>("v_mul_u32_u24");
  run<24>("v_mad_u32_u24");
  run<25>("v_mul_lo_u32");
  run<26>("v_sad_u8");
  run<27>("v_sad_u32");
  run<28>("v_dot4_u32_u8");
  run<29>("v_dot8_u32_u4");
  run<30>("v_pk_add_u16");
  run<31>("v_pk_max_u16");
  run<32>("v_pk_lshlrev_b16");
  run<33>("v_med3_u32");
  run<34>("v_fma_f32");
  run<35>("v_pk_fma_f32");
  run<36>("v_fma_f64");
  run<37>("v_mul_f64");
  run<38>("v_add_f64");
  run<39>("v_rcp_f64");
  run<40>("v_cvt_f32_f64");
  run<41>("v_cvt_f64_f32");
  run<42>("v_cvt_f64_u32");
  run<43>("v_div_scale_f64");
  run<44>("v_div_fmas_f64");
  run<45>("v_div_fixup_f64");
  run<46>("v_lshl_add_u64");
  run<100>("bitop3, 1 chain");
  run<101>("bitop3, 2 chains");
  run<102>("bitop3, 4 chains");
  run<103>("v_and, 1 chain");
  run<104>("v_and, 2 chains");
  return 0;
}
