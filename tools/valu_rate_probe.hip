// Micro-benchmark: issue rate of the VALU operations the BF16X3 split uses on gfx950, per SIMD:
// v_cvt_pk_bf16_f32 against v_add_f32 / v_and_b32 / v_perm_b32 / v_lshlrev_b32 (8 independent chains per
// wave, 1 or 2 waves per SIMD, one workgroup per CU).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate_probe.bin tools/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  unsigned v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 2654435761u + i * 40503u + 0x3f800000u;
  const unsigned k = 0xFFFF0000u + (threadIdx.x & 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        if (OP == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
        if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(k));
        if (OP == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[i]));
        if (OP == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        if (OP == 6) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(v[i]) : "v"(k));
        if (OP == 7) asm volatile("v_lshl_add_u32 %0, %0, 16, %1" : "+v"(v[i]) : "v"(k));
        if (OP == 8) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(k));
        if (OP == 9) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(k), "v"(v[(i + 1) & 7]));
        if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(k), "v"(v[(i + 1) & 7]));
      }
  }
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s ^= v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
}

template <int OP>
void run(const char* name, int threads, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<OP>), dim3(256), dim3(threads), 0, 0, out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((probe<OP>), dim3(256), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_wave = (double)iters * 16 * 8;
  const double waves_per_simd = threads / 256.0;
  // nanoseconds per instruction per SIMD; at 2.4 GHz a full-rate wave64 VALU op is 4 cycles = 1.67 ns
  printf("%-22s %d waves/SIMD: %.3f ms, %.2f ns per instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, (int)waves_per_simd, ms,
         ms * 1e6 / (insts_per_wave * waves_per_simd), ms * 1e6 / (insts_per_wave * waves_per_simd) * 2.4);
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    run<0>("v_cvt_pk_bf16_f32", threads, out);
    run<1>("v_add_f32", threads, out);
    run<5>("v_sub_f32", threads, out);
    run<2>("v_and_b32", threads, out);
    run<3>("v_perm_b32", threads, out);
    run<4>("v_lshlrev_b32", threads, out);
    run<6>("v_lshl_or_b32", threads, out);
    run<7>("v_lshl_add_u32", threads, out);
    run<8>("v_mul_u32_u24", threads, out);
    run<9>("v_and_or_b32", threads, out);
    run<10>("v_fma_f32", threads, out);
  }
  (void)hipFree(out);
  return 0;
}
