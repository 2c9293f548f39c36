// What does the card sustain in v_mfma_f32_32x32x16_f16 (the F16X2 kernels' instruction) against the bf16 form, alone and
// with the A-halo gather kernel's LDS reads beside it (8 ds_read_b128 per 12 MFMAs)?  Long launches, random operands
// (fp16: both pieces of scaled random floats; bf16: three pieces), two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f16_probe.bin tools/mfma_f16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// F16: 1 = fp16 MFMA, 0 = bf16 MFMA.  NR: ds_read_b128 per 12 MFMAs (fresh operands), 0 = registers only.
template <int F16, int NR>
__global__ __launch_bounds__(512) void probe(float* out, const float4* src, int iters) {
  __shared__ float4 smem[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 512) smem[i] = src[(blockIdx.x * 8192 + i) & 0x7FFFF];   // the source holds 512 K float4 (8 MB)
  __syncthreads();
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float4 pc[8];
  for (int i = 0; i < 8; ++i) pc[i] = smem[lane * 16 + i];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = (it * 5 + i * 32 + lane) & 1023;
      pc[i & 7] = smem[r * 8 + ((i & 7) ^ (r & 7))];
    }
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      const float4 a = pc[(g * 5) & 7], b = pc[(g * 3 + 1) & 7];
      if (F16) acc[g & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc[g & 7], 0, 0, 0);
      else acc[g & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[g & 7], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 512 + tid] = s;
}

template <int F16, int NR>
void run(const char* name, float* out, const float4* src, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double last = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<F16, NR>), dim3(256), dim3(512), 0, 0, out, src, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    last = 256.0 * 8 * iters * 12 * 2.0 * 32 * 32 * 16 * 1e-9 / ms;
  }
  printf("%-52s %7.1f TF = %.3f of 2500\n", name, last, last / 2500.0);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const size_t n = 4u << 20;      // 16-bit values
  unsigned short* h = (unsigned short*)malloc(n * 2 * 2);
  unsigned short* hb = h + n;
  unsigned x = 12345;
  for (size_t i = 0; i < n; ++i) {
    x = x * 1664525u + 1013904223u;
    // fp16: sign, exponent 9..17 (2^-6 .. 2^2), random mantissa; bf16: sign, exponent 121..129, random mantissa
    h[i] = (unsigned short)(((x >> 31) << 15) | ((9 + ((x >> 20) % 9)) << 10) | ((x >> 8) & 0x3FF));
    hb[i] = (unsigned short)(((x >> 31) << 15) | ((121 + ((x >> 20) % 9)) << 7) | ((x >> 8) & 0x7F));
  }
  float4 *sf, *sb;
  hipMalloc(&sf, n * 2); hipMalloc(&sb, n * 2);
  hipMemcpy(sf, h, n * 2, hipMemcpyHostToDevice);
  hipMemcpy(sb, hb, n * 2, hipMemcpyHostToDevice);
  const int it = 100000;
  run<0, 0>("bf16 MFMA only", out, sb, it);
  run<1, 0>("fp16 MFMA only", out, sf, it);
  run<0, 0>("bf16 MFMA only (again)", out, sb, it);
  run<1, 0>("fp16 MFMA only (again)", out, sf, it);
  run<1, 8>("fp16 MFMA + 8 ds_read_b128 per 12 (F16X2 gather)", out, sf, it);
  run<0, 6>("bf16 MFMA + 6 ds_read_b128 per 12 (BF16X3 ratio)", out, sb, it);
  run<1, 4>("fp16 MFMA + 4 ds_read_b128 per 12", out, sf, it);
  run<1, 0>("fp16 MFMA only (last)", out, sf, it);
  hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
