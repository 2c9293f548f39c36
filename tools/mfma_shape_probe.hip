// Under the card's power limit, does v_mfma_f32_16x16x32_f16 deliver more than v_mfma_f32_32x32x16_f16 for the SAME work and
// the SAME LDS fragment traffic (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15x in bare bf16 loops)?  Per iteration
// both bodies read NR x ds_read_b128 and execute 12 x 32x32x16 or 24 x 16x16x32 MFMAs (equal FLOPs); random fp16 operands,
// two waves per SIMD, long launches.  8 reads per iteration is the A-halo gather kernel's ratio.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_shape_probe.bin tools/mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
constexpr unsigned SRC_F4 = 1u << 19;   // the source holds 512 K float4 (8 MB); every index is masked with SRC_F4 - 1

template <int SMALL, int NR>
__global__ __launch_bounds__(512) void probe(float* out, const float4* src, int iters) {
  __shared__ float4 smem[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 512) smem[i] = src[(blockIdx.x * 8192u + i) & (SRC_F4 - 1)];
  __syncthreads();
  f32x16 acc[8];
  f32x4 acc4[16];
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int t = 0; t < 16; ++t) acc4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 pc[8];
  for (int i = 0; i < 8; ++i) pc[i] = smem[lane * 16 + i];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = (it * 5 + i * 32 + lane) & 1023;
      pc[i & 7] = smem[r * 8 + ((i & 7) ^ (r & 7))];
    }
    if (SMALL) {
#pragma unroll
      for (int g = 0; g < 24; ++g) {
        const float4 a = pc[(g * 5) & 7], b = pc[(g * 3 + 1) & 7];
        acc4[g & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc4[g & 15], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 12; ++g) {
        const float4 a = pc[(g * 5) & 7], b = pc[(g * 3 + 1) & 7];
        acc[g & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc[g & 7], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int t = 0; t < 16; ++t)
    for (int r = 0; r < 4; ++r) s += acc4[t][r];
  out[blockIdx.x * 512 + tid] = s;
}

template <int SMALL, int NR>
void run(const char* name, float* out, const float4* src, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double last = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<SMALL, NR>), dim3(256), dim3(512), 0, 0, out, src, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    last = 256.0 * 8 * iters * 12 * 2.0 * 32 * 32 * 16 * 1e-9 / ms;
  }
  printf("%-60s %7.1f TF = %.3f of 2517\n", name, last, last / 2516.8);
  fflush(stdout);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const size_t n = (size_t)SRC_F4 * 8;      // 16-bit values
  unsigned short* h = (unsigned short*)malloc(n * 2);
  unsigned x = 12345;
  for (size_t i = 0; i < n; ++i) {
    x = x * 1664525u + 1013904223u;
    h[i] = (unsigned short)(((x >> 31) << 15) | ((9 + ((x >> 20) % 9)) << 10) | ((x >> 8) & 0x3FF));
  }
  float4* sf;
  hipMalloc(&sf, n * 2);
  hipMemcpy(sf, h, n * 2, hipMemcpyHostToDevice);
  const int it = 60000;
  run<0, 0>("32x32x16 f16 only", out, sf, it);
  run<1, 0>("16x16x32 f16 only", out, sf, it);
  run<0, 8>("32x32x16 f16 + 8 ds_read_b128 per 12 (A-halo gather)", out, sf, it);
  run<1, 8>("16x16x32 f16 + 8 ds_read_b128 per 24 (same work)", out, sf, it);
  run<0, 8>("32x32x16 f16 + 8 reads (again)", out, sf, it);
  run<1, 8>("16x16x32 f16 + 8 reads (again)", out, sf, it);
  run<1, 12>("16x16x32 f16 + 12 ds_read_b128 per 24", out, sf, it);
  hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
