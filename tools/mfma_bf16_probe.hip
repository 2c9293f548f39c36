// What bf16 MFMA rate does the card SUSTAIN (power included), with nothing else, and with the split / LDS
// work of the BF16X3 kernels beside it?  Long launches (~0.3 s each) of v_mfma_f32_32x32x16_bf16 on random
// operands, 8 waves per CU (two per SIMD) unless stated.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_bf16_probe.bin tools/mfma_bf16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ void split3(const f32x8 f, bf16x8_t& h, bf16x8_t& m, bf16x8_t& l) {
  u32x4 hp, mp, lp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a0 = f[2 * i], a1 = f[2 * i + 1];
    hp[i] = cvt_pk_bf16(a0, a1);
    const float r0 = a0 - __uint_as_float(hp[i] << 16), r1 = a1 - __uint_as_float(hp[i] & 0xFFFF0000u);
    mp[i] = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(mp[i] << 16), s1 = r1 - __uint_as_float(mp[i] & 0xFFFF0000u);
    lp[i] = cvt_pk_bf16(s0, s1);
  }
  h = __builtin_bit_cast(bf16x8_t, hp);
  m = __builtin_bit_cast(bf16x8_t, mp);
  l = __builtin_bit_cast(bf16x8_t, lp);
}

// MODE bit 0: split3 of two fp32 fragments per 24 MFMAs (the gather kernel's ratio); bit 1: the fragments
// come from LDS (2 x 2 ds_read_b128 per 24 MFMAs) plus 6 ds_read_b128 of "weights"; bit 2: a barrier per 48 MFMAs
template <int NS, int NB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe_mix(float* out, const float* src, int iters) {
  __shared__ float4 smem[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 64 * WAVES) smem[i] = reinterpret_cast<const float4*>(src)[(blockIdx.x * 8192 + i) & 0xFFFFF];
  __syncthreads();
  f32x16 acc[4], acc2[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = acc2[t][r] = 0.f;
  bf16x8_t pc[12];       // piece pool: 4 fragments x 3 planes
  for (int i = 0; i < 12; ++i) pc[i] = __builtin_bit_cast(bf16x8_t, smem[lane * 16 + i]);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int r = (it * 7 + i * 64 + lane) & 1023;
      const float4 lo = smem[r * 8 + ((2 * i) ^ ((r >> 1) & 7))], hi = smem[r * 8 + ((2 * i + 1) ^ ((r >> 1) & 7))];
      split3(f32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}, pc[(3 * i) % 12], pc[(3 * i + 1) % 12], pc[(3 * i + 2) % 12]);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int r = (it * 5 + i * 32 + lane) & 1023;
      pc[(3 * NS + i) % 12] = __builtin_bit_cast(bf16x8_t, smem[r * 8 + ((i & 7) ^ (r & 7))]);
    }
#pragma unroll
    for (int g = 0; g < 24; ++g) {
      const bf16x8_t a = pc[(g * 5) % 12], b = pc[(g * 7 + 3) % 12];
      if (g % 6 != 5) acc2[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2[g & 3], 0, 0, 0);
      else acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g & 3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r] + acc2[t][r];
  out[blockIdx.x * 64 * WAVES + tid] = s;
}

template <int NS, int NB, int WAVES>
void run_mix(const char* name, float* out, const float* src, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double last = 0;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe_mix<NS, NB, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, src, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    last = (double)blocks * WAVES * iters * 24 * 2.0 * 32 * 32 * 16 * 1e-9 / ms;
  }
  printf("mix: %d split3 (+%d b128) and %2d bf16 b128 per 24 MFMAs, %d waves: %-22s %7.1f TF = %.3f of 2500 (as fp32x3 %.1f TF)\n", NS, 2 * NS, NB,
         WAVES, name, last, last / 2500.0, last / 6);
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(float* out, const float* src, int iters) {
  __shared__ float4 smem[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 64 * WAVES) smem[i] = reinterpret_cast<const float4*>(src)[(blockIdx.x * 8192 + i) & 0xFFFFF];
  __syncthreads();
  f32x16 acc[4], acc2[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = acc2[t][r] = 0.f;
  bf16x8_t ah[2], am[2], al[2], bh[2], bm[2], bl[2];
  {
    const float4* s4 = smem + lane * 4;
    for (int i = 0; i < 2; ++i) {
      const float4 lo = s4[i * 512], hi = s4[i * 512 + 1];
      split3(f32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}, ah[i], am[i], al[i]);
      const float4 l2 = s4[i * 512 + 2], h2 = s4[i * 512 + 3];
      split3(f32x8{l2.x, l2.y, l2.z, l2.w, h2.x, h2.y, h2.z, h2.w}, bh[i], bm[i], bl[i]);
    }
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE & 4) { if ((it & 1) == 0) __builtin_amdgcn_s_barrier(); }
    if (MODE & 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float4 lo, hi;
        if (MODE & 2) {
          const int r = (it * 7 + i * 64 + lane) & 1023;
          lo = smem[r * 8 + ((2 * i) ^ ((r >> 1) & 7))];
          hi = smem[r * 8 + ((2 * i + 1) ^ ((r >> 1) & 7))];
        } else {
          lo = make_float4(acc2[i][0], acc2[i][1], acc2[i][2], acc2[i][3]);
          hi = make_float4(acc2[i][4], acc2[i][5], acc2[i][6], acc2[i][7]);
        }
        split3(f32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}, ah[i], am[i], al[i]);
      }
    }
    if (MODE & 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (it * 5 + i * 32 + lane) & 1023;
        bh[i] = __builtin_bit_cast(bf16x8_t, smem[r * 8 + (1 ^ (r & 7))]);
        bm[i] = __builtin_bit_cast(bf16x8_t, smem[r * 8 + (3 ^ (r & 7))]);
        bl[i] = __builtin_bit_cast(bf16x8_t, smem[r * 8 + (5 ^ (r & 7))]);
      }
    }
#pragma unroll
    for (int pp = 0; pp < 6; ++pp)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const bf16x8_t a = pp == 0 ? al[mi] : (pp == 2 || pp == 3) ? am[mi] : ah[mi];
          const bf16x8_t b = pp == 1 ? bl[ni] : (pp == 2 || pp == 4) ? bm[ni] : bh[ni];
          if (pp < 5) acc2[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2[mi * 2 + ni], 0, 0, 0);
          else acc[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[mi * 2 + ni], 0, 0, 0);
        }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r] + acc2[t][r];
  out[blockIdx.x * 64 * WAVES + tid] = s;
}

template <int MODE, int WAVES>
void run(const char* name, float* out, const float* src, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 0, last = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, src, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * WAVES * iters * 24;
    last = mfma * 2.0 * 32 * 32 * 16 * 1e-9 / ms;      // executed bf16 TFLOP/s
    if (last > best) best = last;
  }
  printf("%-58s blocks %4d  last %7.1f TF  best %7.1f TF  (last = %.3f of 2500; as fp32x3: %.1f TF)\n", name, blocks, last, best,
         last / 2500.0, last / 6);
}

int main() {
  float* out; hipMalloc(&out, 2048 * 1024 * 4);
  float* src; const size_t sb = 16u << 20; hipMalloc(&src, sb);
  {
    float* h = (float*)malloc(sb);
    unsigned x = 12345;
    for (size_t i = 0; i < sb / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    hipMemcpy(src, h, sb, hipMemcpyHostToDevice);
    free(h);
  }
  const int it = 60000;
  run<0, 8>("mfma only, 8 waves/CU", out, src, 256, it);
  run<0, 4>("mfma only, 4 waves/CU", out, src, 256, it);
  run<1, 8>("mfma + 2 split3 per 24 (registers), 8 waves", out, src, 256, it);
  run<3, 8>("mfma + 2 split3 + 10 ds_read_b128 per 24, 8 waves", out, src, 256, it);
  run<7, 8>("  .. + barrier per 48", out, src, 256, it);
  run<3, 4>("mfma + 2 split3 + 10 ds_read_b128 per 24, 4 waves", out, src, 256, it);
  run<2, 8>("mfma + 6 ds_read_b128 per 24 (pre-split both), 8 waves", out, src, 256, it);
  run<0, 8>("mfma only, 8 waves/CU (again)", out, src, 256, it);
  run_mix<0, 0, 8>("mfma only", out, src, 256, it);
  run_mix<2, 6, 8>("256x128 4x2 now", out, src, 256, it);
  run_mix<1, 12, 8>("256x128 8x1", out, src, 256, it);
  run_mix<4, 3, 8>("256x128 2x4", out, src, 256, it);
  run_mix<0, 12, 8>("4x2, A planes", out, src, 256, it);
  run_mix<0, 6, 8>("6 reads", out, src, 256, it);
  run_mix<0, 15, 8>("8x1, A planes", out, src, 256, it);
  run_mix<1, 6, 8>("one split, 6 reads", out, src, 256, it);
  run_mix<2, 0, 8>("two splits only", out, src, 256, it);
  run_mix<4, 0, 8>("four splits only (wgrad)", out, src, 256, it);
  run_mix<0, 0, 8>("mfma only (again)", out, src, 256, it);
  hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
