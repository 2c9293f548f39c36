// Micro-benchmark: what fp32 MFMA rate can a 256-thread block per CU sustain on gfx950, alone
// and with the LDS fragment reads / barrier of the conv kernels next to it?
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe.bin tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r.x = (int)(unsigned)a; r.y = (int)((unsigned)(a >> 32) & 0xFFFFu); r.z = (int)bytes; r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void ld16(const i32x4 rsrc, const unsigned voff) {
  i32x4 d;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(d) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void dma16_nom0(const i32x4 rsrc, const unsigned voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void dma16(const i32x4 rsrc, const unsigned voff, const unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(float* out, int tiles, const float* src, unsigned src_bytes, unsigned long long* stamps = nullptr) {
  __shared__ float4 smem[3 * 2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((MODE & 8) && tid == 0) stamps[2048 + blockIdx.x * 4] = wall_clock64();
  for (int i = tid; i < 3 * 2048; i += THREADS) smem[i] = make_float4(i * 1e-6f, 1.f, 0.5f, 0.25f);
  __syncthreads();
  constexpr int NT = THREADS == 256 ? 4 : THREADS == 512 ? 2 : 1;   // 32x32 accumulators per wave
  f32x16 acc[NT];
  for (int t = 0; t < NT; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float4 a0 = smem[lane], a1 = smem[lane + 64], b0 = smem[lane + 128], b1 = smem[lane + 192];
  int stage = 0;
  // MODE&4: the conv kernel's DMA stream: 8 x 1 KiB per wave per tile into the ring, 2 tiles ahead
  const i32x4 rs = make_rsrc(src, src_bytes);
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)smem;
  unsigned goff = (unsigned)((blockIdx.x * 128 + 8 * wave + (lane >> 3)) * 1024 + (lane & 7) * 16);   // row-major [rows][256 floats]
  constexpr int ND = THREADS == 256 ? 8 : THREADS == 512 ? 4 : 2;
  if (MODE & 4) {
    for (int s = 0; s < 2; ++s) {
      for (int q = 0; q < ND; ++q) dma16(rs, goff + q * 32 * 1024, lds0 + (s * 2048 + (wave * ND + q) * 64) * 16);
      goff += 128;
    }
  }
  int nstage = 2;
  if (MODE & 256) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(lds0 + wave * 1024) : "memory");
  for (int it = 0; it < tiles; ++it) {
    if ((MODE & 4) && !(MODE & 32)) {
      if (ND == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (ND == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    if (MODE & 2) __builtin_amdgcn_s_barrier();
    if ((MODE & 8) && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && wave == 0)
      stamps[(blockIdx.x ? 1024 : 0) + it] = wall_clock64();
    const float4* S = smem + stage * 2048;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE & 1) {
        const int r = (wave & 1) * 64 + (lane & 31), c = 2 * j + (lane >> 5);
        a0 = S[r * 8 + (c ^ ((r >> 1) & 7))];
        a1 = S[(r + 32) * 8 + (c ^ (((r + 32) >> 1) & 7))];
        b0 = S[1024 + r * 8 + (c ^ ((r >> 1) & 7))];
        if (NT == 4) b1 = S[1024 + (r + 32) * 8 + (c ^ (((r + 32) >> 1) & 7))];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = e == 0 ? a0.x : e == 1 ? a0.y : e == 2 ? a0.z : a0.w;
        const float x1 = e == 0 ? a1.x : e == 1 ? a1.y : e == 2 ? a1.z : a1.w;
        const float y0 = e == 0 ? b0.x : e == 1 ? b0.y : e == 2 ? b0.z : b0.w;
        const float y1 = e == 0 ? b1.x : e == 1 ? b1.y : e == 2 ? b1.z : b1.w;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0], 0, 0, 0);
        if ((MODE & 4) && !(MODE & 16) && (j * 4 + e) % (16 / ND) == 1) {
          const int q = (j * 4 + e) / (16 / ND);
          if (MODE & 256) dma16_nom0(rs, (goff + q * 32 * 1024) % src_bytes);
          else if (MODE & 64) ld16(rs, (goff + q * 32 * 1024) % src_bytes);
          else dma16(rs, (goff + q * 32 * 1024) % src_bytes, lds0 + (nstage * 2048 + (wave * ND + q) * 64) * 16);
        }
        if ((MODE & 4) && (MODE & 16) && j == 0 && e == 0) {
#pragma unroll
          for (int q = 0; q < ND; ++q) dma16(rs, (goff + q * 32 * 1024) % src_bytes, lds0 + (nstage * 2048 + (wave * ND + q) * 64) * 16);
        }
        if (NT >= 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[1], 0, 0, 0);
        if (NT == 4) {
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[3], 0, 0, 0);
        }
      }
    }
    stage = stage == 2 ? 0 : stage + 1;
    nstage = nstage == 2 ? 0 : nstage + 1;
    goff += 128;
    if ((it & 7) == 7) goff += 128 * 1024 - 1024;   // next "tap": another set of rows
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((MODE & 8) && tid == 0) stamps[2048 + blockIdx.x * 4 + 1] = wall_clock64();
  float s = 0.f;
  for (int t = 0; t < NT; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * THREADS + tid] = s;
  if ((MODE & 8) && tid == 0) stamps[2048 + blockIdx.x * 4 + 2] = wall_clock64();
}

// 4 consumer waves (2x2 MFMA tiles each, never touch VMEM) + PW producer waves that only issue the DMA stream
template <int PW>
__global__ __launch_bounds__(256 + 64 * PW) void probe_ws(float* out, int tiles, const float* src, unsigned src_bytes) {
  __shared__ float4 smem[3 * 2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 3 * 2048; i += 256 + 64 * PW) smem[i] = make_float4(i * 1e-6f, 1.f, 0.5f, 0.25f);
  __syncthreads();
  if (wave >= 4) {
    const int pw = wave - 4;
    constexpr int ND = 32 / PW;
    const i32x4 rs = make_rsrc(src, src_bytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)smem;
    unsigned goff = (unsigned)((blockIdx.x * 128 + (lane >> 3)) * 1024 + (lane & 7) * 16);
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int q = 0; q < ND; ++q) dma16(rs, goff + (pw * ND + q) * 8 * 1024, lds0 + (s * 2048 + (pw * ND + q) * 64) * 16);
      goff += 128;
    }
    int nstage = 2;
    for (int it = 0; it < tiles; ++it) {
      if (ND == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (ND == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int q = 0; q < ND; ++q)
        dma16(rs, (goff + (pw * ND + q) * 8 * 1024) % src_bytes, lds0 + (nstage * 2048 + (pw * ND + q) * 64) * 16);
      nstage = nstage == 2 ? 0 : nstage + 1;
      goff += 128;
      if ((it & 7) == 7) goff += 128 * 1024 - 1024;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  int stage = 0;
  for (int it = 0; it < tiles; ++it) {
    __builtin_amdgcn_s_barrier();
    const float4* S = smem + stage * 2048;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = (wave & 1) * 64 + (lane & 31), c = 2 * j + (lane >> 5);
      const float4 a0 = S[r * 8 + (c ^ ((r >> 1) & 7))];
      const float4 a1 = S[(r + 32) * 8 + (c ^ (((r + 32) >> 1) & 7))];
      const float4 b0 = S[1024 + r * 8 + (c ^ ((r >> 1) & 7))];
      const float4 b1 = S[1024 + (r + 32) * 8 + (c ^ (((r + 32) >> 1) & 7))];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = e == 0 ? a0.x : e == 1 ? a0.y : e == 2 ? a0.z : a0.w;
        const float x1 = e == 0 ? a1.x : e == 1 ? a1.y : e == 2 ? a1.z : a1.w;
        const float y0 = e == 0 ? b0.x : e == 1 ? b0.y : e == 2 ? b0.z : b0.w;
        const float y1 = e == 0 ? b1.x : e == 1 ? b1.y : e == 2 ? b1.z : b1.w;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[3], 0, 0, 0);
      }
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  float sum = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) sum += acc[t][r];
  out[blockIdx.x * 256 + tid] = sum;
}

template <int PW>
void run_ws(const char* name, float* out, int blocks, int tiles, const float* src, unsigned sb) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe_ws<PW>), dim3(blocks), dim3(256 + 64 * PW), 0, 0, out, tiles, src, sb);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe_ws<PW>), dim3(blocks), dim3(256 + 64 * PW), 0, 0, out, tiles, src, sb);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20;
  const double mfma = (double)blocks * 4 * tiles * 64;
  printf("%-34s blocks %4d tiles %4d  %8.1f us  %6.1f TF\n", name, blocks, tiles, us, mfma * 4096 * 1e-6 / us);
}

template <int MODE, int THREADS>
void run(const char* name, float* out, int blocks, int tiles, const float* src = nullptr, unsigned sb = 0, unsigned long long* st = nullptr) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, tiles, src, sb, st);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, tiles, src, sb, st);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20;
  const double mfma = (double)blocks * (THREADS / 64) * tiles * (THREADS == 256 ? 64 : THREADS == 512 ? 32 : 16);
  printf("%-34s blocks %4d tiles %4d  %8.1f us  %6.1f TF  cycles/MFMA/SIMD@2.4GHz %.1f\n", name, blocks, tiles, us,
         mfma * 4096 * 1e-6 / us, us * 1e-6 * 2.4e9 / (mfma / (256.0 * 4)));
}

int main() {
  float* out; hipMalloc(&out, 1024 * 512 * 4);
  float* src; const unsigned sb = 64u << 20; hipMalloc(&src, sb);
  {
    float* h = (float*)malloc(sb);
    unsigned x = 12345;
    for (size_t i = 0; i < sb / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    hipMemcpy(src, h, sb, hipMemcpyHostToDevice);
    free(h);
  }
  unsigned long long* stx; hipMalloc(&stx, 4096 * 8);
  for (int tiles : {72, 720}) {
    run<15, 256>("mfma+lds+barrier+DMA+stamps", out, 256, tiles, src, sb, stx);
    run<7, 256>("mfma+lds+barrier+DMA, 4 waves/CU", out, 256, tiles, src, sb);
    run_ws<4>("specialised: 4 MFMA + 4 DMA waves", out, 256, tiles, src, sb);
    run_ws<2>("specialised: 4 MFMA + 2 DMA waves", out, 256, tiles, src, sb);
    run_ws<1>("specialised: 4 MFMA + 1 DMA wave", out, 256, tiles, src, sb);
    run<7 + 256, 256>("  .. DMA without M0 traffic", out, 256, tiles, src, sb);
    run<7 + 256, 512>("  .. DMA without M0, 8 w/block", out, 256, tiles, src, sb);
    run<7 + 16, 256>("  .. DMAs issued together", out, 256, tiles, src, sb);
    run<7 + 32, 256>("  .. no vmcnt wait", out, 256, tiles, src, sb);
    run<7 + 64, 256>("  .. loads to VGPR not LDS", out, 256, tiles, src, sb);
    run<7 + 64 + 32, 256>("  .. loads to VGPR, no vmcnt", out, 256, tiles, src, sb);
    run<5, 256>("mfma+lds+DMA no barrier, 4 w/CU", out, 256, tiles, src, sb);
    run<7, 512>("mfma+lds+barrier+DMA, 8 w/block", out, 256, tiles, src, sb);
    run<7, 1024>("mfma+lds+barrier+DMA, 16 w/block", out, 256, tiles, src, sb);
    run<3, 1024>("mfma+lds+barrier, 16 w/block", out, 256, tiles, src, sb);
    run<0, 256>("mfma only, 4 waves/CU", out, 256, tiles);
    run<1, 256>("mfma + lds reads, 4 waves/CU", out, 256, tiles);
    run<3, 256>("mfma + lds + barrier, 4 waves/CU", out, 256, tiles);
    run<0, 256>("mfma only, 8 waves/CU (2 blocks)", out, 512, tiles);
    run<3, 256>("mfma+lds+barrier, 8 waves/CU 2blk", out, 512, tiles);
    run<0, 512>("mfma only, 8 waves/block", out, 256, tiles);
    run<3, 512>("mfma+lds+barrier, 8 waves/block", out, 256, tiles);
  }
  {
    unsigned long long* st; hipMalloc(&st, 4096 * 8); hipMemset(st, 0, 4096 * 8);
    static unsigned long long h[4096];
    for (int rep = 0; rep < 3; ++rep)
      hipLaunchKernelGGL((probe<15, 256>), dim3(256), dim3(256), 0, 0, out, 72, src, sb, st);
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; ++b) {
      printf("block %d per-tile wall_clock64 deltas (100 MHz ticks):", b ? 100 : 0);
      for (int i = 1; i < 72; ++i) printf(" %llu", h[b * 1024 + i] - h[b * 1024 + i - 1]);
      printf("\n");
    }
    {
      unsigned long long t0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0, f1 = 0;
      for (int b = 0; b < 256; ++b) {
        const unsigned long long* q = h + 2048 + b * 4;
        if (q[0] < t0) t0 = q[0];
        if (q[0] > s1) s1 = q[0];
        if (q[1] < e0) e0 = q[1];
        if (q[1] > e1) e1 = q[1];
        if (q[2] > f1) f1 = q[2];
      }
      printf("DMA kernel timeline (10 ns ticks from first block start): last block start %llu, loop ends %llu..%llu, last block done %llu; block0 first tile stamp %llu\n",
             s1 - t0, e0 - t0, e1 - t0, f1 - t0, h[0] - t0);
    }
    for (unsigned bytes : {64u << 20, 8u << 20, 1u << 20}) {
      unsigned long long* sts[4];
      for (int i = 0; i < 4; ++i) { hipMalloc(&sts[i], 4096 * 8); hipMemset(sts[i], 0, 4096 * 8); }
      hipDeviceSynchronize();
      for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((probe<15, 256>), dim3(256), dim3(256), 0, 0, out, 72, src, bytes, sts[i]);
      hipDeviceSynchronize();
      unsigned long long pe = 0;
      for (int i = 0; i < 4; ++i) {
        hipMemcpy(h, sts[i], sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, f1 = 0;
        for (int b = 0; b < 256; ++b) { const unsigned long long* q = h + 2048 + b * 4; if (q[0] < t0) t0 = q[0]; if (q[2] > f1) f1 = q[2]; }
        printf("src %3u MiB launch %d: in-kernel %.2f us, gap from previous kernel's last store %.2f us\n", bytes >> 20, i, (f1 - t0) * 0.01, pe ? (t0 - pe) * 0.01 : 0.0);
        pe = f1;
      }
    }
    hipLaunchKernelGGL((probe<11, 256>), dim3(256), dim3(256), 0, 0, out, 72, src, sb, st);
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    printf("no-DMA block 0 deltas:");
    for (int i = 1; i < 72; ++i) printf(" %llu", h[i] - h[i - 1]);
    printf("\n");
  }
  hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
