// How fast can waves of one CU issue 1 KiB LDS-DMA instructions (L2-resident source)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r; r.x = (int)(unsigned)a; r.y = (int)((unsigned)(a >> 32) & 0xFFFFu); r.z = (int)bytes; r.w = 0x00020000; return r;
}
__device__ __forceinline__ void dma16(const i32x4 rsrc, const unsigned voff, const unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma4(const i32x4 rsrc, const unsigned voff, const unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
template <int WAVES, int BYTES>
__global__ __launch_bounds__(64 * WAVES) void k(const float* src, unsigned sb, unsigned long long* out, int iters) {
  __shared__ float4 smem[WAVES * 8 * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const i32x4 rs = make_rsrc(src, sb);
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)smem + wave * 8 * 1024;
  unsigned goff = (unsigned)(((blockIdx.x * WAVES + wave) * 64 + (lane >> 3)) * 1024 + (lane & 7) * BYTES);
  // issue cost only: 32 back-to-back DMAs with nothing waited for inside the timed span
  unsigned long long acc = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (BYTES == 16) dma16(rs, (goff + q * 8 * 1024) % sb, lds0 + q * 1024);
        else dma4(rs, (goff + q * 8 * 1024) % sb, lds0 + q * 256);
      }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += t1 - t0;
    goff += 128;
  }
  if (lane == 0) out[blockIdx.x * WAVES + wave] = acc;
}
template <int WAVES, int BYTES>
void run(const float* src, unsigned sb, unsigned long long* d, int blocks) {
  const int iters = 200;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<WAVES, BYTES>), dim3(blocks), dim3(64 * WAVES), 0, 0, src, sb, d, iters);
  unsigned long long h[8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double cyc = (double)h[0] / (iters * 32);
  printf("waves/CU %d  bytes/lane %2d  blocks %4d: %.1f counter ticks per DMA instr issue per wave\n", WAVES, BYTES, blocks, cyc);
}
int main() {
  const unsigned sb = 16u << 20;
  float* src; hipMalloc(&src, sb); hipMemset(src, 0, sb);
  unsigned long long* d; hipMalloc(&d, 8192 * 8);
  for (int blocks : {1, 256}) {
    run<1, 16>(src, sb, d, blocks); run<2, 16>(src, sb, d, blocks); run<4, 16>(src, sb, d, blocks); run<8, 16>(src, sb, d, blocks);
    run<1, 4>(src, sb, d, blocks); run<4, 4>(src, sb, d, blocks);
  }
  hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(hipGetLastError()));
}
