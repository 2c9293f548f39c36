// Probe: does an out-of-range `buffer_load_dwordx4 ... lds` lane write zeros to LDS, or leave it?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, float* out, int bytes) {
  __shared__ float4 lds[64];
  lds[threadIdx.x] = make_float4(7.f, 7.f, 7.f, 7.f);
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, bytes, 0x00020000);
  int off = threadIdx.x * 16;
  if (threadIdx.x & 1) off = 0x7fffff00;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float4 v = lds[threadIdx.x];
  out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y;
  out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
  float *in, *out; float h[256], o[256];
  for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
  hipMalloc(&in, 1024); hipMalloc(&out, 1024);
  hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out, 1024);
  hipMemcpy(o, out, 1024, hipMemcpyDeviceToHost);
  for (int t = 0; t < 6; ++t) printf("lane %d: %g %g %g %g\n", t, o[t*4], o[t*4+1], o[t*4+2], o[t*4+3]);
  return 0;
}
