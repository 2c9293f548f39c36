// InstanceNorm2d (+activation, +residual) forward/backward on NHWC fp32, and the column
// reductions they share with the bias gradients.  All of these are HBM-bound: every tensor
// is touched with 16-byte accesses along the channel axis, partial sums are combined in
// double precision in a fixed order (bitwise reproducible, no atomics).
#include "fcd_common.h"
#include <mutex>

namespace {

constexpr int MODE_STATS = 0;  // a0 = z, a1 = z*z
constexpr int MODE_BWD = 1;    // g1 = gy*act'(y): a0 = g1, a1 = g1*xhat
constexpr int MODE_SUM = 2;    // a0 = x

struct ColPlan {
  int c4;       // float4 columns
  int c4b;      // float4 columns per block (<= 256)
  int cblocks;  // column blocks (grid.z)
  int rows;     // pixel rows handled concurrently by one block
  int chunks;   // pixel chunks per image
  int chunk_len;
};

// Shared by the workspace query and the launch.
ColPlan col_plan(int N, long HW, int C) {
  ColPlan p;
  p.c4 = C / 4;
  p.c4b = p.c4 < 256 ? p.c4 : 256;
  p.cblocks = (p.c4 + 255) / 256;
  p.rows = 256 / p.c4b;
  if (p.rows < 1) p.rows = 1;
  long want = 1024 / (N > 0 ? N : 1);
  if (want < 1) want = 1;
  long maxc = (HW + p.rows - 1) / p.rows;   // at least one row per thread
  long chunks = want < maxc ? want : maxc;
  if (chunks < 1) chunks = 1;
  p.chunk_len = (int)((HW + chunks - 1) / chunks);
  p.chunks = (int)((HW + p.chunk_len - 1) / p.chunk_len);
  return p;
}

// The normalised + affine value of the forward pass.  in_apply_kernel and both backward kernels
// evaluate this one expression, so the backward can re-derive the activation mask (value > 0) from
// z instead of reading the forward's output again (tests compare against the mask taken from y).
__device__ __forceinline__ float in_forward_value(float z, float mu, float rs, float ga, float be) {
  return (z - mu) * rs * ga + be;
}

// act_value / act_slope_at: fcd_common.h
// partial[n][chunk][c][2] (double)
template <int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                          const float* __restrict__ x2, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, double* __restrict__ partial,
                                                          long HW, int C, ColPlan pl, float act_slope,
                                                          const float* __restrict__ gamma = nullptr,
                                                          const float* __restrict__ beta = nullptr) {
  extern __shared__ double sm[];   // [rows][C][2]
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int c4b = pl.c4b, CB = c4b * 4;
  const int cbase = blockIdx.z * 1024;
  const int tx = threadIdx.x % c4b, ty = threadIdx.x / c4b;
  const bool active = ty < pl.rows && cbase / 4 + tx < pl.c4;
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    const long p0 = (long)chunk * pl.chunk_len;
    const long p1 = min(HW, p0 + pl.chunk_len);
    float mu[4] = {0, 0, 0, 0}, rs[4] = {0, 0, 0, 0}, ga[4] = {0, 0, 0, 0}, be[4] = {0, 0, 0, 0};
    // x1 (the forward output y) is optional: with beta given the mask is recomputed from z, and a
    // slope of 1 needs no mask at all
    const bool need_mask = MODE == MODE_BWD && act_slope != 1.f;
    const bool recompute = need_mask && beta != nullptr;
    if (MODE == MODE_BWD) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mu[e] = mean[(long)n * C + cbase + tx * 4 + e];
        rs[e] = rstd[(long)n * C + cbase + tx * 4 + e];
        if (recompute) {
          ga[e] = gamma[cbase + tx * 4 + e];
          be[e] = beta[cbase + tx * 4 + e];
        }
      }
    }
    // MODE_STATS sums z - sh and (z - sh)^2 with sh = the thread's first value: a channel whose mean is
    // large against its spread (|mean| >> std) would otherwise lose its variance in the fp32 partial
    // sums of z^2; the shift is undone in double when the partial is written
    float sh[4] = {0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    if (MODE == MODE_STATS && p0 + ty < p1) {
      const float4 f = *reinterpret_cast<const float4*>(x0 + ((long)n * HW + p0 + ty) * C + cbase + tx * 4);
      sh[0] = f.x; sh[1] = f.y; sh[2] = f.z; sh[3] = f.w;
    }
    for (long px = p0 + ty; px < p1; px += pl.rows) {
      const long off = ((long)n * HW + px) * C + cbase + tx * 4;
      const float4 a = *reinterpret_cast<const float4*>(x0 + off);
      const float av[4] = {a.x, a.y, a.z, a.w};
      if (MODE == MODE_STATS) {
        ++cnt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dv = av[e] - sh[e];
          s0[e] += dv;
          s1[e] += dv * dv;
        }
      } else if (MODE == MODE_SUM) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s0[e] += av[e];
      } else {
        const float4 z = *reinterpret_cast<const float4*>(x2 + off);
        const float zv[4] = {z.x, z.y, z.z, z.w};
        float yv[4] = {1.f, 1.f, 1.f, 1.f};
        if (recompute) {
#pragma unroll
          for (int e = 0; e < 4; ++e) yv[e] = in_forward_value(zv[e], mu[e], rs[e], ga[e], be[e]);
        } else if (need_mask) {
          const float4 y = *reinterpret_cast<const float4*>(x1 + off);
          yv[0] = y.x; yv[1] = y.y; yv[2] = y.z; yv[3] = y.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g1 = av[e] * act_slope_at(yv[e], act_slope);
          s0[e] += g1;
          s1[e] += g1 * ((zv[e] - mu[e]) * rs[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double d0 = (double)s0[e], d1 = (double)s1[e];
      if (MODE == MODE_STATS) {      // sum z = sum d + cnt sh;  sum z^2 = sum d^2 + 2 sh sum d + cnt sh^2
        const double shd = (double)sh[e];
        d1 += 2.0 * shd * d0 + (double)cnt * shd * shd;
        d0 += (double)cnt * shd;
      }
      sm[((long)ty * CB + tx * 4 + e) * 2 + 0] = d0;
      sm[((long)ty * CB + tx * 4 + e) * 2 + 1] = d1;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < CB && cbase + c < C; c += 256) {
    double t0 = 0.0, t1 = 0.0;
    for (int r = 0; r < pl.rows; ++r) {
      t0 += sm[((long)r * CB + c) * 2 + 0];
      t1 += sm[((long)r * CB + c) * 2 + 1];
    }
    double* dst = partial + (((long)n * pl.chunks + chunk) * C + cbase + c) * 2;
    dst[0] = t0;
    dst[1] = t1;
  }
}

// Second stage of every column reduction: sums the per-chunk partials of 32 channels with 8
// chunk lanes per channel (fixed order: lane-strided sums, then lanes 0..7), then finishes.
//   FIN_STATS: mean / rstd / unbiased var per (n, c)
//   FIN_BWD:   s12[n][c] = (sum g1, sum g1*xhat)
//   FIN_SUM:   out[c] (+)= sum
constexpr int FIN_STATS = 0, FIN_BWD = 1, FIN_SUM = 2;

template <int FIN>
__global__ __launch_bounds__(256) void finalize_kernel(const double* __restrict__ partial, float* __restrict__ o0,
                                                       float* __restrict__ o1, float* __restrict__ o2, long HW, int C,
                                                       int chunks, float eps, int accumulate) {
  __shared__ double sm[8][32][2];
  const int n = blockIdx.y;
  const int cl = threadIdx.x & 31, ln = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0, ss = 0.0;
  if (c < C) {
    // eight chunks' loads in flight per lane, added in chunk order (the bias-gradient sums of the discriminator walk
    // 1024 chunks with 8 workgroups: 40 us of load latency when issued one by one)
    int k = ln;
    for (; k + 56 < chunks; k += 64) {
      double t0[8], t1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double* src = partial + (((long)n * chunks + k + 8 * u) * C + c) * 2;
        t0[u] = src[0];
        t1[u] = FIN != FIN_SUM ? src[1] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s += t0[u];
        if (FIN != FIN_SUM) ss += t1[u];
      }
    }
    for (; k < chunks; k += 8) {
      const double* src = partial + (((long)n * chunks + k) * C + c) * 2;
      s += src[0];
      if (FIN != FIN_SUM) ss += src[1];
    }
  }
  sm[ln][cl][0] = s;
  sm[ln][cl][1] = ss;
  __syncthreads();
  if (ln != 0 || c >= C) return;
  s = 0.0;
  ss = 0.0;
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    s += sm[l][cl][0];
    ss += sm[l][cl][1];
  }
  const long idx = (long)n * C + c;
  if (FIN == FIN_STATS) {
    const double m = s / (double)HW;
    double var = ss / (double)HW - m * m;
    if (var < 0.0) var = 0.0;
    o0[idx] = (float)m;
    o1[idx] = (float)(1.0 / sqrt(var + (double)eps));
    if (o2) o2[idx] = (float)(HW > 1 ? var * ((double)HW / (double)(HW - 1)) : var);
  } else if (FIN == FIN_BWD) {
    o0[idx * 2 + 0] = (float)s;
    o0[idx * 2 + 1] = (float)ss;
  } else {
    o0[c] = accumulate ? o0[c] + (float)s : (float)s;
  }
}

// max |v| of a block's values into the block's slot of amax (FCD_AMAX_SLOTS floats; the value is the maximum of the
// slots).  No atomics on the usual path: a launch of at most FCD_AMAX_SLOTS blocks gives every block a slot of its own,
// updated by one thread with a plain read-max-write (launches on a stream are ordered).  Device-scope atomics on this
// part go to the memory side and serialise per cache line: one per wave cost 300 us per launch, spread over 64 words
// of two lines still 190 us.  Larger grids fall back to an atomic per block.  Every thread of the block must call it.
// FCD_PREC_F16X2 scales its operands by the result.
__device__ __forceinline__ void block_absmax(float m, unsigned* __restrict__ amax, unsigned block, unsigned blocks) {
  __shared__ float wave_max[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
    const unsigned b = __float_as_uint(m);          // non-negative floats order like their bit patterns
    unsigned* slot = amax + (block & (FCD_AMAX_SLOTS - 1));
    if (blocks <= FCD_AMAX_SLOTS) {
      if (b > *slot) *slot = b;
    } else if (b > __builtin_nontemporal_load(slot)) {
      atomicMax(slot, b);
    }
  }
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n4, unsigned* __restrict__ amax) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  block_absmax(m, amax, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void in_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ res,
                                                       float* __restrict__ y, long HW, int C, long total4, float act_slope,
                                                       unsigned* __restrict__ amax) {
  const int c4 = C / 4;
  float am = 0.f;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
    const int cc = (int)(idx % c4) * 4;
    const long n = idx / ((long)HW * c4);
    const float4 zv = *reinterpret_cast<const float4*>(z + idx * 4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + n * C + cc);
    const float4 rs = *reinterpret_cast<const float4*>(rstd + n * C + cc);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + cc);
    const float4 be = *reinterpret_cast<const float4*>(beta + cc);
    float4 o;
    o.x = in_forward_value(zv.x, mu.x, rs.x, ga.x, be.x);
    o.y = in_forward_value(zv.y, mu.y, rs.y, ga.y, be.y);
    o.z = in_forward_value(zv.z, mu.z, rs.z, ga.z, be.z);
    o.w = in_forward_value(zv.w, mu.w, rs.w, ga.w, be.w);
    o.x = act_value(o.x, act_slope);
    o.y = act_value(o.y, act_slope);
    o.z = act_value(o.z, act_slope);
    o.w = act_value(o.w, act_slope);
    if (res) {
      const float4 r = *reinterpret_cast<const float4*>(res + idx * 4);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    *reinterpret_cast<float4*>(y + idx * 4) = o;
    am = fmaxf(fmaxf(fmaxf(am, fabsf(o.x)), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)));
  }
  if (amax) block_absmax(am, amax, blockIdx.x, gridDim.x);
}

// One block = one pixel chunk of one image, all channels: a thread keeps the per-(n, c) constants
// of its 4 channels in registers and walks the chunk's pixels, so the streams are gy, z (and y only
// when the mask is not recomputed) in, dz out.
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                           const float* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ s12, float* __restrict__ dz, long HW,
                                                           int C, ColPlan pl, float act_slope, int use_running,
                                                           const float* __restrict__ beta, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int N, int accumulate,
                                                           unsigned* __restrict__ amax) {
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int c4b = pl.c4b;
  const int cbase = blockIdx.z * 1024;
  const int tx = threadIdx.x % c4b, ty = threadIdx.x / c4b;
  if ((dgamma || dbeta) && chunk == pl.chunks - 1 && n == 0) {
    // The affine gradients (sums of s12 over the images, in double, in image order) ride in one block per channel
    // block instead of a launch of their own between the finalize and this kernel: one thread per channel, the
    // loads of 16 images in flight at a time.
    for (int cl = threadIdx.x; cl < c4b * 4; cl += 256) {
      const int c = cbase + cl;
      if (c >= C) break;
      double a = 0.0, b = 0.0;
      for (int i0 = 0; i0 < N; i0 += 16) {
        float2 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (i0 + u < N) v[u] = *reinterpret_cast<const float2*>(s12 + ((long)(i0 + u) * C + c) * 2);
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (i0 + u < N) {
            b += (double)v[u].x;
            a += (double)v[u].y;
          }
      }
      if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)a : (float)a;
      if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)b : (float)b;
    }
  }
  const bool active = ty < pl.rows && cbase / 4 + tx < pl.c4;
  float am = 0.f;
  if (active) {
  const float inv_hw = 1.f / (float)HW;
  const bool need_mask = act_slope != 1.f;
  const bool recompute = need_mask && beta != nullptr;
  const int c0 = cbase + tx * 4;
  float mu[4], rs[4], ga[4], be[4], k0[4], k1[4], k2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long nc = (long)n * C + c0 + e;
    mu[e] = mean[nc];
    rs[e] = rstd[nc];
    ga[e] = gamma[c0 + e];
    be[e] = recompute ? beta[c0 + e] : 0.f;
    k0[e] = ga[e] * rs[e];
    k1[e] = use_running ? 0.f : s12[nc * 2] * inv_hw;
    k2[e] = use_running ? 0.f : s12[nc * 2 + 1] * inv_hw;
  }
  const long p0 = (long)chunk * pl.chunk_len;
  const long p1 = min(HW, p0 + pl.chunk_len);
  for (long px = p0 + ty; px < p1; px += pl.rows) {
    const long off = ((long)n * HW + px) * C + c0;
    const float4 g = *reinterpret_cast<const float4*>(gy + off);
    const float4 zv = *reinterpret_cast<const float4*>(z + off);
    const float gv[4] = {g.x, g.y, g.z, g.w}, zz[4] = {zv.x, zv.y, zv.z, zv.w};
    float yy[4] = {1.f, 1.f, 1.f, 1.f};
    if (need_mask && !recompute) {
      const float4 yv = *reinterpret_cast<const float4*>(y + off);
      yy[0] = yv.x; yy[1] = yv.y; yy[2] = yv.z; yy[3] = yv.w;
    }
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (recompute) yy[e] = in_forward_value(zz[e], mu[e], rs[e], ga[e], be[e]);
      const float g1 = gv[e] * act_slope_at(yy[e], act_slope);
      if (use_running) {
        o[e] = k0[e] * g1;
      } else {
        const float xh = (zz[e] - mu[e]) * rs[e];
        o[e] = k0[e] * (g1 - k1[e] - xh * k2[e]);
      }
    }
    *reinterpret_cast<float4*>(dz + off) = make_float4(o[0], o[1], o[2], o[3]);
    am = fmaxf(fmaxf(fmaxf(am, fabsf(o[0])), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
  }
  }
  if (amax)
    block_absmax(am, amax, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
}

// InstanceNorm backward in ONE pass for 32 x 32 images (the residual blocks: 34 of the 47 norm backwards of a step).
// The three-kernel form above reads gy and z twice (sums, then dz) with a finalize launch in between: 167 MB and three
// launches per 33.5 MB tensor.  Here a workgroup owns (image, 32 channels) = 1024 pixels x 128 bytes of each operand and
// keeps both slices in registers (1024 threads, 16 float4 each, all loads in flight at once): sums through shuffles and LDS in
// a fixed order (doubles), then dz straight from the registers -- 100 MB, one launch (plus a tiny one for the affine
// gradients, which need the sums of all images).  Needs the mask recomputed from z (beta given) or no mask at all, and a
// (Leaky)ReLU slope: the SiLU / Mish derivatives keep the three-kernel form.
constexpr int IN1P_HW = 1024, IN1P_J = 8, IN1P_WAVES = 16;      // 128 pixel lanes x 8 channel quads
__global__ __launch_bounds__(64 * IN1P_WAVES) void in_bwd_onepass_kernel(const float* __restrict__ gy, const float* __restrict__ z,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ dz, float* __restrict__ s12, int C,
                                                             float act_slope, int use_running, unsigned* __restrict__ amax) {
  __shared__ double wsum[IN1P_WAVES][32][2];       // per wave: (sum g1, sum g1 * xhat) of the workgroup's 32 channels
  __shared__ float kk[32][2];             // the finished sums / HW
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y, cq = tid & 7, pl = tid >> 3;
  const int c0 = blockIdx.x * 32 + 4 * cq;
  const bool recompute = act_slope != 1.f;
  float mu[4], rs[4], ga[4], be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long nc = (long)n * C + c0 + e;
    mu[e] = mean[nc];
    rs[e] = rstd[nc];
    ga[e] = gamma[c0 + e];
    be[e] = recompute ? beta[c0 + e] : 0.f;
  }
  // buffer addressing: one descriptor per operand on this image (1024 x C floats), one 32-bit lane offset, the 16
  // pixel groups as scalar offsets -- 64-bit per-load addresses would cost 64 registers and spill
  const size_t img = (size_t)n * IN1P_HW * C * 4;
  const int img_bytes = IN1P_HW * C * 4;
  __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((char*)gy + img, (short)0, img_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((char*)z + img, (short)0, img_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((char*)dz + img, (short)0, img_bytes, 0x00020000);
  const int voff = (pl * C + c0) * 4, jstep = 128 * C * 4;
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  float4 g[IN1P_J], zz[IN1P_J];
#pragma unroll
  for (int j = 0; j < IN1P_J; ++j) g[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rg, voff, j * jstep, 0));
#pragma unroll
  for (int j = 0; j < IN1P_J; ++j) zz[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rz, voff, j * jstep, 0));
  // g <- g1 = gy * act'(.), zz <- xhat; per-thread sums over its 8 pixels
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < IN1P_J; ++j) {
    float gv[4] = {g[j].x, g[j].y, g[j].z, g[j].w}, zv[4] = {zz[j].x, zz[j].y, zz[j].z, zz[j].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float yv = recompute ? in_forward_value(zv[e], mu[e], rs[e], ga[e], be[e]) : 1.f;
      gv[e] *= yv > 0.f ? 1.f : act_slope;          // slopes in [0, 1] only (the host routes SiLU / Mish elsewhere)
      zv[e] = (zv[e] - mu[e]) * rs[e];
      s1[e] += gv[e];
      s2[e] += gv[e] * zv[e];
    }
    g[j] = make_float4(gv[0], gv[1], gv[2], gv[3]);
    zz[j] = make_float4(zv[0], zv[1], zv[2], zv[3]);
  }
  // lanes l, l + 8, ..., l + 56 of a wave share a channel quad: xor tree over lane bits 3..5, then the 16 waves in order
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    double d1 = (double)s1[e], d2 = (double)s2[e];
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      d1 += __shfl_xor(d1, o);
      d2 += __shfl_xor(d2, o);
    }
    if (lane < 8) {
      wsum[wave][4 * cq + e][0] = d1;
      wsum[wave][4 * cq + e][1] = d2;
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int c = tid & 31, w = tid >> 5;
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < IN1P_WAVES; ++k) t += wsum[k][c][w];
    s12[((long)n * C + blockIdx.x * 32 + c) * 2 + w] = (float)t;
    kk[c][w] = use_running ? 0.f : (float)t * (1.f / (float)IN1P_HW);
  }
  __syncthreads();
  float k0[4], k1[4], k2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    k0[e] = ga[e] * rs[e];
    // (the three-kernel form rounds the sum to fp32 before the division by HW: same here)
    k1[e] = kk[4 * cq + e][0];
    k2[e] = kk[4 * cq + e][1];
  }
  float am = 0.f;
#pragma unroll
  for (int j = 0; j < IN1P_J; ++j) {
    const float gv[4] = {g[j].x, g[j].y, g[j].z, g[j].w}, xh[4] = {zz[j].x, zz[j].y, zz[j].z, zz[j].w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = k0[e] * (gv[e] - k1[e] - xh[e] * k2[e]);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, make_float4(o[0], o[1], o[2], o[3])), rd, voff, j * jstep, 0);
    am = fmaxf(fmaxf(fmaxf(am, fabsf(o[0])), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
  }
  if (amax) {
    // (block_absmax is written for 4 waves: the same scheme for 16)
    __shared__ float wmax[IN1P_WAVES];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    if (lane == 0) wmax[wave] = am;
    __syncthreads();
    if (tid == 0) {
      float m = 0.f;
#pragma unroll
      for (int k = 0; k < IN1P_WAVES; ++k) m = fmaxf(m, wmax[k]);
      const unsigned b = __float_as_uint(m), blocks = gridDim.x * gridDim.y;
      unsigned* slot = amax + ((blockIdx.y * gridDim.x + blockIdx.x) & (FCD_AMAX_SLOTS - 1));
      if (blocks <= FCD_AMAX_SLOTS) {
        if (b > *slot) *slot = b;
      } else if (b > __builtin_nontemporal_load(slot)) {
        atomicMax(slot, b);
      }
    }
  }
}

// InstanceNorm forward behind a convolution whose epilogue left the column partials (fcd_conv_gather_stats_apply), for
// 32 x 32 images: a workgroup owns (image, 32 channels), issues the loads of its z (and residual) slice, finishes the
// statistics of its 32 channels from the partials while they fly (same sums, in the same order, as finalize_kernel: the
// results are bit-identical to the two-kernel form), then normalises from registers.  Saves the finalize launch between
// the convolution and the apply (4.8 us on the critical path, 39 times a step).  (Leaky)ReLU slopes only.
constexpr int IN1F_MAX_CHUNKS = 32;
// PARTS: workgroups per (image, 32 channels), each with 1024 / PARTS pixels (blockIdx.z; the statistics are finished by
// every part, written by part 0) -- two when one per slice would leave CUs idle (16-sample launches: 128 slices)
template <int PARTS>
__global__ __launch_bounds__(64 * IN1P_WAVES) void in_fwd_onepass_kernel(const float* __restrict__ z, const double* __restrict__ partial,
                                                             int chunks, float eps, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ res,
                                                             float* __restrict__ y, float* __restrict__ mean,
                                                             float* __restrict__ rstd, float* __restrict__ varu, int C,
                                                             float act_slope, unsigned* __restrict__ amax) {
  __shared__ double psum[IN1F_MAX_CHUNKS][32][2];
  __shared__ float st[32][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y, cq = tid & 7, pl = tid >> 3;
  const int c0 = blockIdx.x * 32 + 4 * cq;
  const size_t img = (size_t)n * IN1P_HW * C * 4;
  const int img_bytes = IN1P_HW * C * 4;
  __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((char*)z + img, (short)0, img_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((char*)(res ? res : z) + img, (short)0, img_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((char*)y + img, (short)0, img_bytes, 0x00020000);
  constexpr int NJ = IN1P_J / PARTS;
  const int jstep = 128 * C * 4, voff = (pl * C + c0) * 4 + (int)blockIdx.z * NJ * jstep;
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  float4 zz[NJ], rv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) zz[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rz, voff, j * jstep, 0));
  if (res) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) rv[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, voff, j * jstep, 0));
  }
  {
    const int c = tid & 31, k = tid >> 5;
    if (k < chunks) {
      const double* src = partial + (((long)n * chunks + k) * C + blockIdx.x * 32 + c) * 2;
      psum[k][c][0] = src[0];
      psum[k][c][1] = src[1];
    }
  }
  __syncthreads();
  if (tid < 32) {
    // finalize_kernel's order: eight lane-strided sums, then the lanes in order
    double s = 0.0, ss = 0.0;
    for (int l = 0; l < 8; ++l) {
      double a = 0.0, b = 0.0;
      for (int k = l; k < chunks; k += 8) {
        a += psum[k][tid][0];
        b += psum[k][tid][1];
      }
      s += a;
      ss += b;
    }
    const double m = s / (double)IN1P_HW;
    double var = ss / (double)IN1P_HW - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
    const long idx = (long)n * C + blockIdx.x * 32 + tid;
    if (blockIdx.z == 0) {
      mean[idx] = mf;
      rstd[idx] = rf;
      if (varu) varu[idx] = (float)(var * ((double)IN1P_HW / (double)(IN1P_HW - 1)));
    }
    st[tid][0] = mf;
    st[tid][1] = rf;
  }
  __syncthreads();
  float mu[4], rs[4], ga[4], be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mu[e] = st[4 * cq + e][0];
    rs[e] = st[4 * cq + e][1];
    ga[e] = gamma[c0 + e];
    be[e] = beta[c0 + e];
  }
  float am = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float zv[4] = {zz[j].x, zz[j].y, zz[j].z, zz[j].w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = in_forward_value(zv[e], mu[e], rs[e], ga[e], be[e]);
      o[e] = v > 0.f ? v : v * act_slope;
    }
    if (res) { o[0] += rv[j].x; o[1] += rv[j].y; o[2] += rv[j].z; o[3] += rv[j].w; }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, make_float4(o[0], o[1], o[2], o[3])), ry, voff, j * jstep, 0);
    am = fmaxf(fmaxf(fmaxf(am, fabsf(o[0])), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
  }
  if (amax) {
    __shared__ float wmax[IN1P_WAVES];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    if (lane == 0) wmax[wave] = am;
    __syncthreads();
    if (tid == 0) {
      float m = 0.f;
#pragma unroll
      for (int k = 0; k < IN1P_WAVES; ++k) m = fmaxf(m, wmax[k]);
      const unsigned b = __float_as_uint(m), blocks = gridDim.x * gridDim.y * gridDim.z;
      unsigned* slot = amax + (((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) & (FCD_AMAX_SLOTS - 1));
      if (blocks <= FCD_AMAX_SLOTS) {
        if (b > *slot) *slot = b;
      } else if (b > __builtin_nontemporal_load(slot)) {
        atomicMax(slot, b);
      }
    }
  }
}

// dgamma / dbeta from s12[n][c] = (sum g1, sum g1 * xhat): sums over the images in double, in image order
__global__ __launch_bounds__(256) void affine_from_s12_kernel(const float* __restrict__ s12, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int N, int C, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int i0 = 0; i0 < N; i0 += 16) {
    float2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (i0 + u < N) v[u] = *reinterpret_cast<const float2*>(s12 + ((long)(i0 + u) * C + c) * 2);
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (i0 + u < N) {
        b += (double)v[u].x;
        a += (double)v[u].y;
      }
  }
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)a : (float)a;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)b : (float)b;
}

// affine_from_s12_kernel for up to FCD_AFFINE_BATCH_MAX norms of one backward pass in one launch (blockIdx.y = norm):
// same sums in the same order
struct AffineBatch {
  const float* s12[FCD_AFFINE_BATCH_MAX];
  float* dgamma[FCD_AFFINE_BATCH_MAX];
  float* dbeta[FCD_AFFINE_BATCH_MAX];
};
__global__ __launch_bounds__(256) void affine_batch_kernel(const AffineBatch bt, int N, int C, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float* __restrict__ s12 = bt.s12[blockIdx.y];
  float* __restrict__ dgamma = bt.dgamma[blockIdx.y];
  float* __restrict__ dbeta = bt.dbeta[blockIdx.y];
  double a = 0.0, b = 0.0;
  for (int i0 = 0; i0 < N; i0 += 16) {
    float2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (i0 + u < N) v[u] = *reinterpret_cast<const float2*>(s12 + ((long)(i0 + u) * C + c) * 2);
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (i0 + u < N) {
        b += (double)v[u].x;
        a += (double)v[u].y;
      }
  }
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)a : (float)a;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)b : (float)b;
}

__global__ __launch_bounds__(256) void stats_from_running_kernel(const float* __restrict__ rm, const float* __restrict__ rv,
                                                                 float* __restrict__ mean, float* __restrict__ rstd, int N,
                                                                 int C, float eps) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * C) return;
  const int c = (int)(idx % C);
  mean[idx] = rm[c];
  rstd[idx] = 1.f / sqrtf(rv[c] + eps);
}

__global__ __launch_bounds__(256) void update_running_kernel(float* __restrict__ rm, float* __restrict__ rv,
                                                             const float* __restrict__ mean, const float* __restrict__ varu,
                                                             int N, int C, int groups, float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int gn = N / groups;
  float m = rm[c], v = rv[c];
  for (int g = 0; g < groups; ++g) {       // one update per reference call, in call order
    float sm = 0.f, sv = 0.f;
    for (int n = g * gn; n < (g + 1) * gn; ++n) {
      sm += mean[(long)n * C + c];
      sv += varu[(long)n * C + c];
    }
    m = (1.f - momentum) * m + momentum * (sm / (float)gn);
    v = (1.f - momentum) * v + momentum * (sv / (float)gn);
  }
  rm[c] = m;
  rv[c] = v;
}

int check_nhwc(const char* what, int N, long HW, int C) {
  FCD_REQUIRE(N > 0 && HW > 0 && C > 0 && C % 4 == 0, "%s: N=%d HW=%ld C=%d (C must be a multiple of 4)", what, N, HW, C);
  return FCD_OK;
}

size_t partial_bytes(int N, long HW, int C) {
  ColPlan pl = col_plan(N, HW, C);
  return (size_t)N * pl.chunks * C * 2 * sizeof(double);
}

}  // namespace

int fcd_finalize_stats(const double* partial, float* mean, float* rstd, float* var_unbiased, int N, long HW, int C,
                       int chunks, float eps, hipStream_t st) {
  hipLaunchKernelGGL((finalize_kernel<FIN_STATS>), dim3(fcd_cdiv(C, 32), N), dim3(256), 0, st, partial, mean, rstd,
                     var_unbiased, HW, C, chunks, eps, 0);
  return fcd_check_launch("instnorm_stats/finalize");
}

// Statistics from the column partials a convolution's epilogue left, then normalise + affine + activation (+ residual):
// one launch where in_fwd_onepass_kernel applies, finalize + in_apply otherwise.  Arguments already validated by the caller
// (fcd_conv_gather_stats_apply).
int fcd_finalize_apply(const double* partial, const float* z, float* mean, float* rstd, float* var_unbiased, int N, long HW,
                       int C, int chunks, float eps, const float* gamma, const float* beta, const float* residual, float* y,
                       float act_slope, float* y_amax, hipStream_t st) {
  if (HW == IN1P_HW && C % 32 == 0 && chunks <= IN1F_MAX_CHUNKS && act_slope <= 1.f && (long)N * (C / 32) >= 64 &&
      N <= 65535 && !getenv("FCD_NO_IN1F")) {
    if ((long)N * (C / 32) < 256)
      hipLaunchKernelGGL(in_fwd_onepass_kernel<2>, dim3(C / 32, N, 2), dim3(64 * IN1P_WAVES), 0, st, z, partial, chunks, eps, gamma,
                         beta, residual, y, mean, rstd, var_unbiased, C, act_slope, (unsigned*)y_amax);
    else
      hipLaunchKernelGGL(in_fwd_onepass_kernel<1>, dim3(C / 32, N), dim3(64 * IN1P_WAVES), 0, st, z, partial, chunks, eps, gamma,
                         beta, residual, y, mean, rstd, var_unbiased, C, act_slope, (unsigned*)y_amax);
    return fcd_check_launch("instnorm_fwd/onepass");
  }
  int rc = fcd_finalize_stats(partial, mean, rstd, var_unbiased, N, HW, C, chunks, eps, st);
  if (rc) return rc;
  return fcd_instnorm_apply_amax(z, mean, rstd, gamma, beta, residual, y, N, (int)HW, C, act_slope, y_amax, st);
}

extern "C" size_t fcd_instnorm_workspace_bytes(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0 || C % 4) return 0;
  return partial_bytes(N, HW, C) + (size_t)N * C * 2 * sizeof(float) + 64;
}

namespace {
// Both statistics entry points: column partials into `partial`, then the finalize kernel.  Every
// partial[n][chunk][c] entry the second kernel reads is written by the first, so the buffer needs no
// clearing between calls.
int stats_launch(const float* z, float* mean, float* rstd, float* var_unbiased, int N, int HW, int C, float eps,
                 double* partial, hipStream_t st) {
  ColPlan pl = col_plan(N, HW, C);
  const size_t shm = (size_t)pl.rows * pl.c4b * 4 * 2 * sizeof(double);
  hipLaunchKernelGGL((col_partial_kernel<MODE_STATS>), dim3(pl.chunks, N, pl.cblocks), dim3(256), shm, st, z, nullptr, nullptr,
                     nullptr, nullptr, partial, (long)HW, C, pl, 0.f);
  int rc = fcd_check_launch("instnorm_stats/partial");
  if (rc) return rc;
  hipLaunchKernelGGL((finalize_kernel<FIN_STATS>), dim3(fcd_cdiv(C, 32), N), dim3(256), 0, st, (const double*)partial, mean,
                     rstd, var_unbiased, (long)HW, C, pl.chunks, eps, 0);
  return fcd_check_launch("instnorm_stats/finalize");
}
}  // namespace

extern "C" int fcd_instnorm_stats(const float* z, float* mean, float* rstd, float* var_unbiased, int N, int HW,
                                  int C, float eps, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_nhwc("instnorm_stats", N, HW, C);
  if (rc) return rc;
  FCD_REQUIRE(z && mean && rstd, "instnorm_stats: null tensor");
  if (!workspace || workspace_bytes < partial_bytes(N, HW, C)) {
    fcd_set_error("instnorm_stats: workspace too small");
    return FCD_EWORKSPACE;
  }
  return stats_launch(z, mean, rstd, var_unbiased, N, HW, C, eps, (double*)workspace, fcd_stream(stream));
}

// ------------------------------------------------------------------------------------------------
// fcd_instnorm_stats_noalloc: the same two kernels with the column partials in a file-scope device
// array instead of a caller-provided workspace, for callers that cannot allocate (HIP graph capture,
// serving loops).  `scratch` is a host-side stand-in for the symbol, not a device address: host code
// uses the per-device address hipGetSymbolAddress returned (each GPU holds its own copy).
// ------------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[FCD_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[MAX_DEVICES] = {};
}  // namespace

// With `device` current: resolve this device's copy of `scratch` once.  Called by fcd_init, and by the
// entry point's first call on a device that fcd_init was not called for.
int fcd_noalloc_scratch_init(int device) {
  if (device < 0 || device >= MAX_DEVICES) {
    fcd_set_error("noalloc scratch: device %d out of range", device);
    return FCD_EINVAL;
  }
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (g_scratch_ptr[device]) return FCD_OK;
  void* p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !p) {
    fcd_set_error("noalloc scratch: hipGetSymbolAddress failed on device %d (%s)", device, hipGetErrorString(e));
    return e != hipSuccess ? (int)e : FCD_ENODEVICE;
  }
  g_scratch_ptr[device] = p;
  return FCD_OK;
}

extern "C" int fcd_instnorm_stats_noalloc(const float* z, float* mean, float* rstd, float* var_unbiased, int N, int HW,
                                          int C, float eps, void* stream) {
  int rc = check_nhwc("instnorm_stats_noalloc", N, HW, C);
  if (rc) return rc;
  FCD_REQUIRE(z && mean && rstd, "instnorm_stats_noalloc: null tensor");
  const size_t need = partial_bytes(N, HW, C);
  if (need > (size_t)FCD_NOALLOC_SCRATCH_BYTES) {
    fcd_set_error("instnorm_stats_noalloc: N=%d HW=%d C=%d needs %zu bytes of column partials, the built-in scratch "
                  "holds %zu; use fcd_instnorm_stats with a workspace", N, HW, C, need, (size_t)FCD_NOALLOC_SCRATCH_BYTES);
    return FCD_EWORKSPACE;
  }
  int device = -1;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess || device < 0 || device >= MAX_DEVICES) {
    fcd_set_error("instnorm_stats_noalloc: no current device (%s)", hipGetErrorString(e));
    return FCD_ENODEVICE;
  }
  void* p;
  {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    p = g_scratch_ptr[device];
  }
  if (!p) {
    rc = fcd_noalloc_scratch_init(device);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    p = g_scratch_ptr[device];
  }
  return stats_launch(z, mean, rstd, var_unbiased, N, HW, C, eps, (double*)p, fcd_stream(stream));
}

extern "C" int fcd_instnorm_stats_from_running(const float* running_mean, const float* running_var, float* mean,
                                               float* rstd, int N, int C, float eps, void* stream) {
  FCD_REQUIRE(running_mean && running_var && mean && rstd && N > 0 && C > 0, "stats_from_running: bad arguments");
  hipLaunchKernelGGL(stats_from_running_kernel, dim3(fcd_cdiv((long)N * C, 256)), dim3(256), 0, fcd_stream(stream),
                     running_mean, running_var, mean, rstd, N, C, eps);
  return fcd_check_launch("stats_from_running");
}

namespace {
struct RunningBatch {
  float* rm[FCD_RUNNING_BATCH_MAX];
  float* rv[FCD_RUNNING_BATCH_MAX];
  const float* mean[FCD_RUNNING_BATCH_MAX];
  const float* varu[FCD_RUNNING_BATCH_MAX];
};
// blockIdx.y = norm; same arithmetic, in the same order, as update_running_kernel
__global__ __launch_bounds__(256) void update_running_batch_kernel(const RunningBatch b, int N, int C, int groups,
                                                                   float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float* __restrict__ rm = b.rm[blockIdx.y];
  float* __restrict__ rv = b.rv[blockIdx.y];
  const float* __restrict__ mean = b.mean[blockIdx.y];
  const float* __restrict__ varu = b.varu[blockIdx.y];
  const int gn = N / groups;
  float m = rm[c], v = rv[c];
  for (int g = 0; g < groups; ++g) {
    float sm = 0.f, sv = 0.f;
    for (int n = g * gn; n < (g + 1) * gn; ++n) {
      sm += mean[(long)n * C + c];
      sv += varu[(long)n * C + c];
    }
    m = (1.f - momentum) * m + momentum * (sm / (float)gn);
    v = (1.f - momentum) * v + momentum * (sv / (float)gn);
  }
  rm[c] = m;
  rv[c] = v;
}
}  // namespace

extern "C" int fcd_instnorm_update_running_batch(float* const* running_mean, float* const* running_var,
                                                 const float* const* mean, const float* const* var_unbiased,
                                                 int count, int N, int C, int groups, float momentum, void* stream) {
  FCD_REQUIRE(running_mean && running_var && mean && var_unbiased, "update_running_batch: null array");
  FCD_REQUIRE(count >= 1 && count <= FCD_RUNNING_BATCH_MAX, "update_running_batch: count %d not in 1..%d", count,
              FCD_RUNNING_BATCH_MAX);
  FCD_REQUIRE(N > 0 && C > 0 && groups >= 1 && N % groups == 0, "update_running_batch: bad N / C / groups");
  RunningBatch b;
  for (int i = 0; i < FCD_RUNNING_BATCH_MAX; ++i) {
    const int j = i < count ? i : 0;
    FCD_REQUIRE(running_mean[j] && running_var[j] && mean[j] && var_unbiased[j], "update_running_batch: null tensor");
    b.rm[i] = running_mean[j]; b.rv[i] = running_var[j]; b.mean[i] = mean[j]; b.varu[i] = var_unbiased[j];
  }
  hipLaunchKernelGGL(update_running_batch_kernel, dim3(fcd_cdiv(C, 256), count), dim3(256), 0, fcd_stream(stream), b, N, C,
                     groups, momentum);
  return fcd_check_launch("update_running_batch");
}

extern "C" int fcd_instnorm_update_running(float* running_mean, float* running_var, const float* mean,
                                           const float* var_unbiased, int N, int C, int groups, float momentum,
                                           void* stream) {
  FCD_REQUIRE(running_mean && running_var && mean && var_unbiased && N > 0 && C > 0, "update_running: bad arguments");
  FCD_REQUIRE(groups >= 1 && N % groups == 0, "update_running: N=%d is not a multiple of groups=%d", N, groups);
  hipLaunchKernelGGL(update_running_kernel, dim3(fcd_cdiv(C, 256)), dim3(256), 0, fcd_stream(stream), running_mean,
                     running_var, mean, var_unbiased, N, C, groups, momentum);
  return fcd_check_launch("update_running");
}

namespace {
// max |x| of up to FCD_ABSMAX_BATCH_MAX tensors in one launch (blockIdx.y = tensor), OVERWRITING the tensor's row of
// FCD_AMAX_SLOTS partial maxima: block b stores slot b and clears slots b + 256, b + 512, b + 768 (no zeroing pass)
struct AbsmaxBatch {
  const float* x[FCD_ABSMAX_BATCH_MAX];
  size_t n4[FCD_ABSMAX_BATCH_MAX];
};
__global__ __launch_bounds__(256) void absmax_batch_kernel(const AbsmaxBatch bt, float* __restrict__ rows) {
  __shared__ float wave_max[4];
  const float* __restrict__ x = bt.x[blockIdx.y];
  const size_t n4 = bt.n4[blockIdx.y];
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  float* row = rows + (size_t)blockIdx.y * FCD_AMAX_SLOTS;
  if (threadIdx.x == 0) row[blockIdx.x] = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
  else if (threadIdx.x < FCD_AMAX_SLOTS / 256) row[blockIdx.x + 256 * threadIdx.x] = 0.f;
}
}  // namespace

extern "C" int fcd_absmax_batch(const float* const* x, const size_t* n, int count, float* amax_rows, void* stream) {
  FCD_REQUIRE(x && n && amax_rows, "absmax_batch: null array");
  FCD_REQUIRE(count >= 1 && count <= FCD_ABSMAX_BATCH_MAX, "absmax_batch: count %d not in 1..%d", count, FCD_ABSMAX_BATCH_MAX);
  AbsmaxBatch b;
  for (int i = 0; i < FCD_ABSMAX_BATCH_MAX; ++i) {
    const int j = i < count ? i : 0;
    FCD_REQUIRE(x[j] && n[j] % 4 == 0 && ((uintptr_t)x[j] % 16) == 0,
                "absmax_batch: null tensor, or a length / address that is not a multiple of 4 floats");
    b.x[i] = x[j]; b.n4[i] = n[j] / 4;
  }
  static_assert(FCD_AMAX_SLOTS == 1024, "absmax_batch_kernel clears slots in four strides of 256");
  hipLaunchKernelGGL(absmax_batch_kernel, dim3(256, count), dim3(256), 0, fcd_stream(stream), b, amax_rows);
  return fcd_check_launch("absmax_batch");
}

extern "C" int fcd_absmax(const float* x, size_t n, float* amax, void* stream) {
  FCD_REQUIRE(x && amax && n % 4 == 0 && ((uintptr_t)x % 16) == 0, "absmax: null tensor, or a length / address that is not a multiple of 4 floats");
  const int blocks = (int)std::min<size_t>((n / 4 + 255) / 256, FCD_AMAX_SLOTS);
  if (n) hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, fcd_stream(stream), x, n / 4, (unsigned*)amax);
  return fcd_check_launch("absmax");
}

extern "C" int fcd_instnorm_apply(const float* z, const float* mean, const float* rstd, const float* gamma,
                                  const float* beta, const float* residual, float* y, int N, int HW, int C,
                                  float act_slope, void* stream) {
  return fcd_instnorm_apply_amax(z, mean, rstd, gamma, beta, residual, y, N, HW, C, act_slope, nullptr, stream);
}

extern "C" int fcd_instnorm_apply_amax(const float* z, const float* mean, const float* rstd, const float* gamma,
                                       const float* beta, const float* residual, float* y, int N, int HW, int C,
                                       float act_slope, float* y_amax, void* stream) {
  int rc = check_nhwc("instnorm_apply", N, HW, C);
  if (rc) return rc;
  FCD_REQUIRE(z && mean && rstd && gamma && beta && y, "instnorm_apply: null tensor");
  FCD_REQUIRE((act_slope >= 0.f && act_slope <= 1.f) || act_slope == FCD_ACT_SILU || act_slope == FCD_ACT_MISH,
              "instnorm_apply: act_slope %g is neither a slope in [0,1] nor an activation code", act_slope);
  const long total4 = (long)N * HW * (C / 4);
  const int blocks = (int)std::min<long>(fcd_cdiv(total4, 256), y_amax ? FCD_AMAX_SLOTS : 8192);   // a slot per block
  hipLaunchKernelGGL(in_apply_kernel, dim3(blocks), dim3(256), 0, fcd_stream(stream), z, mean, rstd, gamma, beta,
                     residual, y, (long)HW, C, total4, act_slope, (unsigned*)y_amax);
  return fcd_check_launch("instnorm_apply");
}

extern "C" int fcd_instnorm_bwd(const float* gy, const float* y, const float* z, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, float* dz, float* dgamma, float* dbeta,
                                int accumulate, int N,
                                int HW, int C, float act_slope, int use_running, void* workspace,
                                size_t workspace_bytes, void* stream) {
  return fcd_instnorm_bwd_amax(gy, y, z, mean, rstd, gamma, beta, dz, dgamma, dbeta, accumulate, N, HW, C, act_slope,
                               use_running, nullptr, workspace, workspace_bytes, stream);
}

namespace {
// s12_out given: the per-image sums go there instead of the workspace and no affine gradient is computed (the caller
// sums them later: fcd_instnorm_affine_batch)
int instnorm_bwd_impl(const float* gy, const float* y, const float* z, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, float* dz, float* dgamma, float* dbeta, int accumulate, int N,
                      int HW, int C, float act_slope, int use_running, float* dz_amax, float* s12_out, void* workspace,
                      size_t workspace_bytes, void* stream) {
  int rc = check_nhwc("instnorm_bwd", N, HW, C);
  if (rc) return rc;
  FCD_REQUIRE(gy && z && mean && rstd && gamma && dz, "instnorm_bwd: null tensor");
  FCD_REQUIRE(y || beta || act_slope == 1.f, "instnorm_bwd: the activation mask needs y or beta");
  FCD_REQUIRE(act_slope <= 1.f || beta, "instnorm_bwd: SiLU / Mish need beta (their slope is recomputed from z)");
  FCD_REQUIRE((act_slope >= 0.f && act_slope <= 1.f) || act_slope == FCD_ACT_SILU || act_slope == FCD_ACT_MISH,
              "instnorm_bwd: act_slope %g is neither a slope in [0,1] nor an activation code", act_slope);
  const size_t pb = partial_bytes(N, HW, C);
  if (!workspace || workspace_bytes < pb + (size_t)N * C * 2 * sizeof(float)) {
    fcd_set_error("instnorm_bwd: workspace too small");
    return FCD_EWORKSPACE;
  }
  ColPlan pl = col_plan(N, HW, C);
  hipStream_t st = fcd_stream(stream);
  double* partial = (double*)workspace;
  float* s12 = s12_out ? s12_out : (float*)((char*)workspace + pb);
  if (HW == IN1P_HW && C % 32 == 0 && act_slope <= 1.f && (beta || act_slope == 1.f) && (long)N * (C / 32) >= 64 &&
      N <= 65535 && !getenv("FCD_NO_IN1P")) {
    // one pass: (image, 32 channels) per workgroup, both operand slices in registers
    hipLaunchKernelGGL(in_bwd_onepass_kernel, dim3(C / 32, N), dim3(64 * IN1P_WAVES), 0, st, gy, z, mean, rstd, gamma, beta, dz, s12, C,
                       act_slope, use_running, (unsigned*)dz_amax);
    rc = fcd_check_launch("instnorm_bwd/onepass");
    if (rc || !(dgamma || dbeta)) return rc;
    hipLaunchKernelGGL(affine_from_s12_kernel, dim3(fcd_cdiv(C, 256)), dim3(256), 0, st, (const float*)s12, dgamma, dbeta, N, C,
                       accumulate);
    return fcd_check_launch("instnorm_bwd/affine");
  }
  const size_t shm = (size_t)pl.rows * pl.c4b * 4 * 2 * sizeof(double);
  hipLaunchKernelGGL((col_partial_kernel<MODE_BWD>), dim3(pl.chunks, N, pl.cblocks), dim3(256), shm, st, gy, y, z, mean, rstd,
                     partial, (long)HW, C, pl, act_slope, gamma, beta);
  rc = fcd_check_launch("instnorm_bwd/partial");
  if (rc) return rc;
  hipLaunchKernelGGL((finalize_kernel<FIN_BWD>), dim3(fcd_cdiv(C, 32), N), dim3(256), 0, st, (const double*)partial, s12,
                     (float*)nullptr, (float*)nullptr, (long)HW, C, pl.chunks, 0.f, 0);
  rc = fcd_check_launch("instnorm_bwd/finalize");
  if (rc) return rc;
  hipLaunchKernelGGL(in_bwd_apply_kernel, dim3(pl.chunks, N, pl.cblocks), dim3(256), 0, st, gy, y, z, mean, rstd, gamma, s12,
                     dz, (long)HW, C, pl, act_slope, use_running, beta, dgamma, dbeta, N, accumulate, (unsigned*)dz_amax);
  return fcd_check_launch("instnorm_bwd/apply");
}
}  // namespace

extern "C" int fcd_instnorm_bwd_amax(const float* gy, const float* y, const float* z, const float* mean, const float* rstd,
                                     const float* gamma, const float* beta, float* dz, float* dgamma, float* dbeta,
                                     int accumulate, int N, int HW, int C, float act_slope, int use_running,
                                     float* dz_amax, void* workspace, size_t workspace_bytes, void* stream) {
  return instnorm_bwd_impl(gy, y, z, mean, rstd, gamma, beta, dz, dgamma, dbeta, accumulate, N, HW, C, act_slope, use_running,
                           dz_amax, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int fcd_instnorm_bwd_s12(const float* gy, const float* y, const float* z, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, float* dz, float* s12, int N, int HW, int C,
                                    float act_slope, int use_running, float* dz_amax, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  FCD_REQUIRE(s12 != nullptr, "instnorm_bwd_s12: null s12");
  return instnorm_bwd_impl(gy, y, z, mean, rstd, gamma, beta, dz, nullptr, nullptr, 0, N, HW, C, act_slope, use_running,
                           dz_amax, s12, workspace, workspace_bytes, stream);
}

extern "C" int fcd_instnorm_affine_batch(const float* const* s12, float* const* dgamma, float* const* dbeta, int count,
                                         int N, int C, int accumulate, void* stream) {
  FCD_REQUIRE(s12 && dgamma && dbeta, "affine_batch: null array");
  FCD_REQUIRE(count >= 1 && count <= FCD_AFFINE_BATCH_MAX, "affine_batch: count %d not in 1..%d", count, FCD_AFFINE_BATCH_MAX);
  FCD_REQUIRE(N > 0 && C > 0, "affine_batch: bad N / C");
  AffineBatch b;
  for (int i = 0; i < FCD_AFFINE_BATCH_MAX; ++i) {
    const int j = i < count ? i : 0;
    FCD_REQUIRE(s12[j] && (dgamma[j] || dbeta[j]), "affine_batch: null tensor");
    b.s12[i] = s12[j]; b.dgamma[i] = dgamma[j]; b.dbeta[i] = dbeta[j];
  }
  hipLaunchKernelGGL(affine_batch_kernel, dim3(fcd_cdiv(C, 256), count), dim3(256), 0, fcd_stream(stream), b, N, C, accumulate);
  return fcd_check_launch("affine_batch");
}

extern "C" size_t fcd_colsum_workspace_bytes(size_t rows, int C) {
  if (rows == 0 || C <= 0 || C % 4) return 0;
  return partial_bytes(1, (long)rows, C) + 64;
}

extern "C" int fcd_colsum(const float* x, float* out, int accumulate, size_t rows, int C, void* workspace,
                          size_t workspace_bytes, void* stream) {
  int rc = check_nhwc("colsum", 1, (long)rows, C);
  if (rc) return rc;
  FCD_REQUIRE(x && out, "colsum: null tensor");
  if (!workspace || workspace_bytes < partial_bytes(1, (long)rows, C)) {
    fcd_set_error("colsum: workspace too small");
    return FCD_EWORKSPACE;
  }
  ColPlan pl = col_plan(1, (long)rows, C);
  hipStream_t st = fcd_stream(stream);
  const size_t shm = (size_t)pl.rows * pl.c4b * 4 * 2 * sizeof(double);
  hipLaunchKernelGGL((col_partial_kernel<MODE_SUM>), dim3(pl.chunks, 1, pl.cblocks), dim3(256), shm, st, x, nullptr, nullptr, nullptr,
                     nullptr, (double*)workspace, (long)rows, C, pl, 0.f);
  rc = fcd_check_launch("colsum/partial");
  if (rc) return rc;
  hipLaunchKernelGGL((finalize_kernel<FIN_SUM>), dim3(fcd_cdiv(C, 32), 1), dim3(256), 0, st, (const double*)workspace, out,
                     (float*)nullptr, (float*)nullptr, (long)rows, C, pl.chunks, 0.f, accumulate);
  return fcd_check_launch("colsum/finalize");
}
