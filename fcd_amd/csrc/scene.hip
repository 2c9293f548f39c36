// Scene path and threshold search on the device:
//   * tile a uint16 scene with numpy "reflect" padding and normalise to [-1, 1] fp32
//     (data_loader.py:74-103, :375; evaluate.py:90-93),
//   * stitch the per-tile difference maps into a float64 raster (evaluate.py:100-122),
//   * zero nodata pixels and binarise in float64 (evaluate.py:49-51, fcd_solver.py:566-567),
//   * min/max and per-threshold confusion counts of difference maps against ground truth
//     (fcd_solver.py:680-725, metrics.py:4-15).
// All HBM-bound byte/integer work: one pass over the data, coalesced along the pixel axis.
#include "fcd_common.h"

#include <cstdint>

namespace {

__global__ __launch_bounds__(256) void scene_tiles_kernel(const uint16_t* __restrict__ scene, int H, int W, int C, int patch,
                                                          const int* __restrict__ origins, int nb, float* __restrict__ x,
                                                          int* __restrict__ any_valid) {
  const float rcp = 1.0f / 32767.5f;            // fp32(1/32767.5), as numpy.reciprocal(float32)
  const long total = (long)nb * patch * patch;
  int seen = 0;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int px = (int)(idx % patch);
    const long t = idx / patch;
    const int py = (int)(t % patch), b = (int)(t / patch);
    int r = origins[2 * b] + py, c = origins[2 * b + 1] + px;
    r = r < H ? r : 2 * (H - 1) - r;            // numpy.pad(..., 'reflect') past the end
    c = c < W ? c : 2 * (W - 1) - c;
    const uint16_t* src = scene + ((long)r * W + c) * C;
    for (int k = 0; k < C; ++k) {
      const uint16_t v = src[k];
      seen |= v != 0;
      float f = (float)v;
      f = f - 32767.5f;
      f = f * rcp;
      x[(((long)b * C + k) * patch + py) * patch + px] = f;
    }
  }
  if (__any(seen) && (threadIdx.x & 63) == 0) atomicOr(any_valid, 1);
}

__global__ __launch_bounds__(256) void scene_stitch_kernel(const float* __restrict__ delta, const int* __restrict__ origins,
                                                           int nb, int patch, double* __restrict__ raster, int H, int W) {
  const long total = (long)nb * patch * patch;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int px = (int)(idx % patch);
    const long t = idx / patch;
    const int py = (int)(t % patch), b = (int)(t / patch);
    const int r = origins[2 * b] + py, c = origins[2 * b + 1] + px;
    if (r < H && c < W) raster[(long)r * W + c] += (double)delta[idx];
  }
}

__global__ __launch_bounds__(256) void scene_binarize_kernel(double* __restrict__ raster, const uint16_t* __restrict__ scene,
                                                             long npix, int C, double thr, uint8_t* __restrict__ mask) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    int valid = 0;
    for (int k = 0; k < C; ++k) valid |= scene[i * C + k] != 0;
    const double d = valid ? raster[i] : 0.0;
    raster[i] = d;
    mask[i] = d > thr ? 1 : 0;
  }
}

// min / max of diff over pixels with target > 0; values are >= 0, so their bit patterns order
// like unsigned integers and integer atomics give an order-independent (exact) result
__global__ __launch_bounds__(256) void valid_minmax_kernel(const float* __restrict__ diff, const uint8_t* __restrict__ target,
                                                           size_t n, unsigned* __restrict__ out) {
  unsigned lo = 0x7F800000u, hi = 0u;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (target[i] == 0) continue;
    const unsigned u = __float_as_uint(fmaxf(diff[i], 0.f));
    lo = u < lo ? u : lo;
    hi = u > hi ? u : hi;
  }
  __shared__ unsigned slo[256], shi[256];
  slo[threadIdx.x] = lo;
  shi[threadIdx.x] = hi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      slo[threadIdx.x] = min(slo[threadIdx.x], slo[threadIdx.x + s]);
      shi[threadIdx.x] = max(shi[threadIdx.x], shi[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicMin(out, slo[0]);
    atomicMax(out + 1, shi[0]);
  }
}

// hist[cls][j], j = number of (ascending) thresholds strictly below the pixel's value: the pixel is
// predicted 1 for thresholds 0..j-1.  cls = target - 1 in {0, 1}; target 0 = not valid.
constexpr int MAX_THR = 255;
__global__ __launch_bounds__(256) void threshold_hist_kernel(const float* __restrict__ diff, const uint8_t* __restrict__ target,
                                                             size_t n, const float* __restrict__ thr, int T,
                                                             unsigned long long* __restrict__ hist) {
  __shared__ unsigned sh[2][MAX_THR + 1];
  __shared__ float st[MAX_THR];
  for (int i = threadIdx.x; i < 2 * (MAX_THR + 1); i += 256) (&sh[0][0])[i] = 0u;
  for (int i = threadIdx.x; i < T; i += 256) st[i] = thr[i];
  __syncthreads();
  // at most 2^22 pixels per block keep the 32-bit LDS counters exact
  const size_t per_block = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
  const size_t begin = (size_t)blockIdx.x * per_block;
  const size_t end = begin + per_block < n ? begin + per_block : n;
  for (size_t i = begin + threadIdx.x; i < end; i += 256) {
    const int tg = target[i];
    if (tg == 0) continue;
    const float d = diff[i];
    int lo = 0, hi = T;               // first index with !(d > thr[idx])
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (d > st[mid]) lo = mid + 1; else hi = mid;
    }
    atomicAdd(&sh[tg > 1 ? 1 : 0][lo], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * (T + 1); i += 256) {
    const int cls = i / (T + 1), j = i % (T + 1);
    const unsigned v = sh[cls][j];
    if (v) atomicAdd(&hist[cls * (T + 1) + j], (unsigned long long)v);
  }
}

// Training / validation input pipeline: gather a batch of uint16 patches out of a
// patch pool resident in HBM (prepare_landsat8_biome.py:136-138 layout: images (N,H,W,C) uint16,
// masks (N,H,W) uint8), apply the per-sample flips and integer translation drawn by the host
// (data_loader.py:372-374: HorizontalFlip, VerticalFlip, then the shift; uncovered pixels are filled
// with 0 = nodata), normalise like data_loader.py:375 and write NCHW fp32 (ToTensorV2, :376).
// aug[b] = {flags (bit 0: horizontal flip, bit 1: vertical flip), dy, dx}.
__global__ __launch_bounds__(256) void patch_batch_kernel(const uint16_t* __restrict__ images, int H, int W, int C,
                                                          const long* __restrict__ index, const int* __restrict__ aug, int nb,
                                                          float* __restrict__ x, const uint8_t* __restrict__ masks,
                                                          uint8_t* __restrict__ mask_out) {
  const float rcp = 1.0f / 32767.5f;
  const long total = (long)nb * H * W;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int px = (int)(idx % W);
    const long t = idx / W;
    const int py = (int)(t % H), b = (int)(t / H);
    int flags = 0, dy = 0, dx = 0;
    if (aug) { flags = aug[3 * b]; dy = aug[3 * b + 1]; dx = aug[3 * b + 2]; }
    int sy = py - dy, sx = px - dx;
    const bool inside = sy >= 0 && sy < H && sx >= 0 && sx < W;
    if (flags & 2) sy = H - 1 - sy;
    if (flags & 1) sx = W - 1 - sx;
    const long spix = (index[b] * H + sy) * W + sx;
    float* dst = x + ((long)b * C * H + py) * W + px;
    const long plane = (long)H * W;
    if ((C & 1) == 0) {            // a pixel's bands as 32-bit pairs (the row is 4-byte aligned when C is even)
      const uint32_t* src = reinterpret_cast<const uint32_t*>(images + spix * C);
      for (int k = 0; k < C; k += 2) {
        const uint32_t v = inside ? src[k >> 1] : 0u;
        dst[k * plane] = ((float)(v & 0xFFFFu) - 32767.5f) * rcp;
        dst[(k + 1) * plane] = ((float)(v >> 16) - 32767.5f) * rcp;
      }
    } else {
      for (int k = 0; k < C; ++k) {
        const float f = inside ? (float)images[spix * C + k] : 0.f;
        dst[k * plane] = (f - 32767.5f) * rcp;
      }
    }
    if (mask_out) mask_out[idx] = inside ? masks[spix] : (uint8_t)0;
  }
}

// One block per tile: valid[b] = 1 if any band of any pixel of the (reflect-padded) tile is non-zero --
// the all-nodata test of evaluate.py:104 for every tile of a scene in one launch.
__global__ __launch_bounds__(256) void scene_tile_valid_kernel(const uint16_t* __restrict__ scene, int H, int W, int C,
                                                               int patch, const int* __restrict__ origins,
                                                               uint8_t* __restrict__ valid) {
  const int b = blockIdx.x;
  const int r0 = origins[2 * b], c0 = origins[2 * b + 1];
  // reflected rows / columns repeat rows / columns of the scene that the tile already covers or that lie
  // just inside the edge, so the tile's in-scene part plus the mirrored strip decide; walking the
  // padded index space keeps this the same predicate as scene_tiles_kernel's
  const int row_elems = patch * C;
  int seen = 0;
  for (int py = 0; py < patch && !seen; ++py) {
    int r = r0 + py;
    r = r < H ? r : 2 * (H - 1) - r;
    for (int e = threadIdx.x; e < row_elems; e += 256) {
      int c = c0 + e / C;
      c = c < W ? c : 2 * (W - 1) - c;
      seen |= scene[((long)r * W + c) * C + e % C] != 0;
    }
    seen = __syncthreads_or(seen);
  }
  if (threadIdx.x == 0) valid[b] = seen ? 1 : 0;
}

// make_psuedo_masks' per-batch tail (fcd_solver.py:762-772): pseudo = (diff > thr) compared in fp32,
// confusion counts over valid pixels, then pseudo + 1 with invalid pixels (target == 0) set to 0.
__global__ __launch_bounds__(256) void pseudo_mask_kernel(const float* __restrict__ diff, const uint8_t* __restrict__ target,
                                                          size_t n, float thr, uint8_t* __restrict__ out,
                                                          unsigned long long* __restrict__ cm) {
  __shared__ unsigned sc[4];
  if (threadIdx.x < 4) sc[threadIdx.x] = 0u;
  __syncthreads();
  // at most 2^22 pixels per block keep the 32-bit LDS counters exact
  const size_t per_block = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
  const size_t begin = (size_t)blockIdx.x * per_block;
  const size_t end = begin + per_block < n ? begin + per_block : n;
  unsigned c[4] = {0u, 0u, 0u, 0u};
  for (size_t i = begin + threadIdx.x; i < end; i += 256) {
    const int pred = diff[i] > thr ? 1 : 0;
    const int tg = target ? target[i] : 1;
    out[i] = tg == 0 ? (uint8_t)0 : (uint8_t)(pred + 1);
    if (tg != 0) c[(tg > 1 ? 2 : 0) + pred] += 1u;
  }
  if (cm) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned v = c[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sc[k], v);
    }
    __syncthreads();
    if (threadIdx.x < 4 && sc[threadIdx.x]) atomicAdd(&cm[threadIdx.x], (unsigned long long)sc[threadIdx.x]);
  }
}

// write_tifs' byte rasters (evaluate.py:384-390): difference * 255 truncated to uint8, and the
// prediction as 128 (clear) / 255 (cloudy) with invalid pixels 0.
__global__ __launch_bounds__(256) void scene_tif_kernel(const double* __restrict__ raster, const uint8_t* __restrict__ pred,
                                                        const uint8_t* __restrict__ valid, const uint16_t* __restrict__ scene,
                                                        long npix, int C, uint8_t* __restrict__ diff_u8,
                                                        uint8_t* __restrict__ mask_u8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    int ok;
    if (valid) {
      ok = valid[i] != 0;
    } else {
      ok = 0;
      for (int k = 0; k < C; ++k) ok |= scene[i * C + k] != 0;
    }
    if (diff_u8) diff_u8[i] = (uint8_t)(long)(raster[i] * 255.0);     // numpy astype(uint8) of a float64 in [0, 256)
    if (mask_u8) mask_u8[i] = ok ? (pred[i] ? 255 : 128) : 0;
  }
}

inline int grid_for(size_t n, int cap = 8192) {
  const size_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

}  // namespace

extern "C" int fcd_scene_tiles(const uint16_t* scene, int H, int W, int C, int patch, const int* origins_rc, int nb,
                               float* x_nchw, int* any_valid, void* stream) {
  FCD_REQUIRE(scene && origins_rc && x_nchw && any_valid, "scene_tiles: null pointer");
  FCD_REQUIRE(H > 0 && W > 0 && C > 0 && patch > 0 && nb > 0, "scene_tiles: bad extents");
  // single reflection only: the padded region must be shorter than the scene itself
  FCD_REQUIRE(patch <= H - 1 && patch <= W - 1, "scene_tiles: scene %dx%d smaller than patch %d + 1", H, W, patch);
  hipLaunchKernelGGL(scene_tiles_kernel, dim3(grid_for((size_t)nb * patch * patch)), dim3(256), 0, fcd_stream(stream), scene,
                     H, W, C, patch, origins_rc, nb, x_nchw, any_valid);
  return fcd_check_launch("scene_tiles");
}

extern "C" int fcd_scene_tile_valid(const uint16_t* scene, int H, int W, int C, int patch, const int* origins_rc, int nb,
                                    uint8_t* valid, void* stream) {
  FCD_REQUIRE(scene && origins_rc && valid, "scene_tile_valid: null pointer");
  FCD_REQUIRE(H > 0 && W > 0 && C > 0 && patch > 0 && nb > 0, "scene_tile_valid: bad extents");
  FCD_REQUIRE(patch <= H - 1 && patch <= W - 1, "scene_tile_valid: scene %dx%d smaller than patch %d + 1", H, W, patch);
  hipLaunchKernelGGL(scene_tile_valid_kernel, dim3(nb), dim3(256), 0, fcd_stream(stream), scene, H, W, C, patch, origins_rc,
                     valid);
  return fcd_check_launch("scene_tile_valid");
}

extern "C" int fcd_pseudo_mask(const float* diff, const uint8_t* target, size_t n, float thr, uint8_t* out,
                               unsigned long long* cm, void* stream) {
  FCD_REQUIRE(diff && out && n > 0, "pseudo_mask: bad arguments");
  int blocks = grid_for(n, 2048);
  const size_t min_blocks = (n + ((size_t)1 << 22) - 1) >> 22;
  if ((size_t)blocks < min_blocks) blocks = (int)min_blocks;
  hipLaunchKernelGGL(pseudo_mask_kernel, dim3(blocks), dim3(256), 0, fcd_stream(stream), diff, target, n, thr, out, cm);
  return fcd_check_launch("pseudo_mask");
}

extern "C" int fcd_scene_tif_bytes(const double* raster, const uint8_t* prediction, const uint8_t* valid,
                                   const uint16_t* scene, int H, int W, int C, uint8_t* diff_u8, uint8_t* mask_u8,
                                   void* stream) {
  FCD_REQUIRE(raster && (diff_u8 || mask_u8) && H > 0 && W > 0, "scene_tif_bytes: bad arguments");
  FCD_REQUIRE(!mask_u8 || (prediction && (valid || (scene && C > 0))),
              "scene_tif_bytes: the mask needs the prediction and a validity source (valid mask or the scene)");
  hipLaunchKernelGGL(scene_tif_kernel, dim3(grid_for((size_t)H * W)), dim3(256), 0, fcd_stream(stream), raster, prediction,
                     valid, scene, (long)H * W, C, diff_u8, mask_u8);
  return fcd_check_launch("scene_tif_bytes");
}

extern "C" int fcd_scene_stitch(const float* delta, const int* origins_rc, int nb, int patch, double* raster, int H, int W,
                                void* stream) {
  FCD_REQUIRE(delta && origins_rc && raster && nb > 0 && patch > 0 && H > 0 && W > 0, "scene_stitch: bad arguments");
  hipLaunchKernelGGL(scene_stitch_kernel, dim3(grid_for((size_t)nb * patch * patch)), dim3(256), 0, fcd_stream(stream), delta,
                     origins_rc, nb, patch, raster, H, W);
  return fcd_check_launch("scene_stitch");
}

extern "C" int fcd_scene_binarize(double* raster, const uint16_t* scene, int H, int W, int C, double threshold,
                                  uint8_t* mask, void* stream) {
  FCD_REQUIRE(raster && scene && mask && H > 0 && W > 0 && C > 0, "scene_binarize: bad arguments");
  hipLaunchKernelGGL(scene_binarize_kernel, dim3(grid_for((size_t)H * W)), dim3(256), 0, fcd_stream(stream), raster, scene,
                     (long)H * W, C, threshold, mask);
  return fcd_check_launch("scene_binarize");
}

extern "C" int fcd_valid_minmax(const float* diff, const uint8_t* target, size_t n, float* minmax, void* stream) {
  FCD_REQUIRE(diff && target && minmax && n > 0, "valid_minmax: bad arguments");
  // minmax[0], minmax[1] must be initialised by the caller to +inf and 0 (first batch) and are
  // folded across calls
  hipLaunchKernelGGL(valid_minmax_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, fcd_stream(stream), diff, target, n,
                     reinterpret_cast<unsigned*>(minmax));
  return fcd_check_launch("valid_minmax");
}

extern "C" int fcd_threshold_hist(const float* diff, const uint8_t* target, size_t n, const float* thresholds, int T,
                                  unsigned long long* hist, void* stream) {
  FCD_REQUIRE(diff && target && thresholds && hist && n > 0, "threshold_hist: bad arguments");
  FCD_REQUIRE(T >= 1 && T <= MAX_THR, "threshold_hist: T=%d outside 1..%d", T, MAX_THR);
  int blocks = grid_for(n, 2048);
  const size_t min_blocks = (n + ((size_t)1 << 22) - 1) >> 22;
  if ((size_t)blocks < min_blocks) blocks = (int)min_blocks;
  hipLaunchKernelGGL(threshold_hist_kernel, dim3(blocks), dim3(256), 0, fcd_stream(stream), diff, target, n, thresholds, T,
                     hist);
  return fcd_check_launch("threshold_hist");
}

extern "C" int fcd_patch_batch(const uint16_t* images, long n_images, int H, int W, int C, const long* index, const int* aug,
                               int nb, float* x_nchw, const uint8_t* masks, uint8_t* mask_out, void* stream) {
  FCD_REQUIRE(images && index && x_nchw, "patch_batch: null pointer");
  FCD_REQUIRE(n_images > 0 && H > 0 && W > 0 && C > 0 && nb > 0, "patch_batch: bad extents");
  FCD_REQUIRE((masks == nullptr) == (mask_out == nullptr), "patch_batch: masks and mask_out go together");
  // index values are validated by the caller (they live in device memory)
  hipLaunchKernelGGL(patch_batch_kernel, dim3(grid_for((size_t)nb * H * W)), dim3(256), 0, fcd_stream(stream), images, H, W, C,
                     index, aug, nb, x_nchw, masks, mask_out);
  return fcd_check_launch("patch_batch");
}
