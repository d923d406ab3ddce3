// rotate.hip -- radarDriver::Callback for the non-Oxford datasets (radar_driver.cpp:74-90): the sensor image arrives with
// rows = range bins and columns = azimuths and is turned by cv::rotate(ROTATE_90_COUNTERCLOCKWISE) (:84) into the
// rows = azimuth layout the filter works on:   out[i][j] = in[j][in_cols - 1 - i]   (out: in_cols x in_rows).
// Byte transpose through a 64 x 64 LDS tile, reads and writes coalesced; HBM-bound, 2 bytes of traffic per pixel.
// Batched over n images.
#include "common.h"

namespace {

constexpr int TILE = 64;

// VEC = true: rows and columns are multiples of four, every lane moves a dword (256 B per wave instruction) and the bytes are
// regrouped through the tile; VEC = false: byte accesses (any shape).
template <bool VEC>
__global__ __launch_bounds__(256) void rotate90ccw_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int in_rows, int in_cols) {
  constexpr int PITCH = TILE + 4;  // 17 dwords: column walks hit 64 different banks
  __shared__ __attribute__((aligned(4))) uint8_t tile[TILE * PITCH];
  const size_t img = (size_t)blockIdx.z * in_rows * in_cols;
  const int r0 = blockIdx.y * TILE, c0 = blockIdx.x * TILE;  // tile origin in the input
  if (VEC) {
    for (int idx = threadIdx.x; idx < TILE * TILE / 4; idx += 256) {
      const int y = idx >> 4, x = (idx & 15) * 4;
      const int r = r0 + y, c = c0 + x;
      uint32_t v = 0;
      if (r < in_rows && c < in_cols) v = *reinterpret_cast<const uint32_t*>(in + img + (size_t)r * in_cols + c);
      *reinterpret_cast<uint32_t*>(&tile[y * PITCH + x]) = v;
    }
    __syncthreads();
    // output row i = in_cols - 1 - c holds input column c; four consecutive output bytes = four consecutive input rows
    for (int idx = threadIdx.x; idx < TILE * TILE / 4; idx += 256) {
      const int y = idx >> 4, q = (idx & 15) * 4;  // y: column of the tile, q: first of four tile rows
      const int c = c0 + y, r = r0 + q;
      if (r < in_rows && c < in_cols) {
        const uint32_t v = (uint32_t)tile[q * PITCH + y] | ((uint32_t)tile[(q + 1) * PITCH + y] << 8) | ((uint32_t)tile[(q + 2) * PITCH + y] << 16) |
                           ((uint32_t)tile[(q + 3) * PITCH + y] << 24);
        *reinterpret_cast<uint32_t*>(out + img + (size_t)(in_cols - 1 - c) * in_rows + r) = v;
      }
    }
  } else {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4 threads
    for (int y = ty; y < TILE; y += 4) {
      const int r = r0 + y, c = c0 + tx;
      if (r < in_rows && c < in_cols) tile[y * PITCH + tx] = in[img + (size_t)r * in_cols + c];
    }
    __syncthreads();
    for (int y = ty; y < TILE; y += 4) {
      const int c = c0 + y, r = r0 + tx;
      if (r < in_rows && c < in_cols) out[img + (size_t)(in_cols - 1 - c) * in_rows + r] = tile[tx * PITCH + y];
    }
  }
}

}  // namespace

extern "C" {

int cfear_rotate_polar_device(cfear_ctx* ctx, const uint8_t* d_in, int n_images, int in_rows, int in_cols, uint8_t* d_out) {
  if (!ctx || !d_in || !d_out || n_images <= 0 || in_rows <= 0 || in_cols <= 0 || d_in == d_out)
    return cfear_fail(ctx, CFEAR_ERR_INVALID, "rotate_polar: bad argument (in-place rotation is not supported)");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const dim3 grid((in_cols + TILE - 1) / TILE, (in_rows + TILE - 1) / TILE, n_images);
  const bool vec = (in_rows % 4 == 0) && (in_cols % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 3) == 0;
  if (vec) hipLaunchKernelGGL(rotate90ccw_kernel<true>, grid, dim3(256), 0, ctx->stream, d_in, d_out, in_rows, in_cols);
  else hipLaunchKernelGGL(rotate90ccw_kernel<false>, grid, dim3(256), 0, ctx->stream, d_in, d_out, in_rows, in_cols);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

int cfear_rotate_polar(cfear_ctx* ctx, const uint8_t* h_in, int in_rows, int in_cols, uint8_t* h_out) {
  if (!ctx || !h_in || !h_out || in_rows <= 0 || in_cols <= 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "rotate_polar: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t)in_rows * in_cols;
  uint8_t* d = nullptr;
  size_t got = 0;
  { void* blk = nullptr; const int arc = cfear_pool_alloc(ctx, 2 * bytes, &blk, &got); if (arc != CFEAR_OK) return arc; d = static_cast<uint8_t*>(blk); }
  // in through the pinned image staging (pageable sources: cabi.hip cfear_upload_image), out through the same staging: one DMA each way
  int rc = cfear_upload_image(ctx, d, h_in, bytes);
  if (rc == CFEAR_OK) rc = cfear_rotate_polar_device(ctx, d, 1, in_rows, in_cols, d + bytes);
  hipError_t e = hipSuccess;
  const bool staged = rc == CFEAR_OK && ctx->h_img && ctx->h_img_bytes >= bytes && ctx->ev_img_pending;
  if (rc == CFEAR_OK) e = hipMemcpyAsync(staged ? static_cast<void*>(ctx->h_img) : static_cast<void*>(h_out), d + bytes, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (rc == CFEAR_OK && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (rc == CFEAR_OK && e == hipSuccess && staged) { memcpy(h_out, ctx->h_img, bytes); ctx->ev_img_pending = false; }
  cfear_pool_free(ctx, d, got);
  if (rc != CFEAR_OK) return rc;
  if (e != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_HIP, "rotate_polar", e);
  return CFEAR_OK;
}

}  // extern "C"
