// Sliding-window prediction (utils/slide_predict.py:60-117 of the reference): tile extraction,
// per-pixel class, and the ordered Gaussian-weighted blend, all on the device.
//
// The reference blends tile by tile in Python: float32 accumulators, each `+=` evaluated in
// float64 and rounded back to float32, so the final rounding depends on the ORDER of the
// tiles (duplicates of the clamped last row/column included).  The blend kernel below is a
// gather — one thread per output pixel visits the tiles covering it in exactly that order —
// which makes it parallel and still bit-identical.
#include "common.h"

namespace {

// uint8 [H][W][C] crop -> channels-last activation [n][window][window][C], value = u8 / 255
// (float32 division as `tensor.float() / 255.0`, slide_predict.py:86).
template <typename T>
__global__ __launch_bounds__(256) void slide_gather_kernel(const uint8_t* __restrict__ image, int W, int C,
                                                           const int* __restrict__ tile_yx, int window,
                                                           T* __restrict__ out) {
  const int t = blockIdx.z, y = blockIdx.y;
  const int y0 = tile_yx[2 * t], x0 = tile_yx[2 * t + 1];
  const int row = window * C;
  const uint8_t* src = image + ((int64_t)(y0 + y) * W + x0) * C;
  T* dst = out + ((int64_t)t * window + y) * row;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < row; i += gridDim.x * 256)
    dst[i] = from_f32<T>(__fdiv_rn((float)src[i], 255.0f));
}

// torch.argmax(pred, dim=1) (slide_predict.py:92): first maximal class, NaN counts as maximal.
template <typename T>
__global__ __launch_bounds__(256) void argmax_classes_kernel(const T* __restrict__ logits, int64_t sb, int64_t sc,
                                                             int64_t sp, int C, int HW, uint8_t* __restrict__ out) {
  const int b = blockIdx.y;
  const T* lg = logits + (int64_t)b * sb;
  for (int pix = blockIdx.x * 256 + threadIdx.x; pix < HW; pix += gridDim.x * 256) {
    float m = to_f32(lg[(int64_t)pix * sp]);
    int am = 0;
    for (int c = 1; c < C; c++) {
      const float v = to_f32(lg[(int64_t)pix * sp + (int64_t)c * sc]);
      const bool take = (v > m) || (v != v && m == m);
      m = take ? v : m;
      am = take ? c : am;
    }
    out[(int64_t)b * HW + pix] = (uint8_t)am;
  }
}

constexpr int BLEND_MAX_STARTS = 4096;      // tile rows + tile columns held in LDS

// One thread per output pixel.  Tile k = i*nx + j covers rows [ys[i], ys[i]+window) and columns
// [xs[j], xs[j]+window); its class map is maps[map_index[k]].  For each covering tile, in tile
// order:  out = (float)((double)out + (double)class * g),  wsum = (float)((double)wsum + g),
// with g = gauss[y-ys[i]][x-xs[j]] in float64 (slide_predict.py:103-104); then
// round-half-even(out / wsum) as uint8 (slide_predict.py:111-112).
__global__ __launch_bounds__(256) void slide_blend_kernel(const uint8_t* __restrict__ maps,
                                                          const int* __restrict__ map_index,
                                                          const int* __restrict__ ys, const int* __restrict__ xs,
                                                          int ny, int nx, int window,
                                                          const double* __restrict__ gauss, int out_h, int out_w,
                                                          uint8_t* __restrict__ out) {
  __shared__ int starts[BLEND_MAX_STARTS];
  for (int i = threadIdx.x; i < ny + nx; i += 256) starts[i] = i < ny ? ys[i] : xs[i - ny];
  __syncthreads();
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= out_w || y >= out_h) return;
  float acc = 0.f, wsum = 0.f;
  const int64_t ww = (int64_t)window * window;
  for (int i = 0; i < ny; i++) {
    const int dy = y - starts[i];
    if ((unsigned)dy >= (unsigned)window) continue;
    for (int j = 0; j < nx; j++) {
      const int dx = x - starts[ny + j];
      if ((unsigned)dx >= (unsigned)window) continue;
      const int64_t o = (int64_t)dy * window + dx;
      const double g = gauss[o];
      const double cls = (double)maps[(int64_t)map_index[i * nx + j] * ww + o];
      acc = (float)__dadd_rn((double)acc, __dmul_rn(cls, g));
      wsum = (float)__dadd_rn((double)wsum, g);
    }
  }
  // `np.divide(..., where=weight != 0)`: every pixel inside the padded image is covered by a
  // tile and the weights are positive, so the reference's undefined branch cannot be reached
  out[(int64_t)y * out_w + x] = (wsum != 0.f) ? (uint8_t)rintf(__fdiv_rn(acc, wsum)) : (uint8_t)0;
}

}  // namespace

extern "C" {

int mis_slide_gather_tiles(const uint8_t* image, int H, int W, int C, const int32_t* tile_yx, int n_tiles,
                           int window, int dtype, void* out, mis_stream_t stream) {
  MIS_REQUIRE(image && tile_yx && out, "slide_gather_tiles: null pointer");
  MIS_REQUIRE(H >= 1 && W >= 1 && C >= 1 && n_tiles >= 1 && window >= 1 && window <= H && window <= W,
              "slide_gather_tiles: bad geometry H=%d W=%d C=%d n=%d window=%d", H, W, C, n_tiles, window);
  MIS_REQUIRE(n_tiles <= 65535 && window <= 65535, "slide_gather_tiles: at most 65535 tiles of at most 65535 rows");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "slide_gather_tiles: bad dtype %d", dtype);
  dim3 grid(cdiv((int64_t)window * C, 256 * 4), window, n_tiles);
  if (dtype == MIS_F32)
    slide_gather_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(image, W, C, tile_yx, window, (float*)out);
  else
    slide_gather_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>(image, W, C, tile_yx, window, (bf16*)out);
  MIS_CHECK_LAUNCH("slide_gather_kernel");
  return MIS_OK;
}

int mis_argmax_classes(const void* logits, int dtype, int64_t stride_b, int64_t stride_c, int64_t stride_p, int B,
                       int C, int HW, uint8_t* out, mis_stream_t stream) {
  MIS_REQUIRE(logits && out, "argmax_classes: null pointer");
  MIS_REQUIRE(B >= 1 && B <= 65535 && HW >= 1 && C >= 1 && C <= 256, "argmax_classes: need 1 <= C <= 256, B <= 65535");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "argmax_classes: bad dtype %d", dtype);
  int gx = cdiv(HW, 256 * 4);
  if (gx > 4096) gx = 4096;
  dim3 grid(gx, B);
  if (dtype == MIS_F32)
    argmax_classes_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)logits, stride_b, stride_c,
                                                                        stride_p, C, HW, out);
  else
    argmax_classes_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16*)logits, stride_b, stride_c,
                                                                       stride_p, C, HW, out);
  MIS_CHECK_LAUNCH("argmax_classes_kernel");
  return MIS_OK;
}

int mis_slide_blend(const uint8_t* class_maps, const int32_t* map_index, const int32_t* ys, const int32_t* xs,
                    int ny, int nx, int window, const double* gauss, int out_h, int out_w, uint8_t* out,
                    mis_stream_t stream) {
  MIS_REQUIRE(class_maps && map_index && ys && xs && gauss && out, "slide_blend: null pointer");
  MIS_REQUIRE(ny >= 1 && nx >= 1 && ny + nx <= BLEND_MAX_STARTS, "slide_blend: ny + nx must be in [2, %d]",
              BLEND_MAX_STARTS);
  MIS_REQUIRE(window >= 1 && out_h >= 1 && out_w >= 1, "slide_blend: bad geometry");
  MIS_REQUIRE(cdiv(out_h, 4) <= 65535, "slide_blend: image too tall");
  dim3 grid(cdiv(out_w, 64), cdiv(out_h, 4));
  slide_blend_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(class_maps, map_index, ys, xs, ny, nx, window, gauss,
                                                            out_h, out_w, out);
  MIS_CHECK_LAUNCH("slide_blend_kernel");
  return MIS_OK;
}

}  // extern "C"
