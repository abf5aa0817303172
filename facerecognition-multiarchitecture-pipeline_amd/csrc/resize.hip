// Batched bilinear resize of 8-bit RGB images on the device, bit-exact with Pillow's two-pass resampler - the arithmetic
// `transforms.Resize((224, 224))` runs on a PIL image in the reference's input pipeline
// (/root/reference/src/testing.py:99-100, 561-566; src/training.py:305-310; src/app.py:39 for 160x160).
//
// Pillow (libImaging/Resample.c) resizes in two passes, horizontal then vertical, each a small FIR per output sample:
// integer coefficients (22 fractional bits, rounded half away from zero from double-precision weights), int32
// accumulation started at 1 << 21, arithmetic shift, clip to [0, 255], and the horizontal pass's result is ROUNDED TO
// 8 BITS before the vertical pass reads it.  The coefficient tables depend only on (input size, output size); the host
// builds them in float64 with Pillow's operation order (resize.py, cached per size pair) and this kernel does the integer
// part: one workgroup = `rows_per_block` output rows of one image; it runs the horizontal pass for exactly the input rows
// those output rows touch into an LDS image (4 bytes per pixel), then the vertical pass out of LDS.  Images of different
// sizes share one launch (per-image descriptor: byte offset, size, table offsets).  Byte-granular global reads: the step is
// HBM / latency bound and small next to the embed step (the output is 150 KB per face).
#include "frmap_common.h"

struct FrmapResizeItem {       // mirrored by resize.py (numpy structured dtype, 40 bytes)
  unsigned long long src_off;  // byte offset of the image (HWC uint8 RGB, row stride W * 3) in the source pool
  int H, W;
  int bx_off, kx_off, ksx;     // int32 offsets into `tables`: bounds [out_w][2] = (first input column, taps), coefficients [out_w][ksx]; ksx = 0: width unchanged
  int by_off, ky_off, ksy;     // the same for rows; ksy = 0: height unchanged
};

__device__ __forceinline__ int resize_clip8(int v) {
  v >>= 22;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void resize_bilinear_u8_kernel(const unsigned char* __restrict__ pool, const FrmapResizeItem* __restrict__ items,
                                                                 const int* __restrict__ tables, unsigned char* __restrict__ out,
                                                                 int out_h, int out_w, int rows_per_block) {
  extern __shared__ unsigned int s_tmp[];  // [rows][out_w] packed R | G << 8 | B << 16
  const FrmapResizeItem it = items[blockIdx.y];
  const int y0 = blockIdx.x * rows_per_block, y1 = min(y0 + rows_per_block, out_h);
  if (y0 >= out_h) return;
  const unsigned char* src = pool + it.src_off;
  const int* bx = tables + it.bx_off;
  const int* kx = tables + it.kx_off;
  const int* by = tables + it.by_off;
  const int* ky = tables + it.ky_off;
  int row_first = y0, row_last = y1;           // input rows this block's output rows read
  if (it.ksy) {
    row_first = by[2 * y0];
    row_last = by[2 * (y1 - 1)] + by[2 * (y1 - 1) + 1];
  }
  const int nrows = row_last - row_first;
  // ---- horizontal pass (ImagingResampleHorizontal_8bpc) over the needed rows
  for (int idx = threadIdx.x; idx < nrows * out_w; idx += 256) {
    const int r = idx / out_w, xx = idx - r * out_w;
    const unsigned char* rowp = src + ((size_t)(row_first + r) * it.W) * 3;
    unsigned v;
    if (it.ksx) {
      const int xmin = bx[2 * xx], cnt = bx[2 * xx + 1];
      const int* k = kx + xx * it.ksx;
      int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
      const unsigned char* p = rowp + (size_t)xmin * 3;
      for (int x = 0; x < cnt; ++x) {
        const int w = k[x];
        s0 += (int)p[3 * x] * w; s1 += (int)p[3 * x + 1] * w; s2 += (int)p[3 * x + 2] * w;
      }
      v = (unsigned)resize_clip8(s0) | ((unsigned)resize_clip8(s1) << 8) | ((unsigned)resize_clip8(s2) << 16);
    } else {
      const unsigned char* p = rowp + (size_t)xx * 3;
      v = (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16);
    }
    s_tmp[idx] = v;
  }
  __syncthreads();
  // ---- vertical pass (ImagingResampleVertical_8bpc) out of LDS
  unsigned char* dst = out + ((size_t)blockIdx.y * out_h) * out_w * 3;
  for (int idx = threadIdx.x; idx < (y1 - y0) * out_w; idx += 256) {
    const int ry = idx / out_w, xx = idx - ry * out_w, yy = y0 + ry;
    unsigned v;
    if (it.ksy) {
      const int ymin = by[2 * yy] - row_first, cnt = by[2 * yy + 1];
      const int* k = ky + yy * it.ksy;
      int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
      for (int y = 0; y < cnt; ++y) {
        const unsigned t = s_tmp[(ymin + y) * out_w + xx];
        const int w = k[y];
        s0 += (int)(t & 255u) * w; s1 += (int)((t >> 8) & 255u) * w; s2 += (int)((t >> 16) & 255u) * w;
      }
      v = (unsigned)resize_clip8(s0) | ((unsigned)resize_clip8(s1) << 8) | ((unsigned)resize_clip8(s2) << 16);
    } else {
      v = s_tmp[ry * out_w + xx];
    }
    unsigned char* o = dst + ((size_t)yy * out_w + xx) * 3;
    o[0] = (unsigned char)(v & 255u); o[1] = (unsigned char)((v >> 8) & 255u); o[2] = (unsigned char)((v >> 16) & 255u);
  }
}

// src_pool: the images' bytes back to back (device); items: B descriptors (device); tables: int32 coefficient pool (device);
// out: B x out_h x out_w x 3 uint8.  rows_per_block >= 1 and lds_rows = the most input rows any block touches (the host
// knows both from the tables); lds_rows * out_w * 4 bytes must fit in LDS (64 KB are always enough for rows_per_block = 1
// up to a 75x reduction).
extern "C" int frmap_resize_bilinear_u8(const unsigned char* src_pool, const void* items, const int* tables, unsigned char* out,
                                        int B, int out_h, int out_w, int rows_per_block, int lds_rows, void* stream) {
  FRMAP_REQUIRE(src_pool && items && tables && out, "resize_bilinear_u8: null pointer");
  FRMAP_REQUIRE(B > 0 && B <= 65535 && out_h > 0 && out_w > 0 && rows_per_block > 0 && lds_rows > 0, "resize_bilinear_u8: bad shape");
  const size_t lds = (size_t)lds_rows * out_w * sizeof(unsigned int);
  FRMAP_REQUIRE(lds <= 160 * 1024, "resize_bilinear_u8: %d rows x %d columns do not fit in LDS (use fewer rows per block)", lds_rows, out_w);
  if (frmap_big_lds((const void*)resize_bilinear_u8_kernel, 160 * 1024)) return -2;
  const dim3 grid((out_h + rows_per_block - 1) / rows_per_block, B);
  hipLaunchKernelGGL(resize_bilinear_u8_kernel, grid, dim3(256), lds, (hipStream_t)stream, src_pool, (const FrmapResizeItem*)items, tables, out,
                     out_h, out_w, rows_per_block);
  FRMAP_LAUNCH_CHECK();
  return 0;
}
