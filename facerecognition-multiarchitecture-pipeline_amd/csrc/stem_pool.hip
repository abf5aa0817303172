// Fused ResNet stem for gfx950: fp32 NCHW input -> conv 7x7 s2 p3 (3->64) + folded BN + ReLU ->
// maxpool 3x3 s2 p1 -> NHWC B x Hq x Wq x 64 in bf16/f16, ONE kernel.
//
// Replaces `conv1 -> bn1 -> relu -> maxpool` of the torchvision ResNet-18 behind
// /root/reference/src/face_models.py:67,463,658 together with the NCHW->NHWC cast.  Unfused, the
// 64 x 112 x 112 conv output (1.6 MB/face in bf16) makes a round trip through HBM just to be
// reduced 4x by the pool; fused, only the 602 KB fp32 input is read and the 401 KB pooled map
// is written per face.
//
// Work decomposition (256 threads = 4 waves; PERSISTENT: two workgroups per CU walk tiles
// (image, 3 pooled rows, column half), weights stay in LDS; one workgroup stages while the other
// runs its MFMA phase):
//   the workgroup computes conv rows cr0 = 2*py0-1 .. cr0+6 (7 rows -> 3 pooled rows), 4 strips wide;
//   wave s owns the column strip cc = 14s-1 .. 14s+14 (16 conv columns -> 7 pooled columns; strips
//   advance by 14 so every 3-wide pooling window lies inside one strip: 8 strips = 112 columns);
//   so a lane (lr = column in strip, g = k-group) keeps a 7-row x 1-column x 16-channel patch in
//   accumulators: the vertical max is register-local, the horizontal max is two 16-lane shuffles,
//   and the result goes out through the shared LDS-transposed whole-line store.
//   K axis as in conv_small_cin.hip: k = kh*32 + kw*4 + c (7 k-steps of MFMA 16x16x32).
// Recompute: 7/6 rows x 16/14 columns = 1.33x the stem's MACs (6.5 % of the network), paid to drop
// 3.2 MB/face of HBM traffic and two launches.
#include "frmap_common.h"
#include <stdlib.h>

struct StemPoolParams {
  const float* x;           // fp32 NCHW input (U8 = false)
  const unsigned char* x8;  // uint8 HWC RGB input (U8 = true): ToTensor + Normalize(mean, std) happen at staging time
  float mean[3], std[3];
  const void* wpk;
  const float* shift;
  void* out;
  int N, Hi, Wi, Hc, Wc, Hq, Wq;
  int Wl;        // staged row length in pixels (even)
  int nstrips;   // ceil(Wq / 7) <= 8
  int nhalves;   // column halves of 4 strips each (1 or 2)
  int rgroups;   // ceil(Hq / 3)
  uint32_t magic_Wl2;
  int halo_bytes;
};

// POOL3 = true : MaxPool2d(3, 2, 1) (ResNet stem): 7 conv rows -> 3 pooled rows, strips of 16 conv columns
//                advance by 14 (7 pooled columns each), first conv row/column of a tile is 2*p0 - 1.
// POOL3 = false: MaxPool2d(2, 2)    (SiameseNet conv.0-3, face_models.py:115-118): 6 conv rows -> 3 pooled
//                rows, strips of 16 conv columns -> 8 pooled columns, no overlap.
// value of lane i+N of the same 16-lane row (N = 1, 2); the row's last lanes keep their own value.
// A DPP row shift is one VALU op; __shfl_down compiles to ds_bpermute (an LDS round trip per call).
template <int N>
__device__ __forceinline__ float row_down(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x100 + N, 0xF, 0xF, false));
}

// VEC4: the input rows are staged with 16-byte loads (4 pixels of one colour plane per lane and load; needs
// W % 4 == 0 and a 16-byte aligned tensor): 9 loads per thread and tile instead of 30 dword loads, which
// took ~70 cycles each to issue on the vector-memory path.
// U8 (implies the VEC4 item shape): the input is uint8 HWC (`src/testing.py:99-104` BEFORE ToTensor / Normalize): one
// 12-byte load fetches the 4 pixels x RGB of an item, and every byte goes through a 256-entry table per channel that
// holds ((u / 255) - mean) / std already rounded to the storage type - bit-identical to frmap_normalize_u8_hwc
// followed by the fp32 path, without the 602 KB/face fp32 tensor ever existing (150 KB/face are read instead).
// Tiles: the grid is persistent (two workgroups per CU).  Workgroups that share an XCD (blockIdx % 8) walk ONE
// contiguous range of tiles together, so the input rows that vertically adjacent tiles both stage (19 rows per 12
// new ones) are served by that XCD's L2 instead of being fetched again.  The NEXT tile's loads are issued before the
// current tile's MFMA phase and land under it.
// PF: issue the next tile's loads before the current tile's MFMA phase (costs NIT x 3 x 4 staging registers in the
// fp32 path - the 256-register budget then spills ~15 - and NIT x 3 in the uint8 path).
template <typename TT, bool POOL3, bool VEC4, bool U8, bool PF>
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const StemPoolParams p) {
  constexpr int MI = POOL3 ? 7 : 6, NI = 4, KSTEPS = 7, KPAD = 224, WPITCH = (KPAD + 16) * 2;
  constexpr int PADL0 = POOL3 ? 5 : 3, NROWS = 2 * (MI - 1) + 7;
  // VEC4: 4-pixel groups start on multiples of 4 input columns (colbase - (PADL - 1) + 4q) and land at image pixel
  // 4q + 1: the image origin stays an EVEN number of pixels left of the scalar path's, so fragment reads stay 16-byte aligned
  constexpr int PADL = VEC4 ? (POOL3 ? 9 : 5) : PADL0;
  constexpr int NGRP = POOL3 ? 32 : 35;
  constexpr int CSTEP = POOL3 ? 14 : 16;       // conv columns a strip advances by
  constexpr int PPS = POOL3 ? 7 : 8;           // pooled columns per strip
  constexpr int WLH0 = 2 * CSTEP * 4 + (POOL3 ? 10 : 8);  // staged row length (pixels) of a 4-strip half tile
  constexpr int WLH = VEC4 ? 4 * NGRP + 2 : WLH0;
  constexpr int NITEMS = VEC4 ? NROWS * NGRP : NROWS * (WLH / 2);
  constexpr int NIT = (NITEMS + 255) / 256;  // staging items per thread
  constexpr int PITCH = NI * 64 + 16;
  using vec8 = typename TT::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wl = smem + ((NROWS * WLH * 8 + 1023) & ~1023);
  char* scratch_all = wl + 64 * WPITCH;
  typename TT::elem* lut = (typename TT::elem*)(scratch_all + 4 * 16 * PITCH);  // U8: [3][256] normalised values

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const size_t HW = (size_t)p.Hi * p.Wi;
  const int plane_b = (int)(HW * 4);
  const int ntiles = p.N * p.rgroups * p.nhalves;
  // workgroups b, b + 8, b + 16 ... share an XCD: they walk the XCD's contiguous tile range side by side
  // (grids smaller than 8 workgroups: as many ranges as workgroups)
  const int nx = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
  const int xcd = blockIdx.x % nx, wg_in_xcd = blockIdx.x / nx;
  const int wgs_in_xcd = ((int)gridDim.x - xcd + nx - 1) / nx;
  const int t_lo = (int)(((long long)ntiles * xcd) / nx), t_hi = (int)(((long long)ntiles * (xcd + 1)) / nx);
  int tile = t_lo + wg_in_xcd;
  if (U8) {  // the table is needed by every workgroup that stages anything
    for (int i = tid; i < 768; i += 256) {
      const int c = i >> 8, u = i & 255;
      lut[i] = TT::from_f32(((float)u / 255.0f - p.mean[c]) / p.std[c]);
    }
  }
  if (tile >= t_hi) return;

  {  // weights: [64][240] image (same packing as conv_small_cin), resident for the whole kernel
    const u32x4_t* src = (const u32x4_t*)p.wpk;
    constexpr int NV = 64 * WPITCH / 16;
    for (int i = tid; i < NV; i += 256) ((u32x4_t*)wl)[i] = src[i];
  }
  const int pbase0 = (2 * CSTEP * wave + 2 * lr + (PADL - PADL0)) * 8 + g * 16;
  constexpr int rowb = WLH * 8;
  const int woff = lr * WPITCH + g * 16;
  char* scratch = scratch_all + wave * (16 * PITCH);
  typename TT::elem* outp = (typename TT::elem*)p.out;
  const int j = lane >> 3, part = lane & 7;  // store item: pooled column j of the strip, 8-channel run `part`
  const f32x4_t s0 = *(const f32x4_t*)(p.shift + part * 8), s1 = *(const f32x4_t*)(p.shift + part * 8 + 4);

  // ---- staging, split into issue (global loads into registers) and commit (pack + LDS writes) ----------------
  // fp32: per-image buffer descriptor with 32-bit offsets, colour planes via the scalar offset; pixels outside
  // the image get an out-of-range offset -> the bounds check returns 0.0 (zero padding)
  float fs[(!VEC4 && !U8) ? NIT : 1][6];
  u32x4_t fv[(VEC4 && !U8) ? NIT : 1][3];
  u32x4_t fb[U8 ? NIT : 1];
  unsigned okm = 0;  // U8: bit k = item k lies inside the image (padding must be 0 in NORMALISED space)
  auto tile_geom = [&](int t, int& n, int& py0, int& half) {
    half = t % p.nhalves;
    const int t2 = t / p.nhalves;
    n = __builtin_amdgcn_readfirstlane(t2 / p.rgroups);
    py0 = (t2 - n * p.rgroups) * 3;
  };
  auto issue = [&](int t) {
    int n, py0, half;
    tile_geom(t, n, py0, half);
    const int cr0 = POOL3 ? 2 * py0 - 1 : 2 * py0, ir0 = 2 * cr0 - 3;
    const int colbase = 2 * CSTEP * 4 * half;  // input-column origin of this half (4 strips x CSTEP conv columns x stride 2)
    if (U8) {
      const int img_b = (int)(HW * 3);
      auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x8 + (size_t)n * HW * 3), (short)0, img_b, 0x00020000);
      okm = 0;
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int item = tid + k * 256;
        const bool iv = item < NITEMS;
        const int r = (iv ? item : 0) / NGRP;
        const int ix = ((iv ? item : 0) - r * NGRP) * 4 - (PADL - 1) + colbase;  // a multiple of 4: the group is all in or all out
        const int iy = ir0 + r;
        const bool ok = iv && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        const int o = ok ? (iy * p.Wi + ix) * 3 : 0x7FFFFFF0;
        const auto v3 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, o, 0, 0);
        fb[k] = (u32x4_t){v3[0], v3[1], v3[2], 0u};
        okm |= ok ? (1u << k) : 0u;
      }
    } else if (!VEC4) {
      auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)n * 3 * HW), (short)0, 3 * plane_b, 0x00020000);
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int item = tid + k * 256;
        const bool iv = item < NITEMS;
        const int r = (iv ? item : 0) / (WLH / 2);
        const int ix0 = ((iv ? item : 0) - r * (WLH / 2)) * 2 - PADL + colbase, ix1 = ix0 + 1;
        const int iy = ir0 + r;
        const bool rowok = iv && (unsigned)iy < (unsigned)p.Hi;
        const int rowoff = iy * p.Wi;
        const int o0 = (rowok && (unsigned)ix0 < (unsigned)p.Wi) ? (rowoff + ix0) * 4 : 0x7FFFFFF0;
        const int o1 = (rowok && (unsigned)ix1 < (unsigned)p.Wi) ? (rowoff + ix1) * 4 : 0x7FFFFFF0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          fs[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o0, c * plane_b, 0));
          fs[k][3 + c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o1, c * plane_b, 0));
        }
      }
    } else {
      auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)n * 3 * HW), (short)0, 3 * plane_b, 0x00020000);
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int item = tid + k * 256;
        const bool iv = item < NITEMS;
        const int r = (iv ? item : 0) / NGRP;
        const int ix = ((iv ? item : 0) - r * NGRP) * 4 - (PADL - 1) + colbase;  // a multiple of 4: the group is all in or all out
        const int iy = ir0 + r;
        const bool ok = iv && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        const int o = ok ? (iy * p.Wi + ix) * 4 : 0x7FFFFFF0;
#pragma unroll
        for (int c = 0; c < 3; ++c) fv[k][c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, c * plane_b, 0);
      }
    }
  };
  auto commit = [&]() {
    if (U8) {
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int item = tid + k * 256;
        if (item < NITEMS) {
          const int r = item / NGRP, q = item - r * NGRP;
          char* dst = halo + (r * WLH + 4 * q + 1) * 8;
          const bool ok = (okm >> k) & 1u;
          const unsigned w0 = fb[k][0], w1 = fb[k][1], w2 = fb[k][2];
          // 12 bytes = 4 pixels x RGB: byte j of the item is channel j % 3 of pixel j / 3
          const unsigned bytes[12] = {w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, w0 >> 24,
                                      w1 & 255u, (w1 >> 8) & 255u, (w1 >> 16) & 255u, w1 >> 24,
                                      w2 & 255u, (w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            typename TT::elem px[4] = {lut[bytes[3 * e]], lut[256 + bytes[3 * e + 1]], lut[512 + bytes[3 * e + 2]], TT::from_f32(0.f)};
            u32x2_t w;
            __builtin_memcpy(&w, px, 8);
            if (!ok) w = (u32x2_t){0u, 0u};
            *(u32x2_t*)(dst + e * 8) = w;
          }
        }
      }
    } else if (!VEC4) {
#pragma unroll
      for (int k = 0; k < NIT; ++k)
        if (tid + k * 256 < NITEMS) {
          const float q[8] = {fs[k][0], fs[k][1], fs[k][2], 0.f, fs[k][3], fs[k][4], fs[k][5], 0.f};
          *(u32x4_t*)(halo + (tid + k * 256) * 16) = pack8<TT>(q);
        }
    } else {
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int item = tid + k * 256;
        if (item < NITEMS) {
          const int r = item / NGRP, q = item - r * NGRP;
          char* dst = halo + (r * WLH + 4 * q + 1) * 8;
#pragma unroll
          for (int e = 0; e < 4; ++e)
          {
            // (copy the lane first: __builtin_bit_cast applied directly to a vector-element lvalue reads element 0)
            const unsigned r0 = fv[k][0][e], g0 = fv[k][1][e], b0 = fv[k][2][e];
            *(u32x2_t*)(dst + e * 8) = pack4<TT>(__uint_as_float(r0), __uint_as_float(g0), __uint_as_float(b0), 0.f);
          }
        }
      }
    }
  };

  if (U8) __syncthreads();  // the table is complete before the first commit reads it
  if (PF) issue(tile);
  for (; tile < t_hi; tile += wgs_in_xcd) {
    int n, py0, half;
    tile_geom(tile, n, py0, half);
    const int cr0 = POOL3 ? 2 * py0 - 1 : 2 * py0;
    const int strip = half * 4 + wave;
    if (!PF) issue(tile);
    commit();
    __syncthreads();
    if (PF && tile + wgs_in_xcd < t_hi) issue(tile + wgs_in_xcd);  // the next tile's rows land under this tile's MFMA phase

    const bool active = strip < p.nstrips;
    if (active) {
      f32x4_t acc[MI][NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        vec8 wf[NI], pf[MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(wl + ni * 16 * WPITCH + woff + ks * 64);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) pf[mi] = *(const vec8*)(halo + pbase0 + (2 * mi + ks) * rowb);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
      }
      // mask conv positions outside the image (they act as -inf under the max), pool, store
      const int cc = CSTEP * strip + lr - (POOL3 ? 1 : 0);  // this lane's conv column
      const int cc_lo = CSTEP * strip - (POOL3 ? 1 : 0);     // (wave-uniform) first conv column of the strip
      if (cc_lo < 0 || cc_lo + 15 >= p.Wc || cr0 < 0 || cr0 + MI > p.Hc) {  // only border strips / row groups pay for the masking
        const bool colv = (unsigned)cc < (unsigned)p.Wc;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const bool v = colv && (unsigned)(cr0 + mi) < (unsigned)p.Hc;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mi][ni][e] = v ? acc[mi][ni][e] : -INFINITY;
        }
      }
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          f32x4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float m;
            if (POOL3) {
              m = fmaxf(fmaxf(acc[2 * pr][ni][e], acc[2 * pr + 1][ni][e]), acc[2 * pr + 2][ni][e]);  // rows
              m = fmaxf(m, fmaxf(row_down<1>(m), row_down<2>(m)));                                     // columns
            } else {
              m = fmaxf(acc[2 * pr][ni][e], acc[2 * pr + 1][ni][e]);
              m = fmaxf(m, row_down<1>(m));
            }
            v[e] = m;
          }
          *(f32x4_t*)(scratch + lr * PITCH + ni * 64 + g * 16) = v;  // valid where lr is even (and <= 12 for POOL3)
        }
        const int py = py0 + pr, px = PPS * strip + j;
        const f32x4_t a = *(const f32x4_t*)(scratch + (2 * j) * PITCH + part * 32);
        const f32x4_t b = *(const f32x4_t*)(scratch + (2 * j) * PITCH + part * 32 + 16);
        if (j < PPS && px < p.Wq && py < p.Hq) {
          float o[8] = {a[0] + s0[0], a[1] + s0[1], a[2] + s0[2], a[3] + s0[3],
                        b[0] + s1[0], b[1] + s1[1], b[2] + s1[2], b[3] + s1[3]};
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
          *(u32x4_t*)(outp + (((size_t)n * p.Hq + py) * p.Wq + px) * 64 + part * 8) = pack8<TT>(o);
        }
      }
    }
    __syncthreads();  // staged rows are dead: the next tile may overwrite them
  }
}

static int stem_launch(const float* x_nchw, const unsigned char* x_u8, const float* mean3, const float* std3,
                       const void* w_packed_c3, const float* shift, void* out, int B, int Hi, int Wi,
                       int pool3, int dtype, void* stream, const char* who) {
  FRMAP_REQUIRE((x_nchw || x_u8) && w_packed_c3 && shift && out, "%s: null pointer", who);
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "%s: bad dtype %d", who, dtype);
  FRMAP_REQUIRE(B > 0 && Hi >= 7 && Wi >= 7 && (long long)Hi * Wi * 12 < 0x7FFFFF00ll, "%s: bad input size", who);
  StemPoolParams p;
  p.x = x_nchw; p.x8 = x_u8; p.wpk = w_packed_c3; p.shift = shift; p.out = out;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean3 ? mean3[c] : 0.f; p.std[c] = std3 ? std3[c] : 1.f; }
  if (x_u8) {
    FRMAP_REQUIRE(mean3 && std3 && std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "%s: mean / non-zero std required", who);
    FRMAP_REQUIRE(Wi % 4 == 0 && ((uintptr_t)x_u8 & 3) == 0, "%s: uint8 input needs W %% 4 == 0 and a 4-byte aligned tensor (W=%d)", who, Wi);
  }
  if (pool3) {   // the ResNet stem: second-generation kernel (space-to-depth K axis, carried pool row) where it takes the shape
    const int rc = frmap_stem_s2d(x_nchw, x_u8, p.mean, p.std, w_packed_c3, shift, out, B, Hi, Wi, dtype, (hipStream_t)stream);
    if (rc < 0) return rc;
    if (rc == 1) return 0;
  }
  p.N = B; p.Hi = Hi; p.Wi = Wi;
  p.Hc = (Hi + 6 - 7) / 2 + 1; p.Wc = (Wi + 6 - 7) / 2 + 1;
  if (pool3) { p.Hq = (p.Hc + 2 - 3) / 2 + 1; p.Wq = (p.Wc + 2 - 3) / 2 + 1; }
  else { p.Hq = p.Hc / 2; p.Wq = p.Wc / 2; }
  FRMAP_REQUIRE(p.Hq > 0 && p.Wq > 0, "%s: input too small", who);
  const int pps = pool3 ? 7 : 8;
  p.nstrips = (p.Wq + pps - 1) / pps;
  FRMAP_REQUIRE(p.nstrips <= 8, "%s: input wider than ~224 columns (W=%d); use the unfused path", who, Wi);
  p.rgroups = (p.Hq + 2) / 3;
  p.nhalves = (p.nstrips + 3) / 4;
  const int cstep = pool3 ? 14 : 16, mi = pool3 ? 7 : 6;
  p.Wl = 2 * cstep * 4 + (pool3 ? 10 : 8);
  p.magic_Wl2 = frmap_magic((uint32_t)(p.Wl >> 1));
  const long long nb = (long long)B * p.rgroups * p.nhalves;
  FRMAP_REQUIRE(nb < (1ll << 31), "%s: too many tiles", who);
  const int nrows = 2 * (mi - 1) + 7;
  // 16-byte staging loads when rows are 16-byte aligned (W % 4 == 0, aligned base); the uint8 path uses the same item shape
  const bool vec4 = x_u8 ? true : (Wi % 4 == 0 && ((uintptr_t)x_nchw & 15) == 0);
  const int wlh = vec4 ? 4 * (pool3 ? 32 : 35) + 2 : p.Wl;  // must mirror the kernel's WLH
  const int hb = (nrows * wlh * 8 + 1023) & ~1023;
  p.halo_bytes = hb;
  const int wbytes = 64 * 240 * 2;
  const int scratch = 4 * 16 * (4 * 64 + 16);
  const int lds = hb + wbytes + scratch + 3 * 256 * 2;  // + the uint8 normalisation table
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  // persistent: two 4-wave workgroups per CU walk the tiles (one stages while the other computes)
  const unsigned grid = (unsigned)(nb < 2ll * ncu ? nb : 2ll * ncu);
  hipStream_t st = (hipStream_t)stream;
  typedef void (*kern_t)(const StemPoolParams);
  static const kern_t kerns[14] = {
      stem_pool_kernel<BF16, false, false, false, false>, stem_pool_kernel<BF16, false, true, false, false>,
      stem_pool_kernel<BF16, true, false, false, false>,  stem_pool_kernel<BF16, true, true, false, false>,
      stem_pool_kernel<F16, false, false, false, false>,  stem_pool_kernel<F16, false, true, false, false>,
      stem_pool_kernel<F16, true, false, false, false>,   stem_pool_kernel<F16, true, true, false, false>,
      stem_pool_kernel<BF16, false, true, true, true>,    stem_pool_kernel<BF16, true, true, true, true>,
      stem_pool_kernel<F16, false, true, true, true>,     stem_pool_kernel<F16, true, true, true, true>,
      stem_pool_kernel<BF16, true, true, false, true>,    stem_pool_kernel<F16, true, true, false, true>};
  static int pf32 = -1;  // A/B switch: FRMAP_STEM_PF=1 prefetches the next tile in the fp32 ResNet-stem kernel too
  if (pf32 < 0) { const char* e = getenv("FRMAP_STEM_PF"); pf32 = e ? atoi(e) : 0; }
  int ai = x_u8 ? 8 + (dtype == FRMAP_BF16 ? 0 : 2) + (pool3 ? 1 : 0)
                : (dtype == FRMAP_BF16 ? 0 : 4) + (pool3 ? 2 : 0) + (vec4 ? 1 : 0);
  if (!x_u8 && pf32 && pool3 && vec4) ai = dtype == FRMAP_BF16 ? 12 : 13;
  if (frmap_big_lds((const void*)kerns[ai], 160 * 1024)) return -2;
  hipLaunchKernelGGL(kerns[ai], dim3(grid), dim3(256), lds, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

extern "C" int frmap_stem7x7_maxpool(const float* x_nchw, const void* w_packed_c3, const float* shift, void* out,
                                     int B, int Hi, int Wi, int dtype, void* stream) {
  FRMAP_REQUIRE(x_nchw, "stem7x7_maxpool: null input");
  return stem_launch(x_nchw, nullptr, nullptr, nullptr, w_packed_c3, shift, out, B, Hi, Wi, 1, dtype, stream, "stem7x7_maxpool");
}

extern "C" int frmap_stem7x7_maxpool2(const float* x_nchw, const void* w_packed_c3, const float* shift, void* out,
                                      int B, int Hi, int Wi, int dtype, void* stream) {
  FRMAP_REQUIRE(x_nchw, "stem7x7_maxpool2: null input");
  return stem_launch(x_nchw, nullptr, nullptr, nullptr, w_packed_c3, shift, out, B, Hi, Wi, 0, dtype, stream, "stem7x7_maxpool2");
}

extern "C" int frmap_stem7x7_maxpool_u8(const unsigned char* x_u8_hwc, const float* mean3_host, const float* std3_host,
                                        const void* w_packed_c3, const float* shift, void* out, int B, int Hi, int Wi,
                                        int pool3, int dtype, void* stream) {
  FRMAP_REQUIRE(x_u8_hwc, "stem7x7_maxpool_u8: null input");
  return stem_launch(nullptr, x_u8_hwc, mean3_host, std3_host, w_packed_c3, shift, out, B, Hi, Wi, pool3 ? 1 : 0, dtype, stream,
                     "stem7x7_maxpool_u8");
}
