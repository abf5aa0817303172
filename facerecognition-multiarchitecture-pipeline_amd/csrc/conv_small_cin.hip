// First-layer convolution (Cin = 3, stored as NHWC4) on MFMA, gfx950.
//
// Replaces conv1+bn1+relu of the torchvision ResNet-18 stem (7x7 s2 p3, 3->64) used at
// /root/reference/src/face_models.py:67,463,658, SiameseNet conv.0-2 (:115-117) and BaselineNet
// conv1+bn1+relu (3x3 s1 p1, 3->32; :21-22,38).
//
// With only 3(+1 zero) input channels the (kw, c) axis of one kernel row is CONTIGUOUS in an
// NHWC4 image: 7 taps x 4 channels = 28 -> 32 elements, 3 taps x 4 = 12 -> 16.  So the GEMM K axis
// is laid out as  k = kh*Kr + kw*4 + c  (Kr = 32 or 16, zero weights in the pad positions) and one
// 16-byte LDS read of a lane is 2 neighbouring input pixels.  K = 224 (7 k-steps of 32) for the
// 7x7 stem, 64 (2 k-steps) for the 3x3 first layer.  Same tiling as conv_igemm.hip otherwise.
#include "frmap_common.h"
#include <string.h>

struct SmallCinParams {
  const void* in;
  const void* wpk;
  const float* shift;
  void* out;
  int N, Hi, Wi, Ho, Wo, Cout;
  int stride, pad, relu;
  int M, HoWo, Hp, Wl;  // Wl = staged row length in pixels (even, >= Wi + 2*pad + overrun)
  uint32_t magic_Wl2, magic_Hp;
  int halo_bytes;
  int nblocks;
  FrmapPoolOrder pool;  // POOL = true: pool-major pixel order of the fused 2x2 max-pool
};

// POOL = true (stride 1): pool-major pixel order, the epilogue writes the 2x2-max-pooled map (conv_epilogue_pool2).
template <typename TT, int NI, int KH, int KW, int STRIDE, bool POOL = false>
__global__ __launch_bounds__(256, 2) void conv_small_cin_kernel(const SmallCinParams p) {
  constexpr int BM = 256, MI = 4;
  constexpr int KR = (KW * 4 > 16) ? 32 : 16;
  constexpr int KTOT = KH * KR;
  constexpr int KPAD = (KTOT + 31) / 32 * 32;
  constexpr int KSTEPS = KPAD / 32;
  constexpr int WPITCH = (KPAD + 16) * 2;  // bytes per output channel in the weight image: 160 / 480 B keep the ds_read_b128 of a
                                           // lane group (16 channels x this k-group + its neighbours) on distinct banks (144 / 464 B: 2-way)
  constexpr int COUT = NI * 16;
  using vec8 = typename TT::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wl = smem + p.halo_bytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * BM;
  const int mlast = min(m0 + BM, p.M) - 1;

  int n0, oy0, n1, oy1;
  if (POOL) {
    int ox;
    frmap_pool_coords(m0, p.pool, n0, oy0, ox);
    frmap_pool_coords(mlast, p.pool, n1, oy1, ox);
  } else {
    n0 = m0 / p.HoWo;
    oy0 = (m0 - n0 * p.HoWo) / p.Wo;
    n1 = mlast / p.HoWo;
    oy1 = (mlast - n1 * p.HoWo) / p.Wo;
  }
  const int rr0 = oy0 * STRIDE;
  const int nrows = (n1 - n0) * p.Hp + oy1 * STRIDE - rr0 + KH;

  // ---- stage weights (whole [COUT][KPAD+16] image, already in this order in global) -------------
  {
    const u32x4_t* src = (const u32x4_t*)p.wpk;
    constexpr int NV = COUT * WPITCH / 16;
    for (int i = tid; i < NV; i += 256) ((u32x4_t*)wl)[i] = src[i];
  }
  // ---- stage input rows: item = 2 pixels (16 B) ---------------------------------------------
  {
    const typename TT::elem* inp = (const typename TT::elem*)p.in;
    const int wl2 = p.Wl >> 1;
    const int nitems = nrows * wl2;
    for (int it0 = tid; it0 < nitems; it0 += 256 * 4) {
      u32x2_t v[4][2];
      int dst[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int item = it0 + u * 256;
        dst[u] = -1;
        v[u][0] = (u32x2_t){0u, 0u};
        v[u][1] = (u32x2_t){0u, 0u};
        if (item < nitems) {
          const int r = (int)fast_div((uint32_t)item, p.magic_Wl2);
          const int c = (item - r * wl2) * 2;
          const int rr = rr0 + r;
          const int dn = (int)fast_div((uint32_t)rr, p.magic_Hp);
          const int iy = rr - dn * p.Hp - p.pad;
          const int n = n0 + dn;
          dst[u] = item * 16;
          if (n < p.N && (unsigned)iy < (unsigned)p.Hi) {
            const size_t rowbase = ((size_t)n * p.Hi + iy) * p.Wi;
            const int ix0 = c - p.pad, ix1 = ix0 + 1;
            if ((unsigned)ix0 < (unsigned)p.Wi) v[u][0] = *(const u32x2_t*)(inp + (rowbase + ix0) * 4);
            if ((unsigned)ix1 < (unsigned)p.Wi) v[u][1] = *(const u32x2_t*)(inp + (rowbase + ix1) * 4);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (dst[u] >= 0) {
          u32x4_t w = {v[u][0][0], v[u][0][1], v[u][1][0], v[u][1][1]};
          *(u32x4_t*)(halo + dst[u]) = w;
        }
    }
  }

  // per-lane fragment offsets
  int pbase[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = min(m0 + wave * 64 + mi * 16 + lr, p.M - 1);
    int n, oy, ox;
    if (POOL) {
      frmap_pool_coords(m, p.pool, n, oy, ox);
    } else {
      n = m / p.HoWo;
      const int rem = m - n * p.HoWo;
      oy = rem / p.Wo;
      ox = rem - oy * p.Wo;
    }
    pbase[mi] = (((n - n0) * p.Hp + oy * STRIDE - rr0) * p.Wl + ox * STRIDE) * 8;
  }
  int koff[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int e = ks * 32 + g * 8;
    int kh = e / KR;
    const int off = e - kh * KR;
    if (kh >= KH) kh = 0;  // zero-weight pad k-groups: read any staged (finite) data
    koff[ks] = kh * p.Wl * 8 + off * 2;
  }
  const int woff = lr * WPITCH + g * 16;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  __syncthreads();

#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    vec8 wf[NI], pf[MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(wl + ni * 16 * WPITCH + woff + ks * 64);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const char* a = halo + pbase[mi] + koff[ks];
      if (STRIDE == 2) {
        pf[mi] = *(const vec8*)a;  // 16-byte aligned: even column, even row length
      } else {
        u32x2_t lo = *(const u32x2_t*)a, hi = *(const u32x2_t*)(a + 8);
        u32x4_t w = {lo[0], lo[1], hi[0], hi[1]};
        __builtin_memcpy(&pf[mi], &w, 16);
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
  }

  __syncthreads();  // LDS tiles are dead: reuse them for the store transpose
  if (POOL)
    conv_epilogue_pool2<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), (m0 + wave * 64) >> 2, p.M >> 2, COUT, 0, p.shift,
                                    (typename TT::elem*)p.out, p.relu, lane);
  else
    conv_epilogue<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), m0 + wave * 64, p.M, COUT, 0, p.shift,
                              (const typename TT::elem*)nullptr, (typename TT::elem*)p.out, p.relu, lane);
}

static int small_rows_bound(int BM, int Ho, int Wo, int Hp, int stride, int KH) {
  const int rows = (BM + Wo - 2) / Wo + 1;
  const int cross = (BM + Ho * Wo - 2) / (Ho * Wo);
  const int cross_step = Hp - (Ho - 1) * stride;
  const int x = cross < rows - 1 ? cross : rows - 1;
  return (rows - 1 - x) * stride + x * (cross_step > stride ? cross_step : stride) + KH;
}

extern "C" int frmap_small_cin_kpad(int KH, int KW) {
  const int kr = (KW * 4 > 16) ? 32 : 16;
  return (KH * kr + 31) / 32 * 32 + 16;
}

static int small_pool_rows_bound(int BM, int Ho, int Wo, int Hp, int KH) {
  const int nw = BM / 4, Wo2 = Wo / 2, Win = (Ho / 2) * Wo2;
  const int pairs = (nw + Wo2 - 2) / Wo2 + 1, cross = (nw + Win - 2) / Win;
  const int x = cross < pairs - 1 ? cross : pairs - 1;
  return 2 * pairs + (Hp - Ho) * x + KH - 1;
}

template <typename TT, int NI, int KH, int KW, int STRIDE, bool POOL = false>
static int launch_small(SmallCinParams& p, hipStream_t st) {
  constexpr int KR = (KW * 4 > 16) ? 32 : 16;
  constexpr int KPAD = (KH * KR + 31) / 32 * 32;
  const int wbytes = NI * 16 * (KPAD + 16) * 2;
  long long hb = (long long)(POOL ? small_pool_rows_bound(256, p.Ho, p.Wo, p.Hp, KH) : small_rows_bound(256, p.Ho, p.Wo, p.Hp, STRIDE, KH)) * p.Wl * 8;
  hb = (hb + 1023) & ~1023ll;
  FRMAP_REQUIRE(hb + wbytes <= 160 * 1024, "conv_small_cin: rows too wide for LDS (W=%d)", p.Wi);
  p.halo_bytes = (int)hb;
  auto kern = conv_small_cin_kernel<TT, NI, KH, KW, STRIDE, POOL>;
  if (frmap_big_lds((const void*)kern, 160 * 1024)) return -2;
  int lds = (int)hb + wbytes;
  const int scratch = 4 * 16 * (NI * 64 + 16);  // epilogue transpose region (4 waves)
  if (lds < scratch) lds = scratch;
  hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), lds, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

static int small_cin_impl(const void* in_nhwc4, const void* w_packed, const float* shift, void* out,
                          int B, int Hi, int Wi, int Cout, int KH, int KW, int stride, int pad,
                          int relu, int dtype, bool pool, void* stream) {
  FRMAP_REQUIRE(in_nhwc4 && w_packed && shift && out, "conv_small_cin: null pointer");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "conv_small_cin: bad dtype %d", dtype);
  const bool stem7 = (KH == 7 && KW == 7 && stride == 2 && pad == 3 && Cout == 64);
  const bool first3 = (KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cout == 32);
  FRMAP_REQUIRE(stem7 || first3, "conv_small_cin: unsupported geometry k=%dx%d s=%d p=%d Cout=%d", KH, KW, stride,
                pad, Cout);
  FRMAP_REQUIRE(B > 0 && Hi > 0 && Wi > 0, "conv_small_cin: bad input size");
  SmallCinParams p;
  p.in = in_nhwc4; p.wpk = w_packed; p.shift = shift; p.out = out;
  p.N = B; p.Hi = Hi; p.Wi = Wi; p.Cout = Cout;
  p.Ho = (Hi + 2 * pad - KH) / stride + 1; p.Wo = (Wi + 2 * pad - KW) / stride + 1;
  FRMAP_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv_small_cin: empty output");
  p.stride = stride; p.pad = pad; p.relu = relu;
  const long long Mll = (long long)B * p.Ho * p.Wo;
  FRMAP_REQUIRE(Mll < (1ll << 31), "conv_small_cin: too many output pixels");
  p.M = (int)Mll; p.HoWo = p.Ho * p.Wo; p.Hp = Hi + 2 * pad;
  const int kr4 = (KW * 4 > 16) ? 8 : 4;
  p.Wl = (Wi + 2 * pad + (kr4 - KW) + 1) & ~1;
  FRMAP_REQUIRE(p.Wl < 65536 && p.Hp < 32768, "conv_small_cin: image too large");
  p.magic_Wl2 = frmap_magic((uint32_t)(p.Wl >> 1)); p.magic_Hp = frmap_magic((uint32_t)p.Hp);
  p.nblocks = (p.M + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  memset(&p.pool, 0, sizeof(p.pool));
  if (pool) {
    FRMAP_REQUIRE(first3 && p.Ho % 2 == 0 && p.Wo % 2 == 0 && (relu == 0 || relu == 1),
                  "conv_small_cin_pool2: needs the 3x3 s1 p1 3->32 layer with even H and W (H=%d W=%d)", Hi, Wi);
    p.pool.Wo2 = p.Wo / 2; p.pool.Win = (p.Ho / 2) * (p.Wo / 2);
    p.pool.dWo2 = frmap_div_make((uint32_t)p.pool.Wo2); p.pool.dWin = frmap_div_make((uint32_t)p.pool.Win);
    return dtype == FRMAP_BF16 ? launch_small<BF16, 2, 3, 3, 1, true>(p, st) : launch_small<F16, 2, 3, 3, 1, true>(p, st);
  }
  if (stem7) return dtype == FRMAP_BF16 ? launch_small<BF16, 4, 7, 7, 2>(p, st) : launch_small<F16, 4, 7, 7, 2>(p, st);
  return dtype == FRMAP_BF16 ? launch_small<BF16, 2, 3, 3, 1>(p, st) : launch_small<F16, 2, 3, 3, 1>(p, st);
}

extern "C" int frmap_conv_small_cin(const void* in_nhwc4, const void* w_packed, const float* shift, void* out,
                                    int B, int Hi, int Wi, int Cout, int KH, int KW, int stride, int pad,
                                    int relu, int dtype, void* stream) {
  return small_cin_impl(in_nhwc4, w_packed, shift, out, B, Hi, Wi, Cout, KH, KW, stride, pad, relu, dtype, false, stream);
}

// BaselineNet conv1 + bn1 + ReLU + MaxPool2d(2, 2) (face_models.py:38) in one launch: out = [B][Hi/2][Wi/2][32]
extern "C" int frmap_conv_small_cin_pool2(const void* in_nhwc4, const void* w_packed, const float* shift, void* out,
                                          int B, int Hi, int Wi, int Cout, int relu, int dtype, void* stream) {
  return small_cin_impl(in_nhwc4, w_packed, shift, out, B, Hi, Wi, Cout, 3, 3, 1, 1, relu, dtype, true, stream);
}
