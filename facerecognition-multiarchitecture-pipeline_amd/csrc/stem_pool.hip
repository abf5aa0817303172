// Fused ResNet stem for gfx950: fp32 NCHW input -> conv 7x7 s2 p3 (3->64) + folded BN + ReLU ->
// maxpool 3x3 s2 p1 -> NHWC B x Hq x Wq x 64 in bf16/f16, ONE kernel.
//
// Replaces `conv1 -> bn1 -> relu -> maxpool` of the torchvision ResNet-18 behind
// /root/reference/src/face_models.py:67,463,658 together with the NCHW->NHWC cast.  Unfused, the
// 64 x 112 x 112 conv output (1.6 MB/face in bf16) makes a round trip through HBM just to be
// reduced 4x by the pool; fused, only the 602 KB fp32 input is read and the 401 KB pooled map
// is written per face.
//
// Work decomposition (512 threads = 8 waves, one workgroup per (image, 3 pooled rows)):
//   the workgroup computes conv rows cr0 = 2*py0-1 .. cr0+6 (7 rows -> 3 pooled rows) at full width;
//   wave s owns the column strip cc = 14s-1 .. 14s+14 (16 conv columns -> 7 pooled columns; strips
//   advance by 14 so every 3-wide pooling window lies inside one strip: 8 strips = 112 columns);
//   so a lane (lr = column in strip, g = k-group) keeps a 7-row x 1-column x 16-channel patch in
//   accumulators: the vertical max is register-local, the horizontal max is two 16-lane shuffles,
//   and the result goes out through the shared LDS-transposed whole-line store.
//   K axis as in conv_small_cin.hip: k = kh*32 + kw*4 + c (7 k-steps of MFMA 16x16x32).
// Recompute: 7/6 rows x 16/14 columns = 1.33x the stem's MACs (6.5 % of the network), paid to drop
// 3.2 MB/face of HBM traffic and two launches.
#include "frmap_common.h"

struct StemPoolParams {
  const float* x;
  const void* wpk;
  const float* shift;
  void* out;
  int N, Hi, Wi, Hc, Wc, Hq, Wq;
  int Wl;        // staged row length in pixels (even)
  int nstrips;   // ceil(Wq / 7) <= 8
  int rgroups;   // ceil(Hq / 3)
  uint32_t magic_Wl2;
  int halo_bytes;
};

template <typename TT>
__global__ __launch_bounds__(512, 1) void stem_pool_kernel(const StemPoolParams p) {
  constexpr int MI = 7, NI = 4, KSTEPS = 7, KPAD = 224, WPITCH = (KPAD + 8) * 2, PADL = 5, NROWS = 19;
  using vec8 = typename TT::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wl = smem + p.halo_bytes;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int n = blockIdx.x / p.rgroups, rg = blockIdx.x - n * p.rgroups;
  const int py0 = rg * 3;
  const int cr0 = 2 * py0 - 1;   // first conv row of the tile
  const int ir0 = 2 * cr0 - 3;   // first input row of the tile

  // ---- weights: [64][232] image (same packing as conv_small_cin) --------------------------------
  {
    const u32x4_t* src = (const u32x4_t*)p.wpk;
    constexpr int NV = 64 * WPITCH / 16;
    for (int i = tid; i < NV; i += 512) ((u32x4_t*)wl)[i] = src[i];
  }
  // ---- input rows straight from fp32 NCHW: item = 2 pixels -> NHWC4 pair (16 B) ------------------
  {
    const size_t HW = (size_t)p.Hi * p.Wi;
    const float* xb = p.x + (size_t)n * 3 * HW;
    const int wl2 = p.Wl >> 1;
    const int nitems = NROWS * wl2;
    for (int item = tid; item < nitems; item += 512) {
      const int r = (int)fast_div((uint32_t)item, p.magic_Wl2);
      const int c = (item - r * wl2) * 2;
      const int iy = ir0 + r;
      float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if ((unsigned)iy < (unsigned)p.Hi) {
        const float* row = xb + (size_t)iy * p.Wi;
        const int ix0 = c - PADL, ix1 = ix0 + 1;
        if ((unsigned)ix0 < (unsigned)p.Wi) { f[0] = row[ix0]; f[1] = row[HW + ix0]; f[2] = row[2 * HW + ix0]; }
        if ((unsigned)ix1 < (unsigned)p.Wi) { f[4] = row[ix1]; f[5] = row[HW + ix1]; f[6] = row[2 * HW + ix1]; }
      }
      *(u32x4_t*)(halo + item * 16) = pack8<TT>(f);
    }
  }

  const bool active = wave < p.nstrips;
  const int cc = 14 * wave - 1 + lr;  // this lane's conv column
  int koff[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) koff[ks] = ks * p.Wl * 8 + g * 16;
  const int pbase0 = (28 * wave + 2 * lr) * 8;  // (2*cc + 2) pixels from the row start
  const int woff = lr * WPITCH + g * 16;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  __syncthreads();

  if (active) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      vec8 wf[NI], pf[MI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(wl + ni * 16 * WPITCH + woff + ks * 64);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) pf[mi] = *(const vec8*)(halo + pbase0 + 2 * mi * p.Wl * 8 + koff[ks]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
    }
  }
  __syncthreads();  // staged tiles are dead; LDS becomes the per-wave store-transpose scratch
  if (!active) return;

  // ---- mask conv positions outside the image (they behave as -inf under the max), pool -----------
  const bool colv = (unsigned)cc < (unsigned)p.Wc;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const bool v = colv && (unsigned)(cr0 + mi) < (unsigned)p.Hc;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mi][ni][e] = v ? acc[mi][ni][e] : -INFINITY;
  }
  constexpr int PITCH = NI * 64 + 16;
  char* scratch = smem + wave * (16 * PITCH);
  typename TT::elem* outp = (typename TT::elem*)p.out;
  const int j = lane >> 3, part = lane & 7;  // store item: pooled column j of the strip, 8-channel run `part`
  const f32x4_t s0 = *(const f32x4_t*)(p.shift + part * 8), s1 = *(const f32x4_t*)(p.shift + part * 8 + 4);
#pragma unroll
  for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      f32x4_t v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float m = fmaxf(fmaxf(acc[2 * pr][ni][e], acc[2 * pr + 1][ni][e]), acc[2 * pr + 2][ni][e]);   // rows
        m = fmaxf(m, fmaxf(__shfl_down(m, 1, 16), __shfl_down(m, 2, 16)));                                // columns
        v[e] = m;
      }
      *(f32x4_t*)(scratch + lr * PITCH + ni * 64 + g * 16) = v;  // valid where lr is even and <= 12
    }
    const int py = py0 + pr, px = 7 * wave + j;
    const f32x4_t a = *(const f32x4_t*)(scratch + (2 * j) * PITCH + part * 32);
    const f32x4_t b = *(const f32x4_t*)(scratch + (2 * j) * PITCH + part * 32 + 16);
    if (j < 7 && px < p.Wq && py < p.Hq) {
      float o[8] = {a[0] + s0[0], a[1] + s0[1], a[2] + s0[2], a[3] + s0[3],
                    b[0] + s1[0], b[1] + s1[1], b[2] + s1[2], b[3] + s1[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
      *(u32x4_t*)(outp + (((size_t)n * p.Hq + py) * p.Wq + px) * 64 + part * 8) = pack8<TT>(o);
    }
  }
}

extern "C" int frmap_stem7x7_maxpool(const float* x_nchw, const void* w_packed_c3, const float* shift, void* out,
                                     int B, int Hi, int Wi, int dtype, void* stream) {
  FRMAP_REQUIRE(x_nchw && w_packed_c3 && shift && out, "stem7x7_maxpool: null pointer");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "stem7x7_maxpool: bad dtype %d", dtype);
  FRMAP_REQUIRE(B > 0 && Hi >= 7 && Wi >= 7, "stem7x7_maxpool: bad input size");
  StemPoolParams p;
  p.x = x_nchw; p.wpk = w_packed_c3; p.shift = shift; p.out = out;
  p.N = B; p.Hi = Hi; p.Wi = Wi;
  p.Hc = (Hi + 6 - 7) / 2 + 1; p.Wc = (Wi + 6 - 7) / 2 + 1;
  p.Hq = (p.Hc + 2 - 3) / 2 + 1; p.Wq = (p.Wc + 2 - 3) / 2 + 1;
  p.nstrips = (p.Wq + 6) / 7;
  FRMAP_REQUIRE(p.nstrips <= 8, "stem7x7_maxpool: input wider than 224+ columns (W=%d); use the unfused path", Wi);
  p.rgroups = (p.Hq + 2) / 3;
  p.Wl = 28 * p.nstrips + 10;
  p.magic_Wl2 = frmap_magic((uint32_t)(p.Wl >> 1));
  const long long nb = (long long)B * p.rgroups;
  FRMAP_REQUIRE(nb < (1ll << 31), "stem7x7_maxpool: too many tiles");
  int hb = 19 * p.Wl * 8;
  hb = (hb + 1023) & ~1023;
  p.halo_bytes = hb;
  const int wbytes = 64 * 232 * 2;
  int lds = hb + wbytes;
  const int scratch = 8 * 16 * (4 * 64 + 16);
  if (lds < scratch) lds = scratch;
  hipStream_t st = (hipStream_t)stream;
  static bool attr[2] = {false, false};
  const void* kern = dtype == FRMAP_BF16 ? (const void*)stem_pool_kernel<BF16> : (const void*)stem_pool_kernel<F16>;
  if (!attr[dtype]) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      frmap_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return -2;
    }
    attr[dtype] = true;
  }
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(stem_pool_kernel<BF16>, dim3((unsigned)nb), dim3(512), lds, st, p);
  else
    hipLaunchKernelGGL(stem_pool_kernel<F16>, dim3((unsigned)nb), dim3(512), lds, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}
