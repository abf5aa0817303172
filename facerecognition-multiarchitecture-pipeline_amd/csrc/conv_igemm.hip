// Implicit-GEMM convolution for gfx950 (MI355X): NHWC activations, v_mfma_f32_16x16x32 (bf16/f16),
// LDS-staged input halo + LDS-DMA weight tiles, fused shift / residual / ReLU epilogue.
//
// Replaces nn.Conv2d + nn.BatchNorm2d (+ReLU) (+residual) of the reference's backbones
// (/root/reference/src/face_models.py:23-26,39-40 BaselineNet; :121-141 SiameseNet; the torchvision
// ResNet-18 BasicBlocks behind :67,463,658) and, as 1x1 over H=W=1, its wide nn.Linear layers.
//
// Tiling (one workgroup = 256 threads = 4 waves, 2 workgroups per CU when LDS allows):
//   M axis = flattened output pixels (n, oy, ox); a workgroup owns BM consecutive pixels and
//            BN = 64 output channels; wave w owns pixels [w*BM/4, (w+1)*BM/4) x all 64 channels.
//   K axis = (tap, input channel): input channels are walked in chunks of 32; for each chunk the
//            workgroup stages ONCE the input rows its pixels touch ("halo", zero padded) and the
//            KS*KS x 64 x 32 weight slab, then runs KS*KS k-steps of MFMA 16x16x32 out of LDS.
//            A 3x3 tap is just a different LDS offset into the same halo — input bytes come from
//            L2/HBM once per chunk, not nine times.
//   MFMA operand roles: A (rows) = 16 output channels, B (cols) = 16 pixels, so each lane ends
//            up with 4 consecutive output channels of one pixel -> 8-byte NHWC stores.
//   LDS images are unpadded 64 B per pixel (32 channels) with an XOR swizzle of the 16-byte slot
//            chosen per stride so that ds_read_b128 of 16 consecutive pixels is bank-conflict free
//            (stride 1:  slot ^= ((q>>2)&1)<<1 ; stride 2: 256-B row slot ^= ((q>>2)&3)<<1).
//   Weights are pre-packed on the device (pack_weight.hip) in exactly the LDS image order, so the
//            slab is a straight 36 KB copy done with global_load_lds_dwordx4 (no VGPRs).
#include "frmap_common.h"
#include <stdlib.h>

struct ConvParams {
  const void* in;
  const void* wpk;
  const float* shift;
  const void* res;
  void* out;
  int N, Hi, Wi, Cin, Ho, Wo, Cout;
  int stride, pad, relu;
  int M, HoWo, Hp, Wp;
  uint32_t magic_Wp, magic_Hp;
  int nchunks;
  int halo_bytes;
  int nblocks;
  int dbg;  // timing ablations only (FRMAP_CONV_DEBUG): 1 = stage chunk 0 only, 2 = skip the MFMA loop
};

template <int SWZ>
__device__ __forceinline__ int px_off(int q, int slot) {
  if (SWZ == 1) return (q << 6) + ((slot ^ (((q >> 2) & 1) << 1)) << 4);
  return ((q >> 2) << 8) + (((((q & 3) << 2) | slot) ^ (((q >> 2) & 3) << 1)) << 4);
}

template <typename TT, int BM, int KS, int SWZ>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvParams p) {
  constexpr int MI = BM / 64;
  constexpr int NI = 4;
  constexpr int TAPS = KS * KS;
  using vec8 = typename TT::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wl = smem + p.halo_bytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;

  // XCD-aware block remap: blocks b, b+8, b+16.. share an XCD (and its L2); give each XCD a
  // contiguous range of logical tiles so neighbouring pixel tiles / channel tiles share L2 lines.
  int L;
  {
    const int nb = p.nblocks, b = blockIdx.x;
    const int qd = nb >> 3, rm = nb & 7, xcd = b & 7;
    L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (b >> 3);
  }
  const int ntiles = p.Cout >> 6;
  const int mt = L / ntiles, nt = L - mt * ntiles;
  const int m0 = mt * BM;
  const int mlast = min(m0 + BM, p.M) - 1;

  // origin of this tile in the "virtual padded row stack": padded row index G = n*Hp + iy + pad.
  int n0 = 0, rr0 = 0, nrows = 0;
  if (KS > 1) {
    n0 = m0 / p.HoWo;
    const int oy0 = (m0 - n0 * p.HoWo) / p.Wo;
    rr0 = oy0 * p.stride;
    const int n1 = mlast / p.HoWo;
    const int oy1 = (mlast - n1 * p.HoWo) / p.Wo;
    nrows = (n1 - n0) * p.Hp + oy1 * p.stride - rr0 + KS;
  }

  // per-lane LDS byte offsets of the B-operand (pixel) fragments, one per (pixel group, tap)
  int offs[MI][TAPS];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = min(m0 + wave * (BM / 4) + mi * 16 + lr, p.M - 1);
    int qb;
    if (KS == 1) {
      qb = m - m0;
    } else {
      const int n = m / p.HoWo;
      const int rem = m - n * p.HoWo;
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      qb = ((n - n0) * p.Hp + oy * p.stride - rr0) * p.Wp + ox * p.stride;
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) offs[mi][t] = px_off<SWZ>(qb + (t / KS) * p.Wp + (t % KS), g);
  }
  // A-operand (weights) fragment offset inside one tap's 4 KB image
  const int woff = (lr << 6) + ((g ^ (((lr >> 2) & 1) << 1)) << 4);

  // gather-mode (1x1) source element offsets: pixel i of the tile -> input pixel
  size_t gsrc[BM * 4 / 256];
  if (KS == 1) {
#pragma unroll
    for (int it = 0; it < BM * 4 / 256; ++it) {
      const int px = (it * 256 + tid) >> 2;
      const int m = min(m0 + px, p.M - 1);
      const int n = m / p.HoWo;
      const int rem = m - n * p.HoWo;
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      gsrc[it] = (((size_t)n * p.Hi + (size_t)(oy * p.stride)) * p.Wi + (size_t)(ox * p.stride)) * p.Cin +
                 (size_t)((tid & 3) * 8);
    }
  }

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const typename TT::elem* inp = (const typename TT::elem*)p.in;
  const int nitems = nrows * p.Wp * 4;

  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    if (chunk > 0) __syncthreads();  // everyone is done reading the previous chunk's LDS images
    if (!(p.dbg == 1 && chunk > 0)) {
    // ---- weights: straight LDS-DMA copy of the pre-packed slab -------------------------------
    {
      const char* wsrc = (const char*)p.wpk + ((size_t)(nt * p.nchunks + chunk) * TAPS) * 4096 + tid * 16;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + t * 4096),
                                         (__attribute__((address_space(3))) void*)(wl + t * 4096 + wave * 1024),
                                         16, 0, 0);
      }
    }
    // ---- input halo / gathered pixels ---------------------------------------------------------
    if (KS == 1) {
      u32x4_t v[BM * 4 / 256];
#pragma unroll
      for (int it = 0; it < BM * 4 / 256; ++it)
        v[it] = *(const u32x4_t*)(inp + gsrc[it] + (size_t)chunk * 32);
#pragma unroll
      for (int it = 0; it < BM * 4 / 256; ++it) {
        const int px = (it * 256 + tid) >> 2;
        *(u32x4_t*)(halo + px_off<SWZ>(px, tid & 3)) = v[it];
      }
    } else {
      for (int it0 = tid; it0 < nitems; it0 += 256 * 4) {
        u32x4_t v[4];
        int dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int item = it0 + u * 256;
          dst[u] = -1;
          v[u] = (u32x4_t){0u, 0u, 0u, 0u};
          if (item < nitems) {
            const int px = item >> 2, cg = item & 3;
            const int r = (int)fast_div((uint32_t)px, p.magic_Wp);
            const int c = px - r * p.Wp;
            const int rr = rr0 + r;
            const int dn = (int)fast_div((uint32_t)rr, p.magic_Hp);
            const int iy = rr - dn * p.Hp - p.pad;
            const int ix = c - p.pad;
            const int n = n0 + dn;
            dst[u] = px_off<SWZ>(px, cg);
            if (n < p.N && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) {
              const size_t src = (((size_t)n * p.Hi + iy) * p.Wi + ix) * p.Cin + (size_t)(chunk * 32 + cg * 8);
              v[u] = *(const u32x4_t*)(inp + src);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (dst[u] >= 0) *(u32x4_t*)(halo + dst[u]) = v[u];
      }
    }
    }
    __syncthreads();  // (also drains the LDS-DMA: hipcc emits vmcnt(0) ahead of the barrier)
    if (p.dbg == 2) continue;

    // ---- KS*KS k-steps of 32 channels out of LDS -----------------------------------------------
    // Fragment reads run one tap ahead of the MFMAs (two register sets); sched_group_barrier pins
    // the interleave "1 ds_read_b128 : 2 MFMA" so LDS latency hides under the matrix pipe.
    {
      vec8 wf[2][NI], pf[2][MI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[0][ni] = *(const vec8*)(wl + ni * 1024 + woff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) pf[0][mi] = *(const vec8*)(halo + offs[mi][0]);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        if (t + 1 < TAPS) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) wf[nxt][ni] = *(const vec8*)(wl + (t + 1) * 4096 + ni * 1024 + woff);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) pf[nxt][mi] = *(const vec8*)(halo + offs[mi][t + 1]);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = TT::mfma(wf[cur][ni], pf[cur][mi], acc[mi][ni]);
        if (t + 1 < TAPS) {
#pragma unroll
          for (int i = 0; i < NI + MI; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    // 1 DS read
            __builtin_amdgcn_sched_group_barrier(0x008, (MI * NI) / (NI + MI), 0);  // then MFMAs
          }
        }
      }
    }
  }

  // ---- epilogue: + shift (+ residual) (ReLU) -> NHWC, whole-line 16-byte stores via an LDS transpose
  __syncthreads();  // every wave is done reading the staged tiles; LDS is free for the transpose
  conv_epilogue<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), m0 + wave * (BM / 4), p.M, p.Cout, nt << 6,
                            p.shift, (const typename TT::elem*)p.res, (typename TT::elem*)p.out, p.relu, lane);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int halo_rows_bound(int BM, int Ho, int Wo, int Hp, int stride, int KS) {
  // most output rows / image crossings BM consecutive flattened pixels can touch
  const int rows = (BM + Wo - 2) / Wo + 1;
  const int cross = (BM + Ho * Wo - 2) / (Ho * Wo);
  const int cross_step = Hp - (Ho - 1) * stride;  // padded-row jump from last row of n to first of n+1
  const int x = cross < rows - 1 ? cross : rows - 1;
  int gdiff = (rows - 1 - x) * stride + x * (cross_step > stride ? cross_step : stride);
  return gdiff + KS;
}

template <typename TT, int BM, int KS, int SWZ>
static int launch(const ConvParams& p, int lds_bytes, hipStream_t st) {
  auto kern = conv_igemm_kernel<TT, BM, KS, SWZ>;
  static int attr_set = 0;
  if (attr_set < lds_bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      frmap_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return -2;
    }
    attr_set = 160 * 1024;
  }
  const int scratch = 4 * 16 * (4 * 64 + 16);  // epilogue transpose region (4 waves)
  if (lds_bytes < scratch) lds_bytes = scratch;
  hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), lds_bytes, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

extern "C" int frmap_conv_igemm(const void* in, const void* w_packed, const float* shift, const void* residual,
                                void* out, int B, int Hi, int Wi, int Cin, int Cout, int K, int stride, int pad,
                                int relu, int dtype, void* stream) {
  FRMAP_REQUIRE(in && w_packed && shift && out, "conv_igemm: null pointer");
  FRMAP_REQUIRE(K == 1 || K == 3, "conv_igemm: kernel size %d not supported (1 or 3)", K);
  FRMAP_REQUIRE(stride == 1 || stride == 2, "conv_igemm: stride %d not supported", stride);
  FRMAP_REQUIRE((K == 3 && pad == 1) || (K == 1 && pad == 0), "conv_igemm: pad %d with K=%d not supported", pad, K);
  FRMAP_REQUIRE(Cin > 0 && Cin % 32 == 0, "conv_igemm: Cin=%d must be a multiple of 32", Cin);
  FRMAP_REQUIRE(Cout > 0 && Cout % 64 == 0, "conv_igemm: Cout=%d must be a multiple of 64", Cout);
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "conv_igemm: bad dtype %d", dtype);
  FRMAP_REQUIRE(B > 0 && Hi > 0 && Wi > 0, "conv_igemm: empty input");
  const int Ho = (Hi + 2 * pad - K) / stride + 1, Wo = (Wi + 2 * pad - K) / stride + 1;
  FRMAP_REQUIRE(Ho > 0 && Wo > 0, "conv_igemm: empty output");
  const long long Mll = (long long)B * Ho * Wo;
  FRMAP_REQUIRE(Mll < (1ll << 31), "conv_igemm: too many output pixels");
  FRMAP_REQUIRE(Wi + 2 * pad < 32768 && Hi + 2 * pad < 32768, "conv_igemm: image too large");

  ConvParams p;
  p.in = in; p.wpk = w_packed; p.shift = shift; p.res = residual; p.out = out;
  p.N = B; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.stride = stride; p.pad = pad; p.relu = relu;
  p.M = (int)Mll; p.HoWo = Ho * Wo; p.Hp = Hi + 2 * pad; p.Wp = Wi + 2 * pad;
  p.magic_Wp = frmap_magic((uint32_t)p.Wp); p.magic_Hp = frmap_magic((uint32_t)p.Hp);
  p.nchunks = Cin / 32;
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("FRMAP_CONV_DEBUG"); dbg = e ? atoi(e) : 0; }
    p.dbg = dbg;
  }
  hipStream_t st = (hipStream_t)stream;
  const int wbytes = K * K * 4096;
  const int ntiles = Cout / 64;

  if (K == 1) {
    const int BM = 256;
    p.halo_bytes = BM * 64;
    p.nblocks = ((p.M + BM - 1) / BM) * ntiles;
    const int lds = p.halo_bytes + wbytes;
    return dtype == FRMAP_BF16 ? launch<BF16, 256, 1, 1>(p, lds, st) : launch<F16, 256, 1, 1>(p, lds, st);
  }
  // 3x3: pick the pixel tile so the halo fits; prefer 256 pixels
  int BM = 256;
  long long hb = (long long)halo_rows_bound(256, Ho, Wo, p.Hp, stride, 3) * p.Wp * 64;
  if (hb + wbytes > 160 * 1024) {
    BM = 128;
    hb = (long long)halo_rows_bound(128, Ho, Wo, p.Hp, stride, 3) * p.Wp * 64;
  }
  hb = (hb + 1023) & ~1023ll;
  FRMAP_REQUIRE(hb + wbytes <= 160 * 1024, "conv_igemm: input rows too wide for LDS (W=%d)", Wi);
  FRMAP_REQUIRE(hb / 64 < 65536, "conv_igemm: halo too large");
  p.halo_bytes = (int)hb;
  p.nblocks = ((p.M + BM - 1) / BM) * ntiles;
  const int lds = p.halo_bytes + wbytes;
#define FRMAP_DISPATCH(TT)                                                                   \
  (BM == 256 ? (stride == 1 ? launch<TT, 256, 3, 1>(p, lds, st) : launch<TT, 256, 3, 2>(p, lds, st)) \
             : (stride == 1 ? launch<TT, 128, 3, 1>(p, lds, st) : launch<TT, 128, 3, 2>(p, lds, st)))
  return dtype == FRMAP_BF16 ? FRMAP_DISPATCH(BF16) : FRMAP_DISPATCH(F16);
#undef FRMAP_DISPATCH
}
