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
#include <string.h>

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
  FrmapDiv dHoWo, dWo;  // exact divisions of pixel indices < 2^31 by HoWo / Wo (frmap_div)
  int nchunks;
  int halo_bytes;
  int nblocks;
  int ksplit;    // >1: the chunk loop is split over ksplit workgroups per tile, fp32 partials go to `slab`
  float* slab;   // [ksplit][M][Cout] fp32 partial sums (split-K only)
  int dbg;  // timing ablations only (FRMAP_CONV_DEBUG): 1 = stage chunk 0 only, 2 = skip the MFMA loop
  // fused 1x1 projection shortcut (conv3x3_fast_kernel<TT, true>): out += W_ds . x_ds[n, oy*s, ox*s, :]
  const void* ds_in;   // [N][ds_Hi][ds_Wi][ds_Cin]
  const void* ds_w;    // packed like a 1x1 conv: [Cout/64][ds_Cin/32][1][64][4][8]
  int ds_Hi, ds_Wi, ds_Cin, ds_stride, ds_chunks;
  FrmapPoolOrder pool;  // conv_igemm_kernel<..., POOL = true>: pool-major pixel order of the fused 2x2 max-pool
  // conv1x1_kernel<..., MATCH = true>: the GEMM is probes x gallery rows, the epilogue keeps each probe's arg-min distance
  const float* m_stat_a;           // [M][4] = (sum a^2, sum a, 1 / row scale, error band) of the fp32 probes
  const float* m_stat_w;           // [G][4] of the fp32 gallery rows
  MatchRec* m_recs;                // [Cout / 64][M] candidate records (frmap_common.h), one writer each
  int m_G, m_D;                    // real gallery rows (Cout is padded to 64), embedding width
};

// 64 zero bytes: out-of-image (padding) pixels LOAD from here instead of branching around the load —
// a per-element `if (valid) v = load` makes hipcc serialise the loads with a wait each.
__device__ __attribute__((aligned(64))) unsigned int g_zero_page[16];

__device__ __forceinline__ int xcd_remap_fwd(int b, int nb) {
  const int qd = nb >> 3, rm = nb & 7, xcd = b & 7;
  return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (b >> 3);
}

template <int SWZ>
__device__ __forceinline__ int px_off(int q, int slot) {
  if (SWZ == 1) return (q << 6) + ((slot ^ (((q >> 2) & 1) << 1)) << 4);
  return ((q >> 2) << 8) + (((((q & 3) << 2) | slot) ^ (((q >> 2) & 3) << 1)) << 4);
}

// POOL = true (3x3 stride 1 only): pixels are enumerated in pool-major order (frmap_pool_coords) and the epilogue writes
// the 2x2-max-pooled map [N][Ho/2][Wo/2][Cout] instead of the conv output.
template <typename TT, int BM, int KS, int SWZ, bool POOL = false>
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
  const int kslice = L % p.ksplit;  // split-K: which slice of the chunk loop this workgroup owns
  const int Lt = L / p.ksplit;
  const int mt = Lt / ntiles, nt = Lt - mt * ntiles;
  const int m0 = mt * BM;
  const int mlast = min(m0 + BM, p.M) - 1;
  const int chunk_lo = (int)(((long long)kslice * p.nchunks) / p.ksplit);
  const int chunk_hi = (int)(((long long)(kslice + 1) * p.nchunks) / p.ksplit);

  // origin of this tile in the "virtual padded row stack": padded row index G = n*Hp + iy + pad.
  int n0 = 0, rr0 = 0, nrows = 0;
  if (POOL) {  // first pixel = top-left of the first window, last pixel = bottom-right of the last one
    int oy0, ox0, n1, oy1, ox1;
    frmap_pool_coords(m0, p.pool, n0, oy0, ox0);
    frmap_pool_coords(mlast, p.pool, n1, oy1, ox1);
    rr0 = oy0;
    nrows = (n1 - n0) * p.Hp + oy1 - rr0 + KS;
  } else if (KS > 1) {
    n0 = frmap_div(m0, p.dHoWo);
    const int oy0 = frmap_div(m0 - n0 * p.HoWo, p.dWo);
    rr0 = oy0 * p.stride;
    const int n1 = frmap_div(mlast, p.dHoWo);
    const int oy1 = frmap_div(mlast - n1 * p.HoWo, p.dWo);
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
    } else if (POOL) {
      int n, oy, ox;
      frmap_pool_coords(m, p.pool, n, oy, ox);
      qb = ((n - n0) * p.Hp + oy - rr0) * p.Wp + ox;
    } else {
      const int n = frmap_div(m, p.dHoWo);
      const int rem = m - n * p.HoWo;
      const int oy = frmap_div(rem, p.dWo);
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
      const int n = frmap_div(m, p.dHoWo);
      const int rem = m - n * p.HoWo;
      const int oy = frmap_div(rem, p.dWo);
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

  for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
    if (chunk > chunk_lo) __syncthreads();  // everyone is done reading the previous chunk's LDS images
    if (!(p.dbg == 1 && chunk > chunk_lo)) {
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
      constexpr int NB2 = 8;
      for (int it0 = tid; it0 < nitems; it0 += 256 * NB2) {
        u32x4_t v[NB2];
        int dst[NB2];
#pragma unroll
        for (int u = 0; u < NB2; ++u) {
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
            const bool ok = n < p.N && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const size_t src = (((size_t)n * p.Hi + iy) * p.Wi + ix) * p.Cin + (size_t)(chunk * 32 + cg * 8);
            const typename TT::elem* a = ok ? inp + src : (const typename TT::elem*)g_zero_page;
            v[u] = *(const u32x4_t*)a;  // unconditional load (see g_zero_page)
          }
        }
#pragma unroll
        for (int u = 0; u < NB2; ++u)
          if (dst[u] >= 0) *(u32x4_t*)(halo + dst[u]) = v[u];
      }
    }
    }
    __syncthreads();  // (also drains the LDS-DMA: hipcc emits vmcnt(0) ahead of the barrier)
    if (p.dbg == 2) continue;

    // ---- KS*KS k-steps of 32 channels out of LDS -----------------------------------------------
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      vec8 wf[NI], pf[MI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(wl + t * 4096 + ni * 1024 + woff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        pf[mi] = *(const vec8*)(halo + offs[mi][t]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
    }
  }

  // ---- epilogue: + shift (+ residual) (ReLU) -> NHWC, whole-line 16-byte stores via an LDS transpose
  __syncthreads();  // every wave is done reading the staged tiles; LDS is free for the transpose
  if (POOL)
    conv_epilogue_pool2<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), (m0 + wave * (BM / 4)) >> 2, p.M >> 2, p.Cout,
                                    nt << 6, p.shift, (typename TT::elem*)p.out, p.relu, lane);
  else if (p.ksplit > 1)
    conv_epilogue_partial<MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), m0 + wave * (BM / 4), p.M, p.Cout, nt << 6,
                                  p.slab + (size_t)kslice * p.M * p.Cout, lane);
  else
    conv_epilogue<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), m0 + wave * (BM / 4), p.M, p.Cout, nt << 6,
                              p.shift, (const typename TT::elem*)p.res, (typename TT::elem*)p.out, p.relu, lane);
}

// ================================================================================================
// 3x3 stride-1 convolutions whose halo is <= 10 16-byte pieces per thread (every stride-1 layer of the
// ResNet / Siamese / Baseline trunks): register-prefetch, software-pipelined.
//
//  * Two workgroups per CU, one tile each.  The NEXT 32-channel chunk's halo pixels and weight slab
//    (19 pieces per thread) are loaded into registers while the current chunk's MFMAs run, and
//    written to LDS at the next chunk boundary (issue early / write late).
//  * Loads are raw buffer loads: padding pixels carry an out-of-range offset and read 0; an EMPTY
//    descriptor turns the prefetch slots of the last chunk into no-ops without a second code path.
//    (A persistent variant that also prefetched the workgroup's next tile measured no faster - a
//    tile's prologue is already covered by the co-resident workgroup - and lost the hardware's
//    dynamic tile placement: layer3 80 us vs 76 us.)
//  * The 19 loads ride between the MFMAs (one after every ~2 groups of 4): issued as one burst they
//    stall the wave ~1400 cycles on the 64 B/clk vector-memory path.
//  * Fragment pipeline pinned with sched_barriers: pixel fragments of tap t+1 are requested before
//    tap t's MFMAs, weight fragment ni of tap t+1 right after tap t's MFMAs on it (left alone the
//    scheduler sinks every ds_read to just before its use and each group of 4 MFMAs waits out an
//    LDS round trip).  The 36 per-tap pixel addresses are recomputed (3 VALU ops, under the MFMAs).
// ================================================================================================
// DS = true additionally folds a ResNet projection shortcut into the layer: out = act(conv3x3(in) + W_ds . x_ds(strided)
// + shift), i.e. BasicBlock.conv2 + bn2 + downsample(conv1x1 s2 + bn) + add + ReLU in one launch.  The shortcut's K
// dimension is run as extra one-tap stages after the first ds_chunks main chunks (<= nchunks of them): their 4
// gathered-pixel pieces + 1 weight piece per thread are prefetched next to the main chunk's 19, written into the same
// LDS images once the main chunk's MFMAs are done, and cost 16 MFMAs + two barriers each.  Saves the 1x1 launch, its
// output round trip through HBM and the residual read.
template <typename TT, bool DS>
__global__ __launch_bounds__(256, 2) void conv3x3_fast_kernel(const ConvParams p) {
  constexpr int MI = 4, NI = 4, TAPS = 9, NB = 10, ND = DS ? 5 : 0, NL = NB + TAPS + ND, NG = TAPS * NI;
  using vec8 = typename TT::vec8;
  using elem = typename TT::elem;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wl = smem + p.halo_bytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int ntiles = p.Cout >> 6;
  const int wslab = p.nchunks * (TAPS * 4096);
  const int woff = (lr << 6) + ((g ^ (((lr >> 2) & 1) << 1)) << 4);
  const elem* inp = (const elem*)p.in;
  const size_t total_el = (size_t)p.N * p.Hi * p.Wi * p.Cin;

  // ---- load-side state: the tile whose staging data is being fetched
  unsigned soff[NB];
  u32x4_t hv[NB], wv[TAPS];
  int l_nitems = 0, l_n0 = 0, l_rr0 = 0, in_records = 0;
  const elem* in_base = inp;
  const char* w_base = (const char*)p.wpk;
  auto setup_load = [&](int L) {
    const int mt = L / ntiles, nt = L - mt * ntiles;
    const int m0 = mt << 8, mlast = min(m0 + 256, p.M) - 1;
    const int n0 = frmap_div(m0, p.dHoWo), oy0 = frmap_div(m0 - n0 * p.HoWo, p.dWo);
    const int n1 = frmap_div(mlast, p.dHoWo), oy1 = frmap_div(mlast - n1 * p.HoWo, p.dWo);
    const int nrows = (n1 - n0) * p.Hp + oy1 - oy0 + 3;
    l_nitems = nrows * p.Wp * 4; l_n0 = n0; l_rr0 = oy0;
    const size_t base_el = (size_t)n0 * p.Hi * p.Wi * p.Cin;  // image n0: tile-relative byte offsets fit 31 bits
    const size_t rem = (total_el - base_el) * sizeof(elem);
    in_base = inp + base_el;
    in_records = (int)(rem < 0x7FFFFFFFull ? rem : 0x7FFFFFFFull);
    w_base = (const char*)p.wpk + (size_t)nt * wslab;
    const int tl = tid;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int item = u * 256 + tl;
      const int px = item >> 2, cg = item & 3;
      const int r = (int)fast_div((uint32_t)px, p.magic_Wp);
      const int c = px - r * p.Wp;
      const int rr = oy0 + r;
      const int dn = (int)fast_div((uint32_t)rr, p.magic_Hp);
      const int iy = rr - dn * p.Hp - 1, ix = c - 1, n = n0 + dn;
      const bool ok = item < l_nitems && n < p.N && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      soff[u] = ok ? ((unsigned)((dn * p.Hi + iy) * p.Wi + ix) * (unsigned)p.Cin + (unsigned)(cg * 8)) * (unsigned)sizeof(elem)
                   : 0xFFFFFF00u;
    }
  };
  __amdgpu_buffer_rsrc_t rs_in, rs_w;
  auto set_rsrc = [&](bool live) {
    rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in_base, (short)0, live ? in_records : 0, 0x00020000);
    rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, (short)0, live ? wslab : 0, 0x00020000);
  };
  // piece j of a stage: j < NB halo pixels, then the 9 weight taps (already in LDS-image order)
  int pso = 0;  // scalar byte offset of the halo loads (the chunk's 32 channels)
  // shortcut stage dc (DS): 256 gathered pixels x 32 channels (4 pieces per thread) + its 64 x 32 weight block (1 piece)
  unsigned doff[DS ? 4 : 1];
  u32x4_t dv[DS ? 4 : 1], dw = {0u, 0u, 0u, 0u};
  __amdgpu_buffer_rsrc_t rs_dx = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, (short)0, 0, 0x00020000), rs_dw = rs_dx;
  const elem* dbase = nullptr;
  const char* dwbase = nullptr;
  int rec_dx = 0;
  auto prefetch1 = [&](int chunk, int j) {
    if (j < NB) hv[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)soff[j], pso, 0);
    else if (j < NB + TAPS) wv[j - NB] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, tid * 16, (chunk * TAPS + (j - NB)) * 4096, 0);
    else if (DS) {
      // (chunk - 1 = the main chunk whose MFMAs these ride under = the shortcut stage they feed)
      if (j < NB + TAPS + 4) dv[j - NB - TAPS] = __builtin_amdgcn_raw_buffer_load_b128(rs_dx, (int)doff[j - NB - TAPS], (chunk - 1) * 64, 0);
      else dw = __builtin_amdgcn_raw_buffer_load_b128(rs_dw, tid * 16, (chunk - 1) * 4096, 0);
    }
  };

  const int L = xcd_remap_fwd(blockIdx.x, gridDim.x);
  setup_load(L);
  set_rsrc(true);
#pragma unroll
  for (int j = 0; j < NB + TAPS; ++j) prefetch1(0, j);

  {
    const int mt = L / ntiles, nt = L - mt * ntiles;
    const int m0 = mt << 8;
    if (DS) {
      // gather sources: output pixel m -> shortcut input pixel (n, oy * s, ox * s), relative to the tile's first image
      const int n0d = frmap_div(m0, p.dHoWo);
      const size_t img = (size_t)p.ds_Hi * p.ds_Wi * p.ds_Cin;
      const size_t remd = ((size_t)p.N - n0d) * img * sizeof(elem);
      dbase = (const elem*)p.ds_in + (size_t)n0d * img;
      rec_dx = (int)(remd < 0x7FFFFFFFull ? remd : 0x7FFFFFFFull);
      dwbase = (const char*)p.ds_w + (size_t)nt * p.ds_chunks * 4096;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int px = (q * 256 + tid) >> 2, cg = tid & 3;
        const int m = m0 + px;
        const int n = frmap_div(m, p.dHoWo), rem = m - n * p.HoWo, oy = frmap_div(rem, p.dWo), ox = rem - oy * p.Wo;
        doff[q] = m < p.M ? (unsigned)((((n - n0d) * p.ds_Hi + oy * p.ds_stride) * p.ds_Wi + ox * p.ds_stride) * p.ds_Cin + cg * 8) *
                                (unsigned)sizeof(elem)
                          : 0xFFFFFF00u;
      }
    }
    int A[MI];  // (pixel index in the halo image) * 64 + k-group * 16, before the tap offset and the swizzle
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = min(m0 + wave * 64 + mi * 16 + lr, p.M - 1);
      const int n = frmap_div(m, p.dHoWo), rem = m - n * p.HoWo, oy = frmap_div(rem, p.dWo), ox = rem - oy * p.Wo;
      A[mi] = ((((n - l_n0) * p.Hp + oy - l_rr0) * p.Wp + ox) << 6) | (g << 4);
    }
    f32x4_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
      __syncthreads();  // everyone is done with the previous stage's LDS images / the epilogue scratch
#pragma unroll
      for (int t = 0; t < TAPS; ++t) *(u32x4_t*)(wl + t * 4096 + tid * 16) = wv[t];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int item = u * 256 + tid;
        if (item < l_nitems) *(u32x4_t*)(halo + px_off<1>(item >> 2, item & 3)) = hv[u];
      }
      __syncthreads();
      // what the prefetch slots of this stage fetch
      const int pchunk = chunk + 1;
      pso = pchunk * 64;
      set_rsrc(pchunk < p.nchunks);
      if (DS) {  // the shortcut pieces riding under this chunk feed shortcut stage `chunk`; past the last stage: empty descriptors
        const bool dl = chunk < p.ds_chunks;
        rs_dx = __builtin_amdgcn_make_buffer_rsrc((void*)dbase, (short)0, dl ? rec_dx : 0, 0x00020000);
        rs_dw = __builtin_amdgcn_make_buffer_rsrc((void*)dwbase, (short)0, dl ? p.ds_chunks * 4096 : 0, 0x00020000);
      }
      if (!DS && pchunk == p.nchunks && p.res) {
        // last chunk: the idle prefetch slots fetch this thread's 8 residual pieces (the epilogue's layout),
        // so the residual tile lands under the MFMAs instead of being waited for in the epilogue
        pso = 0;
        const size_t rb = ((size_t)(p.M - m0) * p.Cout) * sizeof(elem);
        rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)((const elem*)p.res + (size_t)m0 * p.Cout), (short)0,
                                                  (int)(rb < 0x7FFFFFFFull ? rb : 0x7FFFFFFFull), 0x00020000);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int idx = (u & 1) * 64 + lane, row = wave * 64 + (u >> 1) * 16 + (idx >> 3);
          soff[u] = u < 8 && m0 + row < p.M ? (unsigned)((row * p.Cout + (nt << 6) + (idx & 7) * 8) * (int)sizeof(elem)) : 0xFFFFFF00u;
        }
      }

      vec8 wf[NI], pf[2][MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) asm volatile("" : "+v"(A[mi]));  // keep the per-tap addresses out of registers
      auto paddr = [&](int mi, int t) {
        const int at = A[mi] + (((t / 3) * p.Wp + (t % 3)) << 6);
        return at ^ ((at >> 3) & 32);
      };
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(wl + ni * 1024 + woff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) pf[0][mi] = *(const vec8*)(halo + paddr(mi, 0));
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        if (t + 1 < TAPS) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) pf[(t + 1) & 1][mi] = *(const vec8*)(halo + paddr(mi, t + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT::mfma(wf[ni], pf[t & 1][mi], acc[mi][ni]);
          if (t + 1 < TAPS) wf[ni] = *(const vec8*)(wl + (t + 1) * 4096 + ni * 1024 + woff);
          const int k = t * NI + ni;
          if ((k + 1) * NL / NG != k * NL / NG) prefetch1(pchunk, k * NL / NG);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (DS && chunk < p.ds_chunks) {
        // ---- shortcut stage `chunk`: its operands replace the main chunk's LDS images for 16 MFMAs
        __syncthreads();
        *(u32x4_t*)(wl + tid * 16) = dw;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(u32x4_t*)(halo + px_off<1>((q * 256 + tid) >> 2, tid & 3)) = dv[q];
        __syncthreads();
        vec8 wd[NI], pd[MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wd[ni] = *(const vec8*)(wl + ni * 1024 + woff);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) pd[mi] = *(const vec8*)(halo + px_off<1>(wave * 64 + mi * 16 + lr, g));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = TT::mfma(wd[ni], pd[mi], acc[mi][ni]);
      }
    }
    // ---- epilogue: + shift (+ residual) (activation) -> NHWC, whole-line 16-byte stores via an LDS transpose
    __syncthreads();  // every wave is done reading the staged tiles; LDS is free for the transpose
    if (DS)
      conv_epilogue<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), m0 + wave * 64, p.M, p.Cout, nt << 6, p.shift,
                                (const elem*)nullptr, (elem*)p.out, p.relu, lane);
    else
      conv_epilogue<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), m0 + wave * 64, p.M, p.Cout, nt << 6, p.shift,
                                (const elem*)p.res, (elem*)p.out, p.relu, lane, hv);
  }
}

// ================================================================================================
// 3x3 stride-1 layers with Cin = 64 and H, W multiples of 8 (ResNet layer1; BaselineNet conv3):
// weights-resident kernel with WAVE-AUTONOMOUS tiles.
//
// With two chunks per tile conv3x3_fast_kernel re-stages the layer's whole 74 KB weight slab for
// every 256 output pixels (47 % of all bytes a tile moves) and spends more time in barriers, LDS
// writes and its epilogue than in MFMAs.  Here ONE 8-wave workgroup per CU loads the slab once, and
// every wave then works alone: its tile is an 8x8 output patch x 64 channels, its halo is the
// 10x10 input patch in a wave-private LDS image (pitch 16 pixels: a 16-pixel fragment = two 8-pixel
// row segments lands on disjoint bank quads), and since only that wave ever touches the image the
// in-order LDS pipeline is all the synchronisation there is - no barrier anywhere after start-up.
// Eight independent instruction streams per CU cover each other's staging and epilogues.
// Every global access rides between MFMAs: the next chunk's (or next tile's) 7 halo pieces, this
// tile's 8 residual pieces, and the PREVIOUS tile's 8 packed output pieces (each one an 8-pixel x
// 64-channel row segment = 1 KB contiguous), which the epilogue leaves in registers.
// ================================================================================================
// NCH = channel chunks (2: Cin = 64; 1: Cin = 32, no residual).  POOL = true: MaxPool2d(2, 2) fused (BaselineNet conv2 /
// conv3, face_models.py:39-40): the patch's pixels are walked in pool-major order (fragment row i of pixel group mi =
// window i >> 2 of pooled row mi, pixel i & 3 of it), the transpose takes the max over each window's 4 scratch rows and
// the tile leaves 2 packed pieces (its 4x4 pooled patch) instead of 8.
template <typename TT, int NCH = 2, bool POOL = false>
__global__ __launch_bounds__(512, 1) void conv3x3_c64_wave_kernel(const ConvParams p) {
  constexpr int MI = 4, NI = 4, TAPS = 9, NH = 7, NO = POOL ? 2 : 8, NG = TAPS * NI, NL = NO + NH;
  constexpr int PITCH = NI * 64 + 16, HALO = 10 * 16 * 64;
  using vec8 = typename TT::vec8;
  using elem = typename TT::elem;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  constexpr int wbytes = NCH * TAPS * 4096;  // every chunk
  char* wl = smem;
  char* halo = smem + wbytes + wave * HALO;

  const int ntiles = p.Cout >> 6;
  const int Lb = xcd_remap_fwd(blockIdx.x, gridDim.x);
  const int nt = Lb % ntiles, co0 = nt << 6;
  const int tx = p.Wo >> 3, tpi = (p.Ho >> 3) * tx;  // 8x8 patches per row / per image
  const int total = p.N * tpi;
  // Tile rounds: in the full rounds wave slot s = 8 * workgroup + wave takes tile r * W + s (neighbouring patches on
  // one CU).  The last, partial round is dealt wave-major instead (wave 0 of every workgroup first): its `rem` extra
  // tiles land on `rem` different CUs, one SIMD each, where they finish in about half a round once their SIMD
  // partner is done - instead of 8 extra tiles on each of rem/8 CUs, a whole extra round there (256 faces: 6.5
  // rounds instead of 7; 128 faces per stream: 3.5 instead of 4).
  const int nwg = gridDim.x / ntiles, W = nwg * 8;
  const int full = total / W, rem = total - full * W;
  const int slot = (Lb / ntiles) * 8 + wave, spread = wave * nwg + Lb / ntiles;
  auto tile_of = [&](int r) { return r < full ? r * W + slot : (r == full && spread < rem ? full * W + spread : -1); };
  int rnd = 0;
  int T = tile_of(0);

  {  // the resident weight slab of channel tile nt (both chunks, all taps), already in LDS-image order
    const char* wsrc = (const char*)p.wpk + (size_t)nt * wbytes;
    for (int o = tid * 16; o < wbytes; o += 512 * 16) *(u32x4_t*)(wl + o) = *(const u32x4_t*)(wsrc + o);
  }
  __syncthreads();
  if (T < 0) return;

  const int woff = (lr << 6) + ((g ^ (((lr >> 2) & 1) << 1)) << 4);
  const elem* inp = (const elem*)p.in;
  const int img_bytes = p.Hi * p.Wi * p.Cin * (int)sizeof(elem);
  // POOL: the output map is the pooled one (Wo / 2 wide, 4x4 pooled pixels per patch)
  const int row_in = p.Wi * p.Cin * (int)sizeof(elem), row_out = (POOL ? p.Wo >> 1 : p.Wo) * p.Cout * (int)sizeof(elem);
  // output / residual piece u (= output row u of the patch): lane -> pixel column lane >> 3, channels (lane & 7) * 8 ..+7
  // POOL: piece u = pooled rows 2u, 2u + 1: lane -> pooled row 2u + (lane >> 5), pooled column (lane >> 3) & 3
  const int io_lane = POOL ? (int)(lane >> 5) * row_out + ((((lane >> 3) & 3) * p.Cout + co0 + (lane & 7) * 8) * (int)sizeof(elem))
                           : ((lane >> 3) * p.Cout + co0 + (lane & 7) * 8) * (int)sizeof(elem);
  const int io_step = POOL ? 2 * row_out : row_out;   // byte step between pieces
  float sh[8];
  {
    const f32x4_t s0 = *(const f32x4_t*)(p.shift + co0 + (lane & 7) * 8), s1 = *(const f32x4_t*)(p.shift + co0 + (lane & 7) * 8 + 4);
    sh[0] = s0[0]; sh[1] = s0[1]; sh[2] = s0[2]; sh[3] = s0[3]; sh[4] = s1[0]; sh[5] = s1[1]; sh[6] = s1[2]; sh[7] = s1[3];
  }
  // fragment row i of pixel group mi = output pixel (mi*2 + (i >> 3), i & 7) of the patch = halo pixel (+kh, +kw)
  int A[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
    A[mi] = POOL ? ((((mi * 2 + ((lr >> 1) & 1)) << 4) + ((lr >> 2) * 2 + (lr & 1))) << 6) | (g << 4)
                 : ((((mi * 2 + (lr >> 3)) << 4) + (lr & 7)) << 6) | (g << 4);

  // halo piece u of this lane: q = u*64 + lane -> halo pixel q >> 2 = (pr, pc) of the 10x10 patch, channel group q & 3
  unsigned soff[NH];
  u32x4_t hv[NH], ov[NO];
  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)inp, (short)0, 0, 0x00020000);
  int t_n = 0, t_by = 0, t_bx = 0;  // the tile the halo loads belong to
  auto setup_load = [&](int t, bool live) {
    t_n = t / tpi;
    const int rem = t - t_n * tpi;
    t_by = rem / tx; t_bx = rem - t_by * tx;
    rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(inp + (size_t)t_n * p.Hi * p.Wi * p.Cin), (short)0, live ? img_bytes : 0, 0x00020000);
#pragma unroll
    for (int u = 0; u < NH; ++u) {
      const int q = u * 64 + lane, px = q >> 2, cg = q & 3;
      const int pr = (px * 205) >> 11, pc = px - pr * 10;  // px / 10 for px < 112
      const int iy = t_by * 8 - 1 + pr, ix = t_bx * 8 - 1 + pc;
      const bool ok = px < 100 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      soff[u] = ok ? (unsigned)(iy * row_in + (ix * p.Cin + cg * 8) * (int)sizeof(elem)) : 0xFFFFFF00u;
    }
  };
  auto io_rsrc = [&](const void* base, int n, int by, int bx, bool live) {  // descriptor at the patch's first pixel
    const size_t el = POOL ? (((size_t)n * (p.Ho >> 1) + by * 4) * (p.Wo >> 1) + bx * 4) * p.Cout
                           : (((size_t)n * p.Ho + by * 8) * p.Wo + bx * 8) * p.Cout;
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const elem*)base + el), (short)0, live ? (POOL ? 4 : 8) * row_out : 0, 0x00020000);
  };

  setup_load(T, true);
#pragma unroll
  for (int u = 0; u < NH; ++u) hv[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)soff[u], 0, 0);
  int p_n = -1, p_by = 0, p_bx = 0;  // the tile whose packed outputs sit in ov[]

  for (; T >= 0; T = tile_of(++rnd)) {
    const int c_n = t_n, c_by = t_by, c_bx = t_bx;
    f32x4_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk) {
      // stage: this wave's halo pieces -> its private image (LDS is in order: earlier fragment reads are served first)
#pragma unroll
      for (int u = 0; u < NH; ++u) {
        const int q = u * 64 + lane, px = q >> 2;
        const int pr = (px * 205) >> 11, pc = px - pr * 10;
        if (px < 100) *(u32x4_t*)(halo + px_off<1>(pr * 16 + pc, q & 3)) = hv[u];
      }
      // ---- what rides between this phase's MFMAs
      //  chunk 0: the previous tile's 8 output pieces (stores), then chunk 1's 7 halo pieces
      //  chunk 1: this tile's 8 residual pieces (into ov[]), then the next tile's chunk-0 halo pieces
      //  (NCH = 1: one chunk does both - the previous tile's stores and the next tile's halo pieces; POOL: no residual)
      __amdgpu_buffer_rsrc_t rs_io;
      int pso = 64;
      if (chunk == 0) rs_io = io_rsrc(p.out, p_n < 0 ? 0 : p_n, p_by, p_bx, p_n >= 0);
      else rs_io = io_rsrc(p.res ? p.res : p.out, c_n, c_by, c_bx, p.res != nullptr && !POOL);
      if (chunk == NCH - 1) {
        pso = 0;
        const int Tn = tile_of(rnd + 1);
        const bool more = Tn >= 0;
        setup_load(more ? Tn : T, more);
      }
      auto memop = [&](int i) {
        if (i < NO) {
          if (chunk == 0) __builtin_amdgcn_raw_buffer_store_b128(ov[i], rs_io, io_lane, i * io_step, 0);
          else if (!POOL) ov[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_io, io_lane, i * io_step, 0);
        } else {
          hv[i - NO] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)soff[i - NO], pso, 0);
        }
      };

      const char* wc = wl + chunk * (TAPS * 4096);
      vec8 wf[NI], pf[2][MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) asm volatile("" : "+v"(A[mi]));  // per-tap addresses: 3 VALU ops each, not 36 registers
      auto paddr = [&](int mi, int t) {
        const int at = A[mi] + (((t / 3) * 16 + (t % 3)) << 6);
        return at ^ ((at >> 3) & 32);
      };
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(wc + ni * 1024 + woff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) pf[0][mi] = *(const vec8*)(halo + paddr(mi, 0));
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        if (t + 1 < TAPS) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) pf[(t + 1) & 1][mi] = *(const vec8*)(halo + paddr(mi, t + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT::mfma(wf[ni], pf[t & 1][mi], acc[mi][ni]);
          if (t + 1 < TAPS) wf[ni] = *(const vec8*)(wc + (t + 1) * 4096 + ni * 1024 + woff);
          const int k = t * NI + ni;
          if ((k + 1) * NL / NG != k * NL / NG) memop(k * NL / NG);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (POOL) {
      // ---- pooled epilogue: two pixel groups (= pooled rows mp, mp + 1) per pass; every lane takes one pooled pixel's
      //      8-channel run: the max over the window's 4 scratch rows, + shift, activation, round once
#pragma unroll
      for (int mp = 0; mp < MI; mp += 2) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) *(f32x4_t*)(halo + h * (16 * PITCH) + lr * PITCH + ni * 64 + g * 16) = acc[mp + h][ni];
        __builtin_amdgcn_wave_barrier();
        const char* src = halo + (lane >> 5) * (16 * PITCH) + (((lane >> 3) & 3) * 4) * PITCH + (lane & 7) * 32;
        f32x4_t a = *(const f32x4_t*)src, b = *(const f32x4_t*)(src + 16);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
          const f32x4_t a2 = *(const f32x4_t*)(src + q * PITCH), b2 = *(const f32x4_t*)(src + q * PITCH + 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) { a[e] = fmaxf(a[e], a2[e]); b[e] = fmaxf(b[e], b2[e]); }
        }
        float v[8] = {a[0] + sh[0], a[1] + sh[1], a[2] + sh[2], a[3] + sh[3], b[0] + sh[4], b[1] + sh[5], b[2] + sh[6], b[3] + sh[7]};
        if (p.relu == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        ov[mp >> 1] = pack8<TT>(v);
      }
    } else {
    // ---- epilogue without stores: transpose two pixel groups at a time through the (now free) halo image,
    //      + shift (+ residual from ov[]) (activation), round once, leave the 8 packed pieces in ov[]
#pragma unroll
    for (int mp = 0; mp < MI; mp += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *(f32x4_t*)(halo + h * (16 * PITCH) + lr * PITCH + ni * 64 + g * 16) = acc[mp + h][ni];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int prow = j * 8 + (lane >> 3), part = lane & 7, u = (mp + h) * 2 + j;
          const f32x4_t a = *(const f32x4_t*)(halo + h * (16 * PITCH) + prow * PITCH + part * 32);
          const f32x4_t b = *(const f32x4_t*)(halo + h * (16 * PITCH) + prow * PITCH + part * 32 + 16);
          float v[8] = {a[0] + sh[0], a[1] + sh[1], a[2] + sh[2], a[3] + sh[3], b[0] + sh[4], b[1] + sh[5], b[2] + sh[6], b[3] + sh[7]};
          if (p.res) {
            float r[8];
            unpack8<TT>(ov[u], r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
          }
          if (p.relu == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (p.relu == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
          }
          ov[u] = pack8<TT>(v);
        }
    }
    }
    p_n = c_n; p_by = c_by; p_bx = c_bx;
  }
  // the last tile's outputs
  if (p_n >= 0) {
    const __amdgpu_buffer_rsrc_t rs_o = io_rsrc(p.out, p_n, p_by, p_bx, true);
#pragma unroll
    for (int i = 0; i < NO; ++i) __builtin_amdgcn_raw_buffer_store_b128(ov[i], rs_o, io_lane, i * io_step, 0);
  }
}

// ================================================================================================
// 3x3 stride-2 convolutions: row-parity split staging.
//
// A stride-2 output row oy reads input rows 2oy-1, 2oy, 2oy+1: every input row of the tile is needed,
// 4 input pixels per output pixel, and the full halo of 256 output pixels (~90 KB) leaves room for
// only one workgroup per CU.  But tap rows kh = 0 and 2 touch only EVEN padded rows and kh = 1 only
// ODD ones, so each 32-channel chunk is run as two sub-stages — {even rows, 6 taps} then {odd rows,
// 3 taps} — each with a half-height halo (~48 KB + 24 KB of weights): two workgroups fit per CU again
// and one stages while the other computes.
// ================================================================================================
template <typename TT>
__global__ __launch_bounds__(256, 2) void conv3x3s2_split_kernel(const ConvParams p) {
  constexpr int BM = 256, MI = 4, NI = 4;
  using vec8 = typename TT::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wl = smem + p.halo_bytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int L = xcd_remap_fwd(blockIdx.x, p.nblocks);
  const int ntiles = p.Cout >> 6;
  const int mt = L / ntiles, nt = L - mt * ntiles;
  const int m0 = mt * BM, mlast = min(m0 + BM, p.M) - 1;
  const int Hh = p.Hp >> 1;  // padded rows per image and parity
  const int n0 = frmap_div(m0, p.dHoWo), oy0 = frmap_div(m0 - n0 * p.HoWo, p.dWo);
  const int n1 = frmap_div(mlast, p.dHoWo), oy1 = frmap_div(mlast - n1 * p.HoWo, p.dWo);
  const int span = (n1 - n0) * Hh + oy1 - oy0;  // last pixel's row index within a parity plane
  const int rr0 = 2 * oy0;                       // padded row of the tile's first even row

  int qb[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = min(m0 + wave * 64 + mi * 16 + lr, p.M - 1);
    const int n = frmap_div(m, p.dHoWo), rem = m - n * p.HoWo, oy = frmap_div(rem, p.dWo), ox = rem - oy * p.Wo;
    qb[mi] = ((n - n0) * Hh + oy - oy0) * p.Wp + 2 * ox;
  }
  const int woff = (lr << 6) + ((g ^ (((lr >> 2) & 1) << 1)) << 4);
  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const typename TT::elem* inp = (const typename TT::elem*)p.in;

  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {  // 0: even padded rows, taps kh in {0,2};  1: odd rows, kh = 1
      if (chunk > 0 || par > 0) __syncthreads();
      const int ntap = par ? 3 : 6;
      const char* wsrc = (const char*)p.wpk + ((size_t)(nt * p.nchunks + chunk) * 9) * 4096 + tid * 16;
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (t < ntap) {
          const int tap = par ? 3 + t : (t < 3 ? t : t + 3);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + tap * 4096),
                                           (__attribute__((address_space(3))) void*)(wl + t * 4096 + wave * 1024), 16, 0, 0);
        }
      }
      const int nrows = span + (par ? 1 : 2);
      const int nitems = nrows * p.Wp * 4;
      constexpr int NB = 8;
      for (int it0 = tid; it0 < nitems; it0 += 256 * NB) {
        u32x4_t v[NB];
        int dst[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int item = it0 + u * 256;
          dst[u] = -1;
          v[u] = (u32x4_t){0u, 0u, 0u, 0u};
          if (item < nitems) {
            const int px = item >> 2, cg = item & 3;
            const int r = (int)fast_div((uint32_t)px, p.magic_Wp);
            const int c = px - r * p.Wp;
            const int rr = rr0 + 2 * r + par;  // padded row in the stack of images
            const int dn = (int)fast_div((uint32_t)rr, p.magic_Hp);
            const int iy = rr - dn * p.Hp - 1, ix = c - 1, n = n0 + dn;
            dst[u] = px_off<2>(px, cg);
            const bool ok = n < p.N && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const typename TT::elem* a = ok ? inp + (((size_t)n * p.Hi + iy) * p.Wi + ix) * p.Cin + (size_t)(chunk * 32 + cg * 8)
                                            : (const typename TT::elem*)g_zero_page;
            v[u] = *(const u32x4_t*)a;  // unconditional load (see g_zero_page)
          }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u)
          if (dst[u] >= 0) *(u32x4_t*)(halo + dst[u]) = v[u];
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (t < ntap) {
          const int dq = par ? t : (t / 3) * p.Wp + (t % 3);  // even plane: kh=0 -> row +0, kh=2 -> row +1
          vec8 wf[NI], pf[MI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(wl + t * 4096 + ni * 1024 + woff);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) pf[mi] = *(const vec8*)(halo + px_off<2>(qb[mi] + dq, g));
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
        }
      }
    }
  }
  __syncthreads();
  conv_epilogue<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), m0 + wave * 64, p.M, p.Cout, nt << 6, p.shift,
                            (const typename TT::elem*)p.res, (typename TT::elem*)p.out, p.relu, lane);
}

// ------------------------------------------------------------------------------------------------
// The same row-parity split with the software pipeline of conv3x3_fast_kernel: a stage is (chunk,
// parity plane); the next stage's plane pixels (<= 12 pieces per thread) and its 6 or 3 weight taps
// are buffer-loaded into registers between the current stage's MFMAs and written to LDS at the stage
// boundary.  Used when a plane is <= 12 pieces per thread (all three ResNet stride-2 convs).
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ __launch_bounds__(256, 2) void conv3x3s2_fast_kernel(const ConvParams p) {
  constexpr int MI = 4, NI = 4, NB = 12, NW = 6, NL = NB + NW;
  using vec8 = typename TT::vec8;
  using elem = typename TT::elem;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wl = smem + p.halo_bytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int L = xcd_remap_fwd(blockIdx.x, p.nblocks);
  const int ntiles = p.Cout >> 6;
  const int mt = L / ntiles, nt = L - mt * ntiles;
  const int m0 = mt << 8, mlast = min(m0 + 256, p.M) - 1;
  const int Hh = p.Hp >> 1;  // padded rows per image and parity
  const int n0 = frmap_div(m0, p.dHoWo), oy0 = frmap_div(m0 - n0 * p.HoWo, p.dWo);
  const int n1 = frmap_div(mlast, p.dHoWo), oy1 = frmap_div(mlast - n1 * p.HoWo, p.dWo);
  const int span = (n1 - n0) * Hh + oy1 - oy0;  // last pixel's row index within a parity plane
  const int rr0 = 2 * oy0;                       // padded row of the tile's first even row
  const int woff = (lr << 6) + ((g ^ (((lr >> 2) & 1) << 1)) << 4);
  const elem* inp = (const elem*)p.in;

  int A[MI];  // linear byte address (pixel * 64 + k-group * 16) in a plane image, before tap offset and swizzle
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = min(m0 + wave * 64 + mi * 16 + lr, p.M - 1);
    const int n = frmap_div(m, p.dHoWo), rem = m - n * p.HoWo, oy = frmap_div(rem, p.dWo), ox = rem - oy * p.Wo;
    A[mi] = ((((n - n0) * Hh + oy - oy0) * p.Wp + 2 * ox) << 6) | (g << 4);
  }

  // Source byte offsets (relative to image n0) of this thread's pieces.  Plane row r of the even plane is
  // input row 2j-1 of its image, of the odd plane row 2j: the odd-plane piece is always one input row
  // below the even-plane piece, so one offset per piece + a validity bit per plane suffices
  // (invalid -> out-of-range offset -> the load returns the zero padding).
  unsigned soff[NB];
  unsigned vmask = 0;  // bit u: even-plane piece u is in the image; bit 16+u: odd-plane piece
  const int nitems0 = (span + 2) * p.Wp * 4, nitems1 = (span + 1) * p.Wp * 4;
  const unsigned rowbytes = (unsigned)p.Wi * (unsigned)p.Cin * (unsigned)sizeof(elem);
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int item = u * 256 + tid;
    const int px = item >> 2, cg = item & 3;
    const int r = (int)fast_div((uint32_t)px, p.magic_Wp);
    const int c = px - r * p.Wp;
    const int rr = rr0 + 2 * r;  // even padded row in the stack of images
    const int dn = (int)fast_div((uint32_t)rr, p.magic_Hp);
    const int iy = rr - dn * p.Hp - 1, ix = c - 1, n = n0 + dn;
    const bool colok = n < p.N && (unsigned)ix < (unsigned)p.Wi;
    soff[u] = ((unsigned)((dn * p.Hi + iy) * p.Wi + ix) * (unsigned)p.Cin + (unsigned)(cg * 8)) * (unsigned)sizeof(elem);
    if (colok && item < nitems0 && iy >= 0) vmask |= 1u << u;
    if (colok && item < nitems1 && iy + 1 < p.Hi) vmask |= 1u << (16 + u);
  }
  const size_t base_el = (size_t)n0 * p.Hi * p.Wi * p.Cin;
  const size_t rem_b = ((size_t)p.N * p.Hi * p.Wi * p.Cin - base_el) * sizeof(elem);
  const int in_records = (int)(rem_b < 0x7FFFFFFFull ? rem_b : 0x7FFFFFFFull);
  const int wslab = p.nchunks * (9 * 4096);
  const char* w_base = (const char*)p.wpk + (size_t)nt * wslab;
  __amdgpu_buffer_rsrc_t rs_in, rs_w;
  auto set_rsrc = [&](bool live) {
    rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(inp + base_el), (short)0, live ? in_records : 0, 0x00020000);
    rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, (short)0, live ? wslab : 0, 0x00020000);
  };
  u32x4_t hv[NB], wv[NW];
  // piece j of stage (chunk, par): j < NB plane pixels, then the stage's weight taps (even plane: kh 0 and 2; odd: kh 1)
  auto prefetch1 = [&](int chunk, int par, int j) {
    if (j < NB) {
      const unsigned o = (vmask >> (par * 16 + j)) & 1u ? soff[j] + (par ? rowbytes : 0u) : 0xFFFFFF00u;
      hv[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)o, chunk * 64, 0);
    } else {
      const int t = j - NB;
      if (t < (par ? 3 : 6)) {
        const int tap = par ? 3 + t : (t < 3 ? t : t + 3);
        wv[t] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, tid * 16, (chunk * 9 + tap) * 4096, 0);
      }
    }
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  set_rsrc(true);
#pragma unroll
  for (int j = 0; j < NL; ++j) prefetch1(0, 0, j);

  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {  // 0: even padded rows, taps kh in {0,2};  1: odd rows, kh = 1
      const int ntap = par ? 3 : 6;
      __syncthreads();  // everyone is done reading the previous stage's LDS images
#pragma unroll
      for (int t = 0; t < NW; ++t)
        if (t < ntap) *(u32x4_t*)(wl + t * 4096 + tid * 16) = wv[t];
      int tl = tid;
      asm volatile("" : "+v"(tl));  // recompute the 12 LDS destinations per stage instead of keeping them live
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int item = u * 256 + tl;
        if (item < (par ? nitems1 : nitems0)) *(u32x4_t*)(halo + px_off<2>(item >> 2, item & 3)) = hv[u];
      }
      __syncthreads();
      // next stage: the odd plane of this chunk, or the even plane of the next chunk
      const int pc = par ? chunk + 1 : chunk, pp = par ^ 1;
      set_rsrc(pc < p.nchunks);
      asm volatile("" : "+v"(vmask));  // select each piece's offset at its load, not 24 hoisted copies

      vec8 wf[NI], pf[2][MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) asm volatile("" : "+v"(A[mi]));  // per-tap addresses: 3 VALU ops each, not 36 registers
      auto paddr = [&](int mi, int t) {
        const int dq = par ? t : (t / 3) * p.Wp + (t % 3);  // even plane: kh=0 -> row +0, kh=2 -> row +1
        const int at = A[mi] + (dq << 6);
        return at ^ ((at >> 3) & 0x60);  // == px_off<2>
      };
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(wl + ni * 1024 + woff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) pf[0][mi] = *(const vec8*)(halo + paddr(mi, 0));
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (t < ntap) {
          if (t + 1 < ntap) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) pf[(t + 1) & 1][mi] = *(const vec8*)(halo + paddr(mi, t + 1));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT::mfma(wf[ni], pf[t & 1][mi], acc[mi][ni]);
            if (t + 1 < ntap) wf[ni] = *(const vec8*)(wl + (t + 1) * 4096 + ni * 1024 + woff);
            const int k = t * NI + ni, ng = ntap * NI;
#pragma unroll
            for (int j = 0; j < NL; ++j)
              if (j >= k * NL / ng && j < (k + 1) * NL / ng) prefetch1(pc, pp, j);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }
  __syncthreads();
  conv_epilogue<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), m0 + wave * 64, p.M, p.Cout, nt << 6, p.shift,
                            (const elem*)p.res, (elem*)p.out, p.relu, lane);
}

// ================================================================================================
// 1x1 convolutions and wide Linear layers: a plain gather-GEMM with WIDE stages.
//
// The generic kernel's 32-channel chunk gives a 1x1 conv only 16 MFMAs per wave between barriers.
// Here a stage is CKS x 32 channels (CKS = 4: 128): 256 gathered pixels x 128 channels (64 KB) +
// 64 x 128 weights (16 KB) per stage, 64 MFMAs per wave per stage, both operands of the next stage
// prefetched into registers under the current stage's MFMAs.  Split-K as in frmap_linear_mfma.
// ================================================================================================
// MATCH = true: top-1 gallery match (head_match.hip, frmap_match_top1_packed).  The "pixels" are the probes and the
// "channels" the gallery rows, both split into fp16 (hi, lo) pairs laid out so that one K = 3 D GEMM accumulates
// a_hi.g_hi + a_hi.g_lo + a_lo.g_hi in fp32 (= the fp32 dot product to ~2^-22); the epilogue forms the squared
// F.pairwise_distance from it as gemm_nt_f32_kernel<MODE_DIST> does and writes one candidate record per probe and 64-row slot.
template <typename TT, int CKS, bool MATCH = false>
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(const ConvParams p) {
  constexpr int BM = 256, MI = 4, NI = 4, NIT = BM * 4 / 256;
  using vec8 = typename TT::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;                     // [CKS][256 px][64 B]
  char* wl = smem + CKS * BM * 64;       // [CKS][64 cout][64 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int L = xcd_remap_fwd(blockIdx.x, p.nblocks);
  const int ntiles = p.Cout >> 6;
  const int kslice = L % p.ksplit, Lt = L / p.ksplit;
  const int mt = Lt / ntiles, nt = Lt - mt * ntiles;
  const int m0 = mt * BM;
  const int st_lo = (int)(((long long)kslice * p.nchunks) / p.ksplit);   // nchunks = stages of CKS*32 channels
  const int st_hi = (int)(((long long)(kslice + 1) * p.nchunks) / p.ksplit);

  const typename TT::elem* inp = (const typename TT::elem*)p.in;
  size_t gsrc[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int px = (it * 256 + tid) >> 2;
    const int m = min(m0 + px, p.M - 1);
    const int n = frmap_div(m, p.dHoWo), rem = m - n * p.HoWo, oy = frmap_div(rem, p.dWo), ox = rem - oy * p.Wo;
    gsrc[it] = (((size_t)n * p.Hi + (size_t)(oy * p.stride)) * p.Wi + (size_t)(ox * p.stride)) * p.Cin + (size_t)((tid & 3) * 8);
  }
  int poff[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) poff[mi] = px_off<1>(wave * 64 + mi * 16 + lr, g);
  const int woff = (lr << 6) + ((g ^ (((lr >> 2) & 1) << 1)) << 4);
  const char* wbase = (const char*)p.wpk + (size_t)nt * p.nchunks * CKS * 4096 + tid * 16;

  u32x4_t hv[CKS][NIT], wv[CKS];
  auto prefetch = [&](int stg) {
#pragma unroll
    for (int sc = 0; sc < CKS; ++sc) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) hv[sc][it] = *(const u32x4_t*)(inp + gsrc[it] + (size_t)(stg * CKS + sc) * 32);
      wv[sc] = *(const u32x4_t*)(wbase + (size_t)(stg * CKS + sc) * 4096);
    }
  };
  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (st_lo < st_hi) prefetch(st_lo);
  for (int stg = st_lo; stg < st_hi; ++stg) {
    if (stg > st_lo) __syncthreads();
#pragma unroll
    for (int sc = 0; sc < CKS; ++sc) {
      *(u32x4_t*)(wl + sc * 4096 + tid * 16) = wv[sc];
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        *(u32x4_t*)(halo + sc * (BM * 64) + px_off<1>((it * 256 + tid) >> 2, tid & 3)) = hv[sc][it];
    }
    __syncthreads();
    if (stg + 1 < st_hi) prefetch(stg + 1);
#pragma unroll
    for (int sc = 0; sc < CKS; ++sc) {
      vec8 wf[NI], pf[MI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(wl + sc * 4096 + ni * 1024 + woff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) pf[mi] = *(const vec8*)(halo + sc * (BM * 64) + poff[mi]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
    }
  }
  if constexpr (MATCH) {
    match_epilogue_records<MI>(acc, m0 + wave * 64, p.M, nt << 6, p.m_G, p.m_D, p.M, p.m_stat_a, p.m_stat_w, p.m_recs, lane);
    return;
  }
  __syncthreads();
  if (p.ksplit > 1)
    conv_epilogue_partial<MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), m0 + wave * 64, p.M, p.Cout, nt << 6,
                                  p.slab + (size_t)kslice * p.M * p.Cout, lane);
  else
    conv_epilogue<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), m0 + wave * 64, p.M, p.Cout, nt << 6, p.shift,
                              (const typename TT::elem*)p.res, (typename TT::elem*)p.out, p.relu, lane);
}

template <typename TT, int CKS, bool MATCH = false>
static int launch_1x1(const ConvParams& p, hipStream_t st) {
  auto kern = conv1x1_kernel<TT, CKS, MATCH>;
  if (frmap_big_lds((const void*)kern, 160 * 1024)) return -2;
  int lds = CKS * (256 * 64 + 4096);
  const int scratch = 4 * 16 * (4 * 64 + 16);
  if (lds < scratch) lds = scratch;
  hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), lds, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// p.nchunks must hold Cin/32 on entry; picks the stage width and rewrites it in stages
static int dispatch_1x1(ConvParams& p, int dtype, hipStream_t st) {
  const int c32 = p.Cin / 32;
  if (c32 % 4 == 0) {
    p.nchunks = c32 / 4;
    return dtype == FRMAP_BF16 ? launch_1x1<BF16, 4>(p, st) : launch_1x1<F16, 4>(p, st);
  }
  if (c32 % 2 == 0) {
    p.nchunks = c32 / 2;
    return dtype == FRMAP_BF16 ? launch_1x1<BF16, 2>(p, st) : launch_1x1<F16, 2>(p, st);
  }
  p.nchunks = c32;
  return dtype == FRMAP_BF16 ? launch_1x1<BF16, 1>(p, st) : launch_1x1<F16, 1>(p, st);
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

template <typename TT, int BM, int KS, int SWZ, bool POOL = false>
static int launch(const ConvParams& p, int lds_bytes, hipStream_t st) {
  auto kern = conv_igemm_kernel<TT, BM, KS, SWZ, POOL>;
  if (frmap_big_lds((const void*)kern, 160 * 1024)) return -2;
  const int scratch = 4 * 16 * (4 * 64 + 16);  // epilogue transpose region (4 waves)
  if (lds_bytes < scratch) lds_bytes = scratch;
  hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), lds_bytes, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// conv3x3_c64_wave_kernel: one 8-wave workgroup per CU (per channel tile), each wave walks 8x8 patches
template <int NCH, bool POOL>
static int launch_wave(const ConvParams& p, int dtype, hipStream_t st) {
  const int ntiles = p.Cout / 64, total = p.N * (p.Hi / 8) * (p.Wi / 8);
  int cus = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  int per = cus / ntiles;
  if (per < 1) per = 1;
  if (per > (total + 7) / 8) per = (total + 7) / 8;
  const int grid = per * ntiles;
  const int ldsw = NCH * 9 * 4096 + 8 * (10 * 16 * 64);
  const void* kern = dtype == FRMAP_BF16 ? (const void*)conv3x3_c64_wave_kernel<BF16, NCH, POOL> : (const void*)conv3x3_c64_wave_kernel<F16, NCH, POOL>;
  if (frmap_big_lds(kern, 160 * 1024)) return -2;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL((conv3x3_c64_wave_kernel<BF16, NCH, POOL>), dim3(grid), dim3(512), ldsw, st, p);
  else
    hipLaunchKernelGGL((conv3x3_c64_wave_kernel<F16, NCH, POOL>), dim3(grid), dim3(512), ldsw, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

struct DsArgs {  // fused projection shortcut (conv3x3_fast_kernel<TT, true>); in == nullptr: none
  const void* in;
  const void* w;
  int Hi, Wi, Cin, stride;
};

// does a 3x3 stride-1 layer (with this shortcut) take conv3x3_fast_kernel<TT, true>?
static bool conv_ds_ok(int B, int Hi, int Wi, int Cin, int Cout, const DsArgs& d);

static int conv_igemm_impl(const void* in, const void* w_packed, const float* shift, const void* residual,
                           void* out, int B, int Hi, int Wi, int Cin, int Cout, int K, int stride, int pad,
                           int relu, int dtype, const DsArgs& ds, void* stream) {
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
  p.dHoWo = frmap_div_make((uint32_t)p.HoWo); p.dWo = frmap_div_make((uint32_t)p.Wo);
  p.nchunks = Cin / 32;
  p.ksplit = 1;
  p.slab = nullptr;
  p.ds_in = ds.in; p.ds_w = ds.w; p.ds_Hi = ds.Hi; p.ds_Wi = ds.Wi; p.ds_Cin = ds.Cin; p.ds_stride = ds.stride;
  p.ds_chunks = ds.in ? ds.Cin / 32 : 0;
  if (ds.in) FRMAP_REQUIRE(conv_ds_ok(B, Hi, Wi, Cin, Cout, ds) && !residual && K == 3 && stride == 1,
                           "conv_igemm_ds: this shape does not take the fused-shortcut kernel (check frmap_conv_igemm_ds_supported)");
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
    if (p.dbg == 0 && Cin >= 128) {
      // wide layers: the LDS-DMA ping-pong pipeline when it has enough tiles (conv_pp.hip)
      const int rc = frmap_conv1x1_pp(in, w_packed, shift, residual, out, B, Hi, Wi, Cin, Cout, stride, relu, dtype, st);
      if (rc < 0) return rc;
      if (rc == 1) return 0;
      return dispatch_1x1(p, dtype, st);  // wide stages pay once there are several of them
    }
    const int lds = p.halo_bytes + wbytes;
    return dtype == FRMAP_BF16 ? launch<BF16, 256, 1, 1>(p, lds, st) : launch<F16, 256, 1, 1>(p, lds, st);
  }
  // 3x3 stride 2, even input sizes: the LDS-DMA ping-pong kernel with space-to-depth addressing when it takes the shape
  if (stride == 2 && p.dbg == 0) {
    const int rc = frmap_conv3x3s2_pp(in, w_packed, shift, residual, out, B, Hi, Wi, Cin, Cout, relu, dtype, st);
    if (rc < 0) return rc;
    if (rc == 1) return 0;
  }
  // 3x3 stride 2 (even input height): row-parity split staging, two workgroups per CU
  if (stride == 2 && Hi % 2 == 0 && p.dbg == 0) {
    const int rows = (256 + Wo - 2) / Wo + 1, cross = (256 + Ho * Wo - 2) / (Ho * Wo);
    const int x = cross < rows - 1 ? cross : rows - 1;
    const int step = p.Hp / 2 - (Ho - 1);
    long long hbs = (long long)((rows - 1 - x) + x * (step > 1 ? step : 1) + 2) * p.Wp * 64;
    hbs = (hbs + 1023) & ~1023ll;
    if (hbs + 6 * 4096 <= 80 * 1024 && hbs / 64 < 65536) {
      p.halo_bytes = (int)hbs;
      p.nblocks = ((p.M + 255) / 256) * ntiles;
      const int lds = (int)hbs + 6 * 4096;
      const void* kern = dtype == FRMAP_BF16 ? (const void*)conv3x3s2_split_kernel<BF16> : (const void*)conv3x3s2_split_kernel<F16>;
      if (frmap_big_lds(kern, 160 * 1024)) return -2;
      static int s2fast = -1;
      if (s2fast < 0) { const char* e = getenv("FRMAP_CONV_S2FAST"); s2fast = e ? atoi(e) : 1; }
      const bool fast2 = s2fast && hbs / 16 <= 12 * 256 && (long long)(256 / (Ho * Wo) + 3) * Hi * Wi * Cin * 2 < (1ll << 31);
      if (fast2) {
        const void* k2 = dtype == FRMAP_BF16 ? (const void*)conv3x3s2_fast_kernel<BF16> : (const void*)conv3x3s2_fast_kernel<F16>;
        if (frmap_big_lds(k2, 160 * 1024)) return -2;
        if (dtype == FRMAP_BF16)
          hipLaunchKernelGGL(conv3x3s2_fast_kernel<BF16>, dim3(p.nblocks), dim3(256), lds, st, p);
        else
          hipLaunchKernelGGL(conv3x3s2_fast_kernel<F16>, dim3(p.nblocks), dim3(256), lds, st, p);
      } else if (dtype == FRMAP_BF16)
        hipLaunchKernelGGL(conv3x3s2_split_kernel<BF16>, dim3(p.nblocks), dim3(256), lds, st, p);
      else
        hipLaunchKernelGGL(conv3x3s2_split_kernel<F16>, dim3(p.nblocks), dim3(256), lds, st, p);
      FRMAP_LAUNCH_CHECK();
      return 0;
    }
  }
  // 3x3 stride 1 with Cin >= 128: the LDS-DMA ping-pong kernel (conv_pp.hip) when it takes the shape
  if (stride == 1 && p.dbg == 0) {
    const FrmapPPShortcut sc = {ds.in, ds.w, ds.Hi, ds.Wi, ds.Cin, ds.stride};
    const int rc = frmap_conv3x3_pp(in, w_packed, shift, residual, out, B, Hi, Wi, Cin, Cout, relu, dtype, st, ds.in ? &sc : nullptr);
    if (rc < 0) return rc;
    if (rc == 1) return 0;
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
  // register-prefetch persistent kernel when the whole halo is <= 10 pieces per thread (and no ablation flag)
  const bool fastk = BM == 256 && stride == 1 && p.dbg == 0 && hb / 16 <= 10 * 256 && lds <= 80 * 1024 &&
                     (long long)(256 / (Ho * Wo) + 3) * Hi * Wi * Cin * 2 < (1ll << 31);
  {
    static int wres = -1;
    if (wres < 0) { const char* e = getenv("FRMAP_CONV_WRES"); wres = e ? atoi(e) : 1; }
    if (wres && !ds.in && stride == 1 && p.dbg == 0 && Cin == 64 && Hi % 8 == 0 && Wi % 8 == 0 &&
        (long long)Hi * Wi * Cin * 2 < (1ll << 31) && (long long)8 * Wo * Cout * 2 < (1ll << 31)) {
      return launch_wave<2, false>(p, dtype, st);
    }
  }
  if (ds.in) FRMAP_REQUIRE(fastk, "conv_igemm_ds: layer does not take the register-prefetch kernel");
  if (fastk) {
    const int grid = p.nblocks;
    typedef void (*kern_t)(const ConvParams);
    static const kern_t kerns[4] = {conv3x3_fast_kernel<BF16, false>, conv3x3_fast_kernel<BF16, true>,
                                    conv3x3_fast_kernel<F16, false>, conv3x3_fast_kernel<F16, true>};
    const int ki = (dtype == FRMAP_BF16 ? 0 : 2) + (ds.in ? 1 : 0);
    if (frmap_big_lds((const void*)kerns[ki], 160 * 1024)) return -2;
    const int scratch = 4 * 16 * (4 * 64 + 16);
    const int ldsf = lds < scratch ? scratch : lds;
    hipLaunchKernelGGL(kerns[ki], dim3(grid), dim3(256), ldsf, st, p);
    FRMAP_LAUNCH_CHECK();
    return 0;
  }
#define FRMAP_DISPATCH(TT)                                                                   \
  (BM == 256 ? (stride == 1 ? launch<TT, 256, 3, 1>(p, lds, st) : launch<TT, 256, 3, 2>(p, lds, st)) \
             : (stride == 1 ? launch<TT, 128, 3, 1>(p, lds, st) : launch<TT, 128, 3, 2>(p, lds, st)))
  return dtype == FRMAP_BF16 ? FRMAP_DISPATCH(BF16) : FRMAP_DISPATCH(F16);
#undef FRMAP_DISPATCH
}

static bool conv_ds_ok(int B, int Hi, int Wi, int Cin, int Cout, const DsArgs& d) {
  if (!d.in || !d.w || d.Cin <= 0 || d.Cin % 32 || d.Cin / 32 > Cin / 32 || d.stride < 1) return false;
  if ((d.Hi - 1) / d.stride + 1 != Hi || (d.Wi - 1) / d.stride + 1 != Wi) return false;  // 1x1, pad 0: Ho = (H-1)/s + 1
  const int Hp = Hi + 2, Wp = Wi + 2;
  long long hb = (long long)halo_rows_bound(256, Hi, Wi, Hp, 1, 3) * Wp * 64;
  hb = (hb + 1023) & ~1023ll;
  const long long lds = hb + 9 * 4096;
  return hb >= 256 * 64 && hb / 16 <= 10 * 256 && lds <= 80 * 1024 && hb / 64 < 65536 &&
         (long long)(256 / (Hi * Wi) + 3) * Hi * Wi * Cin * 2 < (1ll << 31) &&
         (long long)(256 / (Hi * Wi) + 3) * d.Hi * d.Wi * d.Cin * 2 < (1ll << 31);
}

extern "C" int frmap_conv_igemm(const void* in, const void* w_packed, const float* shift, const void* residual,
                                void* out, int B, int Hi, int Wi, int Cin, int Cout, int K, int stride, int pad,
                                int relu, int dtype, void* stream) {
  const DsArgs none = {nullptr, nullptr, 0, 0, 0, 0};
  return conv_igemm_impl(in, w_packed, shift, residual, out, B, Hi, Wi, Cin, Cout, K, stride, pad, relu, dtype, none, stream);
}

extern "C" int frmap_conv3x3_pp_layout(int B, int Hi, int Wi, int Cin, int Cout);

// 1 = the fused-shortcut kernel takes the shape.  FRMAP_DS_UNFUSE_SMALL=1 (A/B switch) answers 0 where the
// second-generation kernel (conv_pp.hip, no shortcut stages yet) would take the plain 3x3 layer and the maps are small
// (14x14 / 7x7), so the caller runs the shortcut as its own 1x1 launch and feeds it as the residual.
extern "C" int frmap_conv_igemm_ds_supported(int B, int Hi, int Wi, int Cin, int Cout, int ds_Hi, int ds_Wi, int ds_Cin,
                                             int ds_stride) {
  static int on = -1, unfuse_small = 0;
  if (on < 0) {
    const char* e = getenv("FRMAP_CONV_DSFUSE");
    on = e ? atoi(e) : 1;
    const char* e2 = getenv("FRMAP_DS_UNFUSE_SMALL");
    unfuse_small = e2 ? atoi(e2) : 0;   // measured a wash end to end (eager +0.2 %, graph replay -3 %): off by default
  }
  if (!on || B <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cin % 32 || Cout <= 0 || Cout % 64) return 0;
  const DsArgs d = {(const void*)1, (const void*)1, ds_Hi, ds_Wi, ds_Cin, ds_stride};
  if (!conv_ds_ok(B, Hi, Wi, Cin, Cout, d)) return 0;
  if (unfuse_small && Hi * Wi <= 256 && frmap_conv3x3_pp_layout(B, Hi, Wi, Cin, Cout) != 0) return 0;
  return 1;
}

extern "C" int frmap_conv_igemm_ds(const void* in, const void* w_packed, const float* shift, const void* ds_in,
                                   const void* ds_w_packed, void* out, int B, int Hi, int Wi, int Cin, int Cout,
                                   int ds_Hi, int ds_Wi, int ds_Cin, int ds_stride, int relu, int dtype, void* stream) {
  FRMAP_REQUIRE(ds_in && ds_w_packed, "conv_igemm_ds: null shortcut pointer");
  const DsArgs d = {ds_in, ds_w_packed, ds_Hi, ds_Wi, ds_Cin, ds_stride};
  return conv_igemm_impl(in, w_packed, shift, nullptr, out, B, Hi, Wi, Cin, Cout, 3, 1, 1, relu, dtype, d, stream);
}

// ------------------------------------------------------------------------------------------------
// conv3x3 s1 p1 + shift (+ReLU) + MaxPool2d(2, 2) in one launch (face_models.py:38-40: BaselineNet's
// `self.pool(F.relu(self.bnK(self.convK(x))))`; :121-141: SiameseNet's conv -> BN -> ReLU -> MaxPool2d(2)).  The conv
// map never reaches HBM: out = [B][Hi/2][Wi/2][Cout].
// ------------------------------------------------------------------------------------------------
static int pool_rows_bound(int BM, int Ho, int Wo, int Hp) {
  const int nw = BM / 4, Wo2 = Wo / 2, Win = (Ho / 2) * Wo2;
  const int pairs = (nw + Wo2 - 2) / Wo2 + 1;      // window rows BM/4 consecutive windows can touch
  const int cross = (nw + Win - 2) / Win;          // image crossings (each adds the Hp - Ho = 2 padding rows)
  const int x = cross < pairs - 1 ? cross : pairs - 1;
  return 2 * pairs + 2 * x + 2;
}

// which fused form takes the shape: 3 = ping-pong kernel (conv_pp.hip), 2 = wave-autonomous kernel (Cin 32 / 64, 8-aligned
// maps), 1 = the generic kernel, 0 = none (odd sizes, rows too wide for LDS)
static int pool2_form(int B, int Hi, int Wi, int Cin, int Cout) {
  if (B <= 0 || Hi <= 0 || Wi <= 0 || Hi % 2 || Wi % 2 || Cin <= 0 || Cin % 32 || Cout <= 0 || Cout % 64) return 0;
  if ((long long)B * Hi * Wi >= (1ll << 31) || Wi + 2 >= 32768 || Hi + 2 >= 32768) return 0;
  static int wres = -1, minc = 128;
  if (wres < 0) { const char* e = getenv("FRMAP_POOL_WAVE"); wres = e ? atoi(e) : 1; const char* e2 = getenv("FRMAP_PP_POOL_MIN_CIN"); minc = e2 ? atoi(e2) : 128; }
  if (Cin >= minc && frmap_conv3x3_pp_pool(nullptr, nullptr, nullptr, nullptr, B, Hi, Wi, Cin, Cout, 0, FRMAP_BF16, nullptr) == 1) return 3;
  if (wres && (Cin == 32 || Cin == 64) && Hi % 8 == 0 && Wi % 8 == 0 && (long long)Hi * Wi * Cin * 2 < (1ll << 31) &&
      (long long)4 * (Wi / 2) * Cout * 2 < (1ll << 31))
    return 2;
  const long long hb = (long long)pool_rows_bound(128, Hi, Wi, Hi + 2) * (Wi + 2) * 64;
  return (hb + 9 * 4096 <= 160 * 1024 && hb / 64 < 65536) ? 1 : 0;
}

// 1 = fusing is expected to win (a fast fused form takes the shape, or the layer is narrow enough that the generic fused
// kernel beats conv + pool launches); frmap_conv_igemm_pool2 itself runs every shape pool2_form() accepts.
extern "C" int frmap_conv_igemm_pool2_supported(int B, int Hi, int Wi, int Cin, int Cout) {
  const int f = pool2_form(B, Hi, Wi, Cin, Cout);
  return f >= 2 || (f == 1 && Cin <= 96);
}
extern "C" int frmap_conv_igemm_pool2_form(int B, int Hi, int Wi, int Cin, int Cout) { return pool2_form(B, Hi, Wi, Cin, Cout); }

extern "C" int frmap_conv_igemm_pool2(const void* in, const void* w_packed, const float* shift, void* out, int B, int Hi,
                                      int Wi, int Cin, int Cout, int relu, int dtype, void* stream) {
  FRMAP_REQUIRE(in && w_packed && shift && out, "conv_igemm_pool2: null pointer");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "conv_igemm_pool2: bad dtype %d", dtype);
  FRMAP_REQUIRE(relu == 0 || relu == 1, "conv_igemm_pool2: activation %d does not commute with the max", relu);
  const int form = pool2_form(B, Hi, Wi, Cin, Cout);
  FRMAP_REQUIRE(form != 0,
                "conv_igemm_pool2: shape B=%d %dx%d Cin=%d Cout=%d not taken (even H and W, Cin %% 32 == 0, Cout %% 64 == 0, rows fit LDS)",
                B, Hi, Wi, Cin, Cout);
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.wpk = w_packed; p.shift = shift; p.res = nullptr; p.out = out;
  p.N = B; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Hi; p.Wo = Wi; p.Cout = Cout;
  p.stride = 1; p.pad = 1; p.relu = relu;
  p.M = B * Hi * Wi; p.HoWo = Hi * Wi; p.Hp = Hi + 2; p.Wp = Wi + 2;
  p.magic_Wp = frmap_magic((uint32_t)p.Wp); p.magic_Hp = frmap_magic((uint32_t)p.Hp);
  p.dHoWo = frmap_div_make((uint32_t)p.HoWo); p.dWo = frmap_div_make((uint32_t)p.Wo);
  p.nchunks = Cin / 32; p.ksplit = 1; p.slab = nullptr; p.dbg = 0;
  p.pool.Wo2 = Wi / 2; p.pool.Win = (Hi / 2) * (Wi / 2);
  p.pool.dWo2 = frmap_div_make((uint32_t)p.pool.Wo2); p.pool.dWin = frmap_div_make((uint32_t)p.pool.Win);
  hipStream_t st = (hipStream_t)stream;
  if (form == 3) {
    const int rc = frmap_conv3x3_pp_pool(in, w_packed, shift, out, B, Hi, Wi, Cin, Cout, relu, dtype, st);
    if (rc < 0) return rc;
    if (rc == 1) return 0;
  }
  // Cin = 32 / 64 on 8-aligned maps: the weights-resident wave-autonomous kernel, pooled epilogue (BaselineNet conv2, conv3)
  if (form == 2) return Cin == 64 ? launch_wave<2, true>(p, dtype, st) : launch_wave<1, true>(p, dtype, st);
  const int wbytes = 9 * 4096, ntiles = Cout / 64;
  int BM = 256;
  long long hb = (long long)pool_rows_bound(256, Hi, Wi, p.Hp) * p.Wp * 64;
  if (hb + wbytes > 80 * 1024) {   // two workgroups per CU when the smaller tile allows it
    BM = 128;
    hb = (long long)pool_rows_bound(128, Hi, Wi, p.Hp) * p.Wp * 64;
  }
  hb = (hb + 1023) & ~1023ll;
  p.halo_bytes = (int)hb;
  p.nblocks = ((p.M + BM - 1) / BM) * ntiles;
  const int lds = p.halo_bytes + wbytes;
  if (BM == 256) return dtype == FRMAP_BF16 ? launch<BF16, 256, 3, 1, true>(p, lds, st) : launch<F16, 256, 3, 1, true>(p, lds, st);
  return dtype == FRMAP_BF16 ? launch<BF16, 128, 3, 1, true>(p, lds, st) : launch<F16, 128, 3, 1, true>(p, lds, st);
}

// ------------------------------------------------------------------------------------------------
// Top-1 match GEMM (frmap_match_top1_packed): probes3 = fp16 [B][3 D] rows (a_hi | a_hi | a_lo), gallery_packed = the
// gallery's (g_hi | g_lo | g_hi) rows in conv-weight order (match_pack_gallery_kernel); every row carries its own
// power-of-two scale, whose inverse is the third float of its statistics record.
// ------------------------------------------------------------------------------------------------
int frmap_match_gemm_f16x3(const void* probes3, const void* gallery_packed, const float* stat_a, const float* stat_w,
                           MatchRec* recs, int B, int G, int D, hipStream_t st) {
  const int K3 = 3 * D, Gpad = (G + 255) / 256 * 256;   // (the packed gallery is padded to 256 rows: frmap_match_gallery_pack_bytes)
  FRMAP_REQUIRE(K3 % 32 == 0, "match: D=%d must be a multiple of 32", D);
  {
    const int rc = frmap_match_gemm_pp(probes3, gallery_packed, stat_a, stat_w, recs, B, G, Gpad, D, st);
    if (rc < 0) return rc;
    if (rc == 1) return 0;
  }
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.in = probes3; p.wpk = gallery_packed;
  p.N = B; p.Hi = 1; p.Wi = 1; p.Cin = K3; p.Ho = 1; p.Wo = 1; p.Cout = Gpad;
  p.stride = 1; p.pad = 0;
  p.M = B; p.HoWo = 1; p.Hp = 1; p.Wp = 1;
  p.magic_Wp = frmap_magic(1u); p.magic_Hp = frmap_magic(1u);
  p.dHoWo = frmap_div_make(1u); p.dWo = frmap_div_make(1u);
  p.ksplit = 1;
  p.nblocks = ((B + 255) / 256) * (Gpad / 64);
  p.m_stat_a = stat_a; p.m_stat_w = stat_w; p.m_recs = recs; p.m_G = G; p.m_D = D;
  const int c32 = K3 / 32;
  if (c32 % 4 == 0) { p.nchunks = c32 / 4; return launch_1x1<F16, 4, true>(p, st); }
  if (c32 % 2 == 0) { p.nchunks = c32 / 2; return launch_1x1<F16, 2, true>(p, st); }
  p.nchunks = c32;
  return launch_1x1<F16, 1, true>(p, st);
}

// ------------------------------------------------------------------------------------------------
// Wide nn.Linear (+ folded BatchNorm1d) (+ReLU/GELU) (+residual) on the MFMA kernel above (1x1 over
// H = W = 1), with split-K when the output has too few tiles to fill the GPU (SiameseNet fc.1:
// 256 x 18432 -> 1024 is 16 tiles but 576 channel chunks).  Partials: each K-slice workgroup
// stores its fp32 tile to its own slab (plain coalesced stores, deterministic), a second kernel
// sums the slabs and applies shift / residual / activation.
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ void splitk_finalize_kernel(const float* __restrict__ slab, int ksplit, const float* __restrict__ shift,
                                       const typename TT::elem* __restrict__ res, typename TT::elem* __restrict__ out,
                                       size_t MN, int N, int act) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (; i < MN; i += stride) {
    f32x4_t a = *(const f32x4_t*)(slab + i);
    for (int s = 1; s < ksplit; ++s) {
      const f32x4_t b = *(const f32x4_t*)(slab + (size_t)s * MN + i);
      a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
    }
    const int n = (int)(i % N);
    const f32x4_t sh = *(const f32x4_t*)(shift + n);
    float v[4] = {a[0] + sh[0], a[1] + sh[1], a[2] + sh[2], a[3] + sh[3]};
    if (res) {
      float r[4];
      unpack4<TT>(*(const u32x2_t*)(res + i), r);
      v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (act == 1) v[e] = fmaxf(v[e], 0.f);
      else if (act == 2) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
    }
    *(u32x2_t*)(out + i) = pack4<TT>(v[0], v[1], v[2], v[3]);
  }
}

static int linear_ksplit(int M, int K, int N) {
  // (batch-invariant planning: the split - and with it the fp32 summation order - is that of a single row tile for every M)
  const int tiles = (frmap_batch_invariant() ? 1 : (M + 255) / 256) * (N / 64), nchunks = K / 32;
  int ks = 384 / (tiles > 0 ? tiles : 1);
  if (ks > nchunks / 4) ks = nchunks / 4;
  return ks < 2 ? 1 : ks;
}

extern "C" size_t frmap_linear_mfma_workspace_bytes(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  const int ks = linear_ksplit(M, K, N);
  return ks > 1 ? (size_t)ks * M * N * sizeof(float) : 0;
}

extern "C" int frmap_linear_mfma(const void* x, const void* w_packed, const float* shift, const void* residual, void* out,
                                 void* workspace, int M, int K, int N, int act, int dtype, void* stream) {
  FRMAP_REQUIRE(x && w_packed && shift && out, "linear_mfma: null pointer");
  FRMAP_REQUIRE(M > 0 && K > 0 && K % 32 == 0 && N > 0 && N % 64 == 0, "linear_mfma: need K %% 32 == 0 and N %% 64 == 0 (M=%d K=%d N=%d)", M, K, N);
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "linear_mfma: bad dtype");
  const int ks = linear_ksplit(M, K, N);
  if (ks == 1) return frmap_conv_igemm(x, w_packed, shift, residual, out, M, 1, 1, K, N, 1, 1, 0, act, dtype, stream);
  FRMAP_REQUIRE(workspace, "linear_mfma: workspace required (frmap_linear_mfma_workspace_bytes)");
  ConvParams p;
  p.in = x; p.wpk = w_packed; p.shift = shift; p.res = nullptr; p.out = out;
  p.N = M; p.Hi = 1; p.Wi = 1; p.Cin = K; p.Ho = 1; p.Wo = 1; p.Cout = N;
  p.stride = 1; p.pad = 0; p.relu = 0;
  p.M = M; p.HoWo = 1; p.Hp = 1; p.Wp = 1;
  p.magic_Wp = frmap_magic(1u); p.magic_Hp = frmap_magic(1u);
  p.dHoWo = frmap_div_make(1u); p.dWo = frmap_div_make(1u);
  p.nchunks = K / 32; p.ksplit = ks; p.slab = (float*)workspace; p.dbg = 0;
  p.halo_bytes = 256 * 64;
  p.nblocks = ((M + 255) / 256) * (N / 64) * ks;
  hipStream_t st = (hipStream_t)stream;
  // K slices are counted in kernel stages (up to 128 channels each)
  {
    const int c32 = K / 32, cks = c32 % 4 == 0 ? 4 : (c32 % 2 == 0 ? 2 : 1);
    if (p.ksplit > c32 / cks) p.ksplit = c32 / cks;
    p.nblocks = ((M + 255) / 256) * (N / 64) * p.ksplit;
  }
  if (p.ksplit < 2) return frmap_conv_igemm(x, w_packed, shift, residual, out, M, 1, 1, K, N, 1, 1, 0, act, dtype, stream);
  int rc = dispatch_1x1(p, dtype, st);
  if (rc) return rc;
  const int ks_used = p.ksplit;
  const size_t MN = (size_t)M * N;
  const int blocks = (int)((MN / 4 + 255) / 256 < 4096 ? (MN / 4 + 255) / 256 : 4096);
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(splitk_finalize_kernel<BF16>, dim3(blocks), dim3(256), 0, st, (const float*)workspace, ks_used, shift,
                       (const __bf16*)residual, (__bf16*)out, MN, N, act);
  else
    hipLaunchKernelGGL(splitk_finalize_kernel<F16>, dim3(blocks), dim3(256), 0, st, (const float*)workspace, ks_used, shift,
                       (const _Float16*)residual, (_Float16*)out, MN, N, act);
  FRMAP_LAUNCH_CHECK();
  return 0;
}
