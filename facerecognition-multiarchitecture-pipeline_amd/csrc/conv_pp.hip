// 3x3 stride-1 implicit-GEMM convolution for gfx950, second generation: ONE 8-wave workgroup per CU, both operands
// brought into LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), two wave groups per SIMD
// running half a phase apart ("ping-pong": one group's MFMAs run beside the other group's fragment reads and DMA issue).
//
// Replaces nn.Conv2d(3x3, s1, p1) + nn.BatchNorm2d (+ReLU) (+residual) of the ResNet-18 BasicBlocks behind
// /root/reference/src/face_models.py:67,463,658 (layers 2-4) and the SiameseNet 3x3 layers (face_models.py:121-141)
// where the first-generation kernels (conv_igemm.hip) were bound by the VGPR->LDS staging path and by tile quantisation.
//
// Tile: tile_px consecutive flattened output pixels (n, oy, ox) x BN = NI*64 output channels.  tile_px <= 2*MI*16 is
// chosen on the host as whole output rows (14x14 maps: one image = 196 pixels per workgroup, 256 faces = 256 CUs).
//   Every wave owns MI x 4 MFMA 16x16x32 tiles (MI*16 pixels x 64 channels; A = 16 output channels, B = 16 pixels).
//   WM = 2: 2 pixel slices x 4 channel slices (BN = 256; group = wave >> 2 is the pixel slice);
//   WM = 4: 4 pixel slices x 2 channel slices (BN = 128, for 128-channel layers; a group holds two pixel slices).
// K loop: k-step = (32-channel chunk, tap).  LDS: two halo images (64 B per pixel, the XOR-swizzled layout of
//   conv3x3_fast_kernel; a tap is an LDS offset) and a ring of four weight slabs (BN x 32 channels, pre-packed in
//   LDS-image order by frmap_pack_conv_weight, so a slab is a straight copy).
// Pipeline per k-step and group:   LOAD(k): ds_read the fragments of k-step k; issue DMA {one piece of the NEXT chunk's
//   halo, slab k+2}; s_waitcnt vmcnt(n) so that only THIS phase's DMA is still in flight (slab k+1 has landed);
//   s_barrier; MFMA(k): lgkmcnt(0), MI*NI MFMAs at raised priority; s_barrier.
//   Group B runs one barrier behind group A, so A's MFMA(k) coincides with B's LOAD(k) and B's MFMA(k) with A's LOAD(k+1).
//   Hazards: a slab is rewritten by DMA issued two phases (four barriers) after the phase that read it (ring of 4,
//   prefetch distance 2) and a halo buffer no earlier than tap 1 of the chunk after its last reader, so every fragment
//   read is retired (the lgkmcnt(0) that opens the reader's MFMA segment) at least one barrier before any DMA into its
//   buffer is issued; DMA'd bytes are read only after the issuing waves' counted vmcnt AND a barrier every wave has passed.  DMA is issued from inline asm (hipcc would
//   otherwise drain vmcnt(0) ahead of every LDS read that may alias an outstanding LDS-DMA write), so every wait on it
//   is hand-counted; the number of DMA instructions per phase is static (slabs / halo pieces past the end of the
//   problem are fetched from valid dummy addresses into buffers nobody reads).
#include "frmap_common.h"
#include <stdlib.h>
#include <string.h>

#include <array>
#include <map>
#include <mutex>
#include <utility>

struct PPParams {
  const void* in;
  const void* wpk;
  const float* shift;
  const void* res;
  void* out;
  int N, Hi, Wi, Cin, Cout, relu;
  int M, HoWo, Hp, Wp;
  uint32_t magic_Wp, magic_Hp;
  FrmapDiv dHoWo, dWo;
  int nchunks;    // Cin / 32
  int tile_px;    // output pixels per tile (<= 2 * MI * 16)
  int mtiles, ntiles;
  // fused 1x1 projection shortcut (DS = true): out += W_ds . x_ds[n, oy * s, ox * s, :]  (extra one-tap k-steps at the end)
  const void* ds_in;   // [N][ds_Hi][ds_Wi][ds_Cin]
  const void* ds_w;    // packed like a 1x1 conv: [Cout/64][ds_Cin/32][1][64][4][8]
  int ds_Hi, ds_Wi, ds_Cin, ds_stride, dsc;   // dsc = ds_Cin / 32
  // fused MaxPool2d(2, 2) (PL = true): each wave's 112-pixel slice (whole row pairs) is walked in pool-major order
  int Wo2;         // Wi / 2
  FrmapDiv dWo2;
  // conv1x1_pp_kernel<..., MATCH = true> (top-1 gallery match, head_match.hip): per-row statistics and the arg-min keys
  const float* m_stat_a;        // [M][4] = (sum a^2, sum a, 1 / row scale, error band) of the fp32 probes
  const float* m_stat_w;        // [G][4] of the fp32 gallery rows
  MatchRec* m_recs;             // [Cout / 64][M] candidate records (frmap_common.h), one writer each
  int m_G, m_D;
};

__device__ __attribute__((aligned(4096))) unsigned int g_pp_zero[1024];

__device__ __forceinline__ int pp_xcd_remap(int b, int nb) {
  const int qd = nb >> 3, rm = nb & 7, xcd = b & 7;
  return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (b >> 3);
}

// one LDS-DMA instruction: every lane copies 16 bytes from its own global address to LDS byte address lds_base + lane*16
// (lds_base wave-uniform, in an SGPR).  M0 carries the LDS base; it is saved and restored inside the statement.
__device__ __forceinline__ void pp_dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_base)
               : "memory");
}
template <int N>
__device__ __forceinline__ void pp_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
__device__ __forceinline__ void pp_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>) - the tap index must be a constant expression
// (it selects the immediates of the hand-counted s_waitcnt instructions)
template <int... Is, typename F>
__device__ __forceinline__ void pp_static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// KS = 1: both wave groups walk every 32-channel chunk (they split the tile's PIXELS).
// KS = 2: in-workgroup split-K for layers with few output pixels (7x7 maps, small batches): the tile is MI*32 pixels x 128
//         channels, each group covers ALL of it (2 pixel slices x 2 channel slices) but only every second chunk, with its
//         own halo buffers and slab ring; at the end group B hands its accumulators to group A through LDS.  Twice the
//         tiles of the KS = 1 layouts at the same per-wave MFMA shape and the same bytes staged per MFMA.
// DS = true (KS = 1 only) folds a ResNet projection shortcut into the layer: out = act(conv3x3(in) + W_ds . x_ds(strided) +
// shift), i.e. BasicBlock.conv2 + bn2 + downsample(conv1x1 s2 + bn) + add + ReLU in one launch.  The shortcut's K dimension
// runs as dsc = ds_Cin/32 extra one-tap phases after the main loop.  Their pixel operand is a GATHER image (the tile's own
// pixels, 64 B each, fetched at (oy*s, ox*s) of x_ds) in a ring of three buffers: the halo buffer the last main chunk does
// not use, an extra buffer behind the slabs, and the last main chunk's halo buffer once that chunk is done; image d + 2 is
// issued in shortcut phase d (images 0 and 1 ride in the last main chunk's idle DMA slots), and the phases that read a
// buffer last retire their reads BEFORE their barrier, so the DMA issued one phase later cannot overtake them.
// IM = true (A/B variant, measured 4-6 % slower: an LDS-DMA instruction between MFMAs stalls the issuing wave ~60 cycles and
// the matrix pipe with it): this phase's DMA (one halo piece, slab k+3) is issued BETWEEN the MFMAs of MFMA(k) instead of in LOAD(k): the
// LOAD segment shrinks to the fragment reads and the waits, so it hides under the other group's MFMA segment; the prefetch
// distance grows to 3 k-steps (a slab is re-targeted in the MFMA segment after the phase that last read it has retired
// its reads), and the wait that closes LOAD(k) leaves exactly MFMA(k-1)'s DMA in flight.
// PL = true (pixel-split layouts, tile = WM full slices, 112 % (2 * Wi) == 0): MaxPool2d(2, 2) fused - a slice is whole row
// pairs, its pixels are enumerated window by window (fragment row i of the slice = pixel i & 3 of window i >> 2), the
// epilogue takes the max over each window's 4 transpose rows and writes the pooled map (conv_epilogue_pool2); SiameseNet
// conv.7-10 / conv.14-17 (face_models.py:127-141).
// RI = true (plain layers): the fragment reads of k-step k + 1 ride BETWEEN the MFMAs of
// MFMA(k) (a pixel fragment's registers are refilled right after its four MFMAs, the weight fragments are double-buffered)
// instead of opening LOAD(k + 1) as one burst of 11 reads per wave: the LOAD segment shrinks to the DMA issue and the counted
// wait, so a group's MFMA segment is what the other group waits for.  Measured before the change: LOAD alone 370 cycles,
// MFMA alone 456, both together 620 per segment.  The reads of k + 1 now run one barrier interval EARLIER than the other
// group's wait for its share of slab k + 1, so (KS = 1, shared buffers) slabs are issued three k-steps ahead into a ring of
// five.  The per-tap fragment addresses are recomputed from 7 base registers (3 VALU each, under the MFMAs): hoisted, the
// 63 of them spill.
template <typename TT, int MI, int WM, int NHP, int KS, bool DS, bool IM = false, bool PL = false, bool RI = false>
__global__ __launch_bounds__(512, 2) void conv3x3_pp_kernel(const PPParams p) {
  static_assert(!(DS && KS == 2), "the fused shortcut is built for the pixel-split layouts only");
  static_assert(!RI || (!DS && !IM), "the interleaved fragment reads are built for the plain layers");
  static_assert(!PL || (KS == 1 && !DS && !IM), "the pooled epilogue is built for the plain pixel-split layouts");
  static_assert(!(DS && IM), "the fused shortcut keeps its DMA in the LOAD segments");
  constexpr int NI = 4, WN = KS == 2 ? 2 : 8 / WM;
  constexpr int CAP = WM * MI * 16;                          // pixels a tile can hold (KS = 1)
  constexpr int NGP = DS ? (CAP / 16 + 7) / 8 : 0;           // gather pieces per wave and shortcut image
  constexpr int GBYTES = NGP * 8 * 1024;                     // bytes of one gather image buffer
  constexpr int XT = DS ? 8 - NHP : 1;                       // idle DMA taps of a main chunk (NHP + 1 .. 8) image 1 can ride in
  constexpr int PPT = DS ? (NGP + XT - 1) / XT : 0;          // ... pieces per such tap
  constexpr int TAPS = 9, BN = WN * 64, WB = BN * 64;      // slab bytes: BN channels x 32 channels x 2 B
  constexpr int GW = 8 / KS;                                 // waves that share one set of LDS buffers
  constexpr int NWI = (WB / 1024) / GW;                      // slab DMA instructions per wave and k-step (2, or 1 for BN = 128 / KS = 1)
  constexpr int RING = (RI && KS == 1) ? 5 : 4;              // weight slabs in LDS (prefetch distance 2: a slab is rewritten
                                                             // two full phases after the phase that last read it)
  constexpr int HB = NHP * GW * 1024;                        // bytes of one halo image buffer
  constexpr int GSZ = 2 * HB + RING * WB;                    // LDS of one buffer set: [halo 0][halo 1][slab 0 .. slab 3]
  constexpr int PD = (RI && KS == 1) ? 3 : 2;                // slab prefetch distance (k-steps) of the non-IM forms
  auto slot_of = [](int k) { return RING == 4 ? (k & 3) : k % RING; };
  using vec8 = typename TT::vec8;
  using elem = typename TT::elem;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, q = wave & 3;
  const int gw = KS == 2 ? q : wave;                                                     // index among the waves sharing the buffers
  const int wn = (KS == 2 || WM == 4) ? (wave & 1) : q;                                  // channel slice (64 channels)
  const int mslice = KS == 2 ? (q >> 1) : (WM == 2 ? grp : grp * 2 + (q >> 1));          // pixel slice (MI * 16 pixels)
  const int gbase = KS == 2 ? grp * GSZ : 0;
  const int lr = lane & 15, g = lane >> 4;

  const int L = pp_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = L / p.ntiles, nt = L - mt * p.ntiles;
  const int m0 = mt * p.tile_px, mend = min(m0 + p.tile_px, p.M);
  const int nch = p.nchunks / KS;            // chunks this group walks (KS = 2: chunk = grp + 2 * ci)
  const int nk = nch * TAPS;                 // its k-steps

  // ---- halo geometry of this tile (rows of the virtual padded row stack, as conv3x3_fast_kernel)
  const int n0 = frmap_div(m0, p.dHoWo), oy0 = frmap_div(m0 - n0 * p.HoWo, p.dWo);
  const int n1 = frmap_div(mend - 1, p.dHoWo), oy1 = frmap_div(mend - 1 - n1 * p.HoWo, p.dWo);
  const int nrows = (n1 - n0) * p.Hp + oy1 - oy0 + 3;
  const int nitems = nrows * p.Wp * 4;  // 16-byte items of one 32-channel halo image

  // ---- per-lane DMA sources
  // halo piece j of this wave = piece (gw + GW*j) of the image: lane -> pixel (piece*16 + lane/4), physical slot lane%4
  const char* hsrc[NHP];
#pragma unroll
  for (int j = 0; j < NHP; ++j) {
    const int item = ((gw + GW * j) << 6) + lane;
    const int px = item >> 2, ps = item & 3;
    const int cg = ps ^ (((px >> 2) & 1) << 1);  // logical 8-channel group stored at this physical slot (the read-side swizzle)
    const int r = (int)fast_div((uint32_t)px, p.magic_Wp);
    const int c = px - r * p.Wp;
    const int rr = oy0 + r;
    const int dn = (int)fast_div((uint32_t)rr, p.magic_Hp);
    const int iy = rr - dn * p.Hp - 1, ix = c - 1, n = n0 + dn;
    const bool ok = item < nitems && n < p.N && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
    hsrc[j] = ok ? (const char*)p.in + ((((size_t)n * p.Hi + iy) * p.Wi + ix) * p.Cin + cg * 8) * sizeof(elem)
                 : (const char*)g_pp_zero + cg * 16;   // padding pixels are DMA'd from 4 KB of zeros
  }
  // slab pieces of this wave: the WB/1024 instructions of a k-step are dealt over the GW waves (instruction i = 4-KB block
  // i/4, 1-KB part i%4): NWI = 2: instructions 2*gw, 2*gw + 1; NWI = 1: instruction gw
  const int wblk = NWI == 2 ? (gw >> 1) : (gw >> 2);
  const int wpart = NWI == 2 ? ((gw & 1) << 1) : (gw & 3);
  const char* wsrc = (const char*)p.wpk + ((size_t)(nt * (BN / 64) + wblk) * p.nchunks * TAPS) * 4096 + wpart * 1024 + lane * 16;
  const unsigned wdst = lds0 + gbase + 2 * HB + wblk * 4096 + wpart * 1024;  // + slot * WB
  const unsigned hdst = lds0 + gbase + gw * 1024;                              // + buffer * HB + j * GW * 1024

  // shortcut (DS): gather piece e of this wave = piece (wave + 8e) of the image: lane -> tile pixel (piece*16 + lane/4)
  const char* gsrc[DS ? NGP : 1];
  const char* wsrc_ds = nullptr;
  if (DS) {
    const size_t img = (size_t)p.ds_Hi * p.ds_Wi * p.ds_Cin;
#pragma unroll
    for (int e = 0; e < NGP; ++e) {
      const int item = ((wave + 8 * e) << 6) + lane;
      const int px = item >> 2, ps = item & 3;
      const int cg = ps ^ (((px >> 2) & 1) << 1);
      const int m = m0 + px;
      const int n = frmap_div(min(m, p.M - 1), p.dHoWo), rem = min(m, p.M - 1) - n * p.HoWo, oy = frmap_div(rem, p.dWo), ox = rem - oy * p.Wi;
      gsrc[e] = (px < CAP && m < mend)
                    ? (const char*)p.ds_in + ((size_t)n * img + ((size_t)(oy * p.ds_stride) * p.ds_Wi + ox * p.ds_stride) * p.ds_Cin + cg * 8) * sizeof(elem)
                    : (const char*)g_pp_zero + cg * 16;
    }
    wsrc_ds = (const char*)p.ds_w + ((size_t)(nt * (BN / 64) + wblk) * p.dsc) * 4096 + wpart * 1024 + lane * 16;
  }
  // gather buffer of shortcut image d: ring {idle halo buffer, extra buffer behind the slabs, the last main chunk's halo buffer}
  auto gbuf = [&](int d) -> unsigned {
    const int r = d % 3;
    return r == 0 ? (unsigned)(nch & 1) * HB : (r == 1 ? (unsigned)(2 * HB + RING * WB) : (unsigned)((nch - 1) & 1) * HB);
  };
  auto issue_gather = [&](int d, int e) {  // piece e of shortcut image d (past the last image: zeros into a buffer nobody reads)
    const char* s = d < p.dsc ? gsrc[e] + (size_t)d * 64 : (const char*)g_pp_zero;
    pp_dma16(s, lds0 + gbuf(d) + wave * 1024 + e * 8192);
  };

  // group-local k-step kk = (chunk counter ci, tap t) -> the layer's k-step (chunk * 9 + t); past the main loop: the
  // shortcut's slabs (DS), then a dummy copy of the last slab into a slot nobody reads
  auto issue_slab = [&](int ci, int t, int slot) {
    const unsigned d = wdst + (unsigned)slot * WB;
    const char* s;
    if (DS && ci >= nch) {
      const int dk = (ci - nch) * TAPS + t;   // shortcut phase this slab belongs to
      s = wsrc_ds + (size_t)(dk < p.dsc ? dk : p.dsc - 1) * 4096;
    } else {
      const int cc = ci < nch ? ci : nch - 1;
      const int chunk = KS == 2 ? grp + 2 * cc : cc;
      s = wsrc + (size_t)(chunk * TAPS + (ci < nch ? t : TAPS - 1)) * 4096;
    }
    pp_dma16(s, d);
    if (NWI == 2) pp_dma16(s + 1024, d + 1024);
  };
  auto issue_halo = [&](int ci, int j) {  // piece j of the group's chunk ci; past the last chunk: shortcut image 0 (DS) or zeros
    if (DS && ci >= nch) {
      if (j < NGP) issue_gather(0, j);
      else pp_dma16((const char*)g_pp_zero, hdst + (unsigned)(ci & 1) * HB + j * (GW * 1024));
      return;
    }
    const int chunk = KS == 2 ? grp + 2 * ci : ci;
    const char* s = ci < nch ? hsrc[j] + (size_t)chunk * 64 : (const char*)g_pp_zero;
    pp_dma16(s, hdst + (unsigned)(ci & 1) * HB + j * (GW * 1024));
  };

  // ---- prologue: halo image of the first chunk, slabs 0 and 1
#pragma unroll
  for (int j = 0; j < NHP; ++j) issue_halo(0, j);
  issue_slab(0, 0, 0);
  issue_slab(0, 1, 1);
  if (IM || PD == 3) issue_slab(0, 2, 2);

  // fragment addressing
  int A[MI];  // (pixel index in the halo image) * 64 + k-group * 16, before the tap offset and the swizzle
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int m = m0 + (mslice * MI + mi) * 16 + lr;
    if (PL) {   // pool-major inside the slice: window w = i >> 2 -> (row pair w / Wo2, column pair w % Wo2), pixel i & 3 of it
      const int i = mi * 16 + lr, w = i >> 2, pr = frmap_div(w, p.dWo2);
      m = m0 + mslice * (MI * 16) + (2 * pr + ((i >> 1) & 1)) * p.Wi + 2 * (w - pr * p.Wo2) + (i & 1);
    }
    m = min(m, mend - 1);
    const int n = frmap_div(m, p.dHoWo), rem = m - n * p.HoWo, oy = frmap_div(rem, p.dWo), ox = rem - oy * p.Wi;
    A[mi] = ((((n - n0) * p.Hp + oy - oy0) * p.Wp + ox) << 6) | (g << 4);
  }
  const int woff = (lr << 6) + ((g ^ (((lr >> 2) & 1) << 1)) << 4);
  const char* halo = smem + gbase;
  const char* slabs = smem + gbase + 2 * HB + wn * 4096 + woff;  // this wave's 64 output channels = 4-KB block wn of a slab

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  pp_wait_vm<0>();
  pp_barrier();
  if (grp == 1) pp_barrier();  // group B runs one barrier behind group A from here on

  // one chunk = nine phases.  LAST (DS only): the chunk before the shortcut phases - its idle DMA slots carry shortcut
  // images 0 and 1 (a separate instantiation of the body, so that the other chunks issue no DMA for the shortcut at all)
  auto chunk_body = [&](int ci, auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
    const char* hb = halo + (ci & 1) * HB;
    if (DS) {   // the shortcut variant is short of registers: recompute the 63 per-tap fragment addresses (3 VALU each)
#pragma unroll  // instead of letting the compiler keep them live across the loop
      for (int mi = 0; mi < MI; ++mi) asm volatile("" : "+v"(A[mi]));
    }
    pp_static_for(std::make_integer_sequence<int, TAPS>{}, [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      const int k = ci * TAPS + t;
      // ---------------- LOAD(k): fragment reads first (their latency runs under the DMA issue and the barrier wait)
      vec8 wf[NI], pf[MI];
      {
        const char* sl = slabs + slot_of(k) * WB;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(sl + ni * 1024);
        const int toff = ((t / 3) * p.Wp + (t % 3)) << 6;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int at = A[mi] + toff;
          pf[mi] = *(const vec8*)(hb + (at ^ ((at >> 3) & 32)));
        }
      }
      constexpr bool HP = t >= 1 && t <= NHP;   // (no halo DMA at tap 0: the other group may still be reading that buffer's last tap)
      if (!IM) {
        if (HP) issue_halo(ci + 1, t - 1);
        // DS: shortcut image 1 rides in the idle taps NHP+1.. of the LAST main chunk (image 0 in its halo slots)
        constexpr int XE = (DS && LAST && t > NHP) ? ((t - NHP - 1) * PPT < NGP ? (NGP - (t - NHP - 1) * PPT < PPT ? NGP - (t - NHP - 1) * PPT : PPT) : 0) : 0;
#pragma unroll
        for (int e = 0; e < XE; ++e) issue_gather(1, (t - NHP - 1) * PPT + e);
        issue_slab(t + 2 < TAPS ? ci : ci + 1, (t + 2) % TAPS, slot_of(k + 2));
        pp_wait_vm<NWI + (HP ? 1 : 0) + XE>();  // everything older than this phase's DMA has landed (slab k+1; next halo by t = 8)
        if (DS && LAST && t == TAPS - 1) pp_wait_lgkm0();  // (the first shortcut phase re-targets this chunk's halo buffer: retire its reads here)
      } else {
        // only the DMA issued in the previous phase's MFMA segment may still be in flight (slab k+2; slab k+1 has landed)
        constexpr int tp = (t + TAPS - 1) % TAPS;
        pp_wait_vm<NWI + ((tp >= 1 && tp <= NHP) ? 1 : 0)>();
      }
      pp_barrier();
      // ---------------- MFMA(k)
      pp_wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
      if (!IM) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
      } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
            constexpr int HPOS = 3, S0POS = MI + 3, S1POS = 2 * MI + 3;   // after which MFMAs the DMA instructions go
            const int idx = ni * MI + mi;
            if ((idx == HPOS && HP) || idx == S0POS || (idx == S1POS && NWI == 2)) {
              __builtin_amdgcn_sched_barrier(0);
              if (idx == HPOS) issue_halo(ci + 1, t - 1);
              else {
                const int ci3 = t + 3 < TAPS ? ci : ci + 1;
                const int cc = ci3 < nch ? ci3 : nch - 1;
                const int chunk = KS == 2 ? grp + 2 * cc : cc;
                const char* sp = wsrc + (size_t)(chunk * TAPS + (ci3 < nch ? (t + 3) % TAPS : TAPS - 1)) * 4096;
                const unsigned dp = wdst + (unsigned)slot_of(k + 3) * WB;
                if (idx == S0POS) pp_dma16(sp, dp);
                else pp_dma16(sp + 1024, dp + 1024);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
      }
      pp_barrier();
    });
  };
  if constexpr (RI) {
    // ---- fragment reads interleaved with the MFMAs (see the kernel comment).  Asw[mi][dx]: swizzled address of pixel
    //      group mi at tap column dx; the tap row adds a multiple of 8 pixels (pitch % 8 == 0), which the swizzle ignores
    auto frag_addr = [&](int mi, int toff) {   // 3 VALU under the MFMAs instead of 21 live registers
      const int at = A[mi] + toff;
      return at ^ ((at >> 3) & 32);
    };
    vec8 pf[MI], wf0[NI], wf1[NI];
    {  // operands of k-step 0 (slab 0 and halo image 0 are in place: prologue wait + barrier)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf0[ni] = *(const vec8*)(slabs + ni * 1024);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) pf[mi] = *(const vec8*)(halo + frag_addr(mi, 0));
    }
    // CP = parity of the chunk index: k = 9 ci + t alternates the two weight-fragment sets, so the nine-phase body exists
    // once per parity and the chunk loop walks pairs
    auto chunk_ri = [&](int ci, auto cp_c) {
      constexpr int CP = decltype(cp_c)::value;
      pp_static_for(std::make_integer_sequence<int, TAPS>{}, [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int CUR = (CP + t) & 1;
        const int k = ci * TAPS + t;
        // ---------------- LOAD(k): this phase's DMA and the counted wait only
        constexpr bool HP = t >= 1 && t <= NHP;
        if (HP) issue_halo(ci + 1, t - 1);
        issue_slab(t + PD < TAPS ? ci : ci + 1, (t + PD) % TAPS, slot_of(k + PD));
        // only this phase's DMA stays in flight: slab k + 2 (issued one phase ago) has landed now, so after the NEXT barrier
        // both groups' shares of it are visible - one whole interval before either group reads it (during its MFMA(k + 1))
        pp_wait_vm<NWI + (HP ? 1 : 0)>();
        pp_barrier();
        // ---------------- MFMA(k), refilling the operand registers with k-step k + 1
        pp_wait_lgkm0();
        __builtin_amdgcn_sched_barrier(0);
        constexpr int t1 = (t + 1) % TAPS;
        const char* hb1 = halo + ((t == TAPS - 1 ? ci + 1 : ci) & 1) * HB;
        const int toff1 = ((t1 / 3) * p.Wp + (t1 % 3)) << 6;
        const char* sl1 = slabs + slot_of(k + 1) * WB;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) asm volatile("" : "+v"(A[mi]));   // keep the 7 base addresses, not 63 derived ones
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = TT::mfma(CUR ? wf1[ni] : wf0[ni], pf[mi], acc[mi][ni]);
          __builtin_amdgcn_sched_barrier(0);
          pf[mi] = *(const vec8*)(hb1 + frag_addr(mi, toff1));
          if (mi < NI) {
            if (CUR) wf0[mi] = *(const vec8*)(sl1 + mi * 1024);
            else wf1[mi] = *(const vec8*)(sl1 + mi * 1024);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        pp_barrier();
      });
    };
    int ci = 0;
    for (; ci + 1 < nch; ci += 2) {
      chunk_ri(ci, std::integral_constant<int, 0>{});
      chunk_ri(ci + 1, std::integral_constant<int, 1>{});
    }
    if (ci < nch) chunk_ri(ci, std::integral_constant<int, 0>{});
  } else if (DS) {
    for (int ci = 0; ci + 1 < nch; ++ci) chunk_body(ci, std::false_type{});
    chunk_body(nch - 1, std::true_type{});
  } else {
    for (int ci = 0; ci < nch; ++ci) chunk_body(ci, std::false_type{});
  }
  if (DS) {
    // ---------------- shortcut phases: one 32-channel one-tap k-step each (k = nk + d)
    int Ag[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int at = ((((mslice * MI + mi) << 4) + lr) << 6) | (g << 4);
      Ag[mi] = at ^ ((at >> 3) & 32);
    }
    for (int d = 0; d < p.dsc; ++d) {
      const int k = nk + d;
      vec8 wf[NI], pf[MI];
      {
        const char* sl = slabs + slot_of(k) * WB;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(sl + ni * 1024);
        const char* gb = smem + gbuf(d);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) pf[mi] = *(const vec8*)(gb + Ag[mi]);
      }
#pragma unroll
      for (int e = 0; e < NGP; ++e) issue_gather(d + 2, e);
      issue_slab(nch + (d + 2) / TAPS, (d + 2) % TAPS, slot_of(k + 2));
      pp_wait_vm<NWI + NGP>();
      pp_wait_lgkm0();   // retired before the barrier: the next phase's DMA re-targets the buffer image d - 1 ... d + 2 share
      pp_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
      pp_barrier();
    }
  }
  if (grp == 0) pp_barrier();  // balance group B's extra start barrier
  pp_wait_vm<0>();             // the dummy DMA of the last two phases must not land in the buffers reused below
  pp_barrier();

  if (KS == 2) {
    // group B's partial sums -> LDS -> group A (wave q of B holds exactly the tiles wave q of A holds)
    char* xch = smem + (size_t)q * (MI * NI * 1024) + lane * 16;
    if (grp == 1) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *(f32x4_t*)(xch + (mi * NI + ni) * 1024) = acc[mi][ni];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const f32x4_t o = *(const f32x4_t*)(xch + (mi * NI + ni) * 1024);
          acc[mi][ni][0] += o[0]; acc[mi][ni][1] += o[1]; acc[mi][ni][2] += o[2]; acc[mi][ni][3] += o[3];
        }
    }
    __syncthreads();         // the exchange area is consumed: group A's transpose scratch may overwrite it
    if (grp == 1) return;    // (no barrier below: conv_epilogue works on per-wave scratch)
  }

  // ---- epilogue: + shift (+ residual) (activation) -> NHWC, whole-line 16-byte stores via a per-wave LDS transpose
  if constexpr (PL)
    conv_epilogue_pool2<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), (m0 + mslice * (MI * 16)) >> 2, mend >> 2, p.Cout,
                                    nt * BN + wn * 64, p.shift, (elem*)p.out, p.relu, lane);
  else
    conv_epilogue<TT, MI, NI>(acc, smem + (KS == 2 ? q : wave) * (16 * (NI * 64 + 16)), m0 + mslice * (MI * 16), mend, p.Cout,
                              nt * BN + wn * 64, p.shift, (const elem*)p.res, (elem*)p.out, p.relu, lane);
}

// ================================================================================================
// 1x1 convolutions and wide Linear layers on the same pipeline (the fused-shortcut phases of conv3x3_pp_kernel on their
// own): one k-step = one 32-channel slice; its pixel operand is a GATHER image of the tile's own pixels (64 B each, DMA'd from
// (oy * s, ox * s) of the NHWC input: any stride, no halo) in a ring of three buffers, its weight operand a BN x 32 slab in a
// ring of four; image and slab k + 2 are issued in phase k, a phase retires its fragment reads BEFORE its barrier (the next
// phase's DMA re-targets the buffer image k - 1 used), the two wave groups run half a phase apart.  Against conv1x1_kernel
// (conv_igemm.hip: 256 px x 64 ch tiles staged through registers, bound by the VGPR -> LDS write rate) a tile here is 224 px x
// 256 ch or 448 px x 128 ch: a quarter / half of the pixel bytes per MFMA, none of them through registers.
// KS = 2: 224 px x 128 ch with the two wave groups splitting K (own buffers; accumulators merged through LDS) for outputs with
// few tiles (Linear 2048 -> 512 over 12 544 tokens: 224 tiles instead of 112).
// ================================================================================================
// MATCH = true: the epilogue of conv1x1_kernel<..., MATCH> (conv_igemm.hip): the GEMM is probes x gallery rows in split fp16
// operands, each lane forms the expanded squared F.pairwise_distance of its 16 gallery rows with its error band, the column's
// four lanes meet through two shuffles, one candidate record per probe and 64-row slot (match_epilogue_records).
template <typename TT, int MI, int WM, int KS, bool MATCH = false>
__global__ __launch_bounds__(512, 2) void conv1x1_pp_kernel(const PPParams p) {
  constexpr int NI = 4, WN = KS == 2 ? 2 : 8 / WM;
  constexpr int CAP = (KS == 2 ? 2 : WM) * MI * 16;          // pixels of a tile
  constexpr int BN = WN * 64, WB = BN * 64;
  constexpr int GW = 8 / KS;                                 // waves that share one set of LDS buffers
  constexpr int NWI = (WB / 1024) / GW;                      // slab DMA instructions per wave and k-step
  constexpr int NGP = (CAP / 16 + GW - 1) / GW;              // gather pieces (16 pixels = 1 KB) per wave and image
  constexpr int IMGB = NGP * GW * 1024;                      // bytes of one gather image buffer
  constexpr int RING = 4;
  constexpr int GSZ = 3 * IMGB + RING * WB;                  // LDS of one buffer set: [image 0..2][slab 0..3]
  using vec8 = typename TT::vec8;
  using elem = typename TT::elem;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, q = wave & 3;
  const int gw = KS == 2 ? q : wave;
  const int wn = (KS == 2 || WM == 4) ? (wave & 1) : q;
  const int mslice = KS == 2 ? (q >> 1) : (WM == 2 ? grp : grp * 2 + (q >> 1));
  const int gbase = KS == 2 ? grp * GSZ : 0;
  const int lr = lane & 15, g = lane >> 4;

  const int L = pp_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = L / p.ntiles, nt = L - mt * p.ntiles;
  const int m0 = mt * p.tile_px, mend = min(m0 + p.tile_px, p.M);
  const int nst = p.nchunks / KS;            // k-steps this group walks (KS = 2: chunk = grp + 2 * step)

  // ---- per-lane DMA sources: gather piece e of this wave = piece (gw + GW * e) of the image: lane -> tile pixel piece * 16 + lane / 4
  const char* gsrc[NGP];
#pragma unroll
  for (int e = 0; e < NGP; ++e) {
    const int item = ((gw + GW * e) << 6) + lane;
    const int px = item >> 2, ps = item & 3;
    const int cg = ps ^ (((px >> 2) & 1) << 1);
    const int m = min(m0 + px, p.M - 1);
    const int n = frmap_div(m, p.dHoWo), rem = m - n * p.HoWo, oy = frmap_div(rem, p.dWo), ox = rem - oy * p.Wp;   // (Wp holds Wo here)
    gsrc[e] = (px < CAP && m0 + px < mend)
                  ? (const char*)p.in + ((((size_t)n * p.Hi + (size_t)(oy * p.ds_stride)) * p.Wi + (size_t)(ox * p.ds_stride)) * p.Cin + cg * 8) * sizeof(elem)
                  : (const char*)g_pp_zero + cg * 16;
  }
  const int wblk = NWI == 2 ? (gw >> 1) : (gw >> 2);
  const int wpart = NWI == 2 ? ((gw & 1) << 1) : (gw & 3);
  const char* wsrc = (const char*)p.wpk + ((size_t)(nt * (BN / 64) + wblk) * p.nchunks) * 4096 + wpart * 1024 + lane * 16;
  const unsigned wdst = lds0 + gbase + 3 * IMGB + wblk * 4096 + wpart * 1024;   // + slot * WB
  const unsigned gdst = lds0 + gbase + gw * 1024;                              // + buffer * IMGB + e * GW * 1024

  auto chunk_of = [&](int st) { return KS == 2 ? grp + 2 * st : st; };
  auto issue_step = [&](int st) {   // image and slab of this group's k-step st (past the end: valid dummies into buffers nobody reads)
    const int sc = st < nst ? st : nst - 1;
    const size_t coff = (size_t)chunk_of(sc) * 64;
#pragma unroll
    for (int e = 0; e < NGP; ++e)
      pp_dma16(st < nst ? gsrc[e] + coff : (const char*)g_pp_zero, gdst + (unsigned)(st % 3) * IMGB + e * (GW * 1024));
    const char* sp = wsrc + (size_t)chunk_of(sc) * 4096;
    const unsigned dp = wdst + (unsigned)(st & (RING - 1)) * WB;
    pp_dma16(sp, dp);
    if (NWI == 2) pp_dma16(sp + 1024, dp + 1024);
  };
  issue_step(0);
  issue_step(1);

  int Ag[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int at = ((((mslice * MI + mi) << 4) + lr) << 6) | (g << 4);
    Ag[mi] = at ^ ((at >> 3) & 32);
  }
  const int woff = (lr << 6) + ((g ^ (((lr >> 2) & 1) << 1)) << 4);
  const char* imgs = smem + gbase;
  const char* slabs = smem + gbase + 3 * IMGB + wn * 4096 + woff;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  pp_wait_vm<0>();
  pp_barrier();
  if (grp == 1) pp_barrier();  // group B runs one barrier behind group A from here on

  for (int st = 0; st < nst; ++st) {
    vec8 wf[NI], pf[MI];
    {
      const char* sl = slabs + (st & (RING - 1)) * WB;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(sl + ni * 1024);
      const char* gb = imgs + (st % 3) * IMGB;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) pf[mi] = *(const vec8*)(gb + Ag[mi]);
    }
    issue_step(st + 2);
    pp_wait_vm<NWI + NGP>();   // everything older than this phase's DMA has landed (step st + 1)
    pp_wait_lgkm0();           // retired before the barrier: the next phase's DMA re-targets the image buffer of step st - 1 ... st + 2
    pp_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
    pp_barrier();
  }
  if (grp == 0) pp_barrier();  // balance group B's extra start barrier
  pp_wait_vm<0>();             // the dummy DMA of the last two phases must not land in the buffers reused below
  pp_barrier();

  if (KS == 2) {
    char* xch = smem + (size_t)q * (MI * NI * 1024) + lane * 16;
    if (grp == 1) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *(f32x4_t*)(xch + (mi * NI + ni) * 1024) = acc[mi][ni];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const f32x4_t o = *(const f32x4_t*)(xch + (mi * NI + ni) * 1024);
          acc[mi][ni][0] += o[0]; acc[mi][ni][1] += o[1]; acc[mi][ni][2] += o[2]; acc[mi][ni][3] += o[3];
        }
    }
    __syncthreads();
    if (grp == 1) return;
  }
  if constexpr (MATCH) {
    match_epilogue_records<MI>(acc, m0 + mslice * (MI * 16), mend, nt * BN + wn * 64, p.m_G, p.m_D, p.M, p.m_stat_a, p.m_stat_w,
                               p.m_recs, lane);
    return;
  }
  conv_epilogue<TT, MI, NI>(acc, smem + (KS == 2 ? q : wave) * (16 * (NI * 64 + 16)), m0 + mslice * (MI * 16), mend, p.Cout,
                            nt * BN + wn * 64, p.shift, (const elem*)p.res, (elem*)p.out, p.relu, lane);
}

// ================================================================================================
// 3x3 STRIDE-2 convolutions (ResNet layer{2,3,4}.0.conv1) on the same pipeline, by space-to-depth ADDRESSING.
//
// out(oy, ox) reads in(2oy + kh - 1, 2ox + kw - 1).  Write 2oy + kh - 1 = 2(oy + a) + dy: kh = 0 -> (a, dy) = (-1, 1),
// kh = 1 -> (0, 0), kh = 2 -> (0, 1), and the same for kw -> (b, dx).  For a fixed sub-position s = (dy, dx) the pixels
// in(2y' + dy, 2x' + dx) form a HALF-resolution map, and the taps that read it are a stride-1 stencil over (a, b) in
// {-1, 0}^2: sub-position (0,0) is read by 1 tap, (0,1) and (1,0) by 2 each, (1,1) by 4 - nine k-steps per 32-channel
// chunk, exactly the layer's own, none padded.  So a chunk's k-steps walk FOUR small halo images (one per sub-position,
// (rows + 1) x (Wo + 1) pixels: only a top / left border; 14 KB for a 14x14 output map, where the first-generation
// kernel stages two 28 KB parity planes through registers), each DMA'd straight from the NHWC input with a per-lane base
// pointer (position (2y', 2x')) plus a SCALAR offset ((dy * Wi + dx) * Cin + chunk * 32 elements).
// Images live in four LDS slots (slot = sub-position); schedule per chunk (one phase per tap, t = 0..8):
//   reads : t = 0 -> image 0 | t = 1,2 -> image 1 | t = 3,4 -> image 2 | t = 5..8 -> image 3
//   DMA   : t = 1,2 -> image 3 of THIS chunk | t = 3,4 / 5,6 / 7,8 -> images 0 / 1 / 2 of the NEXT chunk
//   every image is issued >= 2 phases after its slot's last reader and >= 2 phases before its first reader.
// Everything else (slab ring, the two wave groups half a phase apart, counted waits, epilogue) is conv3x3_pp_kernel's.
// ================================================================================================
template <typename TT, int MI, int WM, int NHP2>
__global__ __launch_bounds__(512, 2) void conv3x3s2_pp_kernel(const PPParams p) {
  constexpr int NI = 4, WN = 8 / WM;
  constexpr int TAPS = 9, BN = WN * 64, WB = BN * 64;
  constexpr int NWI = (WB / 1024) / 8;
  constexpr int RING = 4;
  constexpr int HB = NHP2 * 8 * 1024;  // bytes of one sub-position image slot
  constexpr int PPH = NHP2 >= 2 ? NHP2 / 2 : 1;  // halo pieces per wave and issuing phase (an image is issued over two phases)
  using vec8 = typename TT::vec8;
  using elem = typename TT::elem;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS map: [image slot 0..3][slab 0..3]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, q = wave & 3;
  const int wn = WM == 4 ? (wave & 1) : q;
  const int mslice = WM == 2 ? grp : grp * 2 + (q >> 1);
  const int lr = lane & 15, g = lane >> 4;

  const int L = pp_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = L / p.ntiles, nt = L - mt * p.ntiles;
  const int m0 = mt * p.tile_px, mend = min(m0 + p.tile_px, p.M);
  const int nch = p.nchunks, nk = nch * TAPS;
  const int Wo = p.Wi >> 1;
  // (here p.Hp / p.Wp are the padded HALF-resolution sizes Ho + 1 / Wo + 1, p.HoWo = Ho * Wo, p.dWo divides by Wo)

  const int n0 = frmap_div(m0, p.dHoWo), oy0 = frmap_div(m0 - n0 * p.HoWo, p.dWo);
  const int n1 = frmap_div(mend - 1, p.dHoWo), oy1 = frmap_div(mend - 1 - n1 * p.HoWo, p.dWo);
  const int nrows = (n1 - n0) * p.Hp + oy1 - oy0 + 2;
  const int nitems = nrows * p.Wp * 4;

  // ---- per-lane DMA sources: piece j of this wave = piece (wave + 8j) of an image; base = sub-position (0,0), chunk 0
  const char* hsrc[NHP2];
  unsigned hval = 0;  // bit j: the piece's pixel exists (not border / past the problem): the scalar offsets apply
#pragma unroll
  for (int j = 0; j < NHP2; ++j) {
    const int item = ((wave + 8 * j) << 6) + lane;
    const int px = item >> 2, ps = item & 3;
    const int cg = ps ^ (((px >> 2) & 1) << 1);
    const int r = (int)fast_div((uint32_t)px, p.magic_Wp);
    const int c = px - r * p.Wp;
    const int rr = oy0 + r;
    const int dn = (int)fast_div((uint32_t)rr, p.magic_Hp);
    const int yh = rr - dn * p.Hp - 1, xh = c - 1, n = n0 + dn;   // half-resolution position (y', x'); -1 = border
    const bool ok = item < nitems && n < p.N && yh >= 0 && xh >= 0;
    hsrc[j] = ok ? (const char*)p.in + ((((size_t)n * p.Hi + 2 * yh) * p.Wi + 2 * xh) * p.Cin + cg * 8) * sizeof(elem)
                 : (const char*)g_pp_zero + cg * 16;
    hval |= ok ? (1u << j) : 0u;
  }
  const int wblk = NWI == 2 ? (wave >> 1) : (wave >> 2);
  const int wpart = NWI == 2 ? ((wave & 1) << 1) : (wave & 3);
  const char* wsrc = (const char*)p.wpk + ((size_t)(nt * (BN / 64) + wblk) * p.nchunks * TAPS) * 4096 + wpart * 1024 + lane * 16;
  const unsigned wdst = lds0 + 4 * HB + wblk * 4096 + wpart * 1024;
  const unsigned hdst = lds0 + wave * 1024;
  const int rowb = p.Wi * p.Cin * (int)sizeof(elem), pxb = p.Cin * (int)sizeof(elem);

  // phase t of a chunk: original tap (kh*3 + kw) whose weights it uses, sub-position image, (a + 1, b + 1)
  constexpr int TAP_OF[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};
  constexpr int IMG_OF[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
  constexpr int A1_OF[9] = {1, 1, 1, 0, 1, 0, 0, 1, 1};
  constexpr int B1_OF[9] = {1, 0, 1, 1, 1, 0, 1, 0, 1};

  auto issue_slab = [&](int ci, int t, int slot) {
    const int cc = ci < nch ? ci : nch - 1;
    const int tap = ci < nch ? TAP_OF[t] : 8;
    const char* s = wsrc + (size_t)(cc * TAPS + tap) * 4096;
    const unsigned d = wdst + (unsigned)slot * WB;
    pp_dma16(s, d);
    if (NWI == 2) pp_dma16(s + 1024, d + 1024);
  };
  auto issue_img = [&](int ci, int sp, int j) {  // piece j of sub-position image sp of chunk ci (past the end: zeros)
    const int off = (sp >> 1) * rowb + (sp & 1) * pxb + ci * 64;
    const bool live = ci < nch && ((hval >> j) & 1u);
    const char* s = live ? hsrc[j] + off : (ci < nch ? hsrc[j] : (const char*)g_pp_zero);
    pp_dma16(s, hdst + (unsigned)sp * HB + j * 8192);
  };

  // ---- prologue: images 0, 1, 2 of chunk 0 (image 3 rides in phases 1, 2), slabs of phases 0 and 1
#pragma unroll
  for (int sp = 0; sp < 3; ++sp)
#pragma unroll
    for (int j = 0; j < NHP2; ++j) issue_img(0, sp, j);
  issue_slab(0, 0, 0);
  issue_slab(0, 1, 1);

  int A[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = min(m0 + (mslice * MI + mi) * 16 + lr, mend - 1);
    const int n = frmap_div(m, p.dHoWo), rem = m - n * p.HoWo, oy = frmap_div(rem, p.dWo), ox = rem - oy * Wo;
    A[mi] = ((((n - n0) * p.Hp + oy - oy0) * p.Wp + ox) << 6) | (g << 4);
  }
  const int woff = (lr << 6) + ((g ^ (((lr >> 2) & 1) << 1)) << 4);
  const char* slabs = smem + 4 * HB + wn * 4096 + woff;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  pp_wait_vm<0>();
  pp_barrier();
  if (grp == 1) pp_barrier();

  for (int ci = 0; ci < nch; ++ci) {
    pp_static_for(std::make_integer_sequence<int, TAPS>{}, [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      const int k = ci * TAPS + t;
      vec8 wf[NI], pf[MI];
      {
        const char* sl = slabs + (k & (RING - 1)) * WB;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(sl + ni * 1024);
        const char* hb = smem + IMG_OF[t] * HB;
        const int toff = (A1_OF[t] * p.Wp + B1_OF[t]) << 6;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int at = A[mi] + toff;
          pf[mi] = *(const vec8*)(hb + (at ^ ((at >> 3) & 32)));
        }
      }
      // DMA of this phase: t = 1,2 image 3 of this chunk; t = 3,4 / 5,6 / 7,8 images 0 / 1 / 2 of the next chunk
      constexpr int NH = t >= 1 ? ((NHP2 == 1 && ((t - 1) & 1)) ? 0 : PPH) : 0;
      if (t >= 1) {
        constexpr int qi = (t - 1) >> 1, h = (t - 1) & 1;
#pragma unroll
        for (int e = 0; e < NH; ++e) issue_img(qi == 0 ? ci : ci + 1, qi == 0 ? 3 : qi - 1, h * PPH + e);
      }
      issue_slab(t + 2 < TAPS ? ci : ci + 1, (t + 2) % TAPS, (k + 2) & (RING - 1));
      pp_wait_vm<NWI + NH>();
      pp_barrier();
      pp_wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
      pp_barrier();
    });
  }
  if (grp == 0) pp_barrier();
  pp_wait_vm<0>();
  pp_barrier();

  conv_epilogue<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), m0 + mslice * (MI * 16), mend, p.Cout,
                            nt * BN + wn * 64, p.shift, (const elem*)p.res, (elem*)p.out, p.relu, lane);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// most halo rows any tile touches: rows of the stacked padded maps (hp rows per image) between the tile's first and last
// output row, + extra (s1: 3 = one row above, one below; s2 half-resolution maps: 2) - evaluated exactly as the kernels
// do, over one period of the tile start positions
static int pp_max_rows(long long M, int tile_px, int howo, int wo, int hp, int extra) {
  // (memoised: the planner runs on every launch, the scan is up to one image's worth of tile starts)
  static std::mutex mu;
  static std::map<std::array<long long, 6>, int> memo;
  const std::array<long long, 6> key = {M, tile_px, howo, wo, hp, extra};
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
  }
  int best = 0;
  const long long mtiles = (M + tile_px - 1) / tile_px;
  const long long lim = mtiles < howo ? mtiles : howo;
  for (long long mt = 0; mt < lim; ++mt) {
    const long long m0 = mt * tile_px, mend = (m0 + tile_px < M ? m0 + tile_px : M) - 1;
    const long long n0 = m0 / howo, n1 = mend / howo;
    const int oy0 = (int)((m0 - n0 * howo) / wo), oy1 = (int)((mend - n1 * howo) / wo);
    const int rows = (int)(n1 - n0) * hp + oy1 - oy0 + extra;
    if (rows > best) best = rows;
  }
  {
    std::lock_guard<std::mutex> lock(mu);
    if (memo.size() > 4096) memo.clear();
    memo[key] = best;
  }
  return best;
}

static int pp_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// run-time overrides (frmap_conv_pp_tuning): -1 = not set (environment / heuristic decides)
static int g_pp_on = -1, g_pp_px = -1, g_pp_bn = -1, g_pp_ks = -1, g_pp_ds = -1, g_pp_pitch = -1;
extern "C" int frmap_conv_pp_pitch(int v) { g_pp_pitch = v; return 0; }   // A/B hook: conflict-free halo pitch on (1) / off (0)
extern "C" int frmap_conv_pp_ds(int v) { g_pp_ds = v; return 0; }   // A/B hook: fused-shortcut form on (1) / off (0)

extern "C" int frmap_conv_pp_tuning(int enable, int tile_px, int bn) {
  g_pp_on = enable;
  g_pp_px = tile_px;
  g_pp_bn = bn == 1282 ? 128 : bn;   // (1282: the 128-channel tile with the two wave groups splitting K)
  g_pp_ks = bn == 1282 ? 2 : (bn == 128 || bn == 256 ? 1 : -1);
  return 0;
}

static int g_pp_im = -1;   // -1: environment FRMAP_PP_IM (default 0: measured 4-6 % SLOWER than issuing the DMA in the LOAD segments)
extern "C" int frmap_conv_pp_im(int v) { g_pp_im = v; return 0; }
static int g_pp_ri = -1;   // -1: environment FRMAP_PP_RI; 1: fragment reads interleaved with the MFMAs (RI = true) where the layout allows
extern "C" int frmap_conv_pp_ri(int v) { g_pp_ri = v; return 0; }
static bool pp_ri_on() {
  static int env = -1;
  if (env < 0) env = pp_env("FRMAP_PP_RI", 0);
  return (g_pp_ri >= 0 ? g_pp_ri : env) != 0;
}

// RI = true launcher (plain layers): slab ring of 5 in the shared-buffer layouts
template <typename TT, int MI, int WM, int NHP, int KS>
static int pp_launch_ri(const PPParams& p, hipStream_t st) {
  auto kern = conv3x3_pp_kernel<TT, MI, WM, NHP, KS, false, false, false, true>;
  if (frmap_big_lds((const void*)kern, 160 * 1024)) return -2;
  const int wb = (KS == 2 ? 2 : 8 / WM) * 64 * 64;
  int lds = KS * (2 * NHP * (8 / KS) * 1024 + (KS == 1 ? 5 : 4) * wb);
  const int scratch = 8 * 16 * (4 * 64 + 16);
  const int xch = KS == 2 ? 4 * MI * 4 * 1024 : 0;
  if (lds < scratch) lds = scratch;
  if (lds < xch) lds = xch;
  hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles), dim3(512), lds, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

template <typename TT, int MI, int WM, int NHP, int KS, bool DS = false>
static int pp_launch(const PPParams& p, hipStream_t st) {
  static int im_env = -1;
  if (im_env < 0) im_env = pp_env("FRMAP_PP_IM", 0);
  const bool im = !DS && (g_pp_im >= 0 ? g_pp_im : im_env) != 0;
  typedef void (*kern_t)(const PPParams);
  kern_t kern;
  if constexpr (DS) kern = conv3x3_pp_kernel<TT, MI, WM, NHP, KS, true, false>;
  else kern = im ? (kern_t)conv3x3_pp_kernel<TT, MI, WM, NHP, KS, false, true> : (kern_t)conv3x3_pp_kernel<TT, MI, WM, NHP, KS, false, false>;
  if (frmap_big_lds((const void*)kern, 160 * 1024)) return -2;
  const int wb = (KS == 2 ? 2 : 8 / WM) * 64 * 64;
  int lds = KS * (2 * NHP * (8 / KS) * 1024 + 4 * wb) + (DS ? ((WM * MI + 7) / 8) * 8192 : 0);
  const int scratch = 8 * 16 * (4 * 64 + 16);
  const int xch = KS == 2 ? 4 * MI * 4 * 1024 : 0;
  if (lds < scratch) lds = scratch;
  if (lds < xch) lds = xch;
  hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles), dim3(512), lds, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// (in == nullptr: plan only - returns the layout the layer would take: 1 = 224 px x 256 ch, 2 = 448 px x 128 ch,
//  3 = 224 px x 128 ch with the wave groups splitting K, 0 = not taken.)
// Returns 1 if the layer was launched on conv3x3_pp_kernel, 0 if the shape is not taken (caller falls through to the
// first-generation kernels), negative on a launch error.
int frmap_conv3x3_pp(const void* in, const void* w_packed, const float* shift, const void* residual, void* out, int B, int Hi,
                     int Wi, int Cin, int Cout, int relu, int dtype, hipStream_t st, const FrmapPPShortcut* ds) {
  static int on = -1, force_px = 0, force_bn = 0, min_cin = 128, min_tiles = 200;
  if (on < 0) {
    min_tiles = pp_env("FRMAP_PP_MIN_TILES", 200);  // fewer tiles than this leave CUs idle: the first-generation kernels' smaller tiles win
    min_cin = pp_env("FRMAP_PP_MIN_CIN", 128);   // Cin = 64 layers keep the weights-resident wave kernel by default
    on = pp_env("FRMAP_CONV_PP", 1);
    force_px = pp_env("FRMAP_PP_TILE_PX", 0);
    force_bn = pp_env("FRMAP_PP_BN", 0);
  }
  constexpr int MI = 7;
  if (g_pp_on >= 0 ? !g_pp_on : (!on || Cin < min_cin)) return 0;   // (forced on by the hook: every Cin % 32 == 0)
  if (Cin % 32 || Cin > 1024 || Cout % 128) return 0;
  const long long Mll = (long long)B * Hi * Wi;
  if (Mll >= (1ll << 31) || (long long)B * Hi * Wi * Cin * 2 >= (1ll << 46)) return 0;
  PPParams p;
  p.in = in; p.wpk = w_packed; p.shift = shift; p.res = residual; p.out = out;
  p.N = B; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
  p.M = (int)Mll; p.HoWo = Hi * Wi; p.Hp = Hi + 2; p.Wp = Wi + 2;
  // LDS pitch of a halo row (pixels).  A 16-pixel MFMA fragment that wraps an output row continues Wp - Wi + 1 pixels
  // further on; with Wp = Wi (mod 8) that keeps the fragment's 16-byte slots on distinct banks (the swizzle repeats every 8
  // pixels).  Columns past Wi + 1 are just more zero padding for the DMA.  (A/B: FRMAP_PP_PITCH)
  static int pitch_env = -1;
  if (pitch_env < 0) pitch_env = pp_env("FRMAP_PP_PITCH", 0);
  const bool wide_pitch = (g_pp_pitch >= 0 ? g_pp_pitch : pitch_env) != 0;
  if (wide_pitch) { while (p.Wp % 8 != Wi % 8) ++p.Wp; }
  const bool ri_want = pp_ri_on() && ds == nullptr;   // RI form (plain layers): fragment reads under the MFMAs
  p.magic_Wp = frmap_magic((uint32_t)p.Wp); p.magic_Hp = frmap_magic((uint32_t)p.Hp);
  p.dHoWo = frmap_div_make((uint32_t)p.HoWo); p.dWo = frmap_div_make((uint32_t)Wi);
  p.nchunks = Cin / 32;
  p.ds_in = nullptr; p.ds_w = nullptr; p.ds_Hi = p.ds_Wi = p.ds_Cin = p.ds_stride = p.dsc = 0;
  const bool has_ds = ds != nullptr;
  if (has_ds) {
    if (residual || ds->Cin <= 0 || ds->Cin % 32 || ds->stride < 1 || (ds->Hi - 1) / ds->stride + 1 != Hi ||
        (ds->Wi - 1) / ds->stride + 1 != Wi || (long long)B * ds->Hi * ds->Wi * ds->Cin * 2 >= (1ll << 46))
      return 0;
    // A/B switch FRMAP_CONV_PP_DS (on): with the last main chunk peeled (no shortcut DMA in the other chunks) the fused form
    // runs 64.8 / 52.9 us at 28x28 / 14x14 (256 faces) against the first generation's 72.4 / 68.4
    static int ds_on = -1;
    if (ds_on < 0) ds_on = pp_env("FRMAP_CONV_PP_DS", 1);
    if (g_pp_ds >= 0) ds_on = g_pp_ds;
    if (!ds_on && g_pp_on < 0) return 0;
    p.ds_in = ds->in; p.ds_w = ds->w; p.ds_Hi = ds->Hi; p.ds_Wi = ds->Wi; p.ds_Cin = ds->Cin; p.ds_stride = ds->stride;
    p.dsc = ds->Cin / 32;
  }
  // A layout = (pixels a tile can hold, channel tile, split-K groups).  Pixels per tile: whole images when they fit
  // (7x7: 4 per 224, 14x14: 1), else whole rows - a divisor of the image height when one is within 1/8 of the capacity
  // (28 rows, capacity 16 rows: 14), so tiles do not straddle images.  Returns the halo pieces (KB / waves) needed, 0 = no fit.
  auto plan = [&](int cap, int bn, int ks, int& tile_px, int& mtiles, int& ntiles) -> int {
    if (Wi > cap || Cout % bn || (ks == 2 && (Cin / 32) % 2)) return 0;
    if (Hi * Wi <= cap) tile_px = (cap / (Hi * Wi)) * Hi * Wi;
    else {
      int rows = cap / Wi;
      for (int r = rows; r * 8 >= rows * 7 && r >= 1; --r)
        if (Hi % r == 0) { rows = r; break; }
      tile_px = rows * Wi;
    }
    if (force_px > 0 && force_px <= cap) tile_px = force_px;
    if (g_pp_px > 0 && g_pp_px <= cap) tile_px = g_pp_px;
    mtiles = (int)((Mll + tile_px - 1) / tile_px);
    ntiles = Cout / bn;
    const long long hbytes = (long long)pp_max_rows(Mll, tile_px, Hi * Wi, Wi, p.Hp, 3) * p.Wp * 64;
    if (hbytes / 64 >= 65536) return 0;
    const int per = (8 / ks) * 1024;                           // bytes one "piece per wave" adds to the image
    const int nhp = (int)((hbytes + per - 1) / per);
    return nhp <= (ks == 2 ? 6 : 5) ? nhp : 0;
  };
  // candidates: 224 px x 256 ch; 448 px x 128 ch; split-K 224 px x 128 ch (twice the tiles of either)
  int bn_pref = Cout % 256 == 0 ? 256 : 128;
  if (force_bn == 128 || force_bn == 256) bn_pref = (force_bn == 256 && Cout % 256) ? 128 : force_bn;
  if (g_pp_bn == 128 || g_pp_bn == 256) bn_pref = (g_pp_bn == 256 && Cout % 256) ? 128 : g_pp_bn;
  int tpx = 0, mtl = 0, ntl = 0, ks = 1, bn = bn_pref;
  int nhp = plan(bn == 256 ? 2 * MI * 16 : 4 * MI * 16, bn, 1, tpx, mtl, ntl);
  const bool inv = frmap_batch_invariant() != 0;   // layout from the per-image geometry alone: no tile-count rules, no split-K
  const bool want_ks2 = !has_ds && !inv && (g_pp_ks == 2 || (g_pp_ks < 0 && (!nhp || (long long)mtl * ntl < min_tiles)));
  if (want_ks2 && g_pp_ks != 1) {
    int t2 = 0, m2 = 0, n2 = 0;
    const int nhp2 = plan(2 * MI * 16, 128, 2, t2, m2, n2);
    if (nhp2 && (g_pp_ks == 2 || !nhp || (long long)m2 * n2 > (long long)mtl * ntl)) {
      nhp = nhp2; tpx = t2; mtl = m2; ntl = n2; ks = 2; bn = 128;
    }
  }
  if (!nhp) return 0;
  p.tile_px = tpx; p.mtiles = mtl; p.ntiles = ntl;
  if (!inv && g_pp_on < 0 && (long long)mtl * ntl < min_tiles / 2) return 0;   // too few tiles even with split-K: the smaller first-generation tiles win
  if (!inv && has_ds && (long long)mtl * ntl < min_tiles && g_pp_on < 0) return 0;   // (no split-K form of the shortcut kernel)
  if (!in) return ks == 2 ? 3 : (bn == 256 ? 1 : 2);                   // plan-only query (frmap_conv3x3_pp_layout)
  int rc;
  if (has_ds) {   // pixel-split layouts only; the 448-pixel layout needs the 40 KB halo buffers to hold a gather image
#define PPD_GO(TT)                                                                                                       \
  (bn == 256 ? (nhp <= 3 ? pp_launch<TT, MI, 2, 3, 1, true>(p, st) : pp_launch<TT, MI, 2, 5, 1, true>(p, st))              \
             : pp_launch<TT, MI, 4, 5, 1, true>(p, st))
    rc = dtype == FRMAP_BF16 ? PPD_GO(BF16) : PPD_GO(F16);
#undef PPD_GO
    return rc ? rc : 1;
  }
  // (the RI form of the split-K layout with 6 halo pieces needs 258 VGPRs: it would spill inside the DMA-counted loop, so that
  //  one layout keeps the burst-read form; csrc/build.sh rejects any *_pp_kernel with scratch)
  if (ri_want && !(ks == 2 && nhp > 4)) {
#define PPR_GO(TT)                                                                                              \
  (ks == 2 ? pp_launch_ri<TT, MI, 2, 4, 2>(p, st)                                                               \
   : bn == 256 ? (nhp <= 3 ? pp_launch_ri<TT, MI, 2, 3, 1>(p, st) : pp_launch_ri<TT, MI, 2, 5, 1>(p, st))        \
               : (nhp <= 3 ? pp_launch_ri<TT, MI, 4, 3, 1>(p, st) : pp_launch_ri<TT, MI, 4, 5, 1>(p, st)))
    rc = dtype == FRMAP_BF16 ? PPR_GO(BF16) : PPR_GO(F16);
#undef PPR_GO
    return rc ? rc : 1;
  }
#define PP_GO(TT)                                                                                               \
  (ks == 2 ? (nhp <= 4 ? pp_launch<TT, MI, 2, 4, 2>(p, st) : pp_launch<TT, MI, 2, 6, 2>(p, st))                  \
   : bn == 256 ? (nhp <= 3 ? pp_launch<TT, MI, 2, 3, 1>(p, st) : pp_launch<TT, MI, 2, 5, 1>(p, st))              \
               : (nhp <= 3 ? pp_launch<TT, MI, 4, 3, 1>(p, st) : pp_launch<TT, MI, 4, 5, 1>(p, st)))
  rc = dtype == FRMAP_BF16 ? PP_GO(BF16) : PP_GO(F16);
#undef PP_GO
  return rc ? rc : 1;
}

// ------------------------------------------------------------------------------------------------
// conv3x3 s1 p1 + shift (+ReLU) + MaxPool2d(2, 2) on the ping-pong kernel (PL = true).  Takes maps whose row pairs tile a
// wave's 112-pixel slice (Wi in {2, 4, 8, 14, 28, 56}, even Hi), Cin % 32 == 0, Cout % 128 == 0; tiles are WM whole slices.
// 1 = launched, 0 = shape not taken, < 0 = error; in == nullptr: plan only.
// ------------------------------------------------------------------------------------------------
template <typename TT, int WM, int NHP>
static int pp_launch_pool(const PPParams& p, hipStream_t st) {
  auto kern = conv3x3_pp_kernel<TT, 7, WM, NHP, 1, false, false, true>;
  if (frmap_big_lds((const void*)kern, 160 * 1024)) return -2;
  const int wb = (8 / WM) * 64 * 64;
  int lds = 2 * NHP * 8 * 1024 + 4 * wb;
  const int scratch = 8 * 16 * (4 * 64 + 16);
  if (lds < scratch) lds = scratch;
  hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles), dim3(512), lds, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

int frmap_conv3x3_pp_pool(const void* in, const void* w_packed, const float* shift, void* out, int B, int Hi, int Wi, int Cin,
                          int Cout, int relu, int dtype, hipStream_t st) {
  static int on = -1;
  if (on < 0) on = pp_env("FRMAP_CONV_PP", 1) && pp_env("FRMAP_CONV_PP_POOL", 1);
  if (g_pp_on >= 0 ? !g_pp_on : !on) return 0;
  if (Hi % 2 || Wi % 2 || 112 % (2 * Wi) || Cin % 32 || Cin > 1024 || Cout % 128) return 0;
  const long long Mll = (long long)B * Hi * Wi;
  if (Mll >= (1ll << 31) || (long long)B * Hi * Wi * Cin * 2 >= (1ll << 46)) return 0;
  PPParams p;
  p.in = in; p.wpk = w_packed; p.shift = shift; p.res = nullptr; p.out = out;
  p.N = B; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
  p.M = (int)Mll; p.HoWo = Hi * Wi; p.Hp = Hi + 2; p.Wp = Wi + 2;
  p.magic_Wp = frmap_magic((uint32_t)p.Wp); p.magic_Hp = frmap_magic((uint32_t)p.Hp);
  p.dHoWo = frmap_div_make((uint32_t)p.HoWo); p.dWo = frmap_div_make((uint32_t)Wi);
  p.nchunks = Cin / 32;
  p.ds_in = nullptr; p.ds_w = nullptr; p.ds_Hi = p.ds_Wi = p.ds_Cin = p.ds_stride = p.dsc = 0;
  p.Wo2 = Wi / 2; p.dWo2 = frmap_div_make((uint32_t)p.Wo2);
  int bn = Cout % 256 == 0 ? 256 : 128;
  if (g_pp_bn == 128 || g_pp_bn == 256) bn = (g_pp_bn == 256 && Cout % 256) ? 128 : g_pp_bn;
  int nhp = 0;
  for (int attempt = 0; attempt < 2 && !nhp; ++attempt) {
    const int tile_px = (bn == 256 ? 2 : 4) * 112;
    const long long hbytes = (long long)pp_max_rows(Mll, tile_px, Hi * Wi, Wi, p.Hp, 3) * p.Wp * 64;
    const int n = (int)((hbytes + 8191) / 8192);
    if (hbytes / 64 < 65536 && n <= 5) { nhp = n; p.tile_px = tile_px; }
    else bn = bn == 256 ? 128 : 256;   // the other layout (a wider tile has fewer halo rows per pixel, a narrower one fewer rows)
    if (!nhp && Cout % bn) break;
  }
  if (!nhp) return 0;
  p.mtiles = (int)((Mll + p.tile_px - 1) / p.tile_px); p.ntiles = Cout / bn;
  if (!in) return 1;
  int rc;
#define PPP_GO(TT)                                                                                      \
  (bn == 256 ? (nhp <= 3 ? pp_launch_pool<TT, 2, 3>(p, st) : pp_launch_pool<TT, 2, 5>(p, st))           \
             : (nhp <= 3 ? pp_launch_pool<TT, 4, 3>(p, st) : pp_launch_pool<TT, 4, 5>(p, st)))
  rc = dtype == FRMAP_BF16 ? PPP_GO(BF16) : PPP_GO(F16);
#undef PPP_GO
  return rc ? rc : 1;
}

// ------------------------------------------------------------------------------------------------
// 1x1 conv / Linear launcher (conv1x1_pp_kernel): 1 = launched, 0 = shape not taken, < 0 = error; in == nullptr: plan only
// (returns the layout: 1 = 224 px x 256 ch, 2 = 448 px x 128 ch, 3 = 224 px x 128 ch split-K)
// ------------------------------------------------------------------------------------------------
template <typename TT, int WM, int KS, bool MATCH = false>
static int pp1_launch(const PPParams& p, hipStream_t st) {
  constexpr int MI = 7;
  auto kern = conv1x1_pp_kernel<TT, MI, WM, KS, MATCH>;
  if (frmap_big_lds((const void*)kern, 160 * 1024)) return -2;
  constexpr int CAP = (KS == 2 ? 2 : WM) * MI * 16, GW = 8 / KS, NGP = (CAP / 16 + GW - 1) / GW;
  const int wb = (KS == 2 ? 2 : 8 / WM) * 64 * 64;
  int lds = KS * (3 * NGP * GW * 1024 + 4 * wb);
  const int scratch = 8 * 16 * (4 * 64 + 16);
  const int xch = KS == 2 ? 4 * MI * 4 * 1024 : 0;
  if (lds < scratch) lds = scratch;
  if (lds < xch) lds = xch;
  hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles), dim3(512), lds, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

int frmap_conv1x1_pp(const void* in, const void* w_packed, const float* shift, const void* residual, void* out, int B, int Hi,
                     int Wi, int Cin, int Cout, int stride, int relu, int dtype, hipStream_t st) {
  static int on = -1, min_tiles = 200;
  if (on < 0) {
    on = pp_env("FRMAP_CONV_PP", 1) && pp_env("FRMAP_CONV_PP_1X1", 1);
    min_tiles = pp_env("FRMAP_PP_MIN_TILES", 200);
  }
  if (g_pp_on >= 0 ? !g_pp_on : !on) return 0;
  if (frmap_batch_invariant() && g_pp_on < 0) return 0;   // (its layout is a rounds x time estimate over the tile count: the first-generation kernel's is not)
  if (Cin % 32 || Cin > 16384 || Cout % 128 || stride < 1) return 0;
  const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
  const long long Mll = (long long)B * Ho * Wo;
  if (Mll >= (1ll << 31) || (long long)B * Hi * Wi * Cin * 2 >= (1ll << 46)) return 0;
  PPParams p;
  p.in = in; p.wpk = w_packed; p.shift = shift; p.res = residual; p.out = out;
  p.N = B; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
  p.M = (int)Mll; p.HoWo = Ho * Wo; p.Hp = Ho; p.Wp = Wo;          // (output geometry: the kernel needs no padded map)
  p.magic_Wp = frmap_magic((uint32_t)p.Wp); p.magic_Hp = frmap_magic((uint32_t)p.Hp);
  p.dHoWo = frmap_div_make((uint32_t)p.HoWo); p.dWo = frmap_div_make((uint32_t)Wo);
  p.nchunks = Cin / 32;
  p.ds_in = nullptr; p.ds_w = nullptr; p.ds_Hi = p.ds_Wi = p.ds_Cin = p.dsc = 0; p.ds_stride = stride;
  p.Wo2 = 0; p.dWo2 = frmap_div_make(1u);
  // layouts: 1 = 224 px x 256 ch, 2 = 448 px x 128 ch, 3 = split-K 224 px x 128 ch.  One workgroup per CU, so what counts is
  // ROUNDS x time per tile: est = ceil(tiles / CUs) x (k-steps per group x 0.6 us + 7.5 us of prologue and epilogue); 280 tiles
  // on 256 CUs are two rounds.  The first-generation kernel (small tiles, two workgroups per CU) is modelled at 470 TFLOP/s.
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  auto est_us = [&](int lay) -> double {
    if (lay == 1 && Cout % 256) return 1e30;
    if (lay == 3 && p.nchunks % 2) return 1e30;
    const int px = lay == 2 ? 448 : 224, bn = lay == 1 ? 256 : 128;
    const long long tiles = ((Mll + px - 1) / px) * (Cout / bn);
    const long long rounds = (tiles + ncu - 1) / ncu;
    return (double)rounds * ((lay == 3 ? p.nchunks / 2 : p.nchunks) * 0.6 + 7.5);
  };
  int layout = 0;
  double best = 1e30;
  for (int lay = 1; lay <= 3; ++lay) {
    if (g_pp_bn == 256 && g_pp_ks != 2 && lay != 1 && Cout % 256 == 0) continue;   // forced by the tuning hook
    if (g_pp_bn == 128 && g_pp_ks == 1 && lay != 2) continue;
    if (g_pp_ks == 2 && lay != 3 && p.nchunks % 2 == 0) continue;
    const double e = est_us(lay);
    if (e < best) { best = e; layout = lay; }
  }
  if (!layout || best >= 1e30) return 0;
  if (g_pp_on < 0) {
    const double gen1_us = 2.0 * (double)Mll * Cin * Cout / 470e6;
    if (best > 0.9 * gen1_us || min_tiles < 0) return 0;   // (a clear win only: AttentionNet's 640-channel q/k/v conv is 140 tiles - 29 us here, 19 there)
  }
  p.tile_px = layout == 2 ? 448 : 224;
  p.mtiles = (int)((Mll + p.tile_px - 1) / p.tile_px);
  p.ntiles = Cout / (layout == 1 ? 256 : 128);
  if (!in) return layout;
  int rc;
#define PP1_GO(TT) (layout == 1 ? pp1_launch<TT, 2, 1>(p, st) : layout == 2 ? pp1_launch<TT, 4, 1>(p, st) : pp1_launch<TT, 2, 2>(p, st))
  rc = dtype == FRMAP_BF16 ? PP1_GO(BF16) : PP1_GO(F16);
#undef PP1_GO
  return rc ? rc : 1;
}

// top-1 match GEMM on the same kernel (see frmap_match_gemm_f16x3 in conv_igemm.hip): G padded to 256 rows, K3 = 3 D.
// 1 = launched, 0 = not taken
int frmap_match_gemm_pp(const void* probes3, const void* gallery_packed, const float* stat_a, const float* stat_w,
                        MatchRec* recs, int B, int G, int Gpad, int D, hipStream_t st) {
  static int on = -1;
  if (on < 0) on = pp_env("FRMAP_CONV_PP", 1) && pp_env("FRMAP_MATCH_PP", 1);
  const int K3 = 3 * D;
  if (!on || Gpad % 256 || K3 % 32 || K3 > 16384 || B <= 0) return 0;
  PPParams p;
  memset(&p, 0, sizeof(p));
  p.in = probes3; p.wpk = gallery_packed;
  p.N = B; p.Hi = 1; p.Wi = 1; p.Cin = K3; p.Cout = Gpad;
  p.M = B; p.HoWo = 1; p.Hp = 1; p.Wp = 1;
  p.magic_Wp = frmap_magic(1u); p.magic_Hp = frmap_magic(1u);
  p.dHoWo = frmap_div_make(1u); p.dWo = frmap_div_make(1u); p.dWo2 = frmap_div_make(1u);
  p.nchunks = K3 / 32; p.ds_stride = 1;
  p.m_stat_a = stat_a; p.m_stat_w = stat_w; p.m_recs = recs; p.m_G = G; p.m_D = D;
  // 224 probes x 256 gallery rows per tile, or 448 x 128 when that fills the CUs better (one round either way at 1024 probes)
  const long long t1 = ((B + 223) / 224) * (long long)(Gpad / 256), t2 = ((B + 447) / 448) * (long long)(Gpad / 128);
  const long long r1 = (t1 + 255) / 256, r2 = (t2 + 255) / 256;   // rounds on 256 CUs (a tile costs the same in both layouts)
  const bool wide = r2 < r1 || (r2 == r1 && t2 > t1);             // same rounds: the layout that occupies more CUs
  p.tile_px = wide ? 448 : 224;
  p.mtiles = (B + p.tile_px - 1) / p.tile_px;
  p.ntiles = Gpad / (wide ? 128 : 256);
  const int rc = wide ? pp1_launch<F16, 4, 1, true>(p, st) : pp1_launch<F16, 2, 1, true>(p, st);
  return rc ? rc : 1;
}

extern "C" int frmap_conv1x1_pp_layout(int B, int Hi, int Wi, int Cin, int Cout, int stride) {
  if (B <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0) return 0;
  return frmap_conv1x1_pp(nullptr, nullptr, nullptr, nullptr, nullptr, B, Hi, Wi, Cin, Cout, stride, 0, FRMAP_BF16, nullptr);
}

extern "C" int frmap_conv3x3_pp_pool_layout(int B, int Hi, int Wi, int Cin, int Cout) {
  if (B <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0) return 0;
  return frmap_conv3x3_pp_pool(nullptr, nullptr, nullptr, nullptr, B, Hi, Wi, Cin, Cout, 0, FRMAP_BF16, nullptr);
}

extern "C" int frmap_conv3x3_pp_layout(int B, int Hi, int Wi, int Cin, int Cout) {
  if (B <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0) return 0;
  return frmap_conv3x3_pp(nullptr, nullptr, nullptr, nullptr, nullptr, B, Hi, Wi, Cin, Cout, 0, FRMAP_BF16, nullptr, nullptr);
}

// the same question for the layer with a fused 1x1 stride-s projection shortcut (frmap_conv_igemm_ds)
extern "C" int frmap_conv3x3_pp_ds_layout(int B, int Hi, int Wi, int Cin, int Cout, int ds_Hi, int ds_Wi, int ds_Cin, int ds_stride) {
  if (B <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const FrmapPPShortcut d = {nullptr, nullptr, ds_Hi, ds_Wi, ds_Cin, ds_stride};
  return frmap_conv3x3_pp(nullptr, nullptr, nullptr, nullptr, nullptr, B, Hi, Wi, Cin, Cout, 0, FRMAP_BF16, nullptr, &d);
}

// ------------------------------------------------------------------------------------------------
// stride-2 launcher (conv3x3s2_pp_kernel)
// ------------------------------------------------------------------------------------------------
template <typename TT, int MI, int WM, int NHP2>
static int pp2_launch(const PPParams& p, hipStream_t st) {
  auto kern = conv3x3s2_pp_kernel<TT, MI, WM, NHP2>;
  if (frmap_big_lds((const void*)kern, 160 * 1024)) return -2;
  int lds = 4 * NHP2 * 8192 + 4 * (8 / WM) * 64 * 64;
  const int scratch = 8 * 16 * (4 * 64 + 16);
  if (lds < scratch) lds = scratch;
  hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles), dim3(512), lds, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// 3x3 stride-2 pad-1 layer: 1 = launched on conv3x3s2_pp_kernel, 0 = shape not taken, < 0 = error (in == nullptr: plan only)
int frmap_conv3x3s2_pp(const void* in, const void* w_packed, const float* shift, const void* residual, void* out, int B, int Hi,
                       int Wi, int Cin, int Cout, int relu, int dtype, hipStream_t st) {
  static int on = -1, min_tiles = 200, min_cin = 64;
  if (on < 0) {
    min_tiles = pp_env("FRMAP_PP_MIN_TILES", 200);
    min_cin = pp_env("FRMAP_PP_S2_MIN_CIN", 64);
    on = pp_env("FRMAP_CONV_PP", 1) && pp_env("FRMAP_CONV_PP_S2", 1);
  }
  constexpr int MI = 7;
  if (g_pp_on >= 0 ? !g_pp_on : (!on || Cin < min_cin)) return 0;
  if (Hi % 2 || Wi % 2 || Cin % 32 || Cin > 1024 || Cout % 128) return 0;
  const int Ho = Hi / 2, Wo = Wi / 2;
  const long long Mll = (long long)B * Ho * Wo;
  if (Mll >= (1ll << 31) || (long long)B * Hi * Wi * Cin * 2 >= (1ll << 46) || (long long)Hi * Wi * Cin * 2 >= (1ll << 31)) return 0;
  int bn = Cout % 256 == 0 ? 256 : 128;
  if (g_pp_bn == 128 || g_pp_bn == 256) bn = (g_pp_bn == 256 && Cout % 256) ? 128 : g_pp_bn;
  const int cap = (bn == 256 ? 2 : 4) * MI * 16;
  if (Wo > cap) return 0;
  int tile_px;
  if (Ho * Wo <= cap) tile_px = (cap / (Ho * Wo)) * Ho * Wo;
  else {
    int rows = cap / Wo;
    for (int r = rows; r * 8 >= rows * 7 && r >= 1; --r)
      if (Ho % r == 0) { rows = r; break; }
    tile_px = rows * Wo;
  }
  if (g_pp_px > 0 && g_pp_px <= cap) tile_px = g_pp_px;
  PPParams p;
  p.in = in; p.wpk = w_packed; p.shift = shift; p.res = residual; p.out = out;
  p.N = B; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
  p.M = (int)Mll; p.HoWo = Ho * Wo; p.Hp = Ho + 1; p.Wp = Wo + 1;   // half-resolution maps carry a top / left border only
  p.magic_Wp = frmap_magic((uint32_t)p.Wp); p.magic_Hp = frmap_magic((uint32_t)p.Hp);
  p.dHoWo = frmap_div_make((uint32_t)p.HoWo); p.dWo = frmap_div_make((uint32_t)Wo);
  p.nchunks = Cin / 32;
  p.tile_px = tile_px;
  p.mtiles = (int)((Mll + tile_px - 1) / tile_px);
  p.ntiles = Cout / bn;
  const long long hbytes = (long long)pp_max_rows(Mll, tile_px, Ho * Wo, Wo, p.Hp, 2) * p.Wp * 64;
  if (hbytes / 64 >= 65536) return 0;
  const int need = (int)((hbytes + 8191) / 8192);
  const int nhp = need <= 1 ? 1 : (need <= 2 ? 2 : (need <= 4 ? 4 : 0));
  if (!nhp || (nhp == 4 && bn == 256)) return 0;                       // (4 x 32 KB images + 4 x 16 KB slabs would not fit)
  if (!frmap_batch_invariant() && g_pp_on < 0 && (long long)p.mtiles * p.ntiles < min_tiles) return 0;
  if (!in) return bn == 256 ? 1 : 2;
  int rc;
#define PP2_GO(TT)                                                                                                  \
  (bn == 256 ? (nhp == 1 ? pp2_launch<TT, MI, 2, 1>(p, st) : pp2_launch<TT, MI, 2, 2>(p, st))                        \
             : (nhp == 1 ? pp2_launch<TT, MI, 4, 1>(p, st) : nhp == 2 ? pp2_launch<TT, MI, 4, 2>(p, st) : pp2_launch<TT, MI, 4, 4>(p, st)))
  rc = dtype == FRMAP_BF16 ? PP2_GO(BF16) : PP2_GO(F16);
#undef PP2_GO
  return rc ? rc : 1;
}

extern "C" int frmap_conv3x3s2_pp_layout(int B, int Hi, int Wi, int Cin, int Cout) {
  if (B <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0) return 0;
  return frmap_conv3x3s2_pp(nullptr, nullptr, nullptr, nullptr, nullptr, B, Hi, Wi, Cin, Cout, 0, FRMAP_BF16, nullptr);
}
