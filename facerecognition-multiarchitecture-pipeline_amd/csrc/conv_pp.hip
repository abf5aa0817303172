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
//   wave w: group = w >> 2 (pixel half: MFMA column groups [group*MI, group*MI + MI)), wn = w & 3 (channels
//   [wn*NI*16, +NI*16) of the tile): MI x NI MFMA 16x16x32 tiles per k-step, A = 16 output channels, B = 16 pixels.
// K loop: k-step = (32-channel chunk, tap).  LDS: two halo images (64 B per pixel, the XOR-swizzled layout of
//   conv3x3_fast_kernel; a tap is an LDS offset) and a ring of three weight slabs (BN x 32 channels, pre-packed in
//   LDS-image order by frmap_pack_conv_weight, so a slab is a straight copy).
// Pipeline per k-step and group:   LOAD(k): issue DMA {one piece of the NEXT chunk's halo, slab k+2}; ds_read the
//   fragments of k-step k; s_waitcnt vmcnt(n) so that only THIS phase's DMA is still in flight (slab k+1 has landed);
//   lgkmcnt(0); s_barrier; MFMA(k): MI*NI MFMAs; s_barrier.
//   Group B runs one barrier behind group A, so A's MFMA(k) coincides with B's LOAD(k) and B's MFMA(k) with A's LOAD(k+1).
//   Hazards: a slab is rewritten two phases after its last fragment read (ring of 3, prefetch distance 2), every read is
//   retired (lgkmcnt(0)) before the barrier that precedes any DMA into its buffer, and DMA'd bytes are read only after
//   the issuing waves' counted vmcnt AND a barrier every wave has passed.  DMA is issued from inline asm (hipcc would
//   otherwise drain vmcnt(0) ahead of every LDS read that may alias an outstanding LDS-DMA write), so every wait on it
//   is hand-counted; the number of DMA instructions per phase is static (slabs / halo pieces past the end of the
//   problem are fetched from valid dummy addresses into buffers nobody reads).
#include "frmap_common.h"
#include <stdlib.h>

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

template <typename TT, int MI, int NI, int NHP>
__global__ __launch_bounds__(512, 2) void conv3x3_pp_kernel(const PPParams p) {
  constexpr int TAPS = 9, BN = NI * 64, WB = BN * 64;      // slab bytes: BN channels x 32 channels x 2 B
  constexpr int NWI = BN / 128;                              // slab DMA instructions per wave and k-step
  constexpr int HB = NHP * 8 * 1024;                         // bytes of one halo image buffer
  using vec8 = typename TT::vec8;
  using elem = typename TT::elem;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS map: [halo 0][halo 1][slab 0][slab 1][slab 2]; the epilogue's transpose scratch reuses it from 0
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  const int lr = lane & 15, g = lane >> 4;

  const int L = pp_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = L / p.ntiles, nt = L - mt * p.ntiles;
  const int m0 = mt * p.tile_px, mend = min(m0 + p.tile_px, p.M);
  const int nk = p.nchunks * TAPS;

  // ---- halo geometry of this tile (rows of the virtual padded row stack, as conv3x3_fast_kernel)
  const int n0 = frmap_div(m0, p.dHoWo), oy0 = frmap_div(m0 - n0 * p.HoWo, p.dWo);
  const int n1 = frmap_div(mend - 1, p.dHoWo), oy1 = frmap_div(mend - 1 - n1 * p.HoWo, p.dWo);
  const int nrows = (n1 - n0) * p.Hp + oy1 - oy0 + 3;
  const int nitems = nrows * p.Wp * 4;  // 16-byte items of one 32-channel halo image

  // ---- per-lane DMA sources
  // halo piece j of this wave = piece (wave + 8j) of the image: lane -> pixel (piece*16 + lane/4), physical slot lane%4
  const char* hsrc[NHP];
#pragma unroll
  for (int j = 0; j < NHP; ++j) {
    const int item = ((wave + 8 * j) << 6) + lane;
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
  // slab pieces of this wave: BN = 256: instructions 2*wave, 2*wave + 1 of the 16 (block wave/2, 1-KB parts 2*(wave&1), +1);
  //                           BN = 128: instruction wave of the 8 (block wave/4, part wave%4)
  const int wblk = NWI == 2 ? (wave >> 1) : (wave >> 2);
  const int wpart = NWI == 2 ? ((wave & 1) << 1) : (wave & 3);
  const char* wsrc = (const char*)p.wpk + ((size_t)(nt * (BN / 64) + wblk) * p.nchunks * TAPS) * 4096 + wpart * 1024 + lane * 16;
  const unsigned wdst = lds0 + 2 * HB + wblk * 4096 + wpart * 1024;  // + slot * WB
  const unsigned hdst = lds0 + wave * 1024;                            // + buffer * HB + j * 8192

  auto issue_slab = [&](int k, int slot) {  // slab of k-step k into ring slot k % 3 (past the end: a dummy copy nobody reads)
    const int kc = k < nk ? k : nk - 1;
    const char* s = wsrc + (size_t)kc * 4096;
    const unsigned d = wdst + (unsigned)slot * WB;
    pp_dma16(s, d);
    if (NWI == 2) pp_dma16(s + 1024, d + 1024);
  };
  auto issue_halo = [&](int chunk, int j) {  // piece j of chunk `chunk` (past the last chunk: zeros into the idle buffer)
    const char* s = chunk < p.nchunks ? hsrc[j] + (size_t)chunk * 64 : (const char*)g_pp_zero;
    pp_dma16(s, hdst + (unsigned)(chunk & 1) * HB + j * 8192);
  };

  // ---- prologue: halo image of chunk 0, slabs 0 and 1
#pragma unroll
  for (int j = 0; j < NHP; ++j) issue_halo(0, j);
  issue_slab(0, 0);
  issue_slab(1, 1);

  // fragment addressing
  int A[MI];  // (pixel index in the halo image) * 64 + k-group * 16, before the tap offset and the swizzle
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = min(m0 + (grp * MI + mi) * 16 + lr, mend - 1);
    const int n = frmap_div(m, p.dHoWo), rem = m - n * p.HoWo, oy = frmap_div(rem, p.dWo), ox = rem - oy * p.Wi;
    A[mi] = ((((n - n0) * p.Hp + oy - oy0) * p.Wp + ox) << 6) | (g << 4);
  }
  const int woff = (lr << 6) + ((g ^ (((lr >> 2) & 1) << 1)) << 4);
  // this wave's 16*NI output channels inside the slab: BN = 256: block wn; BN = 128: block wn/2, second half for odd wn
  const int wfo = NI == 4 ? wn * 4096 : (wn >> 1) * 4096 + (wn & 1) * 2048;
  const char* halo = smem;
  const char* slabs = smem + 2 * HB + wfo + woff;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  pp_wait_vm<0>();
  pp_barrier();
  if (grp == 1) pp_barrier();  // group B runs one barrier behind group A from here on

  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const char* hb = halo + (chunk & 1) * HB;
    pp_static_for(std::make_integer_sequence<int, TAPS>{}, [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      const int k = chunk * TAPS + t;
      // ---------------- LOAD(k)
      if (t < NHP) issue_halo(chunk + 1, t);
      issue_slab(k + 2, (t + 2) % 3);   // (9 taps per chunk: k % 3 == t % 3)
      vec8 wf[NI], pf[MI];
      {
        const char* sl = slabs + (t % 3) * WB;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(sl + ni * 1024);
        const int toff = ((t / 3) * p.Wp + (t % 3)) << 6;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int at = A[mi] + toff;
          pf[mi] = *(const vec8*)(hb + (at ^ ((at >> 3) & 32)));
        }
      }
      pp_wait_vm<NWI + (t < NHP ? 1 : 0)>();  // everything older than this phase's DMA has landed (slab k+1; next halo by t = 8)
      pp_wait_lgkm0();                         // this phase's fragment reads are retired before the barrier
      pp_barrier();
      // ---------------- MFMA(k)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
      pp_barrier();
    });
  }
  if (grp == 0) pp_barrier();  // balance group B's extra start barrier
  pp_wait_vm<0>();             // the dummy DMA of the last two phases must not land in the epilogue's scratch
  pp_barrier();

  // ---- epilogue: + shift (+ residual) (activation) -> NHWC, whole-line 16-byte stores via a per-wave LDS transpose
  conv_epilogue<TT, MI, NI>(acc, smem + wave * (16 * (NI * 64 + 16)), m0 + grp * (MI * 16), mend, p.Cout,
                            nt * BN + wn * (NI * 16), p.shift, (const elem*)p.res, (elem*)p.out, p.relu, lane);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int pp_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// run-time overrides (frmap_conv_pp_tuning): -1 = not set (environment / heuristic decides)
static int g_pp_on = -1, g_pp_px = -1, g_pp_bn = -1;

extern "C" int frmap_conv_pp_tuning(int enable, int tile_px, int bn) {
  g_pp_on = enable;
  g_pp_px = tile_px;
  g_pp_bn = bn;
  return 0;
}

template <typename TT, int MI, int NI, int NHP>
static int pp_launch(const PPParams& p, hipStream_t st) {
  auto kern = conv3x3_pp_kernel<TT, MI, NI, NHP>;
  if (frmap_big_lds((const void*)kern, 160 * 1024)) return -2;
  int lds = 2 * NHP * 8192 + 3 * NI * 64 * 64;
  const int scratch = 8 * 16 * (NI * 64 + 16);
  if (lds < scratch) lds = scratch;
  hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles), dim3(512), lds, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// Returns 1 if the layer was launched on conv3x3_pp_kernel, 0 if the shape is not taken (caller falls through to the
// first-generation kernels), negative on a launch error.
int frmap_conv3x3_pp(const void* in, const void* w_packed, const float* shift, const void* residual, void* out, int B, int Hi,
                     int Wi, int Cin, int Cout, int relu, int dtype, hipStream_t st) {
  static int on = -1, force_px = 0, force_bn = 0, min_cin = 128;
  if (on < 0) {
    min_cin = pp_env("FRMAP_PP_MIN_CIN", 128);   // Cin = 64 layers keep the weights-resident wave kernel by default
    on = pp_env("FRMAP_CONV_PP", 1);
    force_px = pp_env("FRMAP_PP_TILE_PX", 0);
    force_bn = pp_env("FRMAP_PP_BN", 0);
  }
  constexpr int MI = 7, BM = 2 * MI * 16;
  if (g_pp_on >= 0 ? !g_pp_on : (!on || Cin < min_cin)) return 0;   // (forced on by the hook: every Cin % 32 == 0)
  if (Cin % 32 || Cin > 1024 || Cout % 128 || Wi > BM) return 0;
  const long long Mll = (long long)B * Hi * Wi;
  if (Mll >= (1ll << 31) || (long long)B * Hi * Wi * Cin * 2 >= (1ll << 46)) return 0;
  int rows = BM / Wi;
  int tile_px = rows * Wi;
  if (Hi * Wi <= BM && BM / (Hi * Wi) >= 1) tile_px = (BM / (Hi * Wi)) * Hi * Wi;  // whole images when they fit (7x7: 4, 14x14: 1)
  if (force_px > 0 && force_px <= BM) tile_px = force_px;
  if (g_pp_px > 0 && g_pp_px <= BM) tile_px = g_pp_px;
  PPParams p;
  p.in = in; p.wpk = w_packed; p.shift = shift; p.res = residual; p.out = out;
  p.N = B; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
  p.M = (int)Mll; p.HoWo = Hi * Wi; p.Hp = Hi + 2; p.Wp = Wi + 2;
  p.magic_Wp = frmap_magic((uint32_t)p.Wp); p.magic_Hp = frmap_magic((uint32_t)p.Hp);
  p.dHoWo = frmap_div_make((uint32_t)p.HoWo); p.dWo = frmap_div_make((uint32_t)Wi);
  p.nchunks = Cin / 32;
  p.tile_px = tile_px;
  p.mtiles = (p.M + tile_px - 1) / tile_px;
  // halo rows a tile can touch: its output rows + 2, plus the padded-row jump at every image boundary it crosses
  const int orows = (tile_px + Wi - 2) / Wi + 1;                       // output rows touched (unaligned start)
  const int cross = (tile_px + Hi * Wi - 2) / (Hi * Wi);               // image boundaries crossed
  const long long hrows = orows + 2 + 2ll * cross;
  const long long hbytes = hrows * p.Wp * 64;
  int nhp = (int)((hbytes + 8191) / 8192);
  if (nhp > 5 || hbytes / 64 >= 65536) return 0;
  // channel tile: 256 when that still gives every CU a tile, else 128
  int bn = (Cout % 256 == 0 && (long long)p.mtiles * (Cout / 256) >= 200) ? 256 : 128;
  if (force_bn == 128 || force_bn == 256) bn = (force_bn == 256 && Cout % 256) ? 128 : force_bn;
  if (g_pp_bn == 128 || g_pp_bn == 256) bn = (g_pp_bn == 256 && Cout % 256) ? 128 : g_pp_bn;
  p.ntiles = Cout / bn;
  int rc;
#define PP_GO(TT)                                                                                  \
  (bn == 256 ? (nhp <= 3 ? pp_launch<TT, MI, 4, 3>(p, st) : pp_launch<TT, MI, 4, 5>(p, st))        \
             : (nhp <= 3 ? pp_launch<TT, MI, 2, 3>(p, st) : pp_launch<TT, MI, 2, 5>(p, st)))
  rc = dtype == FRMAP_BF16 ? PP_GO(BF16) : PP_GO(F16);
#undef PP_GO
  return rc ? rc : 1;
}
