// K2/K3/K4/K7/K8: fused GroupNorm(+FiLM)+SiLU prologue -> implicit-GEMM 3x3 / 1x1 convolution on
// MFMA (v_mfma_f32_16x16x32_bf16) -> bias (+ residual) epilogue.  gfx950 only.
//
// Replaces, per call, the reference's GroupNorm32 -> SiLU -> Conv2d (-> skip + h) sequences in
// ResBlock._forward (guided_diffusion/unet.py:236-256), the 1x1 Conv1d projections of
// AttentionBlock._forward (unet.py:299-305), th.cat of the UNet skip (unet.py:662, as a virtual
// concat of two sources) and the fp32 output head (unet.py:612-616, 664-665).
//
// Data layout
//   activations  bf16 NHWC: pixel-major, a pixel's channels are one contiguous run
//   weights      bf16, packed once in MFMA fragment order [Cin/32][taps][ceil(Cout/16)][64][8]
//                (adm_pack_conv_weight) so a K-step's weight tile is one linear HBM/L2 stream
//   GEMM view    D[channel][pixel] = sum_k W[channel][k] * Act[k][pixel]; the weights are the MFMA
//                A operand, the activations the B operand, so each lane ends up with 4 consecutive
//                output channels of one pixel (8-byte bf16 stores into the NHWC row)
//
// Work decomposition
//   block tile  BM = 256 pixels (a TH x TW patch of one image; 128 pixels = two 8x8 images on the smallest
//               maps) x BN = 192 or 128 output channels, 8 waves (2 pixel rows x 4 channel columns of
//               waves, each 8 x {3,2} MFMA tiles of 16x16); the older 4-wave tilings serve odd widths
//   K loop      over 32-channel chunks of the input; per chunk the (TH+2)x(TW+2) halo patch is
//               fetched ONCE from HBM/L2 (register-staged, 16 B per lane), the per-(image,channel)
//               affine + SiLU is applied in fp32, the result is rounded to bf16 and parked in LDS
//               (96-byte pixel rows: conflict-free ds_read_b128 for the 16-pixel fragment), double
//               buffered, and then reused by all 9 taps; zero padding is applied after the activation
//   weights     fragment-ordered in HBM/L2, so each wave loads its own fragments straight into registers
//               (ring of 3 K-steps for 3x3, 4 for 1x1): no LDS, no per-tap barrier
//   epilogue    accumulators (started from the bias) -> bf16 -> LDS (whole tile) -> coalesced 16-byte row
//               stores with the residual added on the way and the output's GroupNorm partial sums
//               (sum, sum of squares per channel) reduced per tile, for the consumer's GroupNorm
//   launch      persistent: one block per CU slot walks an XCD-aware list of tiles and starts the next
//               tile's first chunk by LDS-DMA while the finished tile is stored (see conv_kernel)
#include <stdlib.h>

#include <type_traits>

#include "adm_common.h"
#include "adm_conv_internal.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#ifndef ADM_FOLD_ABL
#define ADM_FOLD_ABL 0
#endif
#ifndef ADM_CONV_1X1_NOEXIT
#define ADM_CONV_1X1_NOEXIT 1   // 1x1 loop of the 128-pixel tiles without early exits (0: the exits, for A/B builds)
#endif
#ifndef ADM_CONV_1X1_DEEP
#define ADM_CONV_1X1_DEEP 4   // 1x1 loop of the 128-pixel tiles: 4 (or 8) activation chunks in flight (8: weight K-steps too); 0: two (A/B builds). Same box, SD: 63.7 (0) / 64.2 (4) / 64.0 (8) latents/s
#endif
#ifndef ADM_CONV_FOLD_RING
#define ADM_CONV_FOLD_RING 2   // skip-connection fold: activation chunks in flight in registers (2; 3 measured 1-5 % slower per tile: profiles/r03/conv_tile_timing_fold.log)
#endif
#ifndef ADM_CONV_RD
#define ADM_CONV_RD 2    // 1x1: activation stages in flight (register ring)
#endif
#ifndef ADM_CONV_KS2
#define ADM_CONV_KS2 0   // 1: 1x1 convs use 64-channel stages where the channel counts allow
#endif
constexpr int KC = 32;           // input channels per chunk (= one MFMA K)
// LDS bytes per halo pixel: 64 data bytes per K-step of the stage + 32 pad (conflict-free ds_read_b128 fragments)
constexpr int conv_rowb(int ks) { return 64 * ks + 32; }
constexpr int ROWB = conv_rowb(1);
constexpr int TI_MAX = 4;
constexpr unsigned OOB = 0x80000000u;  // buffer-load offset beyond num_records -> returns 0

struct ConvK {
  const uint16_t* in0; const uint16_t* in1; const uint16_t* w; const uint16_t* res;
  const float* bias; const float* aa; const float* ab; void* out;
  float* stats;      // optional [N][stat_slabs][Cout][2] per-(image, slab, channel) sum / sum of squares of the OUTPUT
  int stat_slabs;
  int N, H, W, C0, C1, Cout;
  int TH, TW, TI, tiles_x, tiles_y;
  int tw_shift, thw_shift;   // log2(TW), log2(TH*TW)
  unsigned rcp_hpi, rcp_hw2; // ceil(2^20 / halo pixels per image), ceil(2^20 / halo row width): x / d == (x * rcp) >> 20 for x < 512
  int out_mode;
  int in_up, res_up;         // the input (in0, no in1) / the residual is read through a virtual nearest-neighbour 2x upsample
  int ntiles16, nblocks_n;
  int total_tiles;           // pixel tiles x Cout blocks (x K splits) (persistent launch: blocks walk this list)
  unsigned wbytes;
  // split-K (small batches: a launch with few tiles but a long K loop; 3x3, and 1x1 of the wide low-resolution levels): the tile list is multiplied by `ksplit`,
  // split s runs chunks [s * cps, (s + 1) * cps) (cps even: the halo double buffer keeps its parity) from zero
  // accumulators and stores them as fp32 [ksplit][N*H*W][Cout] into `ws`; conv_splitk_reduce adds the splits in a fixed
  // order together with bias and residual, rounds to bf16 and accumulates the output statistics
  int ksplit, cps;
  float* ws;
  // up-conv phase launch (TAPS = 4 instantiations; adm_conv_args.up_phase): the conv3x3 of a nearest-neighbour 2x upsample,
  // for the output pixels of ONE phase (2y + py, 2x + px), is a 2x2-tap conv of the half-resolution source with pre-summed
  // weights: the host passes those weights embedded in a 3x3 window (5 zero taps), `tap_mask` names the 4 live taps --
  // the others skip their fragment reads and MFMAs --, the output is written with pixel stride 2 at offset (ooy, oox)
  // into the [N][2H][2W][Cout] tensor, and the fused statistics go to slab (tile * 4 + slab_off) of 4 x as many slabs
  int tap_mask, ooy, oox, slab_off;
  // up_phase == 5: all four phases in ONE launch -- the Cout-block index of a tile runs over nph * nbp values, phase = nb / nbp,
  // and phase p's packed weights start phase_bytes * p into `w` (the tile list is 4 x as long: small batches still fill the CUs)
  int nph, nbp;
  unsigned phase_bytes;
  // skip-connection fold (PROX = 4; adm_conv_args.fold0): after the nine-tap K loop over the GroupNorm + SiLU input, (FC0 + FC1) / 32
  // one-tap, raw-prologue K-steps over the ResBlock's INPUT (fold0 | fold1, same map) with the skip_connection's 1x1 weights, which
  // follow the 3x3 weights in `w` (K-step index chunks * 9 + j): out = conv3x3(act(GN(h))) + conv1x1(x) + (bias3 + bias1)
  const uint16_t* f0; const uint16_t* f1;
  int FC0, FC1;
  float out_scale;   // fp32 NCHW epilogue only (adm_conv_args.out_scale; 1 if not given)
};

// output-statistics slabs per tile: a 128-pixel tile is two 8x8 images (or two halves of one image)
template <int BM>
constexpr int stat_groups() { return BM == 128 ? 2 : 1; }

// every lane of every staging pass owns an LDS slot: the halo capacity is rounded up to whole passes
template <int NT, int HALO>
constexpr int halo_slots() { return ((HALO * 4 + NT - 1) / NT) * NT / 4; }
template <int NT, int HALO>
constexpr int conv_smem_bytes_k() { return 2 * halo_slots<NT, HALO>() * ROWB + 2 * TI_MAX * 64 * 4; }
// the epilogue restages the whole output tile (BM pixels x BN channels, bf16) in the same LDS
template <int NT, int BN, int HALO, int BM>
constexpr int conv_smem_bytes() {
  constexpr int k = conv_smem_bytes_k<NT, HALO>();
  constexpr int e = BM * (BN * 2 + 16);    // output staging
  constexpr int r = NT * 64;               // statistics reduction: [NT*8/BN rows][BN][2] floats
  return (k > e ? k : e) > r ? (k > e ? k : e) : r;
}

// LDS map of conv_kernel: halo[0] | halo[1] ... | affine tables.  The epilogue's output staging (whole tile,
// bf16 rows + 16 B pad) and the statistics reduction overlay halo[1] onwards, never halo[0] or the affine
// tables: those receive the NEXT tile's first chunk while the current tile is being stored.
template <int NT, int BN, int HALO, int BM, int KS>
struct ConvLds {
  static constexpr int HB = ((HALO * 4 * KS + NT - 1) / NT) * (NT / (4 * KS)) * conv_rowb(KS);  // one halo buffer (whole passes)
  static constexpr int STG = BM * (BN * 2 + 16);             // output staging
  static constexpr int RED = NT * 64;                        // statistics reduction [NT*8/BN rows][BN][2] floats
  static constexpr int OVL = (STG > RED ? STG : RED) > HB ? (STG > RED ? STG : RED) : HB;
  static constexpr int ABUF = (HB + OVL + 15) & ~15;
  static constexpr int BYTES = ABUF + 2 * 2 * TI_MAX * 32 * KS * 4;   // a | b, two stages each
};

#ifdef ADM_CONV_TIMING
// debug build only (make timing): per-block phase time stamps of conv_kernel, read by tools/conv_timing.py
__device__ unsigned long long adm_conv_timing_buf[16 * 16384];
#define ADM_TSTAMP(tile, k)                                                                             \
  do {                                                                                                  \
    if (threadIdx.x == 0 && (tile) < 16384) {                                                           \
      adm_conv_timing_buf[(tile) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();                        \
      if ((k) == 1 || (k) == 2) adm_conv_timing_buf[(tile) * 16 + 7 + (k)] = __builtin_amdgcn_s_memtime(); \
      if ((k) == 0) {                                                                                   \
        adm_conv_timing_buf[(tile) * 16 + 14] = __builtin_amdgcn_s_getreg((31 << 11) | 4);              \
        adm_conv_timing_buf[(tile) * 16 + 15] = __builtin_amdgcn_s_getreg((31 << 11) | 20);             \
      }                                                                                                 \
    }                                                                                                   \
  } while (0)
#else
#define ADM_TSTAMP(tile, k) do {} while (0)
#endif

__device__ __forceinline__ uint4 bufload16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// WM x WN waves, each TM x TN tiles of 16x16; TAPS 9 (3x3 pad 1) or 1; HALO = halo pixels of the tile;
// PRO = prologue (0 raw, 1 affine, 2 affine + SiLU).  Everything the inner loop branches on is a
// template parameter: a K-step is one straight-line block (1 halo load, 1 halo transform + LDS write,
// TN weight loads, TM LDS reads, TM*TN MFMAs) that the scheduler can software-pipeline.
//
// The launch is PERSISTENT: one block per CU slot walks a static, XCD-aware list of output tiles.  A block
// of the 8-wave tilings is alone on its CU (> 128 VGPRs per wave), so nothing else would hide a tile's
// prologue (descriptor / index setup, first halo chunk, affine tables, first weight fragments: 3-4 us of
// dependent latencies) or the block launch gap; here the NEXT tile's first loads are issued before the
// current tile's accumulators are staged and stored, and land in LDS (halo buffer 0, which the output
// staging does not overlay) while the epilogue runs.
// COLD: the build also carries the two rare epilogues (split-K partial sums; fp32 NCHW output).  They are separate instantiations
// because their tile-invariant address math is hoisted out of the tile loop and then sits in registers ACROSS the K loop of every
// launch, used or not: without them the dominant instantiation goes from 28 spilled registers to 2 and the 128-wide tile from 224
// to 192 registers (hipcc -Rpass-analysis=kernel-resource-usage).
template <int WM, int WN, int TM, int TN, int OCC, int TAPS, int HALO, int PROX, int KS, bool COLD>
__global__ void __launch_bounds__(64 * WM * WN, OCC)
conv_kernel(const ConvK p) {
  // PROX 0..2 = the prologue; 3 = a raw input (backward-data conv) whose EPILOGUE is the first half of the GroupNorm(+SiLU)
  // backward of the layer in front (adm_conv_args.prologue == 3): with x = that layer's input (`res`) and its affine (a, b),
  // out = dz = acc * SiLU'(a x + b), statistics = (sum dz, sum dz * x) per (image, slab, channel) -- the partial sums the
  // separate adm_gn_bwd_partial pass would re-read x and dy for
  constexpr int PRO = PROX == 3 ? 0 : (PROX == 4 ? 2 : PROX);
  constexpr bool GNB = PROX == 3;
  constexpr bool FOLD = PROX == 4;   // GN + SiLU prologue, then the ResBlock's skip_connection as extra one-tap K-steps (ConvK::f0)
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = WM * TM * 16;
  constexpr int BN = WN * TN * 16;
  // a STAGE is what one halo buffer holds: KS 32-channel K-steps of every halo pixel (3x3: KS = 1, the 9 taps
  // give a stage its depth; 1x1: KS = 2 where the channel counts allow, or the loop is all barriers)
  constexpr int KCS = KC * KS;             // channels per stage
  constexpr int SEGP = 4 * KS;             // 16-byte segments per halo pixel
  constexpr int SEGSH = KS == 1 ? 2 : 3;   // log2(SEGP)
  constexpr int ROWB = conv_rowb(KS);      // LDS bytes per halo pixel
  constexpr int PASSES = (HALO * SEGP + NT - 1) / NT;
  // OCC == 1: the ONE-WAVE-PER-SIMD build of the 256-pixel tile (4 waves of 128 pixels x TN * 16 channels, 512 registers each:
  // the accumulators live in the AGPR half).  No partner wave hides anything, so the K loop is ONE software-pipelined stream per
  // chunk (chunk1 below): every global load is consumed >= 2-3 taps after its issue, LDS fragment reads run two pixel rows ahead
  // of their MFMAs across tap boundaries, the GroupNorm affine of a lane's 8 channels rides in registers, and the prologue
  // VALU / load / LDS-write instructions are dealt two per MFMA gap (sched_group_barrier) instead of being fenced off
  constexpr bool ONEW = OCC == 1;
  static_assert(!ONEW || (TAPS == 9 && HALO == 324 && WM == 2 && WN == 2 && KS == 1), "the one-wave-per-SIMD build is the 3x3 256-pixel tile");
  constexpr bool T3 = TAPS == 9 || TAPS == 4;   // 3x3 geometry; TAPS == 4: an up-conv phase (4 of the 9 taps live)
  constexpr bool UPPH = TAPS == 4;
  constexpr int PAD = T3 ? 1 : 0;
  constexpr int SGROUPS = stat_groups<BM>();  // statistics slabs per tile
  using Lds = ConvLds<NT, BN, HALO, BM, KS>;
  static_assert(KS == 1 || (KS == 2 && TAPS == 1), "multi-step stages are a 1x1 feature");
  static_assert(TAPS == 1 || PASSES <= (T3 ? 9 : TAPS) - 1, "halo passes must fit in the taps of one chunk");
  static_assert(PASSES <= 8, "ti_pack holds 8 passes");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const halo = smem;               // [2][Lds::HB]
  unsigned char* const stg = smem + Lds::HB;      // output staging / statistics reduction: overlays halo[1]
  float* const abuf = reinterpret_cast<float*>(smem + Lds::ABUF);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
  const int wm = wave / WN, wn = wave % WN;
  const int lc = lane & 15, lq = lane >> 4;

  // XCD-aware static tile list: the dispatcher deals consecutive blocks round-robin to the 8 XCDs (each with
  // its own L2); XCD x owns the contiguous run [tstart, tstart + tcount) of logical tiles and its gx blocks
  // take them interleaved, so the tiles in flight on one L2 are neighbours: all Cout blocks of a pixel tile
  // and the adjacent pixel tiles (shared halo rows)
  const int xcd = blockIdx.x & 7;
  const int gx = (int)(gridDim.x >> 3) + (xcd < (int)(gridDim.x & 7));
  const int tcount = (p.total_tiles >> 3) + (xcd < (p.total_tiles & 7));
  const int tstart = xcd * (p.total_tiles >> 3) + min(xcd, p.total_tiles & 7);
  int tl = blockIdx.x >> 3;
  if (tl >= tcount) return;

  const int HW2 = p.TW + 2 * PAD;
  const int HPI = (p.TH + 2 * PAD) * HW2;
  const int HP = p.TI * HPI;
  const int Cin = p.C0 + p.C1;
  const int HWimg = p.H * p.W;
  const int chunks = Cin / KCS;            // stages of the K loop
  const int c0chunks = p.C0 / KCS;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.Cout * 4, 0x00020000);

  // ---- halo staging geometry: 16-byte segment s = tid + pass*NT -> (halo pixel s / SEGP, segment s % SEGP)
  unsigned ti_pack = 0;    // image-in-tile of each pass, 4 bits each
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int hp = (tid + ps * NT) >> SEGSH;
    if (hp < HP) ti_pack |= (unsigned)(hp / HPI) << (4 * ps);
  }
  const int seg = tid & (SEGP - 1);  // NT % SEGP == 0 -> the same channel segment in every pass
  const int hslot = (tid >> SEGSH) * ROWB + seg * 16;  // LDS byte offset of pass 0's slot; pass ps adds ps*(NT/SEGP)*ROWB

  // ---- MFMA fragment addressing
  // LDS byte offset of a lane's pixel row = lane part (its pixel inside the 16-pixel tile, its
  // 16-byte k slice) + a wave-uniform part per tile (TW divides 16, so a tile starts at column 0)
  const int alane = ((lc / p.TW) * HW2 + lc % p.TW) * ROWB + lq * 16;
  int aoff[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m0 = (wm * TM + i) * 16;
    const int ti = m0 / (p.TH * p.TW), rem = m0 % (p.TH * p.TW);
    aoff[i] = (ti * HPI + (rem / p.TW) * HW2) * ROWB;
  }
  const unsigned wstep = (unsigned)p.ntiles16 * 1024u;  // bytes per K-step

  // ---- per-tile state (rewritten at every tile switch)
  int nb = 0, mt = 0, img0 = 0, y0 = 0, x0 = 0;
  [[maybe_unused]] int ph = p.slab_off;   // up-conv phase of the current tile (fixed per launch unless p.nph == 4)
  int sp = 0, cb = 0, ce = chunks;   // K split of the tile: chunks [cb, ce)
  __amdgpu_buffer_rsrc_t rs0 = rsw, rs1 = rsw;
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rs2 = rsw, rs3 = rsw;   // FOLD: the skip_connection's sources
  int pixrel[PASSES];      // pixel index relative to img0, or -1 (zero padding / beyond the batch / spare slot)
  // weight fragments go straight from the packed image (L2-resident, fragment-ordered: one 1 KB
  // coalesced run per wave-load) into registers -- no LDS, no per-tap barrier
  unsigned wofs[TN];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  // scalar part: tile -> Cout block, pixel-tile index, first image, origin
  auto tile_origin = [&](int lt, int& nb_, int& mt_, int& img0_, int& y0_, int& x0_) {
    if (p.ksplit > 1) {
      sp = lt % p.ksplit;
      lt /= p.ksplit;
      cb = sp * p.cps;
      ce = cb + p.cps;
    }
    nb_ = lt % p.nblocks_n;
    mt_ = lt / p.nblocks_n;
    if constexpr (UPPH) {
      if (p.nph > 1) { ph = nb_ / p.nbp; nb_ -= ph * p.nbp; }
    }
    if (p.TI == 1) {
      const int per_img = p.tiles_x * p.tiles_y;
      img0_ = mt_ / per_img;
      const int r = mt_ % per_img;
      y0_ = (r / p.tiles_x) * p.TH;
      x0_ = (r % p.tiles_x) * p.TW;
    } else {
      img0_ = mt_ * p.TI;
      y0_ = 0;
      x0_ = 0;
    }
  };
  // per-lane part, from (nb, img0, y0, x0)
  auto tile_setup = [&]() {
    const int nimg = min(p.TI, p.N - img0);  // images of this tile that exist
    // buffer descriptors over exactly the images this tile may touch: anything else reads as zero
    // (with in_up the source map is (H/2) x (W/2): output-grid halo pixel (y, x) reads source pixel (y/2, x/2))
    const int Hs = p.H >> p.in_up, Ws = p.W >> p.in_up;
    rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in0 + (long long)img0 * Hs * Ws * p.C0), 0, nimg * Hs * Ws * p.C0 * 2, 0x00020000);
    rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C1 ? p.in1 + (long long)img0 * HWimg * p.C1 : p.in0), 0,
                                            p.C1 ? nimg * HWimg * p.C1 * 2 : 0, 0x00020000);
    if constexpr (FOLD) {
      rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.f0 + (long long)img0 * HWimg * p.FC0), 0, nimg * HWimg * p.FC0 * 2, 0x00020000);
      rs3 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.FC1 ? p.f1 + (long long)img0 * HWimg * p.FC1 : p.f0), 0,
                                              p.FC1 ? nimg * HWimg * p.FC1 * 2 : 0, 0x00020000);
    }
    // the halo pixel of each pass is recomputed per tile from an opaque copy of the thread index: nothing
    // per-lane has to survive the K loop (it runs at the VGPR cap; a reload from scratch costs a dependent
    // memory round trip here).  Divisions by the patch size / width are multiplications by host-made
    // reciprocals (exact for hp < 512), and the tests are evaluated without branches.
    int tid_s = tid;
    asm volatile("" : "+v"(tid_s));
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int hp = (tid_s + ps * NT) >> SEGSH;
      const int ti = (int)(((unsigned)hp * p.rcp_hpi) >> 20), rem = hp - ti * HPI;
      const int ry = (int)(((unsigned)rem * p.rcp_hw2) >> 20), rx = rem - ry * HW2;
      const int y = y0 + ry - PAD, x = x0 + rx - PAD;
      const int ok = (int)(hp < HP) & (int)(ti < nimg) & (int)(y >= 0) & (int)(y < p.H) & (int)(x >= 0) & (int)(x < p.W);
      pixrel[ps] = ok ? (ti * Hs + (y >> p.in_up)) * Ws + (x >> p.in_up) : -1;
    }
    // LDS-DMA of the tile's first chunk (see first_park): raw halo chunk 0, lane-linear into halo[0] ...
    const int seg_s = tid_s & (SEGP - 1), lane_s = tid_s & 63;
    {
      const bool first = cb < c0chunks;   // the tile's first chunk (chunk cb: 0 unless the K loop is split)
      const int cs = first ? p.C0 : p.C1, co = (first ? cb : cb - c0chunks) * KCS;
      const __amdgpu_buffer_rsrc_t rsf = first ? rs0 : rs1;
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const unsigned voff = pixrel[ps] >= 0 ? (unsigned)(pixrel[ps] * cs + co + seg_s * 8) * 2u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (lds_ptr_t)(halo + (ps * NT + wave * 64) * 16), 16, (int)voff, 0, 0, 0);
      }
    }
    // ... and the affine tables of stages 0 and 1 (they are staged two stages ahead): the LDS table is
    // a[buf][ti][KCS] followed by b[buf][ti][KCS]; each half is 2*KCS lanes x 16 B, lane-linear, so KS
    // wave-instructions move it: waves 0..KS-1 the a half, waves KS..2KS-1 the b half (32-bit offsets into
    // a descriptor over the whole [N][Cin] table)
    if constexpr (PRO != 0 && !ONEW) {
      if (wave < 2 * KS) {
        const int half = wave / KS, idx = (wave % KS) * 64 + lane_s;  // idx = (buf * TI_MAX + ti) * (KCS/4) + part
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(half == 0 ? p.aa : p.ab), 0, p.N * Cin * 4, 0x00020000);
        const int buf = idx / (TI_MAX * (KCS / 4)), ti = (idx / (KCS / 4)) % TI_MAX, part = idx % (KCS / 4);
        const int n = min(img0 + ti, p.N - 1);
        const unsigned voff = ti < p.TI ? (unsigned)(n * Cin + min(cb + buf, ce - 1) * KCS + part * 4) * 4u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr_t)(abuf + half * (2 * TI_MAX * KCS) + (wave % KS) * 256), 16, (int)voff, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int tile = nb * (BN / 16) + wn * TN + j;
      wofs[j] = tile < p.ntiles16 ? (unsigned)((tile * 64 + lane) * 16) + (UPPH ? (unsigned)ph * p.phase_bytes : 0u) : OOB;
    }
  };

  f32x4 acc[TM][TN];

  // activation segment `ps` of chunk c (the descriptor select is scalar; out-of-image lanes read zero)
  auto halo_load = [&](int c, int ps) -> uint4 {
    const bool first = c < c0chunks;
    const int cs = first ? p.C0 : p.C1, co = (first ? c : c - c0chunks) * KCS;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const unsigned voff = pixrel[ps] >= 0 ? (unsigned)(pixrel[ps] * cs + co + seg * 8) * 2u : OOB;
    return bufload16(rs, voff, 0);
  };
  // the NOEXIT 1x1 loops run K-steps past the tile's last chunk (on weight fragments the descriptor returns as zeros): their
  // ACTIVATION side reads zeros too -- 0 x Inf of a re-read, overflowed last chunk would be NaN where the exact loop gives +-Inf
  [[maybe_unused]] auto halo_load_pad = [&](int c, int lastc, int ps) -> uint4 {
    const int cc = min(c, lastc);
    const bool first = cc < c0chunks;
    const int cs = first ? p.C0 : p.C1, co = (first ? cc : cc - c0chunks) * KCS;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const unsigned voff = (pixrel[ps] >= 0 && c <= lastc) ? (unsigned)(pixrel[ps] * cs + co + seg * 8) * 2u : OOB;
    return bufload16(rs, voff, 0);
  };
  // FOLD: segment `ps` of chunk k of the skip_connection's input (same halo geometry: only its centre tap is used), and its raw park
  [[maybe_unused]] auto fold_load = [&](int k, int ps) -> uint4 {
    const int f0chunks = p.FC0 / KCS;
    const bool first = k < f0chunks;
    const int cs = first ? p.FC0 : p.FC1, co = (first ? k : k - f0chunks) * KCS;
    const __amdgpu_buffer_rsrc_t rs = first ? rs2 : rs3;
    const unsigned voff = pixrel[ps] >= 0 ? (unsigned)(pixrel[ps] * cs + co + seg * 8) * 2u : OOB;
    return bufload16(rs, voff, 0);
  };
  [[maybe_unused]] auto raw_write = [&](uint4 v, int ps, int buf) {
    *reinterpret_cast<uint4*>(halo + buf * Lds::HB + ps * (NT / SEGP) * ROWB + hslot) = v;
  };
  // affine table of stage c: threads < TI * KCS/2 fetch one float4 of a (parts < KCS/4) or b
  constexpr int APT = KCS / 2;             // threads per image of the tile
  auto affine_load = [&](int c, int im0) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (PRO != 0) {
      if (tid < p.TI * APT) {
        const int n = min(im0 + tid / APT, p.N - 1), part = tid % APT;
        v = *reinterpret_cast<const float4*>((part < APT / 2 ? p.aa : p.ab) + (long long)n * Cin + c * KCS + (part % (APT / 2)) * 4);
      }
    }
    return v;
  };
  auto affine_park = [&](float4 v, int buf) {
    if constexpr (PRO != 0) {
      if (tid < p.TI * APT) {
        const int part = tid % APT;
        *reinterpret_cast<float4*>(abuf + (part < APT / 2 ? 0 : 2 * TI_MAX * KCS) + (buf * TI_MAX + tid / APT) * KCS + (part % (APT / 2)) * 4) = v;
      }
    }
  };
  // ONEW: the affine of this lane's 8 channels (its 16-byte segment of every halo pixel) of ONE chunk, in registers: a | b,
  // fetched straight from the [N][Cin] tables (every lane with the same segment reads the same 64 bytes: L1 / L2 hits)
  [[maybe_unused]] float4 afr[4];
  auto affine_regs = [&](int c, int im0) {
    if constexpr (PRO != 0) {
      const long long o = (long long)min(im0, p.N - 1) * Cin + c * KCS + seg * 8;
      afr[0] = *reinterpret_cast<const float4*>(p.aa + o);
      afr[1] = *reinterpret_cast<const float4*>(p.aa + o + 4);
      afr[2] = *reinterpret_cast<const float4*>(p.ab + o);
      afr[3] = *reinterpret_cast<const float4*>(p.ab + o + 4);
    }
  };
  // transform (fp32 affine [+ SiLU], rounded to bf16) and park a segment; zero padding stays exactly zero
  auto halo_write = [&](uint4 v, int ps, int buf) {
    if constexpr (PRO != 0) {
      float a8[8], b8[8];
      if constexpr (ONEW) {
        *reinterpret_cast<float4*>(a8) = afr[0];
        *reinterpret_cast<float4*>(a8 + 4) = afr[1];
        *reinterpret_cast<float4*>(b8) = afr[2];
        *reinterpret_cast<float4*>(b8 + 4) = afr[3];
      } else {
      const int ti = (ti_pack >> (4 * ps)) & 15;
      const float* ab = abuf + (buf * TI_MAX + ti) * KCS + seg * 8;
      *reinterpret_cast<float4*>(a8) = *reinterpret_cast<const float4*>(ab);
      *reinterpret_cast<float4*>(a8 + 4) = *reinterpret_cast<const float4*>(ab + 4);
      *reinterpret_cast<float4*>(b8) = *reinterpret_cast<const float4*>(ab + 2 * TI_MAX * KCS);
      *reinterpret_cast<float4*>(b8 + 4) = *reinterpret_cast<const float4*>(ab + 2 * TI_MAX * KCS + 4);
      }
      uint32_t u[4] = {v.x, v.y, v.z, v.w};
      const bool valid = pixrel[ps] >= 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        adm_f32x2_t t = {adm_lo_f32(u[j]), adm_hi_f32(u[j])};
        t = __builtin_elementwise_fma(adm_f32x2_t{a8[2 * j], a8[2 * j + 1]}, t, adm_f32x2_t{b8[2 * j], b8[2 * j + 1]});
        if constexpr (PRO == 2) t = adm_silu2(t);
        const uint32_t pk = adm_pack2(t.x, t.y);
        u[j] = valid ? pk : 0u;
      }
      v = make_uint4(u[0], u[1], u[2], u[3]);
    }
    *reinterpret_cast<uint4*>(halo + buf * Lds::HB + ps * (NT / SEGP) * ROWB + hslot) = v;
  };
  auto load_w = [&](int step, uint4 (&dst)[TN]) {
#pragma unroll
    for (int j = 0; j < TN; ++j) dst[j] = bufload16(rsw, wofs[j], (unsigned)step * wstep);
  };
  // LDS fragment reads are issued ahead of the MFMAs that consume them, pinned in the emitted code (the compiler
  // otherwise issues one ds_read_b128, waits lgkmcnt(0), then its TN MFMAs: the read latency was exposed TM times per tap)
  auto mfma_tap = [&](const unsigned char* hp, const uint4 (&w)[TN]) {
    constexpr int AFD = TM;  // measured on MI355X: read-ahead 2 / 4 / 6 tiles equal, all TM reads up front +1 %
    const unsigned char* hl = hp + alane;
    // scheduling fence: the staging work issued above (loads, prologue transform, LDS writes and their
    // own LDS reads) stays out of the pinned read/MFMA pipeline below
    __builtin_amdgcn_sched_barrier(0);
    adm_h8 af[TM];
#pragma unroll
    for (int i = 0; i < AFD; ++i) af[i] = *reinterpret_cast<const adm_h8*>(hl + aoff[i]);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i + AFD < TM) af[i + AFD] = *reinterpret_cast<const adm_h8*>(hl + aoff[i + AFD]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = adm_mfma_16x16x32(__builtin_bit_cast(adm_h8, w[j]), af[i], acc[i][j], 0, 0, 0);
    }
    // pin the interleave in the emitted code: AFD reads up front, then {1 read, TN MFMAs} per pixel tile
    // (LLVM otherwise sinks every read next to its use to save registers)
#pragma unroll
    for (int i = 0; i < AFD; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i + AFD < TM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);
    }
  };

  // ---- a tile's first loads.  The K loop already runs at the 256-VGPR cap, so nothing wide can be held
  // in registers across it or across the staging of the accumulators: the next tile's first halo chunk (raw
  // bf16) and its first two affine tables go global -> LDS by LDS-DMA (buffer_load ... lds / global_load_lds:
  // no VGPR destination) as soon as the K loop has released halo[0] and the affine tables, and land while the
  // accumulators are staged.  The raw chunk is lane-linear in halo[0] (segment s at byte 16 s); first_park
  // reads it back, and after a barrier writes the transformed segments to their padded pixel rows.
  // The narrow register loads (first weight fragments: 3x3 ring of 3 K-steps, two in flight; 1x1 ring of
  // 4, three in flight, plus the activation segments of chunk 1; the bias fragment the accumulators start
  // from) go out once the accumulators' registers are free.
  // 1x1 loop of the 128-pixel tiles (never split-K: chunk 0 first): DEEPN activation chunks and weight K-steps in flight instead of
  // two / three -- a step of these tiles is 8-12 MFMAs per wave, so two steps of prefetch distance are far less than the load latency
  constexpr int DEEPN = ADM_CONV_1X1_DEEP;    // 0: off, else 4 or 8
  constexpr bool DEEP1 = DEEPN != 0 && TAPS == 1 && KS == 1 && TM == 4 && !COLD && ADM_CONV_RD == 2;
  constexpr int WRING = (TAPS == 9 || TAPS == 4) ? 3 : (KS == 1 ? (DEEP1 && DEEPN == 8 ? 8 : 4) : 2 * KS);
  uint4 wr[WRING][TN];
  constexpr int RINGN = FOLD ? ADM_CONV_FOLD_RING : (DEEP1 ? DEEPN : ADM_CONV_RD);
  uint4 ring[RINGN][PASSES];
  float4 bs[TN];
  // 1x1 loops: `last` = the tile's last chunk, ce - 1 (chunks - 1 unless the K loop is split: the split 1x1 tile runs chunks
  // [cb, ce) with cb even, so buffer parities (c & 1) and the first chunk in halo[0] agree)
  auto first_loads = [&](int lq_) {
    const int last = ce - 1;
    if constexpr (T3) {
      load_w(cb * 9, wr[0]);
      load_w(cb * 9 + 1, wr[1]);
    } else {
      if constexpr (KS == 1) {
        constexpr bool NOEXIT1 = ADM_CONV_1X1_NOEXIT && TM == 4 && !COLD;   // see the 1x1 loop: K-steps past the last read zero weights
        load_w(cb, wr[0]);
        load_w(NOEXIT1 ? cb + 1 : min(cb + 1, last), wr[1]);
        load_w(NOEXIT1 ? cb + 2 : min(cb + 2, last), wr[2]);
        if constexpr (WRING == 8) {
#pragma unroll
          for (int q = 3; q < 7; ++q) load_w(NOEXIT1 ? cb + q : min(cb + q, last), wr[q]);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) load_w(ks, wr[ks]);  // stage 0's K-steps
      }
      constexpr bool NOEXIT1P = KS == 1 && ADM_CONV_1X1_NOEXIT && TM == 4 && !COLD;
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) ring[1][ps] = NOEXIT1P ? halo_load_pad(cb + 1, last, ps) : halo_load(min(cb + 1, last), ps);
      if constexpr (DEEP1) {
#pragma unroll
        for (int q = 2; q < RINGN; ++q) {
#pragma unroll
          for (int ps = 0; ps < PASSES; ++ps) ring[q][ps] = NOEXIT1P ? halo_load_pad(cb + q, last, ps) : halo_load(min(cb + q, last), ps);
        }
      }
#if ADM_CONV_RD == 3
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) ring[2][ps] = halo_load(min(2, last), ps);
#endif
    }
    if constexpr (ONEW) affine_regs(cb, img0);   // first_park transforms the tile's first chunk with these
    // bias fragment (whole float4 or nothing: a ragged last fragment only exists with the fp32 NCHW output,
    // which adds those channels' bias at the store)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int bch = nb * BN + (wn * TN + j) * 16 + lq_ * 4;
      const uint4 b = bufload16(rsb, (bch + 3 < p.Cout && p.ksplit <= 1) ? (unsigned)bch * 4u : OOB, 0);  // split-K: the reduce adds the bias
      bs[j] = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
    }
  };
  auto first_park = [&]() {  // after a barrier behind the DMA: raw chunk and affine tables visible
    uint4 raw[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) raw[ps] = *reinterpret_cast<const uint4*>(halo + (ps * NT + tid) * 16);
    __syncthreads();  // every raw segment is in registers: the padded rows may overwrite the linear image
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) halo_write(raw[ps], ps, 0);
  };

  int ltile = tstart + tl;
  tile_origin(ltile, nb, mt, img0, y0, x0);
  tile_setup();
  first_loads(lq);
  __syncthreads();
  first_park();

  for (;;) {
    ADM_TSTAMP(ltile, 0);
    // the accumulators start from the bias (fp32), so the epilogue has no bias operand to wait for
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{bs[j].x, bs[j].y, bs[j].z, bs[j].w};
    __syncthreads();  // halo[0] complete; the previous tile's staging / reduction reads are done
    ADM_TSTAMP(ltile, 1);

    const int tnext = tl + gx;
    const bool more = tnext < tcount;
    if constexpr (T3 && ONEW) {
      // One chunk = 9 taps x 8 pixel rows x TN MFMAs as ONE stream.  Per tap t: the weight fragments of tap t + 2 (ring of 3), then
      // (taps 0-2) two of the six raw halo passes of chunk c + 1 -- issued behind the weights, so that the in-order vmcnt wait for
      // the next tap's weights never drains them --, at tap 0 the next chunk's affine registers; the pass fetched at tap q / 2 is
      // transformed and parked at tap q + 3 (>= 3 taps = ~2300 cycles after its issue).  Fragment row R = 8 t + i is read from LDS
      // two rows (12 MFMAs) ahead of its MFMAs, across tap boundaries; only rows 0 and 1 of a chunk wait for their reads (the
      // other halo buffer becomes readable at the chunk's barrier).
      constexpr int HW2C = 18;                                   // halo row of the 16 x 16 patch (host: TW == TH == 16)
      const unsigned char* const hl0 = halo + alane + aoff[0];   // this lane's fragment row 0 in halo buffer 0
      // The stream is HAND-PLACED: one scheduling region per MFMA (sched_barrier after every slot), each holding that MFMA and
      // at most a few staging instructions chosen here -- hipcc's own placement of the same work bunched the ~65 VALU
      // instructions of a pass transform into two pixel rows and sank every fragment read next to its use.
      //   slot (i, 0)       the LDS read of fragment row R + 2 precedes it
      //   slot (i, 1)       i < TN: weight fragment i of tap t + 2;  i = 6, 7 at taps 0-2: a raw halo pass of chunk c + 1
      //   slots (i, >= 2)   taps 3-8: the transform of pass t - 3, as 64 "pair-ops" (4 channel pairs x 16 dependent single
      //                     instructions: unpack, fma, -log2e, exp2, +1, rcp, multiply, pack, zero-pad select), two pairs
      //                     interleaved so that consecutive instructions of one chain sit one MFMA apart;  (7, TN-1): the LDS write
      auto chunk1 = [&](int c, auto more_) {
        constexpr bool MORE = decltype(more_)::value;
        constexpr int NSLOT = 8 * (TN - 2), OPS = 64 / NSLOT;    // transform slots per tap, pair-ops per slot
        static_assert(TN >= 3 && 64 % NSLOT == 0, "chunk1: TN - 2 transform slots per pixel row");
        const int hb = c & 1;
        const int step0 = c * 9;
        const unsigned char* const hl = hl0 + hb * Lds::HB;
        [[maybe_unused]] uint4 hr[PASSES];
        [[maybe_unused]] float xl[4], xh[4], el[4], eh[4];
        adm_h8 af[3];
        auto rd = [&](int R) -> adm_h8 {   // R = 8 t + i (compile-time in the unrolled stream)
          const int t = R >> 3, i = R & 7;
          return *reinterpret_cast<const adm_h8*>(hl + (((t / 3) * HW2C + (t % 3)) + i * HW2C) * ROWB);
        };
        auto comp = [](uint4& v, int k) -> uint32_t& { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); };
        auto xop = [&](uint4& v, int pr, int op, int ps) {   // pair pr (channels 2 pr, 2 pr + 1 of the lane's segment), instruction op
          uint32_t& u = comp(v, pr);
          const float4 av = afr[pr >> 1], bv = afr[2 + (pr >> 1)];
          const float a_lo = (pr & 1) ? av.z : av.x, a_hi = (pr & 1) ? av.w : av.y;
          const float b_lo = (pr & 1) ? bv.z : bv.x, b_hi = (pr & 1) ? bv.w : bv.y;
          switch (op) {
            case 0: xl[pr] = adm_lo_f32(u); break;
            case 1: xh[pr] = adm_hi_f32(u); break;
            case 2: xl[pr] = __builtin_fmaf(a_lo, xl[pr], b_lo); break;
            case 3: xh[pr] = __builtin_fmaf(a_hi, xh[pr], b_hi); break;
            case 4: if (PRO == 2) el[pr] = xl[pr] * -1.4426950408889634f; break;
            case 5: if (PRO == 2) eh[pr] = xh[pr] * -1.4426950408889634f; break;
            case 6: if (PRO == 2) el[pr] = __builtin_amdgcn_exp2f(el[pr]); break;
            case 7: if (PRO == 2) eh[pr] = __builtin_amdgcn_exp2f(eh[pr]); break;
            case 8: if (PRO == 2) el[pr] = el[pr] + 1.0f; break;
            case 9: if (PRO == 2) eh[pr] = eh[pr] + 1.0f; break;
            case 10: if (PRO == 2) el[pr] = __builtin_amdgcn_rcpf(el[pr]); break;
            case 11: if (PRO == 2) eh[pr] = __builtin_amdgcn_rcpf(eh[pr]); break;
            case 12: if (PRO == 2) xl[pr] = xl[pr] * el[pr]; break;
            case 13: if (PRO == 2) xh[pr] = xh[pr] * eh[pr]; break;
            case 14: u = adm_pack2(xl[pr], xh[pr]); break;
            default: u = pixrel[ps] >= 0 ? u : 0u; break;   // zero padding stays exactly zero
          }
          // pin the instruction into THIS slot: pure arithmetic is otherwise free to move across the sched_barrier fences before
          // instruction selection (hipcc sank whole dependent chains next to their last use); an empty volatile asm that reads
          // and writes the value orders it with the fences at no cost
          if (op == 0 || op == 2 || op == 12) asm volatile("" : "+v"(xl[pr]));
          else if (op == 1 || op == 3 || op == 13) asm volatile("" : "+v"(xh[pr]));
          else if (op >= 14) asm volatile("" : "+v"(u));
          else if (PRO == 2 && !(op & 1)) asm volatile("" : "+v"(el[pr]));
          else if (PRO == 2) asm volatile("" : "+v"(eh[pr]));
        };
        af[0] = rd(0);
        af[1] = rd(1);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int R = t * 8 + i;
            if (R + 2 < 72) af[(R + 2) % 3] = rd(R + 2);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              acc[i][j] = adm_mfma_16x16x32(__builtin_bit_cast(adm_h8, wr[t % 3][j]), af[R % 3], acc[i][j], 0, 0, 0);
              if (j == 1) {
                if (i < TN && (MORE || t + 2 < 9)) wr[(t + 2) % 3][i] = bufload16(rsw, wofs[i], (unsigned)(step0 + t + 2) * wstep);
                if constexpr (MORE) {
                  if (t < 3 && i >= 6 && 2 * t + (i - 6) < PASSES) hr[2 * t + (i - 6)] = halo_load(c + 1, 2 * t + (i - 6));
                }
              }
              if constexpr (MORE) {
                if (PRO != 0 && t == 0 && i == 7 && j == 2) affine_regs(c + 1, img0);   // read by the transforms of taps 3-8
                if (t >= 3 && t - 3 < PASSES) {
                  if (PRO != 0 && j >= 2) {
#pragma unroll
                    for (int q = 0; q < OPS; ++q) {
                      const int m = (i * (TN - 2) + (j - 2)) * OPS + q;    // pair-op 0..63
                      xop(hr[t - 3], (m & 1) + 2 * (m >> 5), (m & 31) >> 1, t - 3);
                    }
                  }
                  if (i == TM - 1 && j == TN - 1)
                    *reinterpret_cast<uint4*>(halo + (hb ^ 1) * Lds::HB + (t - 3) * (NT / SEGP) * ROWB + hslot) = hr[t - 3];
                }
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        __syncthreads();  // halo[hb^1] complete; every wave is done reading halo[hb]
      };
      static_assert(PASSES <= 6, "chunk1 fetches two halo passes per tap over taps 0-2");
      for (int c = cb; c + 1 < ce; ++c) chunk1(c, std::true_type{});
      chunk1(ce - 1, std::false_type{});
    } else if constexpr (T3) {
      // weight ring of 3 K-steps (9 % 3 == 0: the ring slot of tap t is t % 3 in every chunk)
      auto chunk = [&](int c, auto more_, auto rawnext_) {
        constexpr bool MORE = decltype(more_)::value;
        constexpr bool RAWNEXT = decltype(rawnext_)::value;   // FOLD: the chunk staged for the next stage is the skip path's first (raw)
        const int hb = c & 1;
        const int step0 = c * 9;
        uint4 hprev = make_uint4(0, 0, 0, 0);
        float4 affv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          // next chunk's halo: pass t is fetched now; pass t-1 (fetched during the previous tap) is
          // transformed and parked in the other halo buffer
          uint4 hcur = make_uint4(0, 0, 0, 0);
          if constexpr (MORE) {
            if (t < PASSES) hcur = RAWNEXT ? fold_load(0, t) : halo_load(c + 1, t);
          }
          if (MORE || t + 2 < 9) load_w(step0 + t + 2, wr[(t + 2) % 3]);
          // small maps (128-pixel tiles): a mid-chunk rendezvous keeps the waves that share weight fragments
          // in step (L1 reuse); on the 256-pixel tiles it only cost time and is omitted
          if (HALO <= 200 && t == 1) __syncthreads();
          if constexpr (MORE && PRO != 0) {
            // affine table of chunk c+2 -> abuf[hb] (its readers, chunk c-1's transforms, are behind the last
            // barrier): fetched at tap 0, parked at tap 7 so that no wave ever waits on that load
            if (t == 0 && c + 2 < ce) affv = affine_load(c + 2, img0);
            if (t == 7 && c + 2 < ce) affine_park(affv, hb);
          }
          if constexpr (MORE) {
            if (t >= 1 && t - 1 < PASSES) {
              if constexpr (RAWNEXT) raw_write(hprev, t - 1, hb ^ 1);
              else halo_write(hprev, t - 1, hb ^ 1);
            }
          }
          hprev = hcur;
          if (!UPPH || (((0x1b << ((ph >> 1) * 3 + (ph & 1))) >> t) & 1))   // up-conv phase: 5 of the 9 taps carry zero weights (wave-uniform skip)
            mfma_tap(halo + hb * Lds::HB + ((t / 3) * HW2 + (t % 3)) * ROWB, wr[t % 3]);
        }
        __syncthreads();  // halo[hb^1] and abuf[hb] complete; every wave is done reading halo[hb]
      };
      for (int c = cb; c + 1 < ce; ++c) chunk(c, std::true_type{}, std::false_type{});
      if constexpr (!FOLD) {
        chunk(ce - 1, std::false_type{}, std::false_type{});
      } else {
        // the last nine-tap chunk stages skip chunk 0 (raw) into the other halo buffer and its taps 7, 8 fetch the skip path's
        // first two weight K-steps (they follow the 3x3 weights: ring slots 0, 1 again since 9 % 3 == 0)
        chunk(ce - 1, std::true_type{}, std::true_type{});
        // ---- the skip_connection as one-tap K-steps: activations RINGN chunks ahead in registers (chunk k >= 1 in slot k % RINGN),
        // weights two K-steps ahead (ring of 3), one barrier per step like the 1x1 loop; only the centre tap's fragment rows are read
        const int fch = (p.FC0 + p.FC1) / KCS, flast = fch - 1;
        const int fstep0 = chunks * 9;
        const unsigned char* const hc = halo + (HW2 + 1) * ROWB;     // centre tap
        // From skip chunk 1 on only the tile's BM CENTRE pixels are staged (into their slots of the halo geometry; the border slots
        // keep stale data no centre-tap fragment row reads): the skip steps stream their operand at the HBM rate (ablation,
        // profiles/r03/conv_fold_ablation.log: half a step is the wait for these loads, at 20.7 KB per step and CU = 5.3 TB/s over
        // the chip), and the halo ring is 27 % of those bytes.  The per-lane addresses are made here, not in tile_setup: nothing
        // may stay live across the nine-tap loop.
        constexpr int CPASS = BM * SEGP / NT;
        static_assert(CPASS * NT == BM * SEGP && CPASS <= PASSES, "centre staging: whole passes");
        int cpix[CPASS], cslot[CPASS];
        {
          int tid_c = tid;
          asm volatile("" : "+v"(tid_c));
          const int nimg_c = min(p.TI, p.N - img0);
#pragma unroll
          for (int ps = 0; ps < CPASS; ++ps) {
            const int pp = (tid_c + ps * NT) >> SEGSH;                     // centre pixel 0 .. BM - 1
            const int ti = pp >> p.thw_shift, rem = pp & ((1 << p.thw_shift) - 1);
            const int py = rem >> p.tw_shift, px = rem & (p.TW - 1);
            cpix[ps] = ti < nimg_c ? (ti * p.H + y0 + py) * p.W + x0 + px : -1;
            cslot[ps] = (ti * HPI + (py + 1) * HW2 + px + 1) * ROWB + (tid_c & (SEGP - 1)) * 16;
          }
        }
        auto cload = [&](int k, int ps) -> uint4 {
          const int f0chunks = p.FC0 / KCS;
          const bool first = k < f0chunks;
          const int cs = first ? p.FC0 : p.FC1, co = (first ? k : k - f0chunks) * KCS;
          const __amdgpu_buffer_rsrc_t rs = first ? rs2 : rs3;
          const unsigned voff = cpix[ps] >= 0 ? (unsigned)(cpix[ps] * cs + co + seg * 8) * 2u : OOB;
          return bufload16(rs, voff, 0);
        };
#pragma unroll
        for (int q = 1; q <= RINGN; ++q) {
#pragma unroll
          for (int ps = 0; ps < CPASS; ++ps) ring[q % RINGN][ps] = cload(min(q, flast), ps);
        }
        auto fbody = [&](int j, auto s_) {
          constexpr int S = decltype(s_)::value;     // j % 6: ring slots (j + 1) % RINGN (activations), j % 3 (weights)
          if (j > flast) return;
          const int buf = (ce + j) & 1;
#if ADM_FOLD_ABL != 2   // diagnostic builds only (wrong results): 1 no activation loads, 2 no weight loads, 3 no LDS park, 4 no MFMAs
          load_w(fstep0 + min(j + 2, flast), wr[(S + 2) % 3]);
#endif
#if ADM_FOLD_ABL != 4
          mfma_tap(hc + buf * Lds::HB, wr[S % 3]);
#endif
#if ADM_FOLD_ABL != 3
#pragma unroll
          for (int ps = 0; ps < CPASS; ++ps) *reinterpret_cast<uint4*>(halo + (buf ^ 1) * Lds::HB + cslot[ps]) = ring[(S + 1) % RINGN][ps];
#endif
#if ADM_FOLD_ABL != 1
#pragma unroll
          for (int ps = 0; ps < CPASS; ++ps) ring[(S + 1) % RINGN][ps] = cload(min(j + 1 + RINGN, flast), ps);
#endif
          __syncthreads();
        };
        using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>; using J2 = std::integral_constant<int, 2>;
        using J3 = std::integral_constant<int, 3>; using J4 = std::integral_constant<int, 4>; using J5 = std::integral_constant<int, 5>;
        static_assert(6 % RINGN == 0, "skip loop: unrolled by 6, ring slots are compile-time");
        for (int j0 = 0; j0 < fch; j0 += 6) {
          fbody(j0, J0{}); fbody(j0 + 1, J1{}); fbody(j0 + 2, J2{});
          fbody(j0 + 3, J3{}); fbody(j0 + 4, J4{}); fbody(j0 + 5, J5{});
        }
      }
    } else {
      // 1x1: a stage is KS K-steps (KS*TM*TN MFMAs per wave) and ends in the loop's only barrier, so HBM/L2
      // latency must be covered by depth, not by taps: activation segments are fetched 2 stages ahead (register
      // ring of 2), weight fragments 3 K-steps (KS = 1: ring of 4) or one stage (KS = 2: ring of 2 stages) ahead.
      // The loop is unrolled so that every ring slot is a compile-time register; indices past the last stage are
      // clamped (harmless re-reads).
      // (Measured, MI355X: neither 64-channel stages nor running the two waves of a SIMD in opposite phases --
      // MFMAs of stage c against transform + park of stage c+1 -- shortened this loop: ~1 us per K-step either way.
      // Ablations on the timing build: without the transform + park the GN-prologue loop is 43 % shorter, without
      // the activation loads the raw loop is 48 % shorter; weights and the barrier are not the limit.)
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
      const int last = ce - 1;
#if ADM_CONV_RD == 3
      if constexpr (KS == 1) {
        // experiment: three activation stages in flight, weights two K-steps ahead (rings of 3, unrolled by 3)
        auto body = [&](int c, auto s_) {
          constexpr int S = decltype(s_)::value;
          if (c > last) return;
          affine_park(affine_load(min(c + 2, last), img0), c & 1);
#pragma unroll
          for (int ps = 0; ps < PASSES; ++ps) ring[S][ps] = halo_load(min(c + 3, last), ps);
          load_w(min(c + 2, last), wr[(S + 2) % 3]);
          mfma_tap(halo + (c & 1) * Lds::HB, wr[S]);
#pragma unroll
          for (int ps = 0; ps < PASSES; ++ps) halo_write(ring[(S + 1) % 3][ps], ps, (c + 1) & 1);
          __syncthreads();
        };
        for (int c0 = 0; c0 < chunks; c0 += 3) {
          body(c0, I0{});
          body(c0 + 1, I1{});
          body(c0 + 2, I2{});
        }
      } else
#endif
      if constexpr (KS == 1) {
        // NOEXIT (the 128-pixel tiles, never split-K): whole groups of four steps without the early exit.  With it every step is a
        // control-flow merge and hipcc's wait-count insertion drains the prefetch (vmcnt(0) in two of the four steps, vmcnt(5) in
        // the others, where the rings allow 6-8 loads in flight): the load latency was exposed in every other step -- 0.85 us per
        // step on SD v1's 1280-wide projections, whose 8 MFMAs per wave need 0.1.  Steps past the last one multiply the re-read
        // last chunk by weight fragments the buffer descriptor returns as zeros (K-step index beyond wbytes): they add nothing.
        // (The 256-pixel tiles keep the exit: their GN-prologue instantiation spills without it, see below.)
        constexpr bool NOEXIT = ADM_CONV_1X1_NOEXIT && TM == 4 && !COLD;
        auto body = [&](int c, auto sa_, auto sw_) {
          constexpr int SA = decltype(sa_)::value, SW = decltype(sw_)::value;
          if constexpr (!NOEXIT) {
            if (c > last) return;
          }
          const int c2 = min(c + 2, last);
          affine_park(affine_load(c2, img0), c & 1);
          // activation ring: chunk k lives in slot k % RINGN; chunk c's slot is free (parked during step c - 1)
          (void)SA;
#pragma unroll
          for (int ps = 0; ps < PASSES; ++ps) ring[SW % RINGN][ps] = NOEXIT ? halo_load_pad(c + RINGN, last, ps) : halo_load(min(c + RINGN, last), ps);
          load_w(NOEXIT ? c + WRING - 1 : min(c + WRING - 1, last), wr[(SW + WRING - 1) % WRING]);
          mfma_tap(halo + (c & 1) * Lds::HB, wr[SW % WRING]);
#pragma unroll
          for (int ps = 0; ps < PASSES; ++ps) halo_write(ring[(SW + 1) % RINGN][ps], ps, (c + 1) & 1);
          __syncthreads();
        };
        // (Peeling the tail so that the unrolled group has no early exit removes the s_waitcnt vmcnt(0) hipcc puts
        // at the loop header, but the 192-wide GN-prologue instantiation then spills inside the loop: 530 -> 690 us
        // on qkv 384->1152 @32^2; the 128-wide tile, which does not spill, gained 3 %.)
        if constexpr (WRING == 8) {
          using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
          using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>;
          for (int c0 = cb; c0 < ce; c0 += 8) {
            body(c0, I0{}, I0{}); body(c0 + 1, I1{}, I1{}); body(c0 + 2, I0{}, I2{}); body(c0 + 3, I1{}, I3{});
            body(c0 + 4, I0{}, I4{}); body(c0 + 5, I1{}, I5{}); body(c0 + 6, I0{}, I6{}); body(c0 + 7, I1{}, I7{});
          }
        } else
        for (int c0 = cb; c0 < ce; c0 += 4) {   // cb: 0, or even (split-K)
          body(c0, I0{}, I0{});
          body(c0 + 1, I1{}, I1{});
          body(c0 + 2, I0{}, I2{});
          body(c0 + 3, I1{}, I3{});
        }
      } else {
        auto body = [&](int c, auto sa_) {
          constexpr int SA = decltype(sa_)::value;  // parity of c
          if (c > last) return;
          const int c2 = min(c + 2, last);
          // weights first: the wait for them at the next stage must not also drain the (slower, HBM) activation
          // loads issued behind them -- vmcnt retires in order
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) load_w(min(c + 1, last) * KS + ks, wr[(SA ^ 1) * KS + ks]);
          affine_park(affine_load(c2, img0), c & 1);
#pragma unroll
          for (int ps = 0; ps < PASSES; ++ps) ring[SA][ps] = halo_load(c2, ps);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) mfma_tap(halo + SA * Lds::HB + ks * 64, wr[SA * KS + ks]);
#pragma unroll
          for (int ps = 0; ps < PASSES; ++ps) halo_write(ring[SA ^ 1][ps], ps, SA ^ 1);
          __syncthreads();
        };
        for (int c0 = 0; c0 < chunks; c0 += 2) {
          body(c0, I0{});
          body(c0 + 1, I1{});
        }
      }
    }
    ADM_TSTAMP(ltile, 2);

    // ---- tile switch.  Every wave is behind the K loop's last barrier: both halo buffers and abuf are free.
    const int ltile_done = ltile;
    (void)ltile_done;
    if (!COLD || (p.out_mode == 0 && p.ksplit <= 1)) {
      // bf16 NHWC: the whole tile (acc + bias, bf16) is staged in LDS in ONE round and leaves as whole pixel
      // rows with 16-byte lanes (BN*2 contiguous bytes per pixel); the residual operand is fetched with the
      // same coalesced shape before the staging barrier, so that its latency overlaps the LDS round trip.
      // (Fragment-shaped 8-byte stores touch 16 cache lines per instruction and dominated 1x1 convs.)
      constexpr int EROW = BN * 2 + 16;       // staged bytes per pixel
      constexpr int SEGS = BN / 8;            // 16-byte segments per pixel
      constexpr int PR = NT / SEGS;           // pixel rows written per sweep
      constexpr int RG = BM / SGROUPS;        // pixel rows per statistics group
      constexpr int NIT = (RG + PR - 1) / PR; // sweeps per group
      // each thread keeps ONE 16-byte channel segment and walks pixel rows, so that it can also accumulate
      // the GroupNorm statistics of the values it stores (the consumer's adm_gn_partial pass is then unnecessary)
      int tid_e = tid;
      asm volatile("" : "+v"(tid_e));  // opaque per tile: keeps this cold address math out of the K loop's live set
      const int sg = tid_e % SEGS, prow = tid_e / SEGS;
      const int gch = nb * BN + sg * 8;
      const bool act = prow < PR && gch < p.Cout;
      const int thw_mask = (1 << p.thw_shift) - 1, tw_mask = (1 << p.tw_shift) - 1;
      // (a) this tile's output / residual bases and statistics destinations (the per-row offsets follow the
      // tile switch: they only need the first image of the finished tile)
      // element offset of the tile origin (up-conv phase: output map 2H x 2W, this phase's pixels at stride 2)
      const long long ebase = UPPH ? (((long long)img0 * (2 * p.H) + 2 * y0 + (ph >> 1)) * (2 * p.W) + 2 * x0 + (ph & 1)) * p.Cout + gch
                                   : (((long long)img0 * p.H + y0) * p.W + x0) * p.Cout + gch;
      uint16_t* const obase = reinterpret_cast<uint16_t*>(p.out) + ebase;
      // residual through a virtual 2x upsample: source map (H/2) x (W/2); tile origins are even
      const int Hr = p.H >> p.res_up, Wr = p.W >> p.res_up;
      const uint16_t* const rbase = p.res ? p.res + (((long long)img0 * Hr + (y0 >> p.res_up)) * Wr + (x0 >> p.res_up)) * p.Cout + gch : nullptr;
      const int img0_d = img0;
      float* sdst[SGROUPS];
#pragma unroll
      for (int g = 0; g < SGROUPS; ++g) {
        const int n = img0 + ((g * RG) >> p.thw_shift);
        // slab of this group inside its image: (tile of the image) * SGROUPS + g for one-image tiles
        const int slab = p.TI == 1 ? (UPPH ? (mt % (p.tiles_x * p.tiles_y)) * 4 + ph : (mt % (p.tiles_x * p.tiles_y)) * SGROUPS + g) : 0;
        sdst[g] = (p.stats && tid_e < BN && nb * BN + tid_e < p.Cout && n < p.N)
                      ? p.stats + (((long long)n * p.stat_slabs + slab) * p.Cout + nb * BN + tid_e) * 2 : nullptr;
      }
      ADM_TSTAMP(ltile_done, 10);
      // (b) stage the accumulators (bias included since the start) as bf16 for the whole tile
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int ch0 = (wn * TN + j) * 16 + lq * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          uint2 o;
          o.x = adm_pack2(acc[i][j][0], acc[i][j][1]);
          o.y = adm_pack2(acc[i][j][2], acc[i][j][3]);
          *reinterpret_cast<uint2*>(stg + ((wm * TM + i) * 16 + lc) * EROW + ch0 * 2) = o;
        }
      }
      ADM_TSTAMP(ltile_done, 11);
      // (c) switch the tile state; the next tile's first chunk and affine tables start their way into LDS (after the
      // staging writes: the compiler orders every later LDS access behind a pending LDS-DMA)
      if (more) {
        tl = tnext;
        ltile = tstart + tl;
        tile_origin(ltile, nb, mt, img0, y0, x0);
        tile_setup();
      }
      int eoff[SGROUPS][NIT];               // element offset relative to the tile origin, or -1
#pragma unroll
      for (int g = 0; g < SGROUPS; ++g)
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int ml = prow + k * PR, m = g * RG + ml;
          const int ti = m >> p.thw_shift, rem = m & thw_mask;
          eoff[g][k] = (act && ml < RG && img0_d + ti < p.N)
                           ? (UPPH ? ((rem >> p.tw_shift) * (4 * p.W) + 2 * (rem & tw_mask)) * p.Cout   // TI == 1: rows of the 2W-wide map, 2 apart
                                   : ((ti * p.H + (rem >> p.tw_shift)) * p.W + (rem & tw_mask)) * p.Cout)
                           : -1;
        }
      ADM_TSTAMP(ltile_done, 12);
      uint4 rr[SGROUPS][NIT];
      // the residual operand: same coalesced shape as the stores; in flight across the barrier and the parking
      // of the next tile's first chunk (the accumulators' registers are free from here on)
#pragma unroll
      for (int g = 0; g < SGROUPS; ++g)
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          rr[g][k] = make_uint4(0, 0, 0, 0);
          if (rbase && eoff[g][k] >= 0) {
            const int m = g * RG + prow + k * PR;
            const int ti = m >> p.thw_shift, rem = m & thw_mask;
            const int roff = ((ti * Hr + ((rem >> p.tw_shift) >> p.res_up)) * Wr + ((rem & tw_mask) >> p.res_up)) * p.Cout;
            rr[g][k] = *reinterpret_cast<const uint4*>(rbase + roff);
          }
        }
      [[maybe_unused]] float ga[8], gb[8];   // GNB: affine of this thread's 8 channels in the finished tile's image (TI == 1)
      if constexpr (GNB) {
        const long long ao = (long long)img0_d * p.Cout + gch;
        if (act) {
          *reinterpret_cast<float4*>(ga) = *reinterpret_cast<const float4*>(p.aa + ao);
          *reinterpret_cast<float4*>(ga + 4) = *reinterpret_cast<const float4*>(p.aa + ao + 4);
          *reinterpret_cast<float4*>(gb) = *reinterpret_cast<const float4*>(p.ab + ao);
          *reinterpret_cast<float4*>(gb + 4) = *reinterpret_cast<const float4*>(p.ab + ao + 4);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) { ga[e] = 0.f; gb[e] = 0.f; }
        }
      }
      if (more) first_loads((tid_e & 63) >> 4);
      ADM_TSTAMP(ltile_done, 13);
      __syncthreads();
      ADM_TSTAMP(ltile_done, 3);
      if (more) first_park();
      ADM_TSTAMP(ltile_done, 4);
      // (d) sweep the staged tile out: residual add, store and statistics
      // packed fp32 math (v_pk_add_f32 / v_pk_fma_f32): this phase is VALU-bound, not store-bound
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 s1[SGROUPS][4], s2[SGROUPS][4];
#pragma unroll
      for (int g = 0; g < SGROUPS; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[g][q] = f32x2{0.f, 0.f}; s2[g][q] = f32x2{0.f, 0.f}; }
#pragma unroll
      for (int g = 0; g < SGROUPS; ++g)
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          if (eoff[g][k] < 0) continue;
          uint4 v = *reinterpret_cast<const uint4*>(stg + (g * RG + prow + k * PR) * EROW + sg * 16);
          uint32_t a4[4] = {v.x, v.y, v.z, v.w};
          [[maybe_unused]] f32x2 xg[4];
          if constexpr (GNB) {
            // dz = dy * SiLU'(a x + b), SiLU'(z) = s (1 + z (1 - s)), s = sigmoid(z); x rides in the residual registers
            const uint32_t r4[4] = {rr[g][k].x, rr[g][k].y, rr[g][k].z, rr[g][k].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              xg[q] = f32x2{adm_lo_f32(r4[q]), adm_hi_f32(r4[q])};
              const f32x2 z = f32x2{ga[2 * q], ga[2 * q + 1]} * xg[q] + f32x2{gb[2 * q], gb[2 * q + 1]};
              const f32x2 sgm = f32x2{__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z.x * -1.4426950408889634f)),
                                      __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z.y * -1.4426950408889634f))};
              const f32x2 t = f32x2{adm_lo_f32(a4[q]), adm_hi_f32(a4[q])} * (sgm * (1.0f + z * (1.0f - sgm)));
              a4[q] = adm_pack2(t.x, t.y);
            }
            v = make_uint4(a4[0], a4[1], a4[2], a4[3]);
          } else if (rbase) {
            const uint32_t r4[4] = {rr[g][k].x, rr[g][k].y, rr[g][k].z, rr[g][k].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x2 t = f32x2{adm_lo_f32(a4[q]), adm_hi_f32(a4[q])} +
                              f32x2{adm_lo_f32(r4[q]), adm_hi_f32(r4[q])};
              a4[q] = adm_pack2(t.x, t.y);
            }
            v = make_uint4(a4[0], a4[1], a4[2], a4[3]);
          }
          {
            typedef __attribute__((ext_vector_type(4))) unsigned int nt_u32x4;
            __builtin_nontemporal_store(nt_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_u32x4*>(obase + eoff[g][k]));
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 t = f32x2{adm_lo_f32(a4[q]), adm_hi_f32(a4[q])};
            s1[g][q] += t;
            if constexpr (GNB) s2[g][q] = __builtin_elementwise_fma(t, xg[q], s2[g][q]);   // sum dz * x
            else s2[g][q] = __builtin_elementwise_fma(t, t, s2[g][q]);
          }
        }
      ADM_TSTAMP(ltile_done, 5);
      if (p.stats) {
        // reduce the PR row-partials of every channel through LDS (fixed order: bitwise reproducible)
        float* red = reinterpret_cast<float*>(stg);  // [PR][BN][2]
#pragma unroll
        for (int g = 0; g < SGROUPS; ++g) {
          __syncthreads();  // staging buffer (or the previous group's partials) fully consumed
          if (prow < PR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              red[(prow * BN + sg * 8 + e) * 2 + 0] = s1[g][e >> 1][e & 1];
              red[(prow * BN + sg * 8 + e) * 2 + 1] = s2[g][e >> 1][e & 1];
            }
          }
          __syncthreads();
          if (sdst[g]) {
            float t1 = 0.f, t2 = 0.f;
            for (int q = 0; q < PR; ++q) { t1 += red[(q * BN + tid) * 2]; t2 += red[(q * BN + tid) * 2 + 1]; }
            sdst[g][0] = t1;
            sdst[g][1] = t2;
          }
        }
      }
      ADM_TSTAMP(ltile_done, 6);
    } else if constexpr (!COLD) {
      // (not in this build: adm_conv sends split-K and fp32 NCHW launches to the COLD instantiations)
    } else if (p.ksplit > 1) {
      // split-K partial sums: fp32 [split][pixel][Cout], 16 bytes per lane (4 consecutive channels of one pixel)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = (wm * TM + i) * 16 + lc;
        const int ti = m >> p.thw_shift, rem = m & ((1 << p.thw_shift) - 1);
        const int n = img0 + ti, y = y0 + (rem >> p.tw_shift), x = x0 + (rem & ((1 << p.tw_shift) - 1));
        if (n >= p.N) continue;
        float* row = p.ws + ((long long)sp * p.N * HWimg + ((long long)n * p.H + y) * p.W + x) * p.Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int ch0 = nb * BN + (wn * TN + j) * 16 + lq * 4;
          if (ch0 + 3 < p.Cout) *reinterpret_cast<float4*>(row + ch0) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
      }
      if (more) {
        tl = tnext;
        ltile = tstart + tl;
        tile_origin(ltile, nb, mt, img0, y0, x0);
        tile_setup();
        first_loads(lq);
        __syncthreads();
        first_park();
      }
    } else {
      // fp32 NCHW (output head / stem backward): few channels, direct stores
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = (wm * TM + i) * 16 + lc;
        const int ti = m >> p.thw_shift, rem = m & ((1 << p.thw_shift) - 1);
        const int n = img0 + ti, y = y0 + (rem >> p.tw_shift), x = x0 + (rem & ((1 << p.tw_shift) - 1));
        if (n >= p.N) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int ch0 = nb * BN + (wn * TN + j) * 16 + lq * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ch = ch0 + e;
            if (ch >= p.Cout) continue;
            reinterpret_cast<float*>(p.out)[(((long long)n * p.Cout + ch) * p.H + y) * p.W + x] =
                (acc[i][j][e] + (ch0 + 3 < p.Cout ? 0.f : p.bias[ch])) * p.out_scale;
          }
        }
      }
      if (more) {
        tl = tnext;
        ltile = tstart + tl;
        tile_origin(ltile, nb, mt, img0, y0, x0);
        tile_setup();
              first_loads(lq);
        __syncthreads();
        first_park();
      }
    }
    if (!more) break;
  }
}

// ------------------------------------------------------------------------------------------------
// The dominant layer class (3x3, maps >= 16x16, Cout a multiple of 192) on v_mfma_f32_32x32x16_bf16:
// the same pipeline as conv_kernel<2,4,8,3,...,9,324,PRO>, but a 32x32x16 MFMA occupies the SIMD's issue
// port for 8 of its 32 cycles (16x16x32: 8 of 16), which leaves the partner wave twice the issue slots
// for the prologue VALU work, the LDS reads and the loads.  8 waves as 4 (pixels) x 2 (channels); each
// wave owns 64 pixels x 96 channels = 2 x 3 tiles of 32x32 (96 accumulator registers).
// Weights come from the 32x32 fragment-ordered image [Cin/32][9][ceil(Cout/32)][2 k-steps][64][8].
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int PRO>
__global__ void __launch_bounds__(512, 2)
conv32_kernel(const ConvK p) {
  constexpr int NT = 512, WN = 2, TM = 2, TN = 3, BN = 192, HALO = 324, WMR = 4;
  constexpr int PASSES = (HALO * 4 + NT - 1) / NT;  // 3
  constexpr int HSLOT = halo_slots<NT, HALO>();      // 384
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const halo = smem;
  float* const abuf = reinterpret_cast<float*>(smem + 2 * HSLOT * ROWB);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lp = lane & 31, lh = lane >> 5;

  const int nb = blockIdx.x % p.nblocks_n;
  const int mt = blockIdx.x / p.nblocks_n;
  const int HW2 = p.TW + 2;
  const int HPI = (p.TH + 2) * HW2;
  const int Cin = p.C0 + p.C1;
  const int HWimg = p.H * p.W;
  const int per_img = p.tiles_x * p.tiles_y;
  const int img0 = mt / per_img;
  const int y0 = ((mt % per_img) / p.tiles_x) * p.TH, x0 = ((mt % per_img) % p.tiles_x) * p.TW;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.in0 + (long long)img0 * HWimg * p.C0), 0, HWimg * p.C0 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.C1 ? p.in1 + (long long)img0 * HWimg * p.C1 : p.in0), 0, p.C1 ? HWimg * p.C1 * 2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);

  int pixrel[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int hp = (tid + ps * NT) >> 2;
    int off = -1;
    if (hp < HPI) {
      const int y = y0 + hp / HW2 - 1, x = x0 + hp % HW2 - 1;
      if (y >= 0 && y < p.H && x >= 0 && x < p.W) off = y * p.W + x;
    }
    pixrel[ps] = off;
  }
  const int seg = tid & 3;
  const int hslot = (tid >> 2) * ROWB + seg * 16;

  // activation (B operand) fragment of a 32-pixel tile: lane (pixel lp, half lh) reads 16 bytes = k 8*lh..8*lh+7
  const int alane = ((lp / p.TW) * HW2 + lp % p.TW) * ROWB + lh * 16;
  int aoff[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) aoff[i] = (((wm * TM + i) * 32) / p.TW) * HW2 * ROWB;
  // weight (A operand) fragments: tile32 x k-step
  const int ntiles32 = (p.Cout + 31) / 32;
  unsigned wofs[TN][2];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int tile = nb * (BN / 32) + wn * TN + j;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wofs[j][ks] = tile < ntiles32 ? (unsigned)(((tile * 2 + ks) * 64 + lane) * 16) : OOB;
  }
  const unsigned wstep = (unsigned)ntiles32 * 2048u;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int chunks = Cin / KC;
  const int c0chunks = p.C0 / KC;

  auto halo_load = [&](int c, int ps) -> uint4 {
    const bool first = c < c0chunks;
    const int cs = first ? p.C0 : p.C1, co = (first ? c : c - c0chunks) * KC;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const unsigned voff = pixrel[ps] >= 0 ? (unsigned)(pixrel[ps] * cs + co + seg * 8) * 2u : OOB;
    return bufload16(rs, voff, 0);
  };
  auto stage_affine = [&](int c, int buf) {
    if constexpr (PRO != 0) {
      if (tid < 16) {
        const int part = tid & 15;
        const float* s = (part < 8 ? p.aa : p.ab) + (long long)img0 * Cin + c * KC + (part & 7) * 4;
        *reinterpret_cast<float4*>(abuf + buf * 64 + (part < 8 ? 0 : 32) + (part & 7) * 4) = *reinterpret_cast<const float4*>(s);
      }
    }
  };
  auto halo_write = [&](uint4 v, int ps, int buf) {
    if constexpr (PRO != 0) {
      const float* ab = abuf + buf * 64 + seg * 8;
      float a8[8], b8[8];
      *reinterpret_cast<float4*>(a8) = *reinterpret_cast<const float4*>(ab);
      *reinterpret_cast<float4*>(a8 + 4) = *reinterpret_cast<const float4*>(ab + 4);
      *reinterpret_cast<float4*>(b8) = *reinterpret_cast<const float4*>(ab + 32);
      *reinterpret_cast<float4*>(b8 + 4) = *reinterpret_cast<const float4*>(ab + 36);
      uint32_t u[4] = {v.x, v.y, v.z, v.w};
      const bool valid = pixrel[ps] >= 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float lo = adm_lo_f32(u[j]), hi = adm_hi_f32(u[j]);
        lo = a8[2 * j] * lo + b8[2 * j];
        hi = a8[2 * j + 1] * hi + b8[2 * j + 1];
        if constexpr (PRO == 2) { lo = adm_silu(lo); hi = adm_silu(hi); }
        const uint32_t pk = adm_pack2(lo, hi);
        u[j] = valid ? pk : 0u;
      }
      v = make_uint4(u[0], u[1], u[2], u[3]);
    }
    *reinterpret_cast<uint4*>(halo + buf * (HSLOT * ROWB) + ps * (NT / 4) * ROWB + hslot) = v;
  };
  auto load_w = [&](int step, uint4 (&dst)[TN][2]) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) dst[j][ks] = bufload16(rsw, wofs[j][ks], (unsigned)step * wstep);
  };
  auto mfma_tap = [&](const unsigned char* hp, const uint4 (&w)[TN][2]) {
    const unsigned char* hl = hp + alane;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const adm_h8 af = *reinterpret_cast<const adm_h8*>(hl + aoff[i] + ks * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = adm_mfma_32x32x16(__builtin_bit_cast(adm_h8, w[j][ks]), af, acc[i][j], 0, 0, 0);
      }
  };

  stage_affine(0, 0);
  {
    uint4 h0[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) h0[ps] = halo_load(0, ps);
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) halo_write(h0[ps], ps, 0);
  }
  uint4 wreg[3][TN][2];
  load_w(0, wreg[0]);
  load_w(1, wreg[1]);
  __syncthreads();
  auto chunk = [&](int c, auto more_) {
    constexpr bool MORE = decltype(more_)::value;
    const int hb = c & 1;
    const int step0 = c * 9;
    uint4 hprev = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      uint4 hcur = make_uint4(0, 0, 0, 0);
      if constexpr (MORE) {
        if (t == 0) stage_affine(c + 1, hb ^ 1);
        if (t < PASSES) hcur = halo_load(c + 1, t);
      }
      if (MORE || t + 2 < 9) load_w(step0 + t + 2, wreg[(t + 2) % 3]);
      if (t == 1) __syncthreads();
      if constexpr (MORE) {
        if (t >= 1 && t - 1 < PASSES) halo_write(hprev, t - 1, hb ^ 1);
      }
      hprev = hcur;
      mfma_tap(halo + hb * (HSLOT * ROWB) + ((t / 3) * HW2 + (t % 3)) * ROWB, wreg[t % 3]);
    }
    __syncthreads();
  };
  for (int c = 0; c + 1 < chunks; ++c) chunk(c, std::true_type{});
  chunk(chunks - 1, std::false_type{});

  // ---- epilogue (bf16 NHWC): stage one wave-row (64 pixels x 192 channels) at a time, coalesced rows out.
  // 32x32 accumulator: register e -> channel (e&3) + 8*(e>>2) + 4*lh of the tile, pixel lp.
  constexpr int RPX = TM * 32, EROW = BN * 2 + 16, SEGS = BN / 8;
  uint16_t* const outp = reinterpret_cast<uint16_t*>(p.out);
  for (int r = 0; r < WMR; ++r) {
    __syncthreads();
    if (wm == r) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch0 = (wn * TN + j) * 32 + 8 * g + 4 * lh;
          const int gch = nb * BN + ch0;
          float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
          if (gch + 3 < p.Cout) bs = *reinterpret_cast<const float4*>(p.bias + gch);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            uint2 o;
            o.x = adm_pack2(acc[i][j][4 * g + 0] + bs.x, acc[i][j][4 * g + 1] + bs.y);
            o.y = adm_pack2(acc[i][j][4 * g + 2] + bs.z, acc[i][j][4 * g + 3] + bs.w);
            *reinterpret_cast<uint2*>(smem + (i * 32 + lp) * EROW + ch0 * 2) = o;
          }
        }
    }
    __syncthreads();
    for (int u = tid; u < RPX * SEGS; u += NT) {
      const int pl = u / SEGS, sg = u % SEGS;
      const int gch = nb * BN + sg * 8;
      if (gch >= p.Cout) continue;
      const int m = r * RPX + pl;
      const long long pix = ((long long)img0 * p.H + y0 + m / p.TW) * p.W + x0 + m % p.TW;
      uint4 v = *reinterpret_cast<const uint4*>(smem + pl * EROW + sg * 16);
      if (p.res) {
        const uint4 rr = *reinterpret_cast<const uint4*>(p.res + pix * p.Cout + gch);
        uint32_t a4[4] = {v.x, v.y, v.z, v.w};
        const uint32_t r4[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = adm_lo_f32(a4[q]) + adm_lo_f32(r4[q]);
          const float hi = adm_hi_f32(a4[q]) + adm_hi_f32(r4[q]);
          a4[q] = adm_pack2(lo, hi);
        }
        v = make_uint4(a4[0], a4[1], a4[2], a4[3]);
      }
      *reinterpret_cast<uint4*>(outp + pix * p.Cout + gch) = v;
    }
  }
}

// fp32 [cout][cin][taps] -> bf16 [cin/32][taps][ceil(cout/32)][2][lane = h*32 + r][8]
//   channel = tile*32 + r, k = chunk*32 + ks*16 + h*8 + e  (A-operand fragment of v_mfma_f32_32x32x16_bf16)
__global__ void __launch_bounds__(256)
pack_weight32_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cout, int cin, int taps, int ntiles32) {
  const long long total = (long long)(cin / KC) * taps * ntiles32 * 1024;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), ln = (int)((i >> 3) & 63), ks = (int)((i >> 9) & 1);
    long long r = i >> 10;
    const int tile = (int)(r % ntiles32); r /= ntiles32;
    const int tap = (int)(r % taps);
    const int chunk = (int)(r / taps);
    const int ch = tile * 32 + (ln & 31);
    const int k = chunk * KC + ks * 16 + (ln >> 5) * 8 + e;
    float v = 0.0f;
    if (ch < cout) v = w[((long long)ch * cin + k) * taps + tap];
    out[i] = adm_f32_to_h(v);
  }
}

template <int PRO>
int launch_conv32(const ConvK& k, int m_tiles, hipStream_t s) {
  constexpr int smem = conv_smem_bytes<512, 192, 324, 64>();
  ConvK kk = k;
  kk.nblocks_n = (k.Cout + 191) / 192;
  const long long blocks = (long long)m_tiles * kk.nblocks_n;
  ADM_REQUIRE(blocks < (1ll << 31), ADM_E_SHAPE, "adm_conv: grid too large");
  static bool attr_set_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set_dev[dev & 63]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv32_kernel<PRO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) ADM_FAIL((int)e, "adm_conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set_dev[dev & 63] = true;
  }
  hipLaunchKernelGGL((conv32_kernel<PRO>), dim3((unsigned)blocks), dim3(512), smem, s, kk);
  return adm_check_launch("adm_conv");
}

// fp32 [cout][cin][taps] -> bf16 [cin/32][taps][ceil(cout/16)][lane = q*16 + r][8]
//   channel = tile*16 + r, k = chunk*32 + q*8 + e  (A-operand fragment of v_mfma_f32_16x16x32_bf16)
__global__ void __launch_bounds__(256)
pack_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cout, int cin, int taps, int ntiles16) {
  const long long total = (long long)(cin / KC) * taps * ntiles16 * 512;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), ln = (int)((i >> 3) & 63);
    long long r = i >> 9;
    const int tile = (int)(r % ntiles16); r /= ntiles16;
    const int tap = (int)(r % taps);
    const int chunk = (int)(r / taps);
    const int ch = tile * 16 + (ln & 15);
    const int k = chunk * KC + (ln >> 4) * 8 + e;
    float v = 0.0f;
    if (ch < cout) v = w[((long long)ch * cin + k) * taps + tap];
    out[i] = adm_f32_to_h(v);
  }
}

// Second half of a split-K conv: out[pixel][c] = bf16(bias[c] + sum_s ws[s][pixel][c]) (+ res), splits added in
// index order (deterministic), plus the output statistics of the fused-GroupNorm hand-over: per (image, slab, channel)
// sum and sum of squares of the bf16 values written, slab = a contiguous run of HW / slabs pixels (the consumer only adds
// the slabs up).  One block per (64-channel group, slab, image): 8 threads x 8 channels across, 32 pixel lanes down.
__global__ void __launch_bounds__(256)
conv_splitk_reduce(const float* __restrict__ ws, int ksplit, const float* __restrict__ bias, const uint16_t* __restrict__ res,
                   uint16_t* __restrict__ out, float* __restrict__ stats, int n_img, int hw, int cout, int slabs) {
  __shared__ float red[32][64][2];
  const int g8 = threadIdx.x & 7, lane = threadIdx.x >> 3;
  const int ch = blockIdx.x * 64 + g8 * 8;
  const int slab = blockIdx.y, img = blockIdx.z;
  const int per = (hw + slabs - 1) / slabs;
  const int p_begin = slab * per, p_end = min(hw, p_begin + per);
  const long long split_stride = (long long)n_img * hw * cout;
  const bool act = ch < cout;
  float t1[8] = {}, t2[8] = {};
  if (act) {
    float b8[8];
    *reinterpret_cast<float4*>(b8) = *reinterpret_cast<const float4*>(bias + ch);
    *reinterpret_cast<float4*>(b8 + 4) = *reinterpret_cast<const float4*>(bias + ch + 4);
    for (int px = p_begin + lane; px < p_end; px += 32) {
      const long long e = ((long long)img * hw + px) * cout + ch;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = b8[j];
      for (int sidx = 0; sidx < ksplit; ++sidx) {
        const float4 a = *reinterpret_cast<const float4*>(ws + sidx * split_stride + e);
        const float4 b = *reinterpret_cast<const float4*>(ws + sidx * split_stride + e + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
      }
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = adm_pack2(v[2 * j], v[2 * j + 1]);
      if (res) {   // as the one-pass epilogue: the conv result is rounded to bf16 first, then the residual is added
        const uint4 rv = *reinterpret_cast<const uint4*>(res + e);
        const uint32_t r4[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = adm_lo_f32(o[j]) + adm_lo_f32(r4[j]);
          const float hi = adm_hi_f32(o[j]) + adm_hi_f32(r4[j]);
          o[j] = adm_pack2(lo, hi);
        }
      }
      *reinterpret_cast<uint4*>(out + e) = make_uint4(o[0], o[1], o[2], o[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = adm_lo_f32(o[j]), hi = adm_hi_f32(o[j]);
        t1[2 * j] += lo; t2[2 * j] += lo * lo; t1[2 * j + 1] += hi; t2[2 * j + 1] += hi * hi;
      }
    }
  }
  if (!stats) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[lane][g8 * 8 + j][0] = t1[j]; red[lane][g8 * 8 + j][1] = t2[j]; }
  __syncthreads();
  if (threadIdx.x < 128) {   // fixed-order sum over the 32 pixel lanes: bitwise reproducible
    const int c = threadIdx.x >> 1, q = threadIdx.x & 1;
    float t = 0.0f;
    for (int l = 0; l < 32; ++l) t += red[l][c][q];
    if (blockIdx.x * 64 + c < cout) stats[(((long long)img * slabs + slab) * cout + blockIdx.x * 64 + c) * 2 + q] = t;
  }
}

template <int WM, int WN, int TM, int TN, int OCC, int TAPS, int HALO, int PRO, int KS, bool COLD>
int launch_conv_p(const ConvK& k, int m_tiles, hipStream_t s) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BN = WN * TN * 16;
  constexpr int smem = ConvLds<NT, BN, HALO, WM * TM * 16, KS>::BYTES;
  static_assert(smem <= 160 * 1024, "conv_kernel LDS map exceeds the CU's 160 KB");
  // per device: opt in to the LDS size once, and size the persistent grid to the resident blocks
  static int percu_dev[64] = {}, ncu_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& per_cu_d = percu_dev[dev & 63];
  int& ncu_d = ncu_dev[dev & 63];
  const void* fn = reinterpret_cast<const void*>(&conv_kernel<WM, WN, TM, TN, OCC, TAPS, HALO, PRO, KS, COLD>);
  if (per_cu_d == 0) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) ADM_FAIL((int)e, "adm_conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
    int per_cu = 0, ncu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, NT, smem);
    if (e != hipSuccess || per_cu <= 0) ADM_FAIL((int)e, "adm_conv: block of %d threads / %d B LDS is not launchable", NT, smem);
    e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || ncu <= 0) ADM_FAIL((int)e, "adm_conv: hipDeviceGetAttribute: %s", hipGetErrorString(e));
    ncu_d = ncu;
    per_cu_d = per_cu;
  }
  // a CU-masked stream owns fewer CUs: the persistent grid is sized to those (result-neutral: a tile's arithmetic does not depend
  // on which block of the grid walks it)
  const int slots = per_cu_d * adm_stream_cus((void*)s, ncu_d);
  ConvK kk = k;
  kk.nbp = (k.Cout + BN - 1) / BN;
  kk.nblocks_n = kk.nbp * (k.nph > 1 ? k.nph : 1);
  const long long tiles = (long long)m_tiles * kk.nblocks_n * (k.ksplit > 1 ? k.ksplit : 1);
  ADM_REQUIRE(tiles < (1ll << 31), ADM_E_SHAPE, "adm_conv: too many tiles");
  kk.total_tiles = (int)tiles;
  unsigned blocks = (unsigned)(tiles < slots ? tiles : slots);
#ifdef ADM_CONV_TIMING
  // diagnostic build only: cap the persistent grid (how long is a tile's epilogue when few CUs share the HBM?)
  if (const char* e = getenv("ADM_CONV_MAX_BLOCKS")) { const unsigned cap = (unsigned)atoi(e); if (cap > 0 && cap < blocks) blocks = cap; }
#endif
  hipLaunchKernelGGL((conv_kernel<WM, WN, TM, TN, OCC, TAPS, HALO, PRO, KS, COLD>), dim3(blocks), dim3(NT), smem, s, kk);
  return adm_check_launch("adm_conv");
}

template <int WM, int WN, int TM, int TN, int OCC, int TAPS, int HALO, int KS, bool COLD>
int launch_conv(const ConvK& k, int prologue, int m_tiles, hipStream_t s) {
  switch (prologue) {
    case 0: return launch_conv_p<WM, WN, TM, TN, OCC, TAPS, HALO, 0, KS, COLD>(k, m_tiles, s);
    case 1: return launch_conv_p<WM, WN, TM, TN, OCC, TAPS, HALO, 1, KS, COLD>(k, m_tiles, s);
    case 3:   // GroupNorm-backward epilogue: 3x3 backward-data convs on the 256-pixel 8-wave tiles only
      if constexpr (TAPS == 9 && HALO == 324 && WN == 4 && !COLD) return launch_conv_p<WM, WN, TM, TN, OCC, TAPS, HALO, 3, KS, COLD>(k, m_tiles, s);
      else ADM_FAIL(ADM_E_SHAPE, "adm_conv: prologue 3 (GroupNorm-backward epilogue) needs a 3x3 conv on a map >= 16x16");
    case 4:   // GN + SiLU prologue with the skip_connection folded in as one-tap K-steps: the same tiles
      if constexpr (TAPS == 9 && WN == 4 && !COLD) return launch_conv_p<WM, WN, TM, TN, OCC, TAPS, HALO, 4, KS, COLD>(k, m_tiles, s);
      else ADM_FAIL(ADM_E_SHAPE, "adm_conv: the skip-connection fold needs a 3x3 conv on the 8-wave tiles");
    default: return launch_conv_p<WM, WN, TM, TN, OCC, TAPS, HALO, 2, KS, COLD>(k, m_tiles, s);
  }
}

// tile geometry for a BM-pixel tile: returns false if the map does not tile
bool conv_geometry(ConvK& k, int BM, int taps, int halo_max) {
  k.TW = k.W < 16 ? k.W : 16;
  k.TH = k.H < BM / k.TW ? k.H : BM / k.TW;
  if (k.TH <= 0 || BM % (k.TH * k.TW) != 0 || k.H % k.TH != 0 || k.W % k.TW != 0) return false;
  k.TI = BM / (k.TH * k.TW);
  const int pad = taps == 9 ? 1 : 0;
  if (k.TI > TI_MAX || k.TI * (k.TH + 2 * pad) * (k.TW + 2 * pad) > halo_max) return false;
  k.tiles_x = k.W / k.TW;
  k.tiles_y = k.H / k.TH;
  if ((k.TW & (k.TW - 1)) || (k.TH & (k.TH - 1))) return false;
  k.tw_shift = __builtin_ctz(k.TW);
  k.thw_shift = __builtin_ctz(k.TH * k.TW);
  const unsigned hw2 = (unsigned)(k.TW + 2 * pad), hpi = (unsigned)(k.TH + 2 * pad) * hw2;
  k.rcp_hw2 = ((1u << 20) + hw2 - 1) / hw2;
  k.rcp_hpi = ((1u << 20) + hpi - 1) / hpi;
  return true;
}

template <int WM, int WN, int TM, int TN, int OCC, bool COLD = false>
int dispatch_conv(ConvK& k, int taps, int prologue, hipStream_t s) {
  constexpr int BM = WM * TM * 16;
  // halo capacity: one (TH+2)(TW+2) patch of one image for the 256-pixel tiles (maps >= 16x16); TI whole images
  // of a small map for the 128-pixel tiles (adm_conv sends 8x8 maps there)
  if (taps == 9) {
    if constexpr (BM == 256) {
      if (conv_geometry(k, BM, 9, 324) && k.TI == 1) {
        const int m_tiles = k.N * k.tiles_x * k.tiles_y;
        if constexpr (WN == 4) {   // up-conv phase launches exist for the 8-wave tilings only
          if constexpr (!COLD) {
            if (k.tap_mask != 0x1ff) return launch_conv<WM, WN, TM, TN, OCC, 4, 324, 1, false>(k, prologue, m_tiles, s);
          }
        }
        return launch_conv<WM, WN, TM, TN, OCC, 9, 324, 1, COLD>(k, prologue, m_tiles, s);
      }
    } else {
      if (conv_geometry(k, BM, 9, 200)) {
        const int m_tiles = k.TI == 1 ? k.N * k.tiles_x * k.tiles_y : (k.N + k.TI - 1) / k.TI;
        return launch_conv<WM, WN, TM, TN, OCC, 9, 200, 1, COLD>(k, prologue, m_tiles, s);
      }
    }
  } else {   // 1x1 (COLD: split-K on the 8-wave tiles, fp32 NCHW output on the 16-wide tile)
    if (conv_geometry(k, BM, 1, BM)) {
      const int m_tiles = k.TI == 1 ? k.N * k.tiles_x * k.tiles_y : (k.N + k.TI - 1) / k.TI;
      // 64-channel stages (two K-steps per barrier) when neither source straddles a stage
      if (ADM_CONV_KS2 && k.C0 % 64 == 0 && k.C1 % 64 == 0) return launch_conv<WM, WN, TM, TN, OCC, 1, BM, ADM_CONV_KS2 ? 2 : 1, COLD>(k, prologue, m_tiles, s);
      return launch_conv<WM, WN, TM, TN, OCC, 1, BM, 1, COLD>(k, prologue, m_tiles, s);
    }
  }
  ADM_FAIL(ADM_E_SHAPE, "adm_conv: %dx%d feature map does not tile into %d-pixel patches (need >= 8x8, power of two)",
           k.H, k.W, BM);
}


// tiling variant (output-tile width): 5 = 192, 6 = 128 (8 waves); 3 = 16 (4 waves: output head / stem
// backward); 7 = the 32x32x16 MFMA kernel (explicit only).  Auto: the width that pads Cout least, 192 on a tie
// (measured on MI355X, tools/conv_bench.py: the 8-wave tiles share one halo staging + prologue transform among
// twice as many MFMAs as the retired 4-wave 128 / 96 / 64-wide tilings and beat them wherever those padded less).
int pick_variant(const adm_conv_args* a) {
  if (a->variant != 0) return a->variant;
  if (a->cout <= 16 || a->out_mode == 1) return 3;   // fp32 NCHW output is the heads' format (cout <= 16 in every model): the 16-wide tile carries it
  const int w192 = ((a->cout + 191) / 192) * 192, w128 = ((a->cout + 127) / 128) * 128;
  return w192 <= w128 ? 5 : 6;
}

// 16x16 maps on the 128-pixel tiles (the 8x8 maps' instantiations; a 16x16 map is two tiles of 8 rows): deep convs of small-batch
// callers.  SD v1's 1280-wide projections at 6-latent half batches are 6 pixel tiles of 256 x 10 Cout blocks on 256 CUs with 40-160
// K-steps each; 128-pixel tiles double the tiles and halve the bytes per K-step (SD bench 62.9 -> 63.4 latents/s same-box with the 1x1
// convs).  By shape only (K >= 1280: LSUN-256's 1024-channel qkv at batch 64 measured 0.5 % slower with them) -- and the result does
// not depend on the tile: every output element is the same chain of MFMA accumulations (held bitwise by
// tests/test_hip_switches.py).  ADM_CONV_NO_SMALL1X1: A/B switch.  The 3x3 convs of that level were tried on these tiles in place of
// their split-K schedule: 64.1 -> 61.9 latents/s, not kept.
bool small_tiles_16(const adm_conv_args* a) {
  static const bool no_small1x1 = getenv("ADM_CONV_NO_SMALL1X1") != nullptr;
  // (SD's 640-wide 3x3 convs at 32x32 -- 120 tiles of 256 pixels at 6 latents -- were tried on these tiles too: 65.2 -> 64.3 latents/s, not kept)
  if (a->h != 16 || a->w != 16 || a->cout <= 16 || a->ksplit > 1 || a->out_mode != 0 || (a->variant != 0 && a->variant != 5 && a->variant != 6)) return false;
  return a->taps == 1 && !no_small1x1 && a->c0 + a->c1 >= 1280;
}

// slabs of the fused output statistics: one per 256-pixel tile on maps >= 16x16 (four per 16x16 image on the 128-pixel tiles: two
// tiles x two 64-pixel groups), one per image on 8x8 maps; 0 = not offered for this configuration
int stat_slabs_for(const adm_conv_args* a, int variant) {
  if (a->out_mode != 0 || variant == 7) return 0;
  const int hw = a->h * a->w;
  if (a->up_phase) return (a->h >= 16 && a->w >= 16 && hw % 256 == 0) ? hw / 256 * 4 : 0;   // one slab per (source tile, phase)
  if (hw <= 64) return hw == 64 ? 1 : 0;
  if (a->h < 16 || a->w < 16 || hw % 256 != 0) return 0;
  if (small_tiles_16(a)) return hw / 64;   // 128-pixel tiles x two 64-pixel groups
  return hw / 256;
}

}  // namespace

#ifdef ADM_CONV_TIMING
extern "C" int adm_conv_timing_read(unsigned long long* host, int nblocks) {
  if (nblocks > 16384) nblocks = 16384;
  void* dptr = nullptr;
  if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(adm_conv_timing_buf)) != hipSuccess) return -1;
  if (hipMemcpy(host, dptr, (size_t)nblocks * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -2;
  return (int)hipMemset(dptr, 0, sizeof(unsigned long long) * 16 * 16384);
}
#endif

extern "C" int64_t adm_packed_weight_elems(int cout, int cin, int taps) {
  if (cout <= 0 || cin <= 0 || cin % KC != 0 || (taps != 1 && taps != 9)) return -1;
  return (int64_t)(cin / KC) * taps * ((cout + 15) / 16) * 512;
}

extern "C" int adm_pack_conv_weight(const float* w, adm_bf16* out, int cout, int cin, int taps, void* stream) {
  ADM_REQUIRE(w && out, ADM_E_ARG, "adm_pack_conv_weight: null pointer");
  ADM_REQUIRE(cout > 0 && cin > 0 && cin % KC == 0 && (taps == 1 || taps == 9), ADM_E_SHAPE,
              "adm_pack_conv_weight: cout=%d cin=%d taps=%d unsupported (cin %% 32 == 0, taps 1|9)", cout, cin, taps);
  const int nt16 = (cout + 15) / 16;
  const long long total = (long long)(cin / KC) * taps * nt16 * 512;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, out, cout, cin, taps, nt16);
  return adm_check_launch("adm_pack_conv_weight");
}

extern "C" int64_t adm_packed_weight32_elems(int cout, int cin, int taps) {
  if (cout <= 0 || cin <= 0 || cin % KC != 0 || (taps != 1 && taps != 9)) return -1;
  return (int64_t)(cin / KC) * taps * ((cout + 31) / 32) * 1024;
}

extern "C" int adm_pack_conv_weight32(const float* w, adm_bf16* out, int cout, int cin, int taps, void* stream) {
  ADM_REQUIRE(w && out, ADM_E_ARG, "adm_pack_conv_weight32: null pointer");
  ADM_REQUIRE(cout > 0 && cin > 0 && cin % KC == 0 && (taps == 1 || taps == 9), ADM_E_SHAPE,
              "adm_pack_conv_weight32: cout=%d cin=%d taps=%d unsupported", cout, cin, taps);
  const int nt32 = (cout + 31) / 32;
  const long long total = (long long)(cin / KC) * taps * nt32 * 1024;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, out, cout, cin, taps, nt32);
  return adm_check_launch("adm_pack_conv_weight32");
}

extern "C" int adm_conv_stat_slabs(const adm_conv_args* a) {
  if (!a) return 0;
  if (const int sl = adm_conv1x1_resident_slabs(a)) return sl;
  return stat_slabs_for(a, pick_variant(a));
}

extern "C" int adm_conv_pick_variant(const adm_conv_args* a) {
  if (!a) return 0;
  if (adm_conv1x1_resident_cfg(a, nullptr)) return 10;
  return pick_variant(a);
}

extern "C" int adm_conv(const adm_conv_args* a, void* stream) {
  ADM_REQUIRE(a, ADM_E_ARG, "adm_conv: null args");
  ADM_REQUIRE(a->in0 && a->w_packed && a->bias && a->out, ADM_E_ARG, "adm_conv: null pointer");
  ADM_REQUIRE((a->in1 != nullptr) == (a->c1 > 0), ADM_E_ARG, "adm_conv: in1/c1 mismatch");
  ADM_REQUIRE(a->taps == 1 || a->taps == 9, ADM_E_ARG, "adm_conv: taps must be 1 or 9");
  ADM_REQUIRE(a->prologue >= 0 && a->prologue <= 3, ADM_E_ARG, "adm_conv: prologue must be 0..3");
  ADM_REQUIRE(a->prologue != 3 || (a->res && a->out_stats && a->taps == 9 && a->out_mode == 0 && !a->in_up && !a->res_up &&
                                   a->ksplit <= 1 && !a->up_phase && a->h >= 16 && a->w >= 16 && (a->h * a->w) % 256 == 0 && a->cout % 8 == 0),
              ADM_E_ARG, "adm_conv: prologue 3 (GroupNorm-backward epilogue) needs res (= x), out_stats, a 3x3 conv with bf16 output on a map >= 16x16");
  ADM_REQUIRE(a->out_mode == 0 || a->out_mode == 1, ADM_E_ARG, "adm_conv: out_mode must be 0 or 1");
  ADM_REQUIRE((a->prologue == 0) || (a->aff_a && a->aff_b), ADM_E_ARG, "adm_conv: prologue needs aff_a/aff_b");
  ADM_REQUIRE(a->out_mode == 0 || !a->res, ADM_E_ARG, "adm_conv: residual only with bf16 NHWC output");
  ADM_REQUIRE(a->out_mode == 1 || a->cout % 8 == 0, ADM_E_SHAPE, "adm_conv: bf16 NHWC output needs cout %% 8 == 0");
  ADM_REQUIRE(a->n > 0 && a->h > 0 && a->w > 0 && a->cout > 0, ADM_E_ARG, "adm_conv: bad shape");
  ADM_REQUIRE(a->c0 > 0 && a->c0 % KC == 0 && a->c1 >= 0 && a->c1 % KC == 0, ADM_E_SHAPE,
              "adm_conv: channels (%d | %d) must be multiples of 32", a->c0, a->c1);
  ADM_REQUIRE(adm_aligned16(a->in0) && adm_aligned16(a->in1) && adm_aligned16(a->w_packed) &&
              adm_aligned16(a->aff_a) && adm_aligned16(a->aff_b) && adm_aligned16(a->bias) &&
              adm_aligned16(a->res) && adm_aligned16(a->out), ADM_E_ALIGN, "adm_conv: unaligned pointer");
  ADM_REQUIRE((long long)a->n * a->h * a->w < (1ll << 31) / 4, ADM_E_SHAPE, "adm_conv: too many pixels for 32-bit index");

  // 1x1 with the activation tile resident in LDS across all Cout blocks (adm_conv1x1.hip)
  if (adm_conv1x1_resident_cfg(a, nullptr)) return adm_conv1x1_resident_launch(a, stream);
  ADM_REQUIRE(!a->geglu, ADM_E_SHAPE, "adm_conv: geglu needs a 1x1 conv the resident-tile kernel takes (raw input, no residual / statistics, "
              "(c0 + c1) %% 64 == 0, cout %% 16 == 0, cout > 192, h * w %% 64 == 0, the pixel tile x all input channels within 160 KB of LDS)");
  ADM_REQUIRE(a->variant != 10, ADM_E_SHAPE, "adm_conv: variant 10 (resident-tile 1x1) does not take this shape");

  ConvK k{};
  k.in0 = a->in0; k.in1 = a->in1; k.w = a->w_packed; k.res = a->res;
  k.bias = a->bias; k.aa = a->aff_a; k.ab = a->aff_b; k.out = a->out;
  k.stats = a->out_stats; k.stat_slabs = 0;
  k.N = a->n; k.H = a->h; k.W = a->w; k.C0 = a->c0; k.C1 = a->c1; k.Cout = a->cout;
  k.out_mode = a->out_mode;
  k.out_scale = a->out_scale == 0.f ? 1.f : a->out_scale;
  k.in_up = a->in_up ? 1 : 0;
  k.res_up = a->res_up ? 1 : 0;
  ADM_REQUIRE(!(k.in_up || k.res_up) || (a->taps == 9 && a->out_mode == 0 && a->c1 == 0 && a->h % 2 == 0 && a->w % 2 == 0 && a->h >= 16 && a->w >= 16),
              ADM_E_SHAPE, "adm_conv: in_up / res_up need a 3x3 conv with bf16 output, one input source and an even map >= 16x16");
  ADM_REQUIRE(!k.res_up || a->res, ADM_E_ARG, "adm_conv: res_up without a residual operand");
  k.tap_mask = 0x1ff; k.ooy = 0; k.oox = 0; k.slab_off = 0;
  if (a->up_phase) {
    // phase (py, px) = up_phase - 1: out[2y + py][2x + px] = sum over the source window rows {y - 1 + py, y + py} x columns
    // {x - 1 + px, x + px}: taps (ky, kx) in {py, py + 1} x {px, px + 1} of the 3x3 window around source pixel (y, x)
    ADM_REQUIRE(a->up_phase >= 1 && a->up_phase <= 5, ADM_E_ARG, "adm_conv: up_phase must be 0, 1..4 (one phase) or 5 (all four)");
    ADM_REQUIRE(a->taps == 9 && a->out_mode == 0 && a->c1 == 0 && !a->res && !a->in_up && !a->res_up && a->ksplit <= 1 &&
                a->h >= 16 && a->w >= 16, ADM_E_SHAPE,
                "adm_conv: up_phase needs a 3x3 conv with bf16 output, one source >= 16x16, no residual / split-K / in_up");
    const int py = (a->up_phase - 1) >> 1, px = (a->up_phase - 1) & 1;
    k.tap_mask = 0x1b << (py * 3 + px);   // taps (ky, kx) in {py, py + 1} x {px, px + 1}; != 0x1ff selects the TAPS = 4 instantiation
    k.ooy = py; k.oox = px; k.slab_off = a->up_phase == 5 ? 0 : a->up_phase - 1;
    k.nph = a->up_phase == 5 ? 4 : 1;
  }
  k.ksplit = a->ksplit > 1 ? a->ksplit : 1;
  k.cps = 0;
  k.ws = a->ws;
  if (k.ksplit > 1) {
    const int chunks = (a->c0 + a->c1) / KC;
    ADM_REQUIRE(a->out_mode == 0 && !k.res_up && a->ws, ADM_E_ARG, "adm_conv: ksplit needs bf16 output, no res_up and a workspace");
    ADM_REQUIRE(a->taps == 9 || (ADM_CONV_KS2 == 0 && ADM_CONV_RD == 2), ADM_E_ARG, "adm_conv: this build's 1x1 loop does not split K");
    ADM_REQUIRE(chunks % k.ksplit == 0 && (chunks / k.ksplit) % 2 == 0, ADM_E_SHAPE,
                "adm_conv: ksplit %d does not divide the %d 32-channel chunks into even runs", k.ksplit, chunks);
    ADM_REQUIRE(a->cout % 8 == 0 && a->cout <= 2048 && adm_aligned16(a->ws), ADM_E_SHAPE, "adm_conv: ksplit needs cout %% 8 == 0, cout <= 2048, aligned ws");
    k.cps = chunks / k.ksplit;
  }
  k.ntiles16 = (a->cout + 15) / 16;
  k.wbytes = (unsigned)(((long long)(a->c0 + a->c1) / KC) * a->taps * k.ntiles16 * 1024);
  int prologue = a->prologue;
  if (a->fold0) {
    // ResBlock out_layers conv + skip_connection in one K loop (reference unet.py:216-222, 256: `self.skip_connection(x) + h`)
    ADM_REQUIRE(a->taps == 9 && a->prologue == 2 && a->out_mode == 0 && !a->res && !a->in_up && !a->res_up && !a->up_phase && a->ksplit <= 1 &&
                ((a->h >= 16 && a->w >= 16 && (a->h * a->w) % 256 == 0) || (a->h == 8 && a->w == 8)), ADM_E_ARG,
                "adm_conv: fold0 needs a 3x3 conv with the GroupNorm + SiLU prologue, bf16 output, no residual, an 8x8 map or one >= 16x16");
    ADM_REQUIRE(a->fc0 > 0 && a->fc0 % KC == 0 && a->fc1 >= 0 && a->fc1 % KC == 0 && (a->fold1 != nullptr) == (a->fc1 > 0), ADM_E_SHAPE,
                "adm_conv: fold channels (%d | %d) must be multiples of 32", a->fc0, a->fc1);
    ADM_REQUIRE(adm_aligned16(a->fold0) && adm_aligned16(a->fold1), ADM_E_ALIGN, "adm_conv: unaligned fold pointer");
    k.f0 = a->fold0; k.f1 = a->fold1; k.FC0 = a->fc0; k.FC1 = a->fc1;
    k.wbytes += (unsigned)(((long long)(a->fc0 + a->fc1) / KC) * k.ntiles16 * 1024);   // the 1x1 weights follow the 3x3 weights
    prologue = 4;
  }
  if (a->up_phase == 5) {   // four packed weights back to back
    ADM_REQUIRE((long long)k.wbytes * 4 < (1ll << 31), ADM_E_SHAPE, "adm_conv: up_phase 5: weights beyond 2 GiB");
    k.phase_bytes = k.wbytes;
    k.wbytes *= 4;
  }
  hipStream_t s = (hipStream_t)stream;

  const int variant = pick_variant(a);
  if (a->out_stats) {
    k.stat_slabs = stat_slabs_for(a, variant);
    ADM_REQUIRE(k.stat_slabs > 0, ADM_E_SHAPE, "adm_conv: fused output statistics are not offered for this shape / variant");
  }
  if (k.ksplit > 1) {
    ADM_REQUIRE(variant == 5 || variant == 6, ADM_E_ARG, "adm_conv: ksplit needs tiling variant 5 or 6");
    k.res = nullptr;      // bias, residual and statistics belong to the reduce pass
    k.stats = nullptr;
  }
  ADM_REQUIRE(!(k.in_up || k.res_up || a->up_phase) || variant == 5 || variant == 6 || (variant == 8 && !a->up_phase), ADM_E_ARG, "adm_conv: in_up / res_up / up_phase need tiling variant 5 or 6");
  if (variant == 7) {  // 32x32x16 MFMA kernel: 3x3, maps >= 16x16, 256-pixel x 192-channel tile
    ADM_REQUIRE(a->w_packed32 && a->taps == 9 && a->out_mode == 0 && !a->fold0, ADM_E_ARG, "adm_conv: variant 7 needs w_packed32, 3x3, bf16 out, no fold");
    ADM_REQUIRE(conv_geometry(k, 256, 9, 324) && k.TI == 1, ADM_E_SHAPE, "adm_conv: variant 7 needs maps >= 16x16");
    k.w = a->w_packed32;
    k.wbytes = (unsigned)(((long long)(a->c0 + a->c1) / KC) * 9 * ((a->cout + 31) / 32) * 2048);
    const int m_tiles = k.N * k.tiles_x * k.tiles_y;
    switch (a->prologue) {
      case 0: return launch_conv32<0>(k, m_tiles, s);
      case 1: return launch_conv32<1>(k, m_tiles, s);
      default: return launch_conv32<2>(k, m_tiles, s);
    }
  }
  // 8x8 maps use 128-pixel tiles (2 images): halves the halo so that 2 blocks still fit per CU; so do the deep convs of
  // small_tiles_16 on 16x16 maps
  const bool small_map = a->h * a->w <= 64 || small_tiles_16(a);
  if (variant == 8) {
    // structural experiment (explicit only): the 256-pixel x 192-channel tile on FOUR waves, one per SIMD, each 128 pixels x
    // 96 channels (8 x 6 MFMA tiles = 192 accumulator registers of a 512-register wave): half the LDS fragment reads and half
    // the weight-fragment loads per MFMA of the 8-wave tile, and registers to spare for the tile switch
    ADM_REQUIRE(a->taps == 9 && !small_map && a->out_mode == 0 && k.ksplit <= 1 && !a->up_phase && a->prologue != 3 && !a->fold0, ADM_E_ARG,
                "adm_conv: variant 8 takes 3x3 convs with bf16 output on maps >= 16x16 (no fold)");
    ADM_REQUIRE(conv_geometry(k, 256, 9, 324) && k.TI == 1 && k.TW == 16 && k.TH == 16, ADM_E_SHAPE, "adm_conv: variant 8 needs maps >= 16x16");
    return launch_conv<2, 2, 8, 6, 1, 9, 324, 1, false>(k, a->prologue, k.N * k.tiles_x * k.tiles_y, s);
  }
  if (k.ksplit > 1) {
    int rc;
    if (variant == 5) rc = small_map ? dispatch_conv<2, 4, 4, 3, 2, true>(k, a->taps, a->prologue, s) : dispatch_conv<2, 4, 8, 3, 2, true>(k, a->taps, a->prologue, s);
    else rc = small_map ? dispatch_conv<2, 4, 4, 2, 2, true>(k, a->taps, a->prologue, s) : dispatch_conv<2, 4, 8, 2, 2, true>(k, a->taps, a->prologue, s);
    if (rc != 0) return rc;
    const int hw = a->h * a->w;
    const int slabs = a->out_stats ? stat_slabs_for(a, variant) : (hw >= 1024 ? hw / 256 : 1);
    hipLaunchKernelGGL(conv_splitk_reduce, dim3((a->cout + 63) / 64, slabs, a->n), dim3(256), 0, s, (const float*)a->ws, k.ksplit, a->bias,
                       a->res, reinterpret_cast<uint16_t*>(a->out), a->out_stats, a->n, hw, a->cout, slabs);
    return adm_check_launch("adm_conv(split-K reduce)");
  }
  ADM_REQUIRE(a->out_mode == 0 || variant == 3, ADM_E_ARG, "adm_conv: fp32 NCHW output runs on tiling variant 3 (the 16-wide tile)");
  ADM_REQUIRE(!a->fold0 || variant == 5 || variant == 6, ADM_E_ARG, "adm_conv: the skip-connection fold runs on tiling variants 5 / 6");
  switch (variant) {
    case 3: return dispatch_conv<4, 1, 4, 1, 2, true>(k, a->taps, a->prologue, s);  // 256 x 16 (output head / stem backward)
    // 8 waves, 192-wide tile: the prologue transform / halo staging is shared by twice as many MFMAs
    case 5: return small_map ? dispatch_conv<2, 4, 4, 3, 2>(k, a->taps, prologue, s) : dispatch_conv<2, 4, 8, 3, 2>(k, a->taps, prologue, s);
    // 8 waves, 128-wide tile (channel counts that are multiples of 128 but not of 192: the classifier)
    case 6: return small_map ? dispatch_conv<2, 4, 4, 2, 2>(k, a->taps, prologue, s) : dispatch_conv<2, 4, 8, 2, 2>(k, a->taps, prologue, s);
    default: ADM_FAIL(ADM_E_ARG, "adm_conv: unknown variant %d", variant);
  }
}
