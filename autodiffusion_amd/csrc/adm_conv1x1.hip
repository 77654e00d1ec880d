// 1x1 convolutions with the activation tile RESIDENT in LDS (gfx950).  Part of adm_conv (adm_conv.hip dispatches
// here); same operands, weight packing and epilogue contract.
//
// Replaces the reference's Conv1d qkv / proj_out of AttentionBlock._forward (guided_diffusion/unet.py:299-305) and
// the 1x1 skip_connection of ResBlock (unet.py:225-231, 256) -- with the preceding GroupNorm folded in as a
// per-(image, channel) affine, the residual add and the next GroupNorm's partial sums in the epilogue.
//
// Why a second kernel: in the staged kernel (conv_kernel<..., TAPS = 1>) a 32-channel K-step is one barrier interval
// that has to fetch, transform and park 16 KB of new activations for only 24 MFMAs per wave -- ablations on MI355X
// show the GN-prologue loop 43 % shorter without the transform + park and the raw loop 48 % shorter without the
// activation loads -- and every Cout block repeats that work.  Here a block owns BM pixels x ALL input channels:
// the tile is fetched and transformed ONCE, stays in LDS (row = 2K + 32 bytes: conflict-free ds_read_b128
// fragments for K % 64 == 0), and the block then walks every 384-wide Cout block with a K loop that has no
// barrier, no global activation traffic and no VALU work: per K-step and wave 8 (BM = 128) LDS fragment reads,
// 3 weight-fragment loads (fragment-ordered, L2-resident, register ring 3 K-steps ahead) and 24 MFMAs.
//   waves      8 = 1 x 8: a wave owns all BM pixels x 48 output channels (TN = 3), the block BM x 384
//   epilogue   wave-private and barrier-free: accumulators (started from the bias) -> bf16 -> the wave's own 4 KB of
//              LDS -> 16-byte row-segment stores with the residual and the output's GroupNorm partial sums; with
//              no barrier in the Cout loop the waves drift apart and one wave's stores overlap another's MFMAs
//   launch     persistent over pixel tiles (flattened N*H*W; BM <= H*W, so a tile lies in one image)
#include <stdlib.h>

#include <type_traits>

#include "adm_common.h"
#include "adm_conv_internal.h"

#ifndef ADM_C1_ABL
#define ADM_C1_ABL 0   // diagnostic builds only: drop one phase of the tile (results are then wrong) to price it
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned OOB = 0x80000000u;

struct Conv1K {
  const uint16_t* in0; const uint16_t* in1; const uint16_t* w; const uint16_t* res;
  const float* bias; const float* aa; const float* ab; uint16_t* out; float* stats;
  int N, HW, C0, C1, Cout, ntiles16, stat_slabs, m_tiles, nblocks_n;
  int csplit, nbper;   // small launches: the Cout blocks of a pixel tile are shared out over `csplit` consecutive tiles of the list, nbper each
  unsigned wbytes, rcp_seg;   // rcp_seg = ceil(2^32 / (K/8)): idx / (K/8) == umulhi(idx, rcp_seg)
};

__device__ __forceinline__ uint4 bufload16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

template <int BM>
constexpr int c1_stg_bytes() {
  // the larger of: block-epilogue staging (BM rows x 400 B), its statistics reduction ([21][192][2] floats), and the
  // eight wave-private 4 KB regions of the barrier-free epilogue
  constexpr int a = BM * (192 * 2 + 16), b = (512 / 24) * 192 * 8, c = 8 * 4096;
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}
template <int BM>
constexpr int c1_lds_bytes(int k) { return BM * (2 * k + 32) + c1_stg_bytes<BM>() + 2 * k * 4; }

// WP: wave-private barrier-free epilogue (outputs without a residual operand) or the block-wide two-half epilogue
// WM: 1 = waves 1 x 8 (block BM x 384 channels, a wave owns all BM pixels); 2 = waves 2 x 4 (block BM x 192, a wave
//     owns half the pixels: outputs no wider than 192 channels)
// GG: GEGLU epilogue (adm_conv_args.geglu; WP, WM = 1 only): the projection's output channels arrive INTERLEAVED (row 2m = value m,
//     row 2m + 1 = gate m: the host packs the weight and the bias that way), so a lane's four consecutive accumulator channels are
//     two (value, gate) pairs and out[pixel][m] = value_m * gelu(gate_m) is formed in registers from the fp32 accumulators: the
//     [pixels][2 * inner] tensor is never written (SD: ldm/modules/attention.py:37-44, GEGLU.forward)
template <int BM, int PRO, bool WP, int WM, bool GG = false>
__global__ void __launch_bounds__(512, 2)
conv1x1r_kernel(const Conv1K p) {
  constexpr int NT = 512, WN = 8 / WM, BN = WN * 48, TM = BM / 16 / WM, TN = 3, BNH = 192;   // BNH: channels per block-epilogue half (4 waves x 48)
  constexpr int EROW = BNH * 2 + 16, SEGS = BNH / 8, PR = NT / SEGS, NIT = (BM + PR - 1) / PR;
  constexpr int WROW = 112, WST = 4096;   // wave-private restage: 32 pixels x (96 + 16 pad) bytes, in 4 KB per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = p.C0 + p.C1, RB = 2 * K + 32, ksteps = K / 32, segk = K / 8;
  unsigned char* const A = smem;                              // [BM][RB]
  unsigned char* const stg = smem + BM * RB;                  // 8 x WST: wave-private output restage / statistics
  float* const tab = reinterpret_cast<float*>(stg + c1_stg_bytes<BM>());  // a[K] | b[K] of the tile's image

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, lq = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.Cout * 4, 0x00020000);
  const unsigned wstep = (unsigned)p.ntiles16 * 1024u;

  // XCD-aware static tile list (see conv_kernel): XCD x owns a contiguous run of pixel tiles
  const int xcd = blockIdx.x & 7;
  const int gx = (int)(gridDim.x >> 3) + (xcd < (int)(gridDim.x & 7));
  const int tcount = (p.m_tiles >> 3) + (xcd < (p.m_tiles & 7));
  const int tstart = xcd * (p.m_tiles >> 3) + min(xcd, p.m_tiles & 7);

  for (int tl = blockIdx.x >> 3; tl < tcount; tl += gx) {
    const int tq = tstart + tl;
    const int pt = p.csplit > 1 ? tq / p.csplit : tq;
    const int nb0 = p.csplit > 1 ? (tq - pt * p.csplit) * p.nbper : 0;
    const int nb1 = p.csplit > 1 ? min(p.nblocks_n, nb0 + p.nbper) : p.nblocks_n;
    const int pb = pt * BM;                      // first pixel of the tile in the flattened [N*H*W] order
    const int img = pb / p.HW;
    __syncthreads();                             // the previous tile's readers of A / tab / stg are done
    // ---- affine table of this image, then the activation tile: fetched, transformed, parked -- once
    if constexpr (PRO != 0) {
      for (int i = tid; i < 2 * K / 4; i += NT) {
        const int half = i >= K / 4, j = half ? i - K / 4 : i;
        *reinterpret_cast<float4*>(tab + half * K + j * 4) =
            *reinterpret_cast<const float4*>((half ? p.ab : p.aa) + (long long)img * K + j * 4);
      }
      __syncthreads();
    }
    for (int i0 = 0; i0 < (ADM_C1_ABL == 3 ? 0 : BM * segk); i0 += 4 * NT) {
      uint4 v[4];
      int px[4], sg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + u * NT + tid;
        px[u] = (int)__umulhi((unsigned)idx, p.rcp_seg);
        sg[u] = idx - px[u] * segk;
        v[u] = make_uint4(0, 0, 0, 0);
        if (px[u] < BM) {
          const int ch = sg[u] * 8;
          const long long pix = (long long)pb + px[u];
          v[u] = ch < p.C0 ? *reinterpret_cast<const uint4*>(p.in0 + pix * p.C0 + ch)
                           : *reinterpret_cast<const uint4*>(p.in1 + pix * p.C1 + (ch - p.C0));
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (px[u] >= BM) continue;
        uint4 o = v[u];
        if constexpr (PRO != 0) {
          const float* ta = tab + sg[u] * 8;
          float a8[8], b8[8];
          *reinterpret_cast<float4*>(a8) = *reinterpret_cast<const float4*>(ta);
          *reinterpret_cast<float4*>(a8 + 4) = *reinterpret_cast<const float4*>(ta + 4);
          *reinterpret_cast<float4*>(b8) = *reinterpret_cast<const float4*>(ta + K);
          *reinterpret_cast<float4*>(b8 + 4) = *reinterpret_cast<const float4*>(ta + K + 4);
          uint32_t w4[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float lo = adm_lo_f32(w4[j]), hi = adm_hi_f32(w4[j]);
            lo = a8[2 * j] * lo + b8[2 * j];
            hi = a8[2 * j + 1] * hi + b8[2 * j + 1];
            if constexpr (PRO == 2) { lo = adm_silu(lo); hi = adm_silu(hi); }
            w4[j] = adm_pack2(lo, hi);
          }
          o = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
        *reinterpret_cast<uint4*>(A + px[u] * RB + sg[u] * 16) = o;
      }
    }
    __syncthreads();

    const unsigned char* const alane = A + (wm * TM * 16 + lc) * RB + lq * 16;  // fragment row of this lane in its first pixel tile
    for (int nb = nb0; nb < nb1; ++nb) {
      // ---- K loop over the resident tile: no barrier, no global activation traffic
      unsigned wofs[TN];
      f32x4 acc[TM][TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int t16 = nb * (BN / 16) + wn * TN + j;
        wofs[j] = t16 < p.ntiles16 ? (unsigned)((t16 * 64 + lane) * 16) : OOB;
        const int bch = t16 * 16 + lq * 4;
        const uint4 b = bufload16(rsb, bch + 3 < p.Cout ? (unsigned)bch * 4u : OOB, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[i][j] = f32x4{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
      }
      auto load_w = [&](int step, uint4 (&dst)[TN]) {
#pragma unroll
        for (int j = 0; j < TN; ++j) dst[j] = bufload16(rsw, wofs[j], (unsigned)step * wstep);
      };
      // LDS fragment reads run AFD pixel tiles ahead of the MFMAs that consume them, pinned in the emitted code
      // (hipcc otherwise issues all reads, waits lgkmcnt(0), then the MFMAs: the read latency opens every K-step)
      auto kstep = [&](int ks, const uint4 (&w)[TN]) {
        constexpr int AFD = TM < 4 ? TM : 4;
        const unsigned char* ak = alane + ks * 64;
        __builtin_amdgcn_sched_barrier(0);
        adm_h8 af[TM];
#pragma unroll
        for (int i = 0; i < AFD; ++i) af[i] = *reinterpret_cast<const adm_h8*>(ak + i * 16 * RB);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (i + AFD < TM) af[i + AFD] = *reinterpret_cast<const adm_h8*>(ak + (i + AFD) * 16 * RB);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = adm_mfma_16x16x32(__builtin_bit_cast(adm_h8, w[j]), af[i], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < AFD; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (i + AFD < TM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);
        }
      };
      uint4 wr[4][TN];
      const int last = ksteps - 1;
      load_w(0, wr[0]);
      load_w(min(1, last), wr[1]);
      load_w(min(2, last), wr[2]);
      int k0 = 0;
      if (ADM_C1_ABL == 4) k0 = last + 1;
      for (; k0 + 3 <= last; k0 += 4) {  // whole groups of 4: no early exit inside the unrolled group
        load_w(min(k0 + 3, last), wr[3]); kstep(k0, wr[0]);
        load_w(min(k0 + 4, last), wr[0]); kstep(k0 + 1, wr[1]);
        load_w(min(k0 + 5, last), wr[1]); kstep(k0 + 2, wr[2]);
        load_w(min(k0 + 6, last), wr[2]); kstep(k0 + 3, wr[3]);
      }
      if (k0 <= last) kstep(k0, wr[0]);
      if (k0 + 1 <= last) kstep(k0 + 1, wr[1]);
      if (k0 + 2 <= last) kstep(k0 + 2, wr[2]);

      if constexpr (GG) {
      // ---- GEGLU epilogue, wave-private and barrier-free like the one below: 48 accumulator channels = 24 output channels per wave;
      // restaged 32 pixels at a time as 48-byte rows (64-byte pitch) in the wave's private LDS, stored as 16-byte segments
      const int cbw = nb * BN + wn * 48;
      if (cbw < p.Cout) {                               // wave-uniform
        unsigned char* const wst = stg + wave * WST;
        const int Co = p.Cout >> 1, cbo = cbw >> 1;
        const int sgl = lane % 3, rowl = lane / 3;      // lanes 0..62: segment sgl of pixel row rowl (+21)
        const bool lact = lane < 63 && cbo + sgl * 8 < Co;
#pragma unroll
        for (int c = 0; c < TM / 2; ++c) {              // 32 pixels per round
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const f32x4 v = acc[2 * c + ii][j];       // (value, gate, value, gate)
              const float y0 = v[0] * (0.5f * v[1] * (1.0f + erff(v[1] * 0.70710678118654752f)));
              const float y1 = v[2] * (0.5f * v[3] * (1.0f + erff(v[3] * 0.70710678118654752f)));
              *reinterpret_cast<uint32_t*>(wst + (ii * 16 + lc) * 64 + (j * 8 + lq * 2) * 2) = adm_pack2(y0, y1);
            }
#pragma unroll
          for (int q = 0; q < 2; ++q) {                 // rows rowl, +21 (< 32)
            const int r = rowl + q * 21;
            if (!lact || r >= 32) continue;
            const long long eo = ((long long)pb + c * 32 + r) * Co + cbo + sgl * 8;
            *reinterpret_cast<uint4*>(p.out + eo) = *reinterpret_cast<const uint4*>(wst + r * 64 + sgl * 16);
          }
        }
      }
      } else if constexpr (WP) {
      // ---- epilogue, WAVE-PRIVATE and barrier-free: a wave owns all BM pixels of its 48 channels, so it restages
      // its own accumulators (32 pixels at a time, bf16, 112-byte rows in its private 4 KB of LDS), reads them back
      // as 16-byte row segments (6 per pixel, 10 pixels per wave-instruction) and stores them, adding the residual
      // and accumulating the output's GroupNorm sums on the way.  LDS instructions of one wave execute in order, so
      // the restage needs no barrier -- and with none anywhere in the Cout loop the 8 waves drift apart: one wave's
      // stores overlap another's MFMAs (the layer is output-write bound at Cin = 384).
      const int cbw = nb * BN + wn * 48;                // first output channel of this wave
      if (cbw < p.Cout) {                               // wave-uniform
        unsigned char* const wst = stg + wave * WST;
        const int sgl = lane % 6, rowl = lane / 6;      // lanes 0..59: segment sgl of pixel row rowl (+10 per step)
        const bool lact = lane < 60 && cbw + sgl * 8 < p.Cout;
        f32x2 s1[4], s2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[q] = f32x2{0.f, 0.f}; s2[q] = f32x2{0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < TM / 2; ++c) {              // 32 pixels per round
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const f32x4 v = acc[2 * c + ii][j];
              uint2 o;
              o.x = adm_pack2(v[0], v[1]);
              o.y = adm_pack2(v[2], v[3]);
              *reinterpret_cast<uint2*>(wst + (ii * 16 + lc) * WROW + (j * 16 + lq * 4) * 2) = o;
            }
#pragma unroll
          for (int q = 0; q < 4; ++q) {                 // rows rowl, +10, +20, +30 (< 32)
            const int r = rowl + q * 10;
            if (!lact || r >= 32) continue;
            const long long eo = ((long long)pb + wm * TM * 16 + c * 32 + r) * p.Cout + cbw + sgl * 8;
            uint4 v = *reinterpret_cast<const uint4*>(wst + r * WROW + sgl * 16);
            uint32_t a4[4] = {v.x, v.y, v.z, v.w};
            if (p.res && ADM_C1_ABL != 1) {
              const uint4 rr = *reinterpret_cast<const uint4*>(p.res + eo);
              const uint32_t r4[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const f32x2 t = f32x2{adm_lo_f32(a4[e]), adm_hi_f32(a4[e])} +
                                f32x2{adm_lo_f32(r4[e]), adm_hi_f32(r4[e])};
                a4[e] = adm_pack2(t.x, t.y);
              }
              v = make_uint4(a4[0], a4[1], a4[2], a4[3]);
            }
            if (ADM_C1_ABL != 2 || v.x == 0x12345678u) *reinterpret_cast<uint4*>(p.out + eo) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x2 t = f32x2{adm_lo_f32(a4[e]), adm_hi_f32(a4[e])};
              s1[e] += t;
              s2[e] = __builtin_elementwise_fma(t, t, s2[e]);
            }
          }
        }
        if (p.stats) {
          // the 10 row-lanes of every channel segment, summed through the wave's private LDS in a fixed order
          float* red = reinterpret_cast<float*>(wst);   // [10 rows][48 channels][2]
          if (lane < 60) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              red[(rowl * 48 + sgl * 8 + e) * 2 + 0] = s1[e >> 1][e & 1];
              red[(rowl * 48 + sgl * 8 + e) * 2 + 1] = s2[e >> 1][e & 1];
            }
          }
          if (lane < 48 && cbw + lane < p.Cout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int q = 0; q < 10; ++q) { t1 += red[(q * 48 + lane) * 2]; t2 += red[(q * 48 + lane) * 2 + 1]; }
            const int slab = ((pb - img * p.HW) / BM) * WM + wm;   // one slab per (tile, pixel half)
            float* dst = p.stats + (((long long)img * p.stat_slabs + slab) * p.Cout + cbw + lane) * 2;
            dst[0] = t1;
            dst[1] = t2;
          }
        }
      }
      } else {
      // ---- epilogue, two halves of 192 channels (waves 0-3, then 4-7)
      const int sgo = tid % SEGS, prow = tid / SEGS;
#pragma unroll 1
      for (int hf = 0; hf < BN / BNH; ++hf) {
        const int cb = nb * BN + hf * BNH;              // first output channel of this half
        if (cb >= p.Cout) break;                        // block-uniform
        __syncthreads();                                // staging buffer free
        if (WM == 2 || (wave >> 2) == hf) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int ch0 = ((wn & 3) * TN + j) * 16 + lq * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              uint2 o;
              o.x = adm_pack2(acc[i][j][0], acc[i][j][1]);
              o.y = adm_pack2(acc[i][j][2], acc[i][j][3]);
              *reinterpret_cast<uint2*>(stg + ((wm * TM + i) * 16 + lc) * EROW + ch0 * 2) = o;
            }
          }
        }
        const int gch = cb + sgo * 8;
        const bool act = prow < PR && gch < p.Cout;
        uint4 rr[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int m = prow + k * PR;
          rr[k] = make_uint4(0, 0, 0, 0);
          if (p.res && act && m < BM && ADM_C1_ABL != 1) rr[k] = *reinterpret_cast<const uint4*>(p.res + ((long long)pb + m) * p.Cout + gch);
        }
        __syncthreads();
        f32x2 s1[4], s2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[q] = f32x2{0.f, 0.f}; s2[q] = f32x2{0.f, 0.f}; }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int m = prow + k * PR;
          if (!act || m >= BM) continue;
          uint4 v = *reinterpret_cast<const uint4*>(stg + m * EROW + sgo * 16);
          uint32_t a4[4] = {v.x, v.y, v.z, v.w};
          if (p.res) {
            const uint32_t r4[4] = {rr[k].x, rr[k].y, rr[k].z, rr[k].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x2 t = f32x2{adm_lo_f32(a4[q]), adm_hi_f32(a4[q])} +
                              f32x2{adm_lo_f32(r4[q]), adm_hi_f32(r4[q])};
              a4[q] = adm_pack2(t.x, t.y);
            }
            v = make_uint4(a4[0], a4[1], a4[2], a4[3]);
          }
          if (ADM_C1_ABL != 2 || v.x == 0x12345678u) *reinterpret_cast<uint4*>(p.out + ((long long)pb + m) * p.Cout + gch) = v;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 t = f32x2{adm_lo_f32(a4[q]), adm_hi_f32(a4[q])};
            s1[q] += t;
            s2[q] = __builtin_elementwise_fma(t, t, s2[q]);
          }
        }
        if (p.stats) {
          // reduce the PR row-partials of every channel through LDS (fixed order: bitwise reproducible)
          float* red = reinterpret_cast<float*>(stg);  // [PR][BNH][2]
          __syncthreads();
          if (prow < PR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              red[(prow * BNH + sgo * 8 + e) * 2 + 0] = s1[e >> 1][e & 1];
              red[(prow * BNH + sgo * 8 + e) * 2 + 1] = s2[e >> 1][e & 1];
            }
          }
          __syncthreads();
          if (tid < BNH && cb + tid < p.Cout) {
            float t1 = 0.f, t2 = 0.f;
            for (int q = 0; q < PR; ++q) { t1 += red[(q * BNH + tid) * 2]; t2 += red[(q * BNH + tid) * 2 + 1]; }
            const int slab = ((pb - img * p.HW) / BM) * WM;   // WM slabs per tile: the block-wide sum goes to the first
            float* dst = p.stats + (((long long)img * p.stat_slabs + slab) * p.Cout + cb + tid) * 2;
            dst[0] = t1;
            dst[1] = t2;
            if (WM == 2) { dst[2 * p.Cout] = 0.f; dst[2 * p.Cout + 1] = 0.f; }
          }
        }
      }
      }
    }
  }
}

template <int BM, int PRO, bool WP, int WM, bool GG = false>
int launch_c1(const Conv1K& k, int smem, hipStream_t s) {
  static int slots_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& slots = slots_dev[dev & 63];
  const void* fn = reinterpret_cast<const void*>(&conv1x1r_kernel<BM, PRO, WP, WM, GG>);
  if (slots == 0) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) ADM_FAIL((int)e, "adm_conv (1x1 resident): hipFuncSetAttribute: %s", hipGetErrorString(e));
    int ncu = 0;
    e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || ncu <= 0) ADM_FAIL((int)e, "adm_conv: hipDeviceGetAttribute: %s", hipGetErrorString(e));
    slots = ncu;  // one block per CU: the resident tile takes most of the LDS
  }
  const int mine = adm_stream_cus((void*)s, slots);   // a CU-masked stream: one block per CU it owns
  const unsigned blocks = (unsigned)(k.m_tiles < mine ? k.m_tiles : mine);
  hipLaunchKernelGGL((conv1x1r_kernel<BM, PRO, WP, WM, GG>), dim3(blocks), dim3(512), smem, s, k);
  return adm_check_launch("adm_conv");
}

}  // namespace

// tile configuration if these arguments can run on the resident-tile kernel: pixels per tile (128 or 64) and wave
// rows (1: 384-wide Cout blocks; 2: 192-wide, for narrow outputs); returns 0 otherwise
int adm_conv1x1_resident_cfg(const adm_conv_args* a, int* wm_out) {
  static const bool disabled = getenv("ADM_CONV_NO_RESIDENT") != nullptr;  // A/B switch for measurements
  if (disabled && a->variant == 0 && !a->geglu) return 0;   // the GEGLU epilogue exists on this kernel only: the switch leaves it here
  if (a->taps != 1 || a->out_mode != 0 || (a->variant != 0 && a->variant != 10) || a->ksplit > 1) return 0;   // split-K: the staged kernel
  const int k = a->c0 + a->c1, hw = a->h * a->w;
  if (k % 64 != 0 || a->c0 % 8 != 0 || a->c1 % 8 != 0 || a->cout % 8 != 0 || hw % 64 != 0) return 0;
  // narrow outputs: the 2 x 4 wave layout measured no faster than the staged kernel (its K loop is weight-load bound
  // at 12 MFMAs per 3 fragment loads), so the automatic choice leaves them there; variant 10 still takes them
  if (a->variant == 0 && a->cout < 256) return 0;
  if (a->geglu && (a->res || a->out_stats || a->prologue != 0 || a->cout % 16 != 0 || a->cout <= 192)) return 0;
  if (wm_out) *wm_out = a->cout <= 192 ? 2 : 1;
  const int lim = 160 * 1024;
  if (hw % 128 == 0 && c1_lds_bytes<128>(k) <= lim) return 128;
  if (c1_lds_bytes<64>(k) <= lim) return 64;
  // Deep projections of the 16x16 / 8x8 levels (K >= 1280: SD v1's 1280-wide transformer blocks) on 32-PIXEL resident tiles: the
  // [32][K] tile is fetched once by all 512 threads and the K loop runs from LDS with the weights three K-steps ahead, where the
  // staged kernel streams the K loop chunk by chunk through registers and a barrier.  Built, parity-tested (variant 10 takes these
  // shapes explicitly) and MEASURED (round 4, profiles/r04/ab_sd_bm32.log): alone on the chip it wins where a grid entry walks ONE
  // 384-wide Cout block (tools/c1_bench.py SHAPESET=sd at 6 latents: 1280 -> 1280 + residual 16.4 vs 18.7 us, with a GroupNorm prologue
  // 15.6 vs 26.8 us) and loses on wide outputs (4 x the weight traffic of a 128-pixel tile: qkv 1280 -> 3840 35.8 vs 23.5 us) -- and
  // inside the real evaluation, two half batches on two streams, even the narrow-output rule is 0.6 % SLOWER (65.7 vs 66.2 latents/s):
  // its L2 weight streaming lands on the other stream's kernels.  So the automatic choice leaves these shapes on the staged kernel;
  // ADM_C1_BM32=1 opts in (outputs <= 1536 wide), by shape only.
  static const bool bm32 = getenv("ADM_C1_BM32") != nullptr;
  if (k >= 1280 && hw <= 256 && hw % 32 == 0 && a->cout > 192 && (a->variant == 10 || (bm32 && a->cout <= 1536)) && c1_lds_bytes<32>(k) <= lim) return 32;
  return 0;
}

int adm_conv1x1_resident_slabs(const adm_conv_args* a) {
  int wm = 1;
  const int bm = adm_conv1x1_resident_cfg(a, &wm);
  return bm ? a->h * a->w / bm * wm : 0;
}

template <int BM, int WM>
static int launch_c1_cfg(const adm_conv_args* a, const Conv1K& k, int smem, hipStream_t s) {
  const bool wp = a->res == nullptr;
  switch (a->prologue) {
    case 0: return wp ? launch_c1<BM, 0, true, WM>(k, smem, s) : launch_c1<BM, 0, false, WM>(k, smem, s);
    case 1: return wp ? launch_c1<BM, 1, true, WM>(k, smem, s) : launch_c1<BM, 1, false, WM>(k, smem, s);
    default: return wp ? launch_c1<BM, 2, true, WM>(k, smem, s) : launch_c1<BM, 2, false, WM>(k, smem, s);
  }
}

int adm_conv1x1_resident_launch(const adm_conv_args* a, void* stream) {
  int wm = 1;
  const int bm = adm_conv1x1_resident_cfg(a, &wm);
  Conv1K k{};
  k.in0 = a->in0; k.in1 = a->in1; k.w = a->w_packed; k.res = a->res;
  k.bias = a->bias; k.aa = a->aff_a; k.ab = a->aff_b; k.out = (uint16_t*)a->out; k.stats = a->out_stats;
  k.N = a->n; k.HW = a->h * a->w; k.C0 = a->c0; k.C1 = a->c1; k.Cout = a->cout;
  k.ntiles16 = (a->cout + 15) / 16;
  k.stat_slabs = k.HW / bm * wm;
  k.m_tiles = (int)((long long)a->n * k.HW / bm);
  const int bn = wm == 1 ? 384 : 192;
  k.nblocks_n = (a->cout + bn - 1) / bn;
  // Few pixel tiles (small-batch callers: SD v1 at 6-latent half batches has 96 tiles at 32x32) and several Cout blocks per tile: the
  // blocks of a tile are shared out over `csplit` neighbouring entries of the tile list (the same XCD: the second fetch of the
  // activation tile is an L2 hit), so that the launch covers the CUs.  This depends on the batch -- and cannot change a result:
  // every output element is computed by the same instructions on the same operands whichever block of the grid owns it.
  k.csplit = 1;
  {
    static const bool no_csplit = getenv("ADM_C1_NO_CSPLIT") != nullptr;   // A/B switch for measurements
    static int ncu_dev[64] = {};        // CUs per device, queried once
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& ncu_c = ncu_dev[dev & 63];
    if (ncu_c == 0 && (hipDeviceGetAttribute(&ncu_c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu_c <= 0)) ncu_c = 256;
    const int ncu = ncu_c;
    if (!no_csplit && k.nblocks_n > 1 && k.m_tiles < ncu) {
      int want = (ncu + k.m_tiles - 1) / k.m_tiles;
      if (want > k.nblocks_n) want = k.nblocks_n;
      if (want > 4) want = 4;
      k.csplit = want;
    }
  }
  k.nbper = (k.nblocks_n + k.csplit - 1) / k.csplit;
  k.csplit = (k.nblocks_n + k.nbper - 1) / k.nbper;     // no empty parts
  k.m_tiles *= k.csplit;
  const int kk = a->c0 + a->c1;
  k.wbytes = (unsigned)(((long long)kk / 32) * k.ntiles16 * 1024);
  const unsigned segk = (unsigned)kk / 8;
  k.rcp_seg = (unsigned)(((1ull << 32) + segk - 1) / segk);
  hipStream_t s = (hipStream_t)stream;
  const int smem = bm == 128 ? c1_lds_bytes<128>(kk) : (bm == 64 ? c1_lds_bytes<64>(kk) : c1_lds_bytes<32>(kk));
  if (a->geglu)   // raw input, no residual, no statistics, 384-wide Cout blocks (checked by adm_conv1x1_resident_cfg)
    return bm == 128 ? launch_c1<128, 0, true, 1, true>(k, smem, s)
                     : (bm == 64 ? launch_c1<64, 0, true, 1, true>(k, smem, s) : launch_c1<32, 0, true, 1, true>(k, smem, s));
  if (bm == 128) return wm == 1 ? launch_c1_cfg<128, 1>(a, k, smem, s) : launch_c1_cfg<128, 2>(a, k, smem, s);
  if (bm == 32) return launch_c1_cfg<32, 1>(a, k, smem, s);   // cout > 192 (adm_conv1x1_resident_cfg): always the 1 x 8 wave layout
  return wm == 1 ? launch_c1_cfg<64, 1>(a, k, smem, s) : launch_c1_cfg<64, 2>(a, k, smem, s);
}
