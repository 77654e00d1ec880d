// K12: the layers of the FID Inception-v3 pool3 extractor (gfx950) -- general 2-D convolution on the matrix cores,
// 3x3 pooling, global average pooling and the bilinear 299 x 299 input resize.
//
// Replaces, for candidate scoring, the third-party feature extractors the reference calls:
//   * guided_diffusion: `sess.run([pool_features, spatial_features], {image_input: batch})` on a frozen TensorFlow
//     Inception graph (evaluations/evaluator_v1.py:263-269, graph import :665-679)
//   * Stable Diffusion: `pytorch_fid.inception.InceptionV3([3])` (scripts/search_ea.py:95-127, 171-182)
// whose network is BasicConv2d = Conv2d(bias=False) -> BatchNorm(eps 1e-3) -> ReLU with kernels 1x1, 3x3 (stride 1 / 2,
// padded or not), 5x5, 1x7, 7x1, 1x3, 3x1, max / average 3x3 pools and channel concatenation.  The host folds the
// BatchNorm into the packed weights (scale) and a bias (adm_pack_conv2d_weight), every branch writes its channel slice
// of the block's concatenated NHWC output directly (out_stride), so a block is conv launches + one or two pool launches.
//
// adm_conv2d is an implicit GEMM, D[cout][pixel] = sum_k W[cout][k] * X[k][pixel] with k = (tap, channel).  Both operands are
// K-contiguous in memory (NHWC activations with the channel count padded to 32, weights packed [cout][tap][cin_pad]), so a
// lane's MFMA fragment -- 8 consecutive channels of one pixel / one output channel at one tap -- is 16 contiguous bytes;
// out-of-image taps, pixels beyond the tensor and channels beyond cout read as hardware zeros through the buffer descriptor
// (offset bit 31).  The weights are the A operand, so a lane ends up with 4 CONSECUTIVE output channels of one pixel:
// bias + ReLU + one 8-byte store per accumulator, no transposition.  Two kernels:
//   * convg_lds_kernel (layers with more than 32 output channels): a block's K-step of both operands goes global -> LDS by
//     LDS-DMA into a 3-stage ring of XOR-swizzled 64-byte rows, one barrier per step; 230-640 TFLOP/s per layer
//   * convg_kernel (32 output channels or fewer: the first two layers): LDS-free, every wave fetches its own fragments
// These trade the LDS-staged halo tiles of adm_conv (fused GroupNorm prologue, persistent tiles, 1000+ TFLOP/s) for
// generality (any kernel size / stride / padding / channel slice); sized for the ~11 GFLOP/image Inception network.
#include <stdlib.h>

#include "adm_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int cg_u32x4;
constexpr unsigned CG_OOB = 0x80000000u;

struct ConvG {
  const uint16_t* in; const uint16_t* w; const float* bias; uint16_t* out;
  int N, H, W, OH, OW, cin_pad, in_stride, cout, out_stride, KH, KW, stride, pad_h, pad_w, relu;
  unsigned in_bytes, w_bytes;
  long long M;
};

__device__ __forceinline__ adm_h8 cg_load(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  const cg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  return __builtin_bit_cast(adm_h8, v);
}

template <int TN>   // 16-channel tiles per wave (the block's Cout width is TN * 16); 4 pixel tiles of 16 per wave
__global__ void __launch_bounds__(256)
convg_kernel(const ConvG p) {
  constexpr int TM = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  const long long m0 = ((long long)blockIdx.x * 4 + wave) * (TM * 16);
  if (m0 >= p.M) return;   // wave-uniform; the kernel has no barriers
  const int co0 = blockIdx.y * (TN * 16);
  const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  const int taps = p.KH * p.KW, chunks = p.cin_pad >> 5;

  // this lane's 4 pixels (one per pixel tile): image base and the top-left input coordinate of the window
  int pbase[TM], iy0[TM], ix0[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = m0 + i * 16 + lc;
    if (m < p.M) {
      const int ox = (int)(m % p.OW);
      const long long t = m / p.OW;
      const int oy = (int)(t % p.OH), img = (int)(t / p.OH);
      pbase[i] = img * p.H * p.W;
      iy0[i] = oy * p.stride - p.pad_h;
      ix0[i] = ox * p.stride - p.pad_w;
    } else {
      pbase[i] = 0;
      iy0[i] = -(1 << 20);   // every tap out of the image: zeros
      ix0[i] = 0;
    }
  }
  // weight rows of this lane's output channels (rows beyond cout lie beyond the descriptor: zeros)
  unsigned wrow[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) wrow[j] = (unsigned)(co0 + j * 16 + lc) * (unsigned)(taps * p.cin_pad * 2) + lq * 16;

  unsigned aoff[TM];   // byte offset of (pixel, tap, channel lq*8) or OOB
  auto tap_offsets = [&](int ky, int kx) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int iy = iy0[i] + ky, ix = ix0[i] + kx;
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      aoff[i] = ok ? (unsigned)(pbase[i] + iy * p.W + ix) * (unsigned)(p.in_stride * 2) + lq * 16 : CG_OOB;
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  adm_h8 xa[TM], wa[TN], xn[TM], wn[TN];
  int ky = 0, kx = 0, tap = 0, c = 0;
  tap_offsets(0, 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) xa[i] = cg_load(rsi, aoff[i]);
#pragma unroll
  for (int j = 0; j < TN; ++j) wa[j] = cg_load(rsw, wrow[j]);
  const int steps = taps * chunks;
  for (int s = 0; s < steps; ++s) {
    // advance (tap, chunk) and fetch the next step's fragments (the last step re-reads its own: unused)
    if (s + 1 < steps) {
      if (++c == chunks) {
        c = 0;
        ++tap;
        if (++kx == p.KW) { kx = 0; ++ky; }
        tap_offsets(ky, kx);
      }
    }
    const unsigned cb = (unsigned)c * 64u, wb = (unsigned)(tap * p.cin_pad) * 2u + cb;
#pragma unroll
    for (int i = 0; i < TM; ++i) xn[i] = cg_load(rsi, aoff[i] == CG_OOB ? CG_OOB : aoff[i] + cb);
#pragma unroll
    for (int j = 0; j < TN; ++j) wn[j] = cg_load(rsw, wrow[j] + wb);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[j][i] = adm_mfma_16x16x32(wa[j], xa[i], acc[j][i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) xa[i] = xn[i];
#pragma unroll
    for (int j = 0; j < TN; ++j) wa[j] = wn[j];
  }

  // D[cout][pixel]: this lane holds channels co0 + j*16 + 4*lq .. +3 of pixel m0 + i*16 + lc
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = co0 + j * 16 + 4 * lq;
    if (co + 3 >= p.cout) continue;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) b = *reinterpret_cast<const float4*>(p.bias + co);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long long m = m0 + i * 16 + lc;
      if (m >= p.M) continue;
      float v0 = acc[j][i][0] + b.x, v1 = acc[j][i][1] + b.y, v2 = acc[j][i][2] + b.z, v3 = acc[j][i][3] + b.w;
      if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
      uint2 o;
      o.x = adm_pack2(v0, v1);
      o.y = adm_pack2(v2, v3);
      *reinterpret_cast<uint2*>(p.out + m * p.out_stride + co) = o;
    }
  }
}

// The same implicit GEMM with both operand tiles staged through LDS (layers with >= 33 output channels).  In the LDS-free
// kernel above every wave fetches its own 64-pixel and 64-channel fragments: 8 KB through the L1 per 16 MFMAs, 32 FLOP per
// byte against the 64 the CU's 64 B/clk load path needs to keep the matrix pipe busy -- measured 130-320 TFLOP/s.  Here a
// block of 4 waves (WM x WN, each 64 pixels x 64 channels) shares one K-step of both operands: (WM + WN) * 4 KB per step go
// global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR round trip; out-of-image taps, pixels past the tensor and
// channels past cout arrive as zeros), double-buffered, one barrier per step, and each wave reads its 4 + 4 fragments with
// ds_read_b128.  A DMA piece is lane-linear (1 KB = 16 rows of 64 B), so the rows are unpadded and the 16-byte slot of
// k-quarter q in row r is q ^ ((r >> 1) & 3): applied on the global side of the DMA and on every fragment read;
// conflict-free for the four 16-lane groups of ds_read_b128 (checked exhaustively).
template <int WM, int WN>
__global__ void __launch_bounds__(256)
convg_lds_kernel(const ConvG p) {
  constexpr int TM = 4, TN = 4, BM = WM * 64, BN = WN * 64;
  constexpr int NA = BM / 16, NB = BN / 16, NP = NA + NB, NPW = NP / 4;   // 1 KB DMA pieces per K-step: pixels, channels, per wave
  static_assert(WM * WN == 4 && NP % 4 == 0, "4 waves");
  constexpr int STAGE = NP * 1024, NS = 3;   // ring of 3 stages: the DMA of step s + 2 flies during step s (one step of
  __shared__ __attribute__((aligned(1024))) unsigned char tiles[NS * STAGE];   // 16 MFMAs is far shorter than a global round trip)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lc = lane & 15, lq = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const long long m0 = (long long)blockIdx.x * BM;
  const int co0 = blockIdx.y * BN;
  const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  const int taps = p.KH * p.KW, chunks = p.cin_pad >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)tiles;

  // ---- the DMA pieces of this wave: piece q = wave + 4 * u; pieces [0, NA) are pixel rows, [NA, NP) channel rows.
  // lane -> row (lane >> 2) of the piece, LDS slot lane & 3, i.e. global k-quarter (lane & 3) ^ ((row >> 1) & 3)
  const unsigned gq16 = (unsigned)((lane & 3) ^ ((lane >> 3) & 3)) * 16u;
  int pbase[NPW], iy0[NPW], ix0[NPW];
  unsigned wrow[NPW];
#pragma unroll
  for (int u = 0; u < NPW; ++u) {
    const int q = wave + 4 * u;
    pbase[u] = 0; iy0[u] = -(1 << 20); ix0[u] = 0; wrow[u] = CG_OOB;
    if (q < NA) {
      const long long m = m0 + q * 16 + (lane >> 2);
      if (m < p.M) {
        const int ox = (int)(m % p.OW);
        const long long t = m / p.OW;
        const int oy = (int)(t % p.OH), img = (int)(t / p.OH);
        pbase[u] = img * p.H * p.W;
        iy0[u] = oy * p.stride - p.pad_h;
        ix0[u] = ox * p.stride - p.pad_w;
      }
    } else {
      const int co = co0 + (q - NA) * 16 + (lane >> 2);
      if (co < p.cout) wrow[u] = (unsigned)co * (unsigned)(taps * p.cin_pad * 2) + gq16;
    }
  }
  unsigned aoff[NPW];
  auto tap_offsets = [&](int ky, int kx) {
#pragma unroll
    for (int u = 0; u < NPW; ++u) {
      const int iy = iy0[u] + ky, ix = ix0[u] + kx;
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      aoff[u] = ok ? (unsigned)(pbase[u] + iy * p.W + ix) * (unsigned)(p.in_stride * 2) + gq16 : CG_OOB;
    }
  };
  auto dma_step = [&](int buf, int tap, int c) {
    const unsigned cb = (unsigned)c * 64u, wb = (unsigned)(tap * p.cin_pad) * 2u + cb;
#pragma unroll
    for (int u = 0; u < NPW; ++u) {
      const int q = wave + 4 * u;                                  // wave-uniform
      const unsigned dst = lds0 + (unsigned)buf * STAGE + (unsigned)q * 1024u;
      unsigned keep;
      if (q < NA) {
        const unsigned v = aoff[u] == CG_OOB ? CG_OOB : aoff[u] + cb;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(v), "s"(rsi), "s"(dst) : "memory");
      } else {
        const unsigned v = wrow[u] == CG_OOB ? CG_OOB : wrow[u] + wb;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(v), "s"(rsw), "s"(dst) : "memory");
      }
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: row (16-row tile base + lc), slot lq ^ ((lc >> 1) & 3)
  const unsigned frag = (unsigned)lc * 64u + (unsigned)(lq ^ ((lc >> 1) & 3)) * 16u;
  const unsigned xfrag = (unsigned)(wm * 4) * 1024u + frag, wfrag = (unsigned)(NA + wn * 4) * 1024u + frag;

  int ky = 0, kx = 0, tap = 0, c = 0;   // (tap, chunk) of the step whose DMA is issued next
  const int steps = taps * chunks;
  auto advance = [&]() {
    if (++c == chunks) {
      c = 0;
      ++tap;
      if (++kx == p.KW) { kx = 0; ++ky; }
      tap_offsets(ky, kx);
    }
  };
  tap_offsets(0, 0);
  dma_step(0, 0, 0);
  if (steps > 1) { advance(); dma_step(1, tap, c); }
  if (steps > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");   // step 0 has landed, step 1 may still fly
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int stage = 0;
  for (int s = 0; s < steps; ++s) {
    const bool more = s + 2 < steps;
    if (more) {
      advance();
      dma_step(stage >= 1 ? stage - 1 : NS - 1, tap, c);   // stage (s + 2) % 3, last read in step s - 1
    }
    const unsigned char* st = tiles + stage * STAGE;
    adm_h8 xa[TM], wa[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) xa[i] = *reinterpret_cast<const adm_h8*>(st + xfrag + i * 1024);
#pragma unroll
    for (int j = 0; j < TN; ++j) wa[j] = *reinterpret_cast<const adm_h8*>(st + wfrag + j * 1024);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[j][i] = adm_mfma_16x16x32(wa[j], xa[i], acc[j][i], 0, 0, 0);
    // this wave's pieces of step s + 1 have landed (those of step s + 2, issued above, may still fly)
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // ... everyone's have, and everyone is done reading this stage
    stage = stage == NS - 1 ? 0 : stage + 1;
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = co0 + wn * 64 + j * 16 + 4 * lq;
    if (co + 3 >= p.cout) continue;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) b = *reinterpret_cast<const float4*>(p.bias + co);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long long m = m0 + wm * 64 + i * 16 + lc;
      if (m >= p.M) continue;
      float v0 = acc[j][i][0] + b.x, v1 = acc[j][i][1] + b.y, v2 = acc[j][i][2] + b.z, v3 = acc[j][i][3] + b.w;
      if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
      uint2 o;
      o.x = adm_pack2(v0, v1);
      o.y = adm_pack2(v2, v3);
      *reinterpret_cast<uint2*>(p.out + m * p.out_stride + co) = o;
    }
  }
}

// fp32 [cout][cin][kh][kw] (x scale[cout]) -> 16-bit [cout][kh*kw][cin_pad], channels beyond cin zero
__global__ void __launch_bounds__(256)
pack_conv2d_kernel(const float* __restrict__ w, const float* __restrict__ scale, uint16_t* __restrict__ out, int cout,
                   int cin, int taps, int cin_pad) {
  const long long total = (long long)cout * taps * cin_pad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cin_pad);
    const long long r = i / cin_pad;
    const int t = (int)(r % taps), o = (int)(r / taps);
    float v = 0.0f;
    if (c < cin) v = w[((long long)o * cin + c) * taps + t] * (scale ? scale[o] : 1.0f);
    out[i] = adm_f32_to_h(v);
  }
}

// k x k pooling over NHWC channel slices, 8 channels per thread.  mode 0: max (out-of-image taps ignored),
// mode 1: average over the taps INSIDE the image (count_include_pad=False)
__global__ void __launch_bounds__(256)
pool2d_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int n, int h, int w, int c, int in_stride,
              int out_stride, int oh, int ow, int k, int stride, int pad, int mode) {
  const int cg = c / 8;
  const long long items = (long long)n * oh * ow * cg;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(it % cg);
    const long long pix = it / cg;
    const int ox = (int)(pix % ow), oy = (int)((pix / ow) % oh), img = (int)(pix / ((long long)ow * oh));
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = mode == 0 ? -3.0e38f : 0.0f;
    int cnt = 0;
    for (int ky = 0; ky < k; ++ky) {
      const int iy = oy * stride - pad + ky;
      if ((unsigned)iy >= (unsigned)h) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = ox * stride - pad + kx;
        if ((unsigned)ix >= (unsigned)w) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(in + (((long long)img * h + iy) * w + ix) * in_stride + g * 8);
        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = adm_lo_f32(u[q]), hi = adm_hi_f32(u[q]);
          if (mode == 0) { r[2 * q] = fmaxf(r[2 * q], lo); r[2 * q + 1] = fmaxf(r[2 * q + 1], hi); }
          else { r[2 * q] += lo; r[2 * q + 1] += hi; }
        }
        ++cnt;
      }
    }
    if (mode == 1) {
      const float inv = 1.0f / (float)(cnt > 0 ? cnt : 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] *= inv;
    }
    uint4 o;
    o.x = adm_pack2(r[0], r[1]); o.y = adm_pack2(r[2], r[3]); o.z = adm_pack2(r[4], r[5]); o.w = adm_pack2(r[6], r[7]);
    *reinterpret_cast<uint4*>(out + pix * out_stride + g * 8) = o;
  }
}

// mean over the hw pixels of each (image, channel): NHWC 16-bit -> fp32 [n][c]
__global__ void __launch_bounds__(256)
global_avgpool_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, int n, int hw, int c) {
  const int cg = c / 8;
  const int it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= n * cg) return;
  const int g = it % cg, img = it / cg;
  float r[8] = {};
  for (int px = 0; px < hw; ++px) {
    const uint4 v = *reinterpret_cast<const uint4*>(in + ((long long)img * hw + px) * c + g * 8);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) { r[2 * q] += adm_lo_f32(u[q]); r[2 * q + 1] += adm_hi_f32(u[q]); }
  }
  const float inv = 1.0f / (float)hw;
#pragma unroll
  for (int e = 0; e < 8; ++e) out[(long long)img * c + g * 8 + e] = r[e] * inv;
}

// bilinear resize of 3-channel images to [n][oh][ow][cpad] 16-bit NHWC (channels 3.. zero), value * scale + shift.
// kind 0: uint8 NHWC, kind 1: fp32 NCHW, kind 2: fp32 NHWC.  half_pixel 1: src = (dst + 0.5) * in/out - 0.5 clamped at 0
// (torch F.interpolate(mode="bilinear", align_corners=False)); 0: src = dst * in/out (TensorFlow-1 ResizeBilinear).
__global__ void __launch_bounds__(256)
resize_bilinear_kernel(const void* __restrict__ in, uint16_t* __restrict__ out, int n, int h, int w, int oh, int ow,
                       int cpad, int kind, int half_pixel, float scale, float shift) {
  // one thread per 16-byte segment of an output pixel (consecutive lanes write consecutive segments: a pixel-per-thread
  // version wrote its 64 bytes as four 16-byte stores 64 bytes apart and ran at 1.5 TB/s); segment 0 holds the image
  const int segs = cpad / 8;
  const long long items = (long long)n * oh * ow * segs;
  const float ry = (float)h / (float)oh, rx = (float)w / (float)ow;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
    const int sg = (int)(it % segs);
    const long long pix = it / segs;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (sg == 0) {
      const int ox = (int)(pix % ow), oy = (int)((pix / ow) % oh), img = (int)(pix / ((long long)ow * oh));
      float sy = half_pixel ? ((float)oy + 0.5f) * ry - 0.5f : (float)oy * ry;
      float sx = half_pixel ? ((float)ox + 0.5f) * rx - 0.5f : (float)ox * rx;
      sy = fmaxf(sy, 0.0f);
      sx = fmaxf(sx, 0.0f);
      const int y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1);
      const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
      const float fy = sy - (float)y0, fx = sx - (float)x0;
      float v[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        auto at = [&](int y, int x) -> float {
          if (kind == 0) return (float)reinterpret_cast<const uint8_t*>(in)[(((long long)img * h + y) * w + x) * 3 + ch];
          if (kind == 1) return reinterpret_cast<const float*>(in)[(((long long)img * 3 + ch) * h + y) * w + x];
          return reinterpret_cast<const float*>(in)[(((long long)img * h + y) * w + x) * 3 + ch];
        };
        // the operation order of both references: interpolate along x on the two rows, then along y
        const float top = at(y0, x0) * (1.0f - fx) + at(y0, x1) * fx;
        const float bot = at(y1, x0) * (1.0f - fx) + at(y1, x1) * fx;
        v[ch] = (top * (1.0f - fy) + bot * fy) * scale + shift;
      }
      val.x = adm_pack2(v[0], v[1]);
      val.y = adm_pack2(v[2], 0.0f);
    }
    *reinterpret_cast<uint4*>(out + it * 8) = val;
  }
}

int grid_for_items(long long items) {
  long long b = (items + 255) / 256;
  if (b > 8192) b = 8192;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int adm_conv2d(const adm_conv2d_args* a, void* stream) {
  ADM_REQUIRE(a && a->in && a->w && a->out, ADM_E_ARG, "adm_conv2d: null pointer");
  ADM_REQUIRE(a->n > 0 && a->h > 0 && a->w_in > 0 && a->kh > 0 && a->kw > 0 && a->stride > 0 && a->pad_h >= 0 && a->pad_w >= 0,
              ADM_E_ARG, "adm_conv2d: bad geometry");
  ADM_REQUIRE(a->cin_pad > 0 && a->cin_pad % 32 == 0 && a->in_stride >= a->cin_pad && a->in_stride % 8 == 0, ADM_E_SHAPE,
              "adm_conv2d: cin_pad %d must be a multiple of 32 and fit the input's channel stride %d (a multiple of 8)",
              a->cin_pad, a->in_stride);
  ADM_REQUIRE(a->cout > 0 && a->cout % 4 == 0 && a->out_stride >= a->cout && a->out_stride % 4 == 0, ADM_E_SHAPE,
              "adm_conv2d: cout %d must be a multiple of 4 within the output's channel stride %d", a->cout, a->out_stride);
  const int oh = (a->h + 2 * a->pad_h - a->kh) / a->stride + 1, ow = (a->w_in + 2 * a->pad_w - a->kw) / a->stride + 1;
  ADM_REQUIRE(oh > 0 && ow > 0, ADM_E_SHAPE, "adm_conv2d: empty output");
  ADM_REQUIRE(adm_aligned16(a->in) && adm_aligned16(a->w) && (((uintptr_t)a->out) & 7u) == 0 && adm_aligned16(a->bias), ADM_E_ALIGN,
              "adm_conv2d: unaligned pointer");
  const long long in_bytes = (long long)a->n * a->h * a->w_in * a->in_stride * 2;
  const long long w_bytes = (long long)a->cout * a->kh * a->kw * a->cin_pad * 2;
  ADM_REQUIRE(in_bytes < (1ll << 31) && w_bytes < (1ll << 31), ADM_E_SHAPE, "adm_conv2d: operand beyond 2 GiB (32-bit buffer offsets)");
  ConvG k;
  k.in = reinterpret_cast<const uint16_t*>(a->in); k.w = reinterpret_cast<const uint16_t*>(a->w); k.bias = a->bias;
  k.out = reinterpret_cast<uint16_t*>(a->out);
  k.N = a->n; k.H = a->h; k.W = a->w_in; k.OH = oh; k.OW = ow; k.cin_pad = a->cin_pad; k.in_stride = a->in_stride;
  k.cout = a->cout; k.out_stride = a->out_stride; k.KH = a->kh; k.KW = a->kw; k.stride = a->stride; k.pad_h = a->pad_h;
  k.pad_w = a->pad_w; k.relu = a->relu;
  k.in_bytes = (unsigned)in_bytes; k.w_bytes = (unsigned)w_bytes;
  k.M = (long long)a->n * oh * ow;
  const long long mblocks = (k.M + 255) / 256;
  ADM_REQUIRE(mblocks < (1ll << 30), ADM_E_SHAPE, "adm_conv2d: too many pixels");
  hipStream_t s = (hipStream_t)stream;
  // Cout block width by least padding: 128 (LDS kernel, 128 pixels x 128 channels) where rounding cout up to 128 costs no more
  // than rounding to 64, else 64 (LDS kernel, 256 x 64); 32 output channels or fewer: the LDS-free kernel, 256 x 32.
  // (A 128-wide wave tile in the LDS-free kernel -- 12 fragment loads per 32 MFMAs instead of 8 per 16 -- measured +7 % on
  // 288 -> 384 @ 35x35 and -35 % on the 8x8 level, where it halves an already short grid: not built.)
  static const bool no_lds = getenv("ADM_CG_NO_LDS") != nullptr;   // A/B switch for measurements
  const int pad128 = (a->cout + 127) / 128 * 128, pad64 = (a->cout + 63) / 64 * 64, pad32 = (a->cout + 31) / 32 * 32;
  if (a->cout > 32 && !no_lds) {
    if (pad128 == pad64) hipLaunchKernelGGL((convg_lds_kernel<2, 2>), dim3((unsigned)((k.M + 127) / 128), pad128 / 128), dim3(256), 0, s, k);
    else hipLaunchKernelGGL((convg_lds_kernel<4, 1>), dim3((unsigned)mblocks, pad64 / 64), dim3(256), 0, s, k);
  } else if (pad32 < pad64) hipLaunchKernelGGL((convg_kernel<2>), dim3((unsigned)mblocks, pad32 / 32), dim3(256), 0, s, k);
  else hipLaunchKernelGGL((convg_kernel<4>), dim3((unsigned)mblocks, pad64 / 64), dim3(256), 0, s, k);
  return adm_check_launch("adm_conv2d");
}

extern "C" int adm_pack_conv2d_weight(const float* w, const float* scale, adm_bf16* out, int cout, int cin, int kh, int kw,
                                      int cin_pad, void* stream) {
  ADM_REQUIRE(w && out, ADM_E_ARG, "adm_pack_conv2d_weight: null pointer");
  ADM_REQUIRE(cout > 0 && cin > 0 && kh > 0 && kw > 0 && cin_pad >= cin && cin_pad % 32 == 0, ADM_E_SHAPE,
              "adm_pack_conv2d_weight: bad shape cout=%d cin=%d %dx%d cin_pad=%d", cout, cin, kh, kw, cin_pad);
  const long long total = (long long)cout * kh * kw * cin_pad;
  hipLaunchKernelGGL(pack_conv2d_kernel, dim3(grid_for_items(total)), dim3(256), 0, (hipStream_t)stream, w, scale,
                     reinterpret_cast<uint16_t*>(out), cout, cin, kh * kw, cin_pad);
  return adm_check_launch("adm_pack_conv2d_weight");
}

extern "C" int adm_pool2d(const adm_bf16* in, adm_bf16* out, int n, int h, int w, int c, int in_stride, int out_stride, int k,
                          int stride, int pad, int mode, void* stream) {
  ADM_REQUIRE(in && out, ADM_E_ARG, "adm_pool2d: null pointer");
  ADM_REQUIRE(n > 0 && h > 0 && w > 0 && k > 0 && stride > 0 && pad >= 0 && pad < k && (mode == 0 || mode == 1), ADM_E_ARG,
              "adm_pool2d: bad arguments");
  ADM_REQUIRE(c > 0 && c % 8 == 0 && in_stride >= c && out_stride >= c && in_stride % 8 == 0 && out_stride % 8 == 0, ADM_E_SHAPE,
              "adm_pool2d: channels %d (strides %d, %d) must be multiples of 8", c, in_stride, out_stride);
  ADM_REQUIRE(adm_aligned16(in) && adm_aligned16(out), ADM_E_ALIGN, "adm_pool2d: unaligned pointer");
  const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
  ADM_REQUIRE(oh > 0 && ow > 0, ADM_E_SHAPE, "adm_pool2d: empty output");
  hipLaunchKernelGGL(pool2d_kernel, dim3(grid_for_items((long long)n * oh * ow * (c / 8))), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint16_t*>(in), reinterpret_cast<uint16_t*>(out), n, h, w, c, in_stride, out_stride,
                     oh, ow, k, stride, pad, mode);
  return adm_check_launch("adm_pool2d");
}

extern "C" int adm_global_avgpool_f32(const adm_bf16* in, float* out, int n, int hw, int c, void* stream) {
  ADM_REQUIRE(in && out, ADM_E_ARG, "adm_global_avgpool_f32: null pointer");
  ADM_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 8 == 0, ADM_E_SHAPE, "adm_global_avgpool_f32: bad shape");
  ADM_REQUIRE(adm_aligned16(in), ADM_E_ALIGN, "adm_global_avgpool_f32: unaligned pointer");
  const int items = n * (c / 8);
  hipLaunchKernelGGL(global_avgpool_kernel, dim3((items + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint16_t*>(in), out, n, hw, c);
  return adm_check_launch("adm_global_avgpool_f32");
}

extern "C" int adm_resize_bilinear(const void* in, adm_bf16* out, int n, int h, int w, int oh, int ow, int cpad, int kind,
                                   int half_pixel, float scale, float shift, void* stream) {
  ADM_REQUIRE(in && out, ADM_E_ARG, "adm_resize_bilinear: null pointer");
  ADM_REQUIRE(n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && cpad >= 8 && cpad % 8 == 0 && kind >= 0 && kind <= 2, ADM_E_ARG,
              "adm_resize_bilinear: bad arguments");
  ADM_REQUIRE(adm_aligned16(out), ADM_E_ALIGN, "adm_resize_bilinear: unaligned output");
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for_items((long long)n * oh * ow * (cpad / 8))), dim3(256), 0, (hipStream_t)stream,
                     in, reinterpret_cast<uint16_t*>(out), n, h, w, oh, ow, cpad, kind, half_pixel, scale, shift);
  return adm_check_launch("adm_resize_bilinear");
}
