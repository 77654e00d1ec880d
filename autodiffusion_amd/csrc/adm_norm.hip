// Stem convolution, GroupNorm statistics and the up/down resampling of ResBlocks.
//
//  * adm_stem_conv3x3 : input_blocks.0.0 on the fp32 NCHW image (reference unet.py:480-483, 656-658)
//  * adm_gn_partial / adm_gn_finalize : GroupNorm32(32, C) statistics (reference nn.py:17-19, 93-100)
//    folded with the ResBlock FiLM pair into one per-(image, channel) affine that the conv kernel
//    applies while staging its input tile (unet.py:237-252)
//  * adm_resample : h_upd / x_upd of up/down ResBlocks (unet.py:190-195, 237-242)
//
// All HBM-bound: 16-byte (8 x bf16) accesses per lane, pixel-major so that a pixel's channels are
// one contiguous run.
#include "adm_common.h"

namespace {

// ------------------------------------------------------------------------------------ stem
__global__ void __launch_bounds__(256)
stem_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
            uint16_t* __restrict__ out, int n, int cin, int h, int wd, int cout) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [cin*9][cout] + bias[cout]
  const int kk = cin * 9;
  for (int i = threadIdx.x; i < kk * cout; i += blockDim.x) {
    const int o = i % cout, k = i / cout;  // w is [cout][cin][3][3] -> k = ci*9 + tap
    sw[i] = w[(long long)o * kk + k];
  }
  for (int i = threadIdx.x; i < cout; i += blockDim.x) sw[kk * cout + i] = bias[i];
  __syncthreads();
  const int cg = cout / 8;
  const long long items = (long long)n * h * wd * cg;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items;
       it += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(it % cg);
    const long long pix = it / cg;
    const int px = (int)(pix % wd), py = (int)((pix / wd) % h), img = (int)(pix / ((long long)wd * h));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = sw[kk * cout + g * 8 + j];
    for (int ci = 0; ci < cin; ++ci) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = py + ky - 1;
        if (yy < 0 || yy >= h) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = px + kx - 1;
          if (xx < 0 || xx >= wd) continue;
          const float v = x[(((long long)img * cin + ci) * h + yy) * wd + xx];
          const float* wr = sw + (ci * 9 + ky * 3 + kx) * cout + g * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += v * wr[j];
        }
      }
    }
    uint4 pk;
    pk.x = adm_pack2(acc[0], acc[1]);
    pk.y = adm_pack2(acc[2], acc[3]);
    pk.z = adm_pack2(acc[4], acc[5]);
    pk.w = adm_pack2(acc[6], acc[7]);
    *reinterpret_cast<uint4*>(out + pix * cout + g * 8) = pk;
  }
}

// fp32 NCHW image -> bf16 NHWC with the channel dimension zero-padded to cpad (32): the stem then runs on
// the MFMA conv kernel (a direct VALU stem at batch 256 cost 1.3 ms; as a K=32 implicit GEMM it is ~10x faster)
__global__ void __launch_bounds__(256)
nchw_to_nhwc_pad_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, int n, int c, int hw, int cpad) {
  const long long items = (long long)n * hw * (cpad / 8);
  const int cg = cpad / 8;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items;
       it += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(it % cg);
    const long long pix = it / cg;
    const int p = (int)(pix % hw), img = (int)(pix / hw);
    uint32_t u[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = g * 8 + j;
      if (ch < c) u[j >> 1] |= (uint32_t)adm_f32_to_h(x[((long long)img * c + ch) * hw + p]) << ((j & 1) * 16);
    }
    *reinterpret_cast<uint4*>(out + pix * cpad + g * 8) = make_uint4(u[0], u[1], u[2], u[3]);
  }
}

// ------------------------------------------------------------------------------------ GN partial
// grid (slabs, n). Thread -> (pixel lane, 8-channel group). Per-channel (sum, sumsq) over the slab's
// pixels, combined across pixel lanes through LDS, written as partial[n][slab][c][2].
__global__ void __launch_bounds__(256)
gn_partial_kernel(const uint16_t* __restrict__ in0, int c0, const uint16_t* __restrict__ in1, int c1,
                  float* __restrict__ partial, int hw, int slabs, int c_total, int c_off) {
  // (c_total, c_off): this launch covers channels [c_off, c_off + c0 + c1) of rows that are c_total wide
  extern __shared__ __attribute__((aligned(16))) float red[];  // [lanes][c][2]
  const int c = c0 + c1;
  const int groups8 = c / 8;
  const int lanes = blockDim.x / groups8;  // pixel lanes per block (>= 1)
  const int lane = threadIdx.x / groups8, g8 = threadIdx.x % groups8;
  const int slab = blockIdx.x, img = blockIdx.y;
  const int per = (hw + slabs - 1) / slabs;
  const int p_begin = slab * per, p_end = min(hw, p_begin + per);
  float s[8] = {}, ss[8] = {};
  if (lane < lanes) {
    const int ch = g8 * 8;
    const uint16_t* src;
    int cs, co;
    if (ch < c0) { src = in0; cs = c0; co = ch; } else { src = in1; cs = c1; co = ch - c0; }
    for (int p = p_begin + lane; p < p_end; p += lanes) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + ((long long)img * hw + p) * cs + co);
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = adm_lo_f32(u[j]), b = adm_hi_f32(u[j]);
        s[2 * j] += a; ss[2 * j] += a * a;
        s[2 * j + 1] += b; ss[2 * j + 1] += b * b;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[((long long)lane * c + ch + j) * 2 + 0] = s[j];
      red[((long long)lane * c + ch + j) * 2 + 1] = ss[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < c * 2; i += blockDim.x) {
    float t = 0.0f;
    for (int l = 0; l < lanes; ++l) t += red[(long long)l * c * 2 + i];
    partial[(((long long)img * slabs + slab) * c_total + c_off) * 2 + i] = t;
  }
}

// grid (n, 4): a block takes 8 of the 32 groups, one half-wave (32 lanes) per group = 8 channel lanes x 4 slab lanes.  Combine slabs
// and the channels of each group in double, emit the affine.  The launch is a chain of dependent-latency loads, not bandwidth
// (a few hundred 8-byte partial sums per group): one wave-quarter per group with every slab of a channel walked by one lane took
// 5-7 us per call -- 3.4 % of the SD step (61 calls per evaluation on each guidance stream's critical path), 1 % of the guided step.
__global__ void __launch_bounds__(256)
gn_finalize_kernel(const float* __restrict__ part0, int c0, int slabs0, const float* __restrict__ part1, int c1, int slabs1,
                   const float* __restrict__ gamma, const float* __restrict__ beta,
                   const float* __restrict__ film, int film_stride, float* __restrict__ aff_a,
                   float* __restrict__ aff_b, float* __restrict__ stats, int hw, float eps,
                   const float* __restrict__ add, int add_stride) {
  const int c = c0 + c1;
  __shared__ float gmean[8], grstd[8];
  const int img = blockIdx.x;
  const int cpg = c / 32;
  const int gl = threadIdx.x / 32, grp = blockIdx.y * 8 + gl, sub = threadIdx.x % 32;
  const int cl = sub % 8, sl0 = sub / 8;     // channel lane, slab lane
  double s = 0.0, ss = 0.0;
  for (int j = cl; j < cpg; j += 8) {  // channels of the group; each may live in either part of the concat
    const int ch = grp * cpg + j;
    const bool first = ch < c0;
    const float* base = first ? part0 : part1;
    const int cs = first ? c0 : c1, cl_ = first ? ch : ch - c0, slabs = first ? slabs0 : slabs1;
    double cs1 = 0.0, cs2 = 0.0;
    const float2* p = reinterpret_cast<const float2*>(base + (((long long)img * slabs) * cs + cl_) * 2);
    int sl = sl0;
    for (; sl + 12 < slabs; sl += 16) {   // four slabs of this lane in flight
      const float2 v0 = p[(long long)sl * cs], v1 = p[(long long)(sl + 4) * cs], v2 = p[(long long)(sl + 8) * cs],
                   v3 = p[(long long)(sl + 12) * cs];
      cs1 += (double)v0.x; cs2 += (double)v0.y;
      cs1 += (double)v1.x; cs2 += (double)v1.y;
      cs1 += (double)v2.x; cs2 += (double)v2.y;
      cs1 += (double)v3.x; cs2 += (double)v3.y;
    }
    for (; sl < slabs; sl += 4) {
      const float2 v = p[(long long)sl * cs];
      cs1 += (double)v.x;
      cs2 += (double)v.y;
    }
    if (add) {  // statistics of x + e[img, ch] from those of x: sum += hw*e, sum of squares += 2*e*sum + hw*e^2 (linear in this lane's share)
      const double e = (double)add[(long long)img * add_stride + ch];
      cs2 += 2.0 * e * cs1;
      if (sl0 == 0) { cs2 += (double)hw * e * e; cs1 += (double)hw * e; }
    }
    s += cs1;
    ss += cs2;
  }
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {
    s += __shfl_xor(s, off);
    ss += __shfl_xor(ss, off);
  }
  if (sub == 0) {
    const double cnt = (double)cpg * (double)hw;
    const double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    gmean[gl] = (float)mean;
    grstd[gl] = (float)(1.0 / sqrt(var + (double)eps));
    if (stats) {  // kept for the backward-data pass (classifier guidance)
      stats[((long long)img * 32 + grp) * 2 + 0] = gmean[gl];
      stats[((long long)img * 32 + grp) * 2 + 1] = grstd[gl];
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < 8 * cpg; q += blockDim.x) {   // the channels of this block's 8 groups
    const int g = q / cpg, ch = blockIdx.y * 8 * cpg + q;
    float a = grstd[g] * gamma[ch];
    float b = beta[ch] - gmean[g] * a;
    if (add) b += a * add[(long long)img * add_stride + ch];  // y = a*(x + e) + b applied to the stored x
    if (film) {
      const float sc = 1.0f + film[(long long)img * film_stride + ch];
      const float sh = film[(long long)img * film_stride + c + ch];
      a = a * sc;
      b = b * sc + sh;
    }
    aff_a[(long long)img * c + ch] = a;
    aff_b[(long long)img * c + ch] = b;
  }
}

// ------------------------------------------------------------------------------------ resample
template <int MODE, bool ACT>  // MODE 1: avgpool2, 2: nearest x2, 3: every second pixel (stride-2 subsample), 4: zero-insert x2 (out[2y][2x] = in[y][x], 0 elsewhere:
                                // the input of a stride-2 conv's backward-data conv)
__global__ void __launch_bounds__(256)
resample_kernel(const uint16_t* __restrict__ in, const float* __restrict__ aff_a, const float* __restrict__ aff_b,
                uint16_t* __restrict__ out, int n, int h, int w, int c) {
  const int oh = (MODE == 2 || MODE == 4) ? h * 2 : h / 2, ow = (MODE == 2 || MODE == 4) ? w * 2 : w / 2;
  const int cg = c / 8;
  const long long items = (long long)n * oh * ow * cg;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items;
       it += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(it % cg);
    const long long opix = it / cg;
    const int ox = (int)(opix % ow), oy = (int)((opix / ow) % oh), img = (int)(opix / ((long long)ow * oh));
    if (MODE == 4 && ((oy | ox) & 1)) {
      *reinterpret_cast<uint4*>(out + opix * c + g * 8) = make_uint4(0u, 0u, 0u, 0u);
      continue;
    }
    float a8[8], b8[8];
    if (ACT) {
      *reinterpret_cast<float4*>(a8) = *reinterpret_cast<const float4*>(aff_a + (long long)img * c + g * 8);
      *reinterpret_cast<float4*>(a8 + 4) = *reinterpret_cast<const float4*>(aff_a + (long long)img * c + g * 8 + 4);
      *reinterpret_cast<float4*>(b8) = *reinterpret_cast<const float4*>(aff_b + (long long)img * c + g * 8);
      *reinterpret_cast<float4*>(b8 + 4) = *reinterpret_cast<const float4*>(aff_b + (long long)img * c + g * 8 + 4);
    }
    float acc[8] = {};
    constexpr int TAPS = MODE == 1 ? 4 : 1;
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
      const int iy = MODE == 1 ? oy * 2 + tp / 2 : ((MODE == 2 || MODE == 4) ? oy / 2 : oy * 2);
      const int ix = MODE == 1 ? ox * 2 + tp % 2 : ((MODE == 2 || MODE == 4) ? ox / 2 : ox * 2);
      const uint4 v = *reinterpret_cast<const uint4*>(in + (((long long)img * h + iy) * w + ix) * c + g * 8);
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float lo = adm_lo_f32(u[j]), hi = adm_hi_f32(u[j]);
        if (ACT) {
          lo = adm_silu(a8[2 * j] * lo + b8[2 * j]);
          hi = adm_silu(a8[2 * j + 1] * hi + b8[2 * j + 1]);
        }
        acc[2 * j] += lo;
        acc[2 * j + 1] += hi;
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] *= 0.25f;
    }
    uint4 pk;
    pk.x = adm_pack2(acc[0], acc[1]);
    pk.y = adm_pack2(acc[2], acc[3]);
    pk.z = adm_pack2(acc[4], acc[5]);
    pk.w = adm_pack2(acc[6], acc[7]);
    *reinterpret_cast<uint4*>(out + opix * c + g * 8) = pk;
  }
}

}  // namespace

extern "C" int adm_stem_conv3x3(const float* x, const float* w, const float* bias, adm_bf16* out, int n, int cin,
                                int h, int wd, int cout, void* stream) {
  ADM_REQUIRE(x && w && bias && out, ADM_E_ARG, "adm_stem_conv3x3: null pointer");
  ADM_REQUIRE(n > 0 && h > 0 && wd > 0, ADM_E_ARG, "adm_stem_conv3x3: bad shape");
  ADM_REQUIRE(cin >= 1 && cin <= 8 && cout % 8 == 0 && cout <= 512, ADM_E_SHAPE,
              "adm_stem_conv3x3: cin=%d cout=%d unsupported (cin<=8, cout%%8==0, cout<=512)", cin, cout);
  ADM_REQUIRE(adm_aligned16(out), ADM_E_ALIGN, "adm_stem_conv3x3: out not 16-byte aligned");
  const size_t smem = (size_t)(cin * 9 + 1) * cout * sizeof(float);
  const long long items = (long long)n * h * wd * (cout / 8);
  int blocks = (int)((items + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(stem_kernel, dim3(blocks), dim3(256), smem, (hipStream_t)stream, x, w, bias, out, n, cin, h, wd, cout);
  return adm_check_launch("adm_stem_conv3x3");
}

extern "C" int adm_nchw_to_nhwc_pad(const float* x, adm_bf16* out, int n, int c, int h, int w, int cpad, void* stream) {
  ADM_REQUIRE(x && out, ADM_E_ARG, "adm_nchw_to_nhwc_pad: null pointer");
  ADM_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && cpad >= c && cpad % 8 == 0, ADM_E_ARG, "adm_nchw_to_nhwc_pad: bad shape");
  ADM_REQUIRE(adm_aligned16(out), ADM_E_ALIGN, "adm_nchw_to_nhwc_pad: unaligned output");
  const long long items = (long long)n * h * w * (cpad / 8);
  int blocks = (int)((items + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, n, c, h * w, cpad);
  return adm_check_launch("adm_nchw_to_nhwc_pad");
}

extern "C" int adm_gn_partial(const adm_bf16* in0, int c0, const adm_bf16* in1, int c1, float* partial, int n,
                              int hw, int slabs, void* stream) {
  ADM_REQUIRE(in0 && partial, ADM_E_ARG, "adm_gn_partial: null pointer");
  ADM_REQUIRE((in1 != nullptr) == (c1 > 0), ADM_E_ARG, "adm_gn_partial: in1/c1 mismatch");
  const int c = c0 + c1;
  ADM_REQUIRE(n > 0 && hw > 0 && slabs > 0 && slabs <= hw, ADM_E_ARG, "adm_gn_partial: bad n/hw/slabs");
  ADM_REQUIRE(c0 % 8 == 0 && c1 % 8 == 0 && c % 32 == 0 && c0 <= 2048 && c1 <= 2048, ADM_E_SHAPE,
              "adm_gn_partial: channels (%d + %d) unsupported", c0, c1);
  ADM_REQUIRE(adm_aligned16(in0) && adm_aligned16(in1), ADM_E_ALIGN, "adm_gn_partial: unaligned input");
  auto launch = [&](const adm_bf16* a, int ca, const adm_bf16* b, int cb, int c_off) {
    const int cc = ca + cb;
    const int lanes = 256 / (cc / 8);
    const size_t smem = (size_t)lanes * cc * 2 * sizeof(float);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(slabs, n), dim3(256), smem, (hipStream_t)stream, a, ca, b, cb, partial, hw, slabs, c, c_off);
  };
  if (c <= 2048) {
    launch(in0, c0, in1, c1, 0);
  } else {  // a block covers at most 2048 channels: one launch per source of the concat (the SD UNet's 1280 | 1280)
    launch(in0, c0, nullptr, 0, 0);
    launch(in1, c1, nullptr, 0, c0);
  }
  return adm_check_launch("adm_gn_partial");
}

extern "C" int adm_gn_finalize(const float* partial, const float* gamma, const float* beta, const float* film,
                               int film_stride, float* aff_a, float* aff_b, float* stats, int n, int c, int hw,
                               int slabs, float eps, void* stream) {
  ADM_REQUIRE(partial && gamma && beta && aff_a && aff_b, ADM_E_ARG, "adm_gn_finalize: null pointer");
  ADM_REQUIRE(n > 0 && c > 0 && c % 32 == 0 && hw > 0 && slabs > 0, ADM_E_SHAPE, "adm_gn_finalize: bad shape");
  ADM_REQUIRE(!film || film_stride >= 2 * c, ADM_E_ARG, "adm_gn_finalize: film_stride < 2*c");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(n, 4), dim3(256), 0, (hipStream_t)stream, partial, c, slabs,
                     (const float*)nullptr, 0, 0, gamma, beta, film, film_stride, aff_a, aff_b, stats, hw, eps,
                     (const float*)nullptr, 0);
  return adm_check_launch("adm_gn_finalize");
}

extern "C" int adm_gn_finalize_add(const float* partial, const float* gamma, const float* beta, const float* add,
                                   int add_stride, float* aff_a, float* aff_b, float* stats, int n, int c, int hw, int slabs, float eps,
                                   void* stream) {
  ADM_REQUIRE(partial && gamma && beta && add && aff_a && aff_b, ADM_E_ARG, "adm_gn_finalize_add: null pointer");
  ADM_REQUIRE(n > 0 && c > 0 && c % 32 == 0 && hw > 0 && slabs > 0 && add_stride >= c, ADM_E_SHAPE, "adm_gn_finalize_add: bad shape");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(n, 4), dim3(256), 0, (hipStream_t)stream, partial, c, slabs,
                     (const float*)nullptr, 0, 0, gamma, beta, (const float*)nullptr, 0, aff_a, aff_b, stats, hw, eps,
                     add, add_stride);
  return adm_check_launch("adm_gn_finalize_add");
}

extern "C" int adm_gn_finalize2(const float* partial0, int c0, int slabs0, const float* partial1, int c1, int slabs1,
                                const float* gamma, const float* beta, const float* film, int film_stride,
                                float* aff_a, float* aff_b, float* stats, int n, int hw, float eps, void* stream) {
  ADM_REQUIRE(partial0 && gamma && beta && aff_a && aff_b, ADM_E_ARG, "adm_gn_finalize2: null pointer");
  ADM_REQUIRE((partial1 != nullptr) == (c1 > 0), ADM_E_ARG, "adm_gn_finalize2: partial1/c1 mismatch");
  const int c = c0 + c1;
  ADM_REQUIRE(n > 0 && c0 > 0 && c % 32 == 0 && hw > 0 && slabs0 > 0 && (c1 == 0 || slabs1 > 0), ADM_E_SHAPE,
              "adm_gn_finalize2: bad shape");
  ADM_REQUIRE(!film || film_stride >= 2 * c, ADM_E_ARG, "adm_gn_finalize2: film_stride < 2*c");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(n, 4), dim3(256), 0, (hipStream_t)stream, partial0, c0, slabs0, partial1, c1,
                     slabs1, gamma, beta, film, film_stride, aff_a, aff_b, stats, hw, eps, (const float*)nullptr, 0);
  return adm_check_launch("adm_gn_finalize2");
}

extern "C" int adm_resample(const adm_bf16* in, const float* aff_a, const float* aff_b, adm_bf16* out, int n, int h,
                            int w, int c, int mode, void* stream) {
  ADM_REQUIRE(in && out, ADM_E_ARG, "adm_resample: null pointer");
  ADM_REQUIRE((aff_a != nullptr) == (aff_b != nullptr), ADM_E_ARG, "adm_resample: aff_a/aff_b go together");
  ADM_REQUIRE(mode >= 1 && mode <= 4, ADM_E_ARG, "adm_resample: mode must be 1 (avgpool2), 2 (nearest x2), 3 (stride-2 subsample) or 4 (zero-insert x2)");
  ADM_REQUIRE(mode != 4 || !aff_a, ADM_E_ARG, "adm_resample: zero-insert takes no affine");
  ADM_REQUIRE(n > 0 && h > 0 && w > 0 && c % 8 == 0, ADM_E_SHAPE, "adm_resample: bad shape");
  ADM_REQUIRE(mode == 2 || mode == 4 || (h % 2 == 0 && w % 2 == 0), ADM_E_SHAPE, "adm_resample: odd size for a 2x reduction");
  ADM_REQUIRE(adm_aligned16(in) && adm_aligned16(out) && adm_aligned16(aff_a) && adm_aligned16(aff_b), ADM_E_ALIGN,
              "adm_resample: unaligned pointer");
  const int oh = (mode == 2 || mode == 4) ? h * 2 : h / 2, ow = (mode == 2 || mode == 4) ? w * 2 : w / 2;
  const long long items = (long long)n * oh * ow * (c / 8);
  int blocks = (int)((items + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t s = (hipStream_t)stream;
  const bool act = aff_a != nullptr;
#define LAUNCH(M, A) hipLaunchKernelGGL((resample_kernel<M, A>), dim3(blocks), dim3(256), 0, s, in, aff_a, aff_b, out, n, h, w, c)
  if (mode == 1) { if (act) LAUNCH(1, true); else LAUNCH(1, false); }
  else if (mode == 2) { if (act) LAUNCH(2, true); else LAUNCH(2, false); }
  else if (mode == 3) { if (act) LAUNCH(3, true); else LAUNCH(3, false); }
  else           LAUNCH(4, false);
#undef LAUNCH
  return adm_check_launch("adm_resample");
}
