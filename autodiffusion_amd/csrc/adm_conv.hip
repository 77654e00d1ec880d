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
//   block tile  BM = 256 pixels (a TH x TW patch of TI images) x BN output channels, 4 waves
//   K loop      over 32-channel chunks of the input; per chunk the (TH+2)x(TW+2) halo patch is
//               fetched ONCE from HBM/L2 (register-staged, 16 B per lane), the per-(image,channel)
//               affine + SiLU is applied in fp32, the result is rounded to bf16 and parked in LDS
//               (96-byte pixel rows: conflict-free ds_read_b128 for the 16-pixel fragment), and
//               then reused by all 9 taps; zero padding is applied after the activation
//   weights     one [BN x 32] tile per (chunk, tap), double-buffered in LDS, prefetched a step ahead
#include "adm_common.h"

namespace {

constexpr int KC = 32;           // input channels per chunk (= one MFMA K)
constexpr int ROWB = 96;         // LDS bytes per halo pixel: 64 data + 32 pad
constexpr int HALO_MAX = 400;    // 4 images x (8+2)^2, or 1 image x (16+2)^2 = 324
constexpr int TI_MAX = 4;

struct ConvK {
  const uint16_t* in0; const uint16_t* in1; const uint16_t* w; const uint16_t* res;
  const float* bias; const float* aa; const float* ab; void* out;
  int N, H, W, C0, C1, Cout;
  int TH, TW, TI, tiles_x, tiles_y;
  int taps, prologue, out_mode;
  int ntiles16, nblocks_n;
};

template <int BN>
constexpr int conv_smem_bytes() { return HALO_MAX * ROWB + 2 * BN * 64 + 2 * TI_MAX * 64 * 4; }

template <int WM, int WN, int TM, int TN, int OCC>
__global__ void __launch_bounds__(64 * WM * WN, OCC)
conv_kernel(const ConvK p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BN = WN * TN * 16;
  constexpr int PASSES = (HALO_MAX * 4 + NT - 1) / NT;
  constexpr int BUNITS = BN * 4;                       // 16-byte units per weight tile
  constexpr int BPASS = (BUNITS + NT - 1) / NT;
  static_assert(WM * TM * 16 == 256, "pixel tile is 256");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const halo = smem;
  unsigned char* const bbuf = smem + HALO_MAX * ROWB;
  float* const abuf = reinterpret_cast<float*>(bbuf + 2 * BN * 64);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lc = lane & 15, lq = lane >> 4;

  const int nb = blockIdx.x % p.nblocks_n;
  const int mt = blockIdx.x / p.nblocks_n;
  const int PAD = p.taps == 9 ? 1 : 0;
  const int HW2 = p.TW + 2 * PAD;
  const int HPI = (p.TH + 2 * PAD) * HW2;
  const int HP = p.TI * HPI;
  const int Cin = p.C0 + p.C1;

  int img0, y0, x0;
  if (p.TI == 1) {
    const int per_img = p.tiles_x * p.tiles_y;
    img0 = mt / per_img;
    const int r = mt % per_img;
    y0 = (r / p.tiles_x) * p.TH;
    x0 = (r % p.tiles_x) * p.TW;
  } else {
    img0 = mt * p.TI;
    y0 = 0;
    x0 = 0;
  }

  // ---- halo staging geometry: 16-byte segment s = tid + pass*NT -> (halo pixel s>>2, segment s&3)
  int pixoff[PASSES];      // global pixel index or -1 (outside the image / batch -> zero padding)
  unsigned ti_pack = 0;    // image-in-tile of each pass, 4 bits each
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int s = tid + ps * NT;
    const int hp = s >> 2;
    int off = -1;
    if (hp < HP) {
      const int ti = hp / HPI, rem = hp % HPI;
      const int n = img0 + ti, y = y0 + rem / HW2 - PAD, x = x0 + rem % HW2 - PAD;
      if (n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W) off = (n * p.H + y) * p.W + x;
      ti_pack |= (unsigned)ti << (4 * ps);
    }
    pixoff[ps] = off;
  }
  const int seg = tid & 3;  // NT % 4 == 0 -> the same channel segment in every pass

  // ---- MFMA fragment addressing
  int abase[TM];            // LDS byte offset of this lane's pixel row (+ its 16-byte k slice)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = (wm * TM + i) * 16 + lc;
    const int ti = m / (p.TH * p.TW), rem = m % (p.TH * p.TW);
    abase[i] = (ti * HPI + (rem / p.TW) * HW2 + rem % p.TW) * ROWB + lq * 16;
  }
  const int wbase = ((wn * TN) * 64 + lane) * 16;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int chunks = Cin / KC;
  const int nsteps = chunks * p.taps;
  const long long wstep = (long long)p.ntiles16 * 512;  // bf16 elements per K-step in the packed image
  const int tile0 = nb * (BN / 16);

  uint4 hreg[PASSES];
  uint4 breg[BPASS];

  auto load_halo = [&](int c) {
    const int cc = c * KC;
    const uint16_t* src;
    int cs, co;
    if (cc < p.C0) { src = p.in0; cs = p.C0; co = cc; } else { src = p.in1; cs = p.C1; co = cc - p.C0; }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (pixoff[ps] >= 0) v = *reinterpret_cast<const uint4*>(src + (long long)pixoff[ps] * cs + co + seg * 8);
      hreg[ps] = v;
    }
  };
  auto stage_affine = [&](int c, int buf) {
    if (p.prologue == 0) return;
    if (tid < p.TI * 16) {
      const int ti = tid >> 4, part = tid & 15;
      const int n = img0 + ti;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < p.N) {
        const float* s = (part < 8 ? p.aa : p.ab) + (long long)n * Cin + c * KC + (part & 7) * 4;
        v = *reinterpret_cast<const float4*>(s);
      }
      *reinterpret_cast<float4*>(abuf + (buf * TI_MAX + ti) * 64 + (part < 8 ? 0 : 32) + (part & 7) * 4) = v;
    }
  };
  auto write_halo = [&](int buf) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int s = tid + ps * NT;
      const int hp = s >> 2;
      if (hp >= HP) continue;
      uint4 v = hreg[ps];
      if (p.prologue != 0 && pixoff[ps] >= 0) {
        const int ti = (ti_pack >> (4 * ps)) & 15;
        const float* ab = abuf + (buf * TI_MAX + ti) * 64 + seg * 8;
        float a8[8], b8[8];
        *reinterpret_cast<float4*>(a8) = *reinterpret_cast<const float4*>(ab);
        *reinterpret_cast<float4*>(a8 + 4) = *reinterpret_cast<const float4*>(ab + 4);
        *reinterpret_cast<float4*>(b8) = *reinterpret_cast<const float4*>(ab + 32);
        *reinterpret_cast<float4*>(b8 + 4) = *reinterpret_cast<const float4*>(ab + 36);
        uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float lo = __uint_as_float(u[j] << 16), hi = __uint_as_float(u[j] & 0xffff0000u);
          lo = a8[2 * j] * lo + b8[2 * j];
          hi = a8[2 * j + 1] * hi + b8[2 * j + 1];
          if (p.prologue == 2) { lo = adm_silu(lo); hi = adm_silu(hi); }
          u[j] = (uint32_t)adm_f32_to_bf16(lo) | ((uint32_t)adm_f32_to_bf16(hi) << 16);
        }
        v = make_uint4(u[0], u[1], u[2], u[3]);
      }
      *reinterpret_cast<uint4*>(halo + hp * ROWB + seg * 16) = v;
    }
  };
  auto load_b = [&](int step) {
    const uint16_t* src = p.w + (long long)step * wstep;
#pragma unroll
    for (int bp = 0; bp < BPASS; ++bp) {
      const int u = tid + bp * NT;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (u < BUNITS && tile0 + (u >> 6) < p.ntiles16) v = *reinterpret_cast<const uint4*>(src + (long long)(tile0 * 64 + u) * 8);
      breg[bp] = v;
    }
  };
  auto write_b = [&](int buf) {
#pragma unroll
    for (int bp = 0; bp < BPASS; ++bp) {
      const int u = tid + bp * NT;
      if (u < BUNITS) *reinterpret_cast<uint4*>(bbuf + buf * (BN * 64) + u * 16) = breg[bp];
    }
  };

  // ---- prologue
  load_halo(0);
  stage_affine(0, 0);
  load_b(0);
  write_b(0);
  __syncthreads();

  int step = 0;
  for (int c = 0; c < chunks; ++c) {
    write_halo(c & 1);
    if (c + 1 < chunks) {
      load_halo(c + 1);
      stage_affine(c + 1, (c + 1) & 1);
    }
    __syncthreads();
    for (int t = 0; t < p.taps; ++t, ++step) {
      const bool has_next = step + 1 < nsteps;
      if (has_next) load_b(step + 1);
      const int tapoff = p.taps == 9 ? ((t / 3) * HW2 + (t % 3)) * ROWB : 0;
      const unsigned char* bcur = bbuf + (step & 1) * (BN * 64) + wbase;
      bf16x8 af[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(halo + abase[i] + tapoff);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(bcur + j * 1024);
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[i], acc[i][j], 0, 0, 0);
      }
      if (has_next) write_b((step + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: lane (lc, lq) holds channels 4*lq..4*lq+3 of tile j for pixel lc of tile i
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = (wm * TM + i) * 16 + lc;
    const int ti = m / (p.TH * p.TW), rem = m % (p.TH * p.TW);
    const int n = img0 + ti, y = y0 + rem / p.TW, x = x0 + rem % p.TW;
    if (n >= p.N) continue;
    const long long pix = ((long long)n * p.H + y) * p.W + x;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ch0 = nb * BN + (wn * TN + j) * 16 + lq * 4;
      if (ch0 >= p.Cout) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (p.out_mode == 0 && ch0 + 3 < p.Cout) {
        const float4 bs = *reinterpret_cast<const float4*>(p.bias + ch0);
        v[0] += bs.x; v[1] += bs.y; v[2] += bs.z; v[3] += bs.w;
        if (p.res) {
          const uint2 r = *reinterpret_cast<const uint2*>(p.res + pix * p.Cout + ch0);
          v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
          v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
        }
        uint2 o;
        o.x = (uint32_t)adm_f32_to_bf16(v[0]) | ((uint32_t)adm_f32_to_bf16(v[1]) << 16);
        o.y = (uint32_t)adm_f32_to_bf16(v[2]) | ((uint32_t)adm_f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + pix * p.Cout + ch0) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ch = ch0 + e;
          if (ch >= p.Cout) continue;
          float r = v[e] + p.bias[ch];
          if (p.out_mode == 0) {
            if (p.res) r += adm_bf16_to_f32(p.res[pix * p.Cout + ch]);
            reinterpret_cast<uint16_t*>(p.out)[pix * p.Cout + ch] = adm_f32_to_bf16(r);
          } else {
            reinterpret_cast<float*>(p.out)[(((long long)n * p.Cout + ch) * p.H + y) * p.W + x] = r;
          }
        }
      }
    }
  }
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
    out[i] = adm_f32_to_bf16(v);
  }
}

template <int WM, int WN, int TM, int TN, int OCC>
int launch_conv(const ConvK& k, int m_tiles, hipStream_t s) {
  constexpr int BN = WN * TN * 16;
  constexpr int smem = conv_smem_bytes<BN>();
  static bool attr_set_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  bool& attr_set = attr_set_dev[dev & 63];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_kernel<WM, WN, TM, TN, OCC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) ADM_FAIL((int)e, "adm_conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  ConvK kk = k;
  kk.nblocks_n = (k.Cout + BN - 1) / BN;
  const long long blocks = (long long)m_tiles * kk.nblocks_n;
  ADM_REQUIRE(blocks < (1ll << 31), ADM_E_SHAPE, "adm_conv: grid too large");
  hipLaunchKernelGGL((conv_kernel<WM, WN, TM, TN, OCC>), dim3((unsigned)blocks), dim3(64 * WM * WN), smem, s, kk);
  return adm_check_launch("adm_conv");
}

}  // namespace

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

extern "C" int adm_conv(const adm_conv_args* a, void* stream) {
  ADM_REQUIRE(a, ADM_E_ARG, "adm_conv: null args");
  ADM_REQUIRE(a->in0 && a->w_packed && a->bias && a->out, ADM_E_ARG, "adm_conv: null pointer");
  ADM_REQUIRE((a->in1 != nullptr) == (a->c1 > 0), ADM_E_ARG, "adm_conv: in1/c1 mismatch");
  ADM_REQUIRE(a->taps == 1 || a->taps == 9, ADM_E_ARG, "adm_conv: taps must be 1 or 9");
  ADM_REQUIRE(a->prologue >= 0 && a->prologue <= 2, ADM_E_ARG, "adm_conv: prologue must be 0..2");
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

  ConvK k{};
  k.in0 = a->in0; k.in1 = a->in1; k.w = a->w_packed; k.res = a->res;
  k.bias = a->bias; k.aa = a->aff_a; k.ab = a->aff_b; k.out = a->out;
  k.N = a->n; k.H = a->h; k.W = a->w; k.C0 = a->c0; k.C1 = a->c1; k.Cout = a->cout;
  k.taps = a->taps; k.prologue = a->prologue; k.out_mode = a->out_mode;
  k.ntiles16 = (a->cout + 15) / 16;
  // 256-pixel patch: TW = min(W,16), TH = min(H, 256/TW); small maps batch TI images per tile
  const int BM = 256;
  k.TW = a->w < 16 ? a->w : 16;
  k.TH = a->h < BM / k.TW ? a->h : BM / k.TW;
  ADM_REQUIRE(BM % (k.TH * k.TW) == 0 && a->h % k.TH == 0 && a->w % k.TW == 0, ADM_E_SHAPE,
              "adm_conv: %dx%d feature map does not tile into %d-pixel patches", a->h, a->w, BM);
  k.TI = BM / (k.TH * k.TW);
  const int pad = a->taps == 9 ? 1 : 0;
  ADM_REQUIRE(k.TI <= TI_MAX && k.TI * (k.TH + 2 * pad) * (k.TW + 2 * pad) <= HALO_MAX, ADM_E_SHAPE,
              "adm_conv: %dx%d feature map too small for the halo tile (need >= 8x8)", a->h, a->w);
  k.tiles_x = a->w / k.TW;
  k.tiles_y = a->h / k.TH;
  const int m_tiles = k.TI == 1 ? a->n * k.tiles_x * k.tiles_y : (a->n + k.TI - 1) / k.TI;
  hipStream_t s = (hipStream_t)stream;

  int variant = a->variant;
  if (variant == 0) {
    if (a->cout <= 16) variant = 3;
    else if (a->cout <= 64) variant = 4;
    else {
      const int w128 = ((a->cout + 127) / 128) * 128, w96 = ((a->cout + 95) / 96) * 96;
      variant = (w96 <= w128) ? 2 : 1;
    }
  }
  switch (variant) {
    case 1: return launch_conv<2, 2, 8, 4, 1>(k, m_tiles, s);  // 256 x 128
    case 2: return launch_conv<2, 2, 8, 3, 2>(k, m_tiles, s);  // 256 x 96
    case 3: return launch_conv<4, 1, 4, 1, 2>(k, m_tiles, s);  // 256 x 16 (output head)
    case 4: return launch_conv<2, 2, 8, 2, 2>(k, m_tiles, s);  // 256 x 64
    default: ADM_FAIL(ADM_E_ARG, "adm_conv: unknown variant %d", variant);
  }
}
