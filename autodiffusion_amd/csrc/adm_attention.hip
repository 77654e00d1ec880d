// K5: QKVAttention / QKVAttentionLegacy as a flash-style MFMA kernel (gfx950).
//
// Replaces reference guided_diffusion/unet.py:361-393 (new order) and :328-358 (legacy order):
//   w = softmax_fp32((q*s) . (k*s)), s = ch^-1/4;  a = w . v
// without materialising the [T, T] weight matrix.  Input is the token-major output of the qkv 1x1
// projection, bf16 [N][T][3*H*D]; output bf16 [N][T][H*D].
//
// Formulation (all MFMAs are v_mfma_f32_16x16x32_bf16):
//   S^T[key][query] = K . Q^T        A = K rows from LDS, B = Q^T held in registers
//   online softmax over keys: keys live in the 4 accumulator registers and the 4 lane quarters of
//   a query column, so the row max/sum need two cross-lane shuffles (xor 16, 32) per tile
//   O^T[d][query] += V^T . P^T       B = P^T taken straight from the S^T accumulators (the k-order
//   of the contraction is permuted identically on the V^T side, so P never touches LDS);
//   A = V^T read from the row-major LDS tile with the transposing load ds_read_b64_tr_b16
// Block = 4 waves x 32 queries; K/V tiles of 64 keys shared through LDS.
#include <stdlib.h>

#include <type_traits>

#include "adm_attn_common.h"

namespace {

constexpr int KT = 64;        // keys per tile
constexpr int QW = 32;        // queries per wave
constexpr int QB = 128;       // queries per block
constexpr int PADE = 16;      // bf16 elements of row padding: row stride = D/2 + 8 dwords = 8 (mod 16), which keeps both the
                              // ds_read_b128 fragment reads and the ds_read_b64_tr_b16 transposing reads bank-conflict free

#ifdef ADM_ATTN_TIMING
// diagnostic build only (make timing): wave 0 of every block accumulates the shader cycles of each phase of the tile loop
__device__ unsigned long long adm_attn_timing_buf[8 * 65536];
#define ATT_T0() unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tlast = __builtin_amdgcn_s_memtime()
#define ATT_T(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tph[k] += tn_ - tlast; tlast = tn_; } while (0)
#define ATT_TEND() do { if (threadIdx.x == 0) { const unsigned b_ = blockIdx.x + gridDim.x * blockIdx.y; if (b_ < 65536) for (int i_ = 0; i_ < 8; ++i_) adm_attn_timing_buf[b_ * 8 + i_] = tph[i_]; } } while (0)
#else
#define ATT_T0() do {} while (0)
#define ATT_T(k) do {} while (0)
#define ATT_TEND() do {} while (0)
#endif

#ifndef ADM_ATTN_ABL
#define ADM_ATTN_ABL 0   // diagnostic builds only: drop one piece of the tile loop (results are then wrong) to price it
#endif
#define ABL_KEEP(x) asm volatile("" ::"v"(x))

struct AttnK {
  const uint16_t* qkv; uint16_t* out; float* lse;
  const uint16_t* kv;      // keys / values: [N][kv_rows][Ckv], the first Tk rows of an image are attended to
  int T, heads, C3, C;     // T queries per image, row pitch C3 of the query tensor, C = heads * D output columns
  int Tk, kv_rows, Ckv;
  int q_off, k_off, v_off, head_stride, kv_head_stride;
  float scale_log2;  // log2(e) / sqrt(D)
};

// Second build of the tuned kernel (round 2).  Per 64-key tile a wave issues 32 MFMAs (512 cycles of matrix pipe); the
// first build also issued ~130 VALU instructions + 32 v_exp_f32 around them, at 164 registers = 3 waves per SIMD.
// Ablations on MI355X (tools/attn_ablate.py: drop one piece, time the rest) price every piece of the loop at 15-25 % of
// its time with little overlap between them: the loop is LATENCY-bound (LDS round trips, MFMA drain, global loads,
// barrier), so what pays is fewer dependent instructions per tile at the same or higher occupancy:
//   * the running-max subtraction leaves the per-tile stream: -m is the INITIAL ACCUMULATOR of the S chains, so the
//     MFMA result is s - m; m moves only when a lane sees s - m above MOVE_THR (wave-uniform branch: the cross-lane
//     max, the rescale of O and the re-centring of S run in a handful of tiles instead of all of them);
//     p <= 2^MOVE_THR keeps P well inside bf16 / fp32 range and the first tile pins l >= 1;
//   * the row sum l = P . 1 rides the matrix pipe (an all-ones A fragment: 4 MFMAs per tile instead of 16 packed
//     adds and two cross-lane reductions): it is the sum of the SAME bf16 P that multiplies V, rescaled with O;
//   * the tile body is instantiated once for full tiles and once for the last, ragged one (hipcc had hoisted the
//     32 compares + selects of the masking into every iteration);
//   * K and V^T fragments roll through two register sets each instead of being read up front (-48 registers): with the
//     above the 64-wide instantiation still fits 168 registers = 3 waves per SIMD.
// Measured and dropped: issuing S(i+1) before the softmax of tile i (two S sets, loop unrolled by two): 256 registers =
// 2 waves per SIMD, 586 vs 625 TFLOP/s -- an in-order wave cannot overlap its own MFMA burst with its VALU work unless
// the two are interleaved instruction by instruction, and the third wave hides more.
// The logit scale stays an fp32 multiply in front of v_exp_f32 (one v_mul per score): folding it into a bf16 Q costs
// |s| * 2^-9 of absolute error in the exponent, which the large-logit test sees.
// launch bounds: >= 2 waves per SIMD for every width caps the kernel at 256 registers, so hipcc keeps the MFMA accumulators in
// VGPRs (with no bound it parks them in AGPRs and copies them out with v_accvgpr_read: slower, and the 80-wide
// instantiation then returned wrong scores for one query tile); 3 waves (168 registers) up to 64-wide heads
#ifndef ADM_ATTN_WAVES
#define ADM_ATTN_WAVES 3
#endif
template <int D>
__global__ void __launch_bounds__(256, D <= 64 ? ADM_ATTN_WAVES : 2)
attn_kernel(const AttnK p) {
  constexpr int KS = D / 32;   // 32-deep k-steps of QK^T
  constexpr bool R16 = (D % 32) == 16;  // plus one 16-deep step (v_mfma_f32_16x16x16_bf16): head widths 48, 80, 112
  static_assert(D % 16 == 0 && KS >= 1, "head width must be a multiple of 16, at least 32");
  constexpr int DT = D / 16;   // d tiles of the output
  constexpr float MOVE_THR = 8.0f;  // log2 units: the running max follows once some (s - m) * scale_log2 exceeds it
  // LDS row pitch = 8 mod 16 dwords (conflict-free for the b128 fragment reads and the transposing reads): 48- and
  // 80-wide heads have it unpadded (24 / 40 dwords), the multiples of 32 need the 16-element pad
  // 64-wide heads (every ADM / classifier attention of the benchmarked path): K and V tiles go global -> LDS by LDS-DMA
  // (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write, 16 registers fewer).  A DMA piece is lane-linear
  // (1 KB = 8 rows of 128 B), so the rows are unpadded and the bank spread comes from an XOR swizzle instead: the
  // 16-byte segment s of row r is stored in slot s ^ (r & 7) -- applied on the GLOBAL side of the DMA (each lane fetches
  // the segment that belongs in its slot) and on the LDS side of every fragment read; conflict-free for the
  // ds_read_b128 K fragments and for the ds_read_b64_tr_b16 V^T reads.
  constexpr bool DMA = D == 64;
  constexpr int KROW = DMA ? D : D + (((D / 2) % 16 == 8) ? 0 : PADE);
  // K and V tiles row-major, double-buffered: the next tile's global loads fly during this tile's MFMAs
  __shared__ __attribute__((aligned(16))) uint16_t Ks[2][KT * KROW];
  __shared__ __attribute__((aligned(16))) uint16_t Vs[2][KT * KROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  int bx, by;
  adm_xcd_block(bx, by);
  const int n = by / p.heads, hd = by % p.heads;
  const int qbase = bx * QB + wave * QW;
  const uint16_t* base = p.qkv + (long long)n * p.T * p.C3;
  const int qcol = p.q_off + hd * p.head_stride, kcol = p.k_off + hd * p.kv_head_stride,
            vcol = p.v_off + hd * p.kv_head_stride;
  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kv + (long long)n * p.kv_rows * p.Ckv), 0,
                                                                       p.Tk * p.Ckv * 2, 0x00020000);

  // one descriptor over this image's T rows: queries / keys beyond T read as zeros (no bounds branches)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, p.T * p.C3 * 2, 0x00020000);
  // Q^T fragments: lane (query lc, quarter lq) holds Q[query][ks*32 + 8*lq .. +8]
  adm_h8 qf[2][KS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qbase + qt * 16 + lc;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const adm_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (q * p.C3 + qcol + ks * 32 + lq * 8) * 2, 0, 0);
      qf[qt][ks] = __builtin_bit_cast(adm_h8, v);
    }
  }
  adm_s16x4 qf16[2] = {};  // the 16-deep tail: lane (query lc, quarter lq) holds Q[query][32*KS + 4*lq .. +4]
  if constexpr (R16) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int q = qbase + qt * 16 + lc;
      const adm_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (q * p.C3 + qcol + KS * 32 + lq * 4) * 2, 0, 0);
      qf16[qt] = __builtin_bit_cast(adm_s16x4, v);
    }
  }

  f32x4 oacc[DT][2], lacc[2], cin[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    lacc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    cin[qt] = f32x4{0.f, 0.f, 0.f, 0.f};   // -m of the query column (raw logit units): the initial accumulator of its S chains
  }
  float m_run[2] = {0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const adm_h8 ones = __builtin_bit_cast(adm_h8, s16x8{ADM_ONE16, ADM_ONE16, ADM_ONE16, ADM_ONE16, ADM_ONE16, ADM_ONE16, ADM_ONE16, ADM_ONE16});
  const float thr_raw = MOVE_THR / p.scale_log2;

  const int ntiles = (p.Tk + KT - 1) / KT;
  ATT_T0();
  AdmTileRegs<KT, D, 256> kr, vr;
  // LDS-DMA of key tile kt0 into ring slot kt0 & 1: each wave moves pieces 2w, 2w + 1 of K and of V (hand-written: hipcc
  // orders EVERY later LDS access behind a pending LDS-DMA builtin, i.e. it would wait for the next tile's data before
  // reading this tile's fragments; an asm statement is invisible to that pass, so the loop counts the DMA itself --
  // s_waitcnt vmcnt(0) + barrier at the bottom of the tile that issued it; no other vector-memory operation is in flight
  // inside the loop)
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4s;
  const unsigned long long kvbase = (unsigned long long)(p.kv + (long long)n * p.kv_rows * p.Ckv);
  const u32x4s rsk_s = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kvbase),
                        (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(kvbase >> 32)) & 0xffffu,
                        (unsigned)__builtin_amdgcn_readfirstlane(p.Tk * p.Ckv * 2), 0x00020000u};
  const unsigned lds_k = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)&Ks[0][0];
  const unsigned lds_v = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)&Vs[0][0];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_tile = [&](int kt0) {
    const unsigned slot = (unsigned)(kt0 & 1) * (KT * KROW * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int piece = wave_u * 2 + j;
      const int row = kt0 * KT + piece * 8 + (lane >> 3);
      const int gseg = (lane & 7) ^ (lane >> 3);
      const unsigned voff_k = (unsigned)(row * p.Ckv + kcol + gseg * 8) * 2u, voff_v = (unsigned)(row * p.Ckv + vcol + gseg * 8) * 2u;
      const unsigned dst_k = lds_k + slot + piece * 1024, dst_v = lds_v + slot + piece * 1024;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff_k), "s"(rsk_s), "s"(dst_k) : "memory");
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff_v), "s"(rsk_s), "s"(dst_v) : "memory");
    }
  };
  if constexpr (DMA) {
    dma_tile(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    kr.load_buf(rsk, p.Ckv, kcol, 0, tid);
    vr.load_buf(rsk, p.Ckv, vcol, 0, tid);
    kr.store(Ks[0], KROW, tid);
    vr.store(Vs[0], KROW, tid);
  }
  __syncthreads();

  // S - m of key tile `kt0` (its K is in ring slot kt0 & 1): 4 key tiles x 2 query tiles; scheduling fences pin the
  // fragment reads one key tile ahead of their MFMAs (hipcc otherwise sinks each LDS read next to its use)
  auto s_phase = [&](int kt0, f32x4 (&st)[4][2]) {
    const uint16_t* Kc = Ks[kt0 & 1];
    // K fragments roll through two register sets (read tile kt + 1 before the MFMAs of tile kt): 16 live registers
    // instead of 32, which with the rolling V^T fragments below brings the kernel under 168 registers = three waves
    // per SIMD; the loop is latency-bound, and the third wave is worth more than the wider read-ahead.
    // (Head widths with a 16-deep tail keep all four fragment sets: they are not on a hot path.)
    constexpr int NSET = R16 ? 4 : 2;
    adm_h8 kfr[NSET][KS];
    adm_s16x4 kfr16[NSET] = {};
    auto kread = [&](int kt) {
      const int slot = kt % NSET;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#if ADM_ATTN_ABL == 5
        kfr[slot][ks] = qf[0][ks];
#else
        kfr[slot][ks] = *reinterpret_cast<const adm_h8*>(
            &Kc[(kt * 16 + lc) * KROW + (DMA ? ((ks * 4 + lq) ^ (lc & 7)) * 8 : ks * 32 + lq * 8)]);
#endif
      if constexpr (R16) kfr16[slot] = *reinterpret_cast<const adm_s16x4*>(&Kc[(kt * 16 + lc) * KROW + KS * 32 + lq * 4]);
    };
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (R16) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) kread(kt);
    } else {
      kread(0);
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if (!R16 && kt + 1 < 4) kread(kt + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
        st[kt][qt] = adm_mfma_16x16x32(kfr[kt % NSET][0], qf[qt][0], cin[qt], 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < KS; ++ks)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          st[kt][qt] = adm_mfma_16x16x32(kfr[kt % NSET][ks], qf[qt][ks], st[kt][qt], 0, 0, 0);
      if constexpr (R16) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          st[kt][qt] = adm_mfma_16x16x16(kfr16[kt % NSET], qf16[qt], st[kt][qt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // one key tile.  RAG = the tile holds rows beyond Tk (only the last one can), FIRST = no running max yet
  auto tile = [&](int kt0, auto rag_, auto first_) {
    constexpr bool RAG = decltype(rag_)::value, FIRST = decltype(first_)::value;
    const int k0 = kt0 * KT;
#if ADM_ATTN_ABL == 6
    const bool next = false;
#else
    const bool next = kt0 + 1 < ntiles;
#endif
    if (next) {
      if constexpr (DMA) {
        dma_tile(kt0 + 1);
      } else {
        kr.load_buf(rsk, p.Ckv, kcol, k0 + KT, tid);
        vr.load_buf(rsk, p.Ckv, vcol, k0 + KT, tid);
      }
    }
    ATT_T(0);
    f32x4 st[4][2];
    s_phase(kt0, st);
    ATT_T(1);
    const uint16_t* Vc = Vs[kt0 & 1];
    __builtin_amdgcn_sched_barrier(0);
    // V^T fragments (transposing LDS read) roll through two register sets, one output d-tile ahead of the MFMAs; the
    // first set is requested here so that its latency hides under the softmax VALU work
    adm_h8 vfr[2][2];
    auto vread = [&](int dt, int slot) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#if ADM_ATTN_ABL == 4
        vfr[slot][kb] = qf[kb][0];
#else
        vfr[slot][kb] = DMA ? adm_tr_frag_swz(Vc, kb * 32, dt * 16, lc, lq) : adm_tr_frag(Vc, KROW, kb * 32, dt * 16, lc, lq);
#endif
    };
    vread(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    ATT_T(2);
    if constexpr (RAG) {  // keys beyond Tk (their K rows read as zeros): weight 0
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + kt * 16 + lq * 4 + r >= p.Tk) st[kt][qt][r] = -1e30f;
    }
    // ---- the running max moves only when it has to: per-lane maxima of both query tiles (two independent chains),
    //      one wave-uniform test
    float mxl[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float mx = st[0][qt][0];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][qt][r]);
      mxl[qt] = mx;
    }
#if ADM_ATTN_ABL == 8
    if (FIRST) {
#else
    if (FIRST || __any(fmaxf(mxl[0], mxl[1]) > thr_raw)) {
#endif
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float mx = adm_quarter_max(mxl[qt]);
        const float delta = FIRST ? mx : fmaxf(mx, 0.f);   // m_new = m + delta (the first tile sets the baseline)
        m_run[qt] += delta;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) st[kt][qt] -= delta;
        cin[qt] -= delta;
        if constexpr (!FIRST) {  // O and l are still zero in the first tile
          const float alpha = __builtin_amdgcn_exp2f(-delta * p.scale_log2);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) oacc[dt][qt] *= alpha;
          lacc[qt] *= alpha;
        }
      }
    }
    ATT_T(3);
    // ---- P = 2^((S - m) * scale_log2), rounded to bf16 straight into the B fragments of the second product
    //      (this file is built with -fno-slp-vectorize: hipcc otherwise pairs the multiplies into v_pk_mul_f32, which
    //      issues slower than two single multiplies beside MFMAs)
    adm_h8 pf[2][2];  // [query tile][32-key block]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        adm_h8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e)
#if ADM_ATTN_ABL == 1
          f[e] = __builtin_bit_cast(adm_elem_t, (uint16_t)(__float_as_uint(st[2 * kb + (e >> 2)][qt][e & 3]) >> 16));
#elif ADM_ATTN_ABL == 9
          f[e] = (adm_elem_t)(st[2 * kb + (e >> 2)][qt][e & 3] * p.scale_log2);
#else
          f[e] = (adm_elem_t)__builtin_amdgcn_exp2f(st[2 * kb + (e >> 2)][qt][e & 3] * p.scale_log2);
#endif
        pf[qt][kb] = f;
      }
    // ---- O^T += V^T . P^T and l += 1 . P^T; contraction slot k = 8*lq + e  <->  key kb*32 + 16*(e>>2) + 4*lq + (e&3)
    __builtin_amdgcn_sched_barrier(0);
    ATT_T(4);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) lacc[qt] = adm_mfma_16x16x32(ones, pf[qt][kb], lacc[qt], 0, 0, 0);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      if (dt + 1 < DT) vread(dt + 1, (dt + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          oacc[dt][qt] = adm_mfma_16x16x32(vfr[dt & 1][kb], pf[qt][kb], oacc[dt][qt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    ATT_T(5);
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next tile have landed
    } else if (next) {
      kr.store(Ks[(kt0 + 1) & 1], KROW, tid);
      vr.store(Vs[(kt0 + 1) & 1], KROW, tid);
    }
    ATT_T(6);
#if ADM_ATTN_ABL != 7
    __syncthreads();
#endif
    ATT_T(7);
  };
  using TT = std::true_type;
  using FF = std::false_type;
  const bool ragged = (p.Tk % KT) != 0;
  if (ntiles == 1) {
    if (ragged) tile(0, TT{}, TT{}); else tile(0, FF{}, TT{});
  } else {
    tile(0, FF{}, TT{});
    for (int kt0 = 1; kt0 + 1 < ntiles; ++kt0) tile(kt0, FF{}, FF{});
    if (ragged) tile(ntiles - 1, TT{}, FF{}); else tile(ntiles - 1, FF{}, FF{});
  }

  ATT_TEND();
  // ---- normalise and store: lane holds d = dt*16 + 4*lq .. +3 of query lc; every row of lacc is the row sum
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float l = lacc[qt][0];
    const float inv = 1.0f / l;
    const int q = qbase + qt * 16 + lc;
    if (q >= p.T) continue;
    // log2-domain log-sum-exp of the scaled logits: P = exp2(s * scale_log2 - lse)
    if (p.lse && lq == 0) p.lse[((long long)n * p.heads + hd) * p.T + q] = m_run[qt] * p.scale_log2 + log2f(l);
    uint16_t* orow = p.out + ((long long)n * p.T + q) * p.C + hd * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const f32x4 o = oacc[dt][qt] * inv;
      uint2 pk;
      pk.x = adm_pack2(o[0], o[1]);
      pk.y = adm_pack2(o[2], o[3]);
      *reinterpret_cast<uint2*>(orow + dt * 16 + lq * 4) = pk;
    }
  }
}

// ---- wide heads (D = 192, 256: ADM-128's num_heads = 4 gives 128 / 192 / 256 channels per head).  Same
// formulation, sized for the register file instead of for speed: 16 queries per wave (64 per block), 32-key tiles
// single-buffered in LDS, K / V^T fragments streamed one tile at a time.  Not on the benchmarked path.
template <int D>
__global__ void __launch_bounds__(256)
attn_wide_kernel(const AttnK p) {
  constexpr int KT2 = 32, QB2 = 64;
  constexpr int KS = D / 32, DT = D / 16, KROW = D + PADE;
  __shared__ __attribute__((aligned(16))) uint16_t Ks[KT2 * KROW];
  __shared__ __attribute__((aligned(16))) uint16_t Vs[KT2 * KROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  int bx, by;
  adm_xcd_block(bx, by);
  const int n = by / p.heads, hd = by % p.heads;
  const int q = bx * QB2 + wave * 16 + lc;
  const uint16_t* base = p.qkv + (long long)n * p.T * p.C3;
  const int qcol = p.q_off + hd * p.head_stride, kcol = p.k_off + hd * p.kv_head_stride,
            vcol = p.v_off + hd * p.kv_head_stride;
  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kv + (long long)n * p.kv_rows * p.Ckv), 0,
                                                                       p.Tk * p.Ckv * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, p.T * p.C3 * 2, 0x00020000);

  adm_h8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const adm_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (q * p.C3 + qcol + ks * 32 + lq * 8) * 2, 0, 0);
    qf[ks] = __builtin_bit_cast(adm_h8, v);
  }
  f32x4 oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  AdmTileRegs<KT2, D, 256> kr, vr;
  const int ntiles = (p.Tk + KT2 - 1) / KT2;
  for (int kt0 = 0; kt0 < ntiles; ++kt0) {
    const int k0 = kt0 * KT2;
    kr.load_buf(rsk, p.Ckv, kcol, k0, tid);
    vr.load_buf(rsk, p.Ckv, vcol, k0, tid);
    __syncthreads();  // the previous tile's readers are done
    kr.store(Ks, KROW, tid);
    vr.store(Vs, KROW, tid);
    __syncthreads();
    // S^T = K . Q^T: 2 key tiles x 1 query tile
    f32x4 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const adm_h8 kf = *reinterpret_cast<const adm_h8*>(&Ks[(kt * 16 + lc) * KROW + ks * 32 + lq * 8]);
        st[kt] = adm_mfma_16x16x32(kf, qf[ks], st[kt], 0, 0, 0);
      }
    }
    if (k0 + KT2 > p.Tk) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (k0 + kt * 16 + lq * 4 + r >= p.Tk) st[kt][r] = -1e30f;
    }
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
    mx = adm_quarter_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
    const float mneg = -m_new * p.scale_log2;
    m_run = m_new;
    float psum = 0.f;
    adm_h8 pf;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(st[kt][r] * p.scale_log2 + mneg);
        psum += e;
        pf[kt * 4 + r] = (adm_elem_t)e;
      }
    l_run = l_run * alpha + psum;
    // O^T += V^T . P^T; contraction slot k = 8*lq + e <-> key 16*(e>>2) + 4*lq + (e&3)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const adm_h8 vf = adm_tr_frag(Vs, KROW, 0, dt * 16, lc, lq);
      oacc[dt] = adm_mfma_16x16x32(vf, pf, oacc[dt] * alpha, 0, 0, 0);
    }
  }
  const float l = adm_quarter_sum(l_run);
  const float inv = 1.0f / l;
  if (q >= p.T) return;
  if (p.lse && lq == 0) p.lse[((long long)n * p.heads + hd) * p.T + q] = m_run * p.scale_log2 + log2f(l);
  uint16_t* orow = p.out + ((long long)n * p.T + q) * p.C + hd * D;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const f32x4 o = oacc[dt] * inv;
    uint2 pk;
    pk.x = adm_pack2(o[0], o[1]);
    pk.y = adm_pack2(o[2], o[3]);
    *reinterpret_cast<uint2*>(orow + dt * 16 + lq * 4) = pk;
  }
}

int launch_attention(const AttnK& k, int n, int t, int heads, int d, hipStream_t s);

}  // namespace

extern "C" int adm_attention_lse(const adm_bf16* qkv, adm_bf16* out, float* lse, int n, int t, int heads, int d,
                                 int new_order, void* stream);

#ifdef ADM_ATTN_TIMING
extern "C" int adm_attn_timing_read(unsigned long long* host, int nblocks) {
  if (nblocks > 65536) nblocks = 65536;
  void* dptr = nullptr;
  if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(adm_attn_timing_buf)) != hipSuccess) return -1;
  if (hipMemcpy(host, dptr, (size_t)nblocks * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -2;
  return (int)hipMemset(dptr, 0, sizeof(unsigned long long) * 8 * 65536);
}
#endif

extern "C" int adm_attention(const adm_bf16* qkv, adm_bf16* out, int n, int t, int heads, int d, int new_order,
                             void* stream) {
  return adm_attention_lse(qkv, out, nullptr, n, t, heads, d, new_order, stream);
}

extern "C" int adm_attention_lse(const adm_bf16* qkv, adm_bf16* out, float* lse, int n, int t, int heads, int d,
                                 int new_order, void* stream) {
  ADM_REQUIRE(qkv && out, ADM_E_ARG, "adm_attention: null pointer");
  ADM_REQUIRE(n > 0 && t > 0 && heads > 0, ADM_E_ARG, "adm_attention: bad shape n=%d t=%d heads=%d", n, t, heads);
  ADM_REQUIRE(d == 32 || d == 48 || d == 64 || d == 80 || d == 96 || d == 128 || d == 160 || d == 192 || d == 256, ADM_E_SHAPE,
              "adm_attention: head dim %d unsupported (32, 48, 64, 80, 96, 128, 160, 192, 256)", d);
  ADM_REQUIRE(adm_aligned16(qkv) && adm_aligned16(out), ADM_E_ALIGN, "adm_attention: unaligned pointer");
  ADM_REQUIRE((long long)n * heads < 65536, ADM_E_SHAPE, "adm_attention: n*heads exceeds grid.y");
  AttnK k{};
  k.qkv = qkv; k.out = out; k.lse = lse; k.T = t; k.heads = heads;
  k.C = heads * d; k.C3 = 3 * k.C;
  if (new_order) { k.q_off = 0; k.k_off = k.C; k.v_off = 2 * k.C; k.head_stride = d; }
  else           { k.q_off = 0; k.k_off = d;   k.v_off = 2 * d;   k.head_stride = 3 * d; }
  k.kv = qkv; k.Tk = t; k.kv_rows = t; k.Ckv = k.C3; k.kv_head_stride = k.head_stride;
  k.scale_log2 = 1.4426950408889634f / sqrtf((float)d);
  return launch_attention(k, n, t, heads, d, (hipStream_t)stream);
}

extern "C" int adm_attention_cross(const adm_bf16* q, int q_stride, const adm_bf16* kv, int kv_stride, int kv_rows,
                                   adm_bf16* out, int n, int tq, int tk, int heads, int d, float scale, void* stream) {
  ADM_REQUIRE(q && kv && out, ADM_E_ARG, "adm_attention_cross: null pointer");
  ADM_REQUIRE(n > 0 && tq > 0 && tk > 0 && heads > 0 && kv_rows >= tk, ADM_E_ARG,
              "adm_attention_cross: bad shape n=%d tq=%d tk=%d kv_rows=%d heads=%d", n, tq, tk, kv_rows, heads);
  ADM_REQUIRE(d == 32 || d == 48 || d == 64 || d == 80 || d == 96 || d == 128 || d == 160 || d == 192 || d == 256, ADM_E_SHAPE,
              "adm_attention_cross: head dim %d unsupported (32, 48, 64, 80, 96, 128, 160, 192, 256)", d);
  ADM_REQUIRE(q_stride >= heads * d && kv_stride >= 2 * heads * d && q_stride % 8 == 0 && kv_stride % 8 == 0, ADM_E_SHAPE,
              "adm_attention_cross: row pitches %d / %d too small or not multiples of 8", q_stride, kv_stride);
  ADM_REQUIRE(adm_aligned16(q) && adm_aligned16(kv) && adm_aligned16(out), ADM_E_ALIGN, "adm_attention_cross: unaligned pointer");
  ADM_REQUIRE((long long)n * heads < 65536, ADM_E_SHAPE, "adm_attention_cross: n*heads exceeds grid.y");
  ADM_REQUIRE((long long)tq * q_stride < (1ll << 30) && (long long)tk * kv_stride < (1ll << 30), ADM_E_SHAPE,
              "adm_attention_cross: an image's rows exceed the 32-bit byte offsets of a buffer descriptor");
  AttnK k{};
  k.qkv = q; k.out = out; k.lse = nullptr; k.T = tq; k.heads = heads;
  k.C = heads * d; k.C3 = q_stride;
  k.q_off = 0; k.head_stride = d;
  k.kv = kv; k.Tk = tk; k.kv_rows = kv_rows; k.Ckv = kv_stride;
  k.k_off = 0; k.v_off = heads * d; k.kv_head_stride = d;
  k.scale_log2 = 1.4426950408889634f * (scale > 0.f ? scale : 1.0f / sqrtf((float)d));
  return launch_attention(k, n, tq, heads, d, (hipStream_t)stream);
}

namespace {
int launch_attention(const AttnK& k, int n, int t, int heads, int d, hipStream_t s) {
  dim3 grid((t + QB - 1) / QB, n * heads);
  if (d > 128) {
    dim3 gridw((t + 63) / 64, n * heads);
    if (d == 160) hipLaunchKernelGGL((attn_wide_kernel<160>), gridw, dim3(256), 0, s, k);
    else if (d == 192) hipLaunchKernelGGL((attn_wide_kernel<192>), gridw, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((attn_wide_kernel<256>), gridw, dim3(256), 0, s, k);
    return adm_check_launch("adm_attention");
  }
  if (d == 32) hipLaunchKernelGGL((attn_kernel<32>), grid, dim3(256), 0, s, k);
  else if (d == 48) hipLaunchKernelGGL((attn_kernel<48>), grid, dim3(256), 0, s, k);
  else if (d == 80) hipLaunchKernelGGL((attn_kernel<80>), grid, dim3(256), 0, s, k);
  else if (d == 64) hipLaunchKernelGGL((attn_kernel<64>), grid, dim3(256), 0, s, k);
  else if (d == 96) hipLaunchKernelGGL((attn_kernel<96>), grid, dim3(256), 0, s, k);
  else hipLaunchKernelGGL((attn_kernel<128>), grid, dim3(256), 0, s, k);
  return adm_check_launch("adm_attention");
}
}  // namespace
