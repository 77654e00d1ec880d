// ABI version + thread-local error text for libadm_hip.so.
#include <stdarg.h>

#include "adm_common.h"

static thread_local char g_err[512] = "";

void adm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int adm_abi_version(void) { return ADM_ABI_VERSION; }
extern "C" const char* adm_last_error(void) { return g_err; }


// ---------------------------------------------------------------- CU-partitioned streams
// A stream created with a CU mask runs its kernels on that subset of the chip only (hipExtStreamCreateWithCUMask).  The
// persistent kernels size their grid to the CUs they may occupy -- one block per CU slot -- so a masked stream must tell them how
// many CUs it owns: a small stream -> CU-count table, read at every persistent launch (a grid sized to the whole chip on a
// masked stream would queue its surplus blocks behind the resident ones and serialise the tile list's tail).
#include <mutex>

namespace {
struct StreamCus { void* stream; int ncu; };
StreamCus g_cus[32];
int g_ncus = 0;
std::mutex g_cus_mu;
}  // namespace

int adm_stream_cus(void* stream, int dflt) {
  if (g_ncus == 0) return dflt;   // the common case: no partitioned stream anywhere in the process
  std::lock_guard<std::mutex> lk(g_cus_mu);
  for (int i = 0; i < g_ncus; ++i)
    if (g_cus[i].stream == stream) return g_cus[i].ncu < dflt ? g_cus[i].ncu : dflt;
  return dflt;
}

extern "C" int adm_stream_set_cus(void* stream, int ncu) {
  ADM_REQUIRE(ncu >= 0, ADM_E_ARG, "adm_stream_set_cus: ncu must be >= 0 (0 forgets the stream)");
  std::lock_guard<std::mutex> lk(g_cus_mu);
  for (int i = 0; i < g_ncus; ++i) {
    if (g_cus[i].stream == stream) {
      if (ncu == 0) g_cus[i] = g_cus[--g_ncus]; else g_cus[i].ncu = ncu;
      return 0;
    }
  }
  if (ncu == 0) return 0;
  ADM_REQUIRE(g_ncus < 32, ADM_E_ARG, "adm_stream_set_cus: more than 32 partitioned streams");
  g_cus[g_ncus++] = StreamCus{stream, ncu};
  return 0;
}

extern "C" int adm_stream_create_cumask(const uint32_t* mask, int words, void** stream_out) {
  ADM_REQUIRE(mask && words > 0 && words <= 32 && stream_out, ADM_E_ARG, "adm_stream_create_cumask: mask / words / stream_out");
  int ncu = 0;
  for (int i = 0; i < words; ++i) ncu += __builtin_popcount(mask[i]);
  ADM_REQUIRE(ncu > 0, ADM_E_ARG, "adm_stream_create_cumask: empty mask");
  hipStream_t s = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
  if (e != hipSuccess) ADM_FAIL((int)e, "hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
  *stream_out = (void*)s;
  return adm_stream_set_cus((void*)s, ncu);
}

extern "C" int adm_stream_destroy(void* stream) {
  (void)adm_stream_set_cus(stream, 0);
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) ADM_FAIL((int)e, "hipStreamDestroy: %s", hipGetErrorString(e));
  return 0;
}

// where the blocks of a launch on `stream` land: out[b] = XCC_ID | HW_ID << 8 of block b (one wave per block; HW_REG_XCC_ID = 20,
// HW_REG_HW_ID = 4: CU_ID bits 11:8, SH_ID 12, SE_ID 15:13).  Diagnostic for CU-masked streams (tools/cumask_probe.py).
namespace {
__global__ void probe_kernel(uint32_t* out, int spin) {
  const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  // hold the CU for a while so that the blocks of the grid spread over every CU the stream may use
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 15u) | (hw << 8);
}
}  // namespace

extern "C" int adm_stream_probe(uint32_t* out_dev, int nblocks, void* stream) {
  ADM_REQUIRE(out_dev && nblocks > 0 && nblocks <= 65536, ADM_E_ARG, "adm_stream_probe: out / nblocks");
  // 1024 threads x 64 KB LDS: at most two blocks per CU, so a grid of ~2 x CUs covers the stream's CUs
  hipLaunchKernelGGL(probe_kernel, dim3(nblocks), dim3(1024), 64 * 1024, (hipStream_t)stream, out_dev, 20000 /* 100 MHz ticks = 200 us */);
  return adm_check_launch("adm_stream_probe");
}
