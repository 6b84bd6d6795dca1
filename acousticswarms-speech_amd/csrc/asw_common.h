// Shared host-side helpers for libasw_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/asw_hip.h"

namespace asw {

// thread-local error message returned by asw_last_error()
char* err_buf();
int set_error(int code, const char* fmt, ...);

#define ASW_CHECK_ARG(cond, ...)                                   \
  do {                                                             \
    if (!(cond)) return ::asw::set_error(ASW_ERR_ARG, __VA_ARGS__); \
  } while (0)

#define ASW_HIP(call)                                                                 \
  do {                                                                                \
    hipError_t _e = (call);                                                           \
    if (_e != hipSuccess)                                                             \
      return ::asw::set_error(ASW_ERR_HIP, "%s failed: %s (%s:%d)", #call,            \
                              hipGetErrorString(_e), __FILE__, __LINE__);             \
  } while (0)

#define ASW_LAUNCH_CHECK()                                                            \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess)                                                             \
      return ::asw::set_error(ASW_ERR_HIP, "kernel launch failed: %s (%s:%d)",        \
                              hipGetErrorString(_e), __FILE__, __LINE__);             \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel.  One
// SmemAttr per launch site remembers the largest size already set on each device, so a process
// that serves several GPUs (SpotModel.to("cuda:1"), one stream per device) sets it on each.
struct SmemAttr {
  static constexpr int kMaxDev = 64;
  size_t bytes[kMaxDev] = {};
  int ensure(const void* kern, size_t want);     // 0 or a negative asw_status
};
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// GroupNorm(2) + GLU of one (value, gate) pair: (a - m0) r0 ga + ba, gated by the sigmoid of the normalised
// gate.  One definition with every rounding spelled out, because two kernels apply it (gn_glu_kernel and the
// residual layer that normalises while it loads) and must agree to the bit whatever the compiler would
// contract around them.
__device__ __forceinline__ float gn_glu_value(float a, float g, float m0, float r0, float m1, float r1, float ga, float ba,
                                              float gg, float bg) {
  const float av = __fmaf_rn(__fmul_rn(__fsub_rn(a, m0), r0), ga, ba);
  const float gv = __fmaf_rn(__fmul_rn(__fsub_rn(g, m1), r1), gg, bg);
  // sigmoid on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each, about 3e-7 overall): the IEEE expf +
  // division sequence is some 35 VALU instructions per element, and in the layers that normalise while they stage
  // their rows (resstack.hip, resconv16) that was 10 k of a workgroup's 70 k cycles with the matrix pipe idle
  const float e = __builtin_amdgcn_exp2f(__fmul_rn(gv, -1.4426950408889634f));
  return __fmul_rn(av, __builtin_amdgcn_rcpf(__fadd_rn(1.0f, e)));
}
#endif

}  // namespace asw

// ---- optional in-library launch profiler (HIP events on the launch stream) -----------
// bench.py enables it for the timed region so per-kernel durations are measured live on
// the stream the kernels run on; disabled (the default) it costs one branch per launch.
#include <string>
namespace asw {
std::string prof_name(const char* base, int bm, int bn, int bk, bool ln, bool stats);
bool prof_detail();
// `work` = algorithmic FLOPs of the launch (GEMM-class kernels), `bytes` = algorithmic HBM bytes (the
// memory-bound passes); either may be 0.
struct ProfScope {
  ProfScope(hipStream_t s, const std::string& name, double work, double bytes = 0.0);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};
}  // namespace asw
