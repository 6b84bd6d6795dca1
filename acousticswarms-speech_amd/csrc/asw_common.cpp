#include "asw_common.h"

namespace asw {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
int SmemAttr::ensure(const void* kern, size_t want) {
  int dev = 0;
  ASW_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDev) return set_error(ASW_ERR_ARG, "device ordinal %d out of range", dev);
  if (bytes[dev] >= want) return ASW_OK;
  ASW_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
  bytes[dev] = want;
  return ASW_OK;
}
}  // namespace asw

extern "C" const char* asw_last_error(void) { return asw::err_buf(); }
extern "C" int asw_abi_version(void) { return 3; }   // 2: + joint separation network (asw_sep_*); 3: + asw_resstack64_f16x3

// ---- launch profiler -----------------------------------------------------------------
#include <map>
#include <mutex>
#include <vector>
namespace asw {
namespace {
struct Rec { std::string name; double work, bytes; hipEvent_t e0, e1; };
std::mutex g_mu;
bool g_on = false;
bool g_detail = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

std::string prof_name(const char* base, int bm, int bn, int bk, bool ln, bool stats) {
  char b[96];
  snprintf(b, sizeof b, "%s<%d,%d,%d,%s>", base, bm, bn, bk, ln ? "ln" : stats ? "stats" : "plain");
  return b;
}
bool prof_detail() { return g_detail; }
ProfScope::ProfScope(hipStream_t s, const std::string& name, double work, double bytes) : slot(-1), stream(s) {
  if (!g_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r{name, work, bytes, take_event(), take_event()};
  (void)hipEventRecord(r.e0, s);
  g_recs.push_back(r);
  slot = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (slot < (int)g_recs.size()) (void)hipEventRecord(g_recs[slot].e1, stream);
}
}  // namespace asw

extern "C" int asw_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(asw::g_mu);
  for (auto& r : asw::g_recs) { asw::g_pool.push_back(r.e0); asw::g_pool.push_back(r.e1); }
  asw::g_recs.clear();
  asw::g_on = on != 0;
  asw::g_detail = on == 2;                 // 2: GEMM launch names carry their shape
  return ASW_OK;
}

extern "C" int asw_profile_report(char* buf, size_t cap) {
  if (!buf || cap < 64) return asw::set_error(ASW_ERR_ARG, "profile_report: buffer too small");
  std::lock_guard<std::mutex> lk(asw::g_mu);
  struct Agg { long n = 0; double ms = 0, work = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : asw::g_recs) {
    if (hipEventSynchronize(r.e1) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    Agg& a = agg[r.name];
    a.n += 1; a.ms += ms; a.work += r.work; a.bytes += r.bytes;
  }
  size_t off = 0;
  off += snprintf(buf + off, cap - off, "{");
  bool first = true;
  for (auto& kv : agg) {
    if (off + 256 >= cap) break;
    off += snprintf(buf + off, cap - off, "%s\"%s\":{\"launches\":%ld,\"ms\":%.6f,\"work\":%.6e,\"bytes\":%.6e}", first ? "" : ",",
                    kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.work, kv.second.bytes);
    first = false;
  }
  snprintf(buf + off, cap - off, "}");
  return ASW_OK;
}
