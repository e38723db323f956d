// Proves csrc/refmath.h == the installed C library, argument by argument, on the ranges the prefix beam search uses.
//   g++ -O2 -mfma -ffp-contract=off -std=c++17 -pthread tools/refmath_check.cpp -o /tmp/refmath_check && /tmp/refmath_check [stride]
// stride 1 (default) = every float of each range (about a minute of CPU over all threads' sum); tests/test_host.py runs a
// coarser stride.  Prints one line per function: "<name> checked <n> differ <m>", exit status 1 if any m > 0.
//   expf: all floats in [-17.5, -0] (log_sum_exp never calls it below: beam_device.hip lse, decoder_utils.h:41-49)
//   logf: all floats in [1, 2]      (1 + expf(d))
//   log : (double)p + FLT_MIN for all floats p in [0, 1] (ctc_beam_search_decoder.cpp:57-59)
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../tensorflowasr_amd/csrc/refmath.h"

static const uint64_t kExpTab[32] = REFMATH_EXP2F_TAB;
static const double kLogfTab[32] = REFMATH_LOGF_TAB;
static const double kLogTab[256] = REFMATH_LOG_TAB;

template <class F>
static long sweep(uint32_t lo, uint32_t hi, uint32_t stride, F differs) {   // bit patterns lo..hi inclusive
  const unsigned nt = std::max(1u, std::thread::hardware_concurrency());
  std::vector<long> bad(nt, 0);
  std::vector<std::thread> th;
  const uint64_t span = (uint64_t)hi - lo + 1, chunk = (span + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      const uint64_t b = lo + t * chunk, e = std::min<uint64_t>(b + chunk, (uint64_t)hi + 1);
      const uint64_t first = b + (stride - (b - lo) % stride) % stride;
      for (uint64_t u = first; u < e; u += stride) bad[t] += differs((uint32_t)u);
    });
  for (auto& x : th) x.join();
  long s = 0;
  for (long v : bad) s += v;
  return s;
}

int main(int argc, char** argv) {
  const uint32_t stride = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 1u;
  using namespace refmath;
  int rc = 0;
  auto report = [&](const char* name, uint32_t lo, uint32_t hi, long bad) {
    std::printf("%s checked %llu differ %ld\n", name, (unsigned long long)(((uint64_t)hi - lo) / stride + 1), bad);
    if (bad) rc = 1;
  };
  {  // negative floats: bit patterns ascend with magnitude
    const uint32_t lo = as_u32(-0.0f), hi = as_u32(-17.5f);
    report("expf", lo, hi, sweep(lo, hi, stride, [](uint32_t u) {
             const float x = as_float(u);
             volatile float xv = x;
             return (long)(as_u32(std::exp((float)xv)) != as_u32(ref_expf(x, kExpTab)));
           }));
  }
  {
    const uint32_t lo = as_u32(1.0f), hi = as_u32(2.0f);
    report("logf", lo, hi, sweep(lo, hi, stride, [](uint32_t u) {
             const float x = as_float(u);
             volatile float xv = x;
             return (long)(as_u32(std::log((float)xv)) != as_u32(ref_logf(x, kLogfTab)));
           }));
  }
  {
    const uint32_t lo = as_u32(0.0f), hi = as_u32(1.0f);
    report("log(p+FLT_MIN)", lo, hi, sweep(lo, hi, stride, [](uint32_t u) {
             const double x = (double)as_float(u) + (double)FLT_MIN;
             volatile double xv = x;
             return (long)(as_u64(std::log((double)xv)) != as_u64(ref_log(x, kLogTab)));
           }));
  }
  {  // the composition as the search uses it
    const uint32_t lo = as_u32(-0.0f), hi = as_u32(-17.5f);
    report("logf(1+expf(d))", lo, hi, sweep(lo, hi, stride, [](uint32_t u) {
             const float d = as_float(u);
             volatile float dv = d;
             const float a = std::log(1.0f + std::exp((float)dv));
             const float b = ref_logf(1.0f + ref_expf(d, kExpTab), kLogfTab);
             return (long)(as_u32(a) != as_u32(b));
           }));
  }
  return rc;
}
