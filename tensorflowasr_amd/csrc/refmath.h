// expf / logf / log exactly as the HOST's C library returns them, for code that runs on the device.
//
// The reference's prefix beam search keeps float32 scores and combines them with
//     log_sum_exp<float>(x, y) = std::log(std::exp(x - m) + std::exp(y - m)) + m        (decoder_utils.h:41-49)
// and takes the candidates' log-probabilities as (float)log((double)p + FLT_MIN) (ctc_beam_search_decoder.cpp:57-59):
// its scores ARE the C library's roundings.  glibc's float routines are fast, not correctly rounded -- on this image
// logf differs from the exact value rounded to float for 97 842 of the 8 388 609 floats in [1, 2], expf for 85 067 of the
// 2.5e8 floats in [-17.5, 0), and logf(1 + expf(d)) for 0.5 % of all d -- so a device search that evaluates "the exact
// function, rounded" (rounds 2-3 did) drifts from the reference by an ulp every couple of hundred log_sum_exp calls, and on
// a 750-frame utterance an ulp eventually reorders two beam entries.  Round 4 therefore restates the library's own
// evaluation: glibc 2.35, x86-64, the FMA variants its ifunc resolvers select on every CPU with AVX2 + FMA
// (sysdeps/ieee754/flt-32/e_expf.c, e_logf.c, sysdeps/ieee754/dbl-64/e_log.c as compiled into __expf_fma / __logf_fma /
// __log_fma; the fused operations below are the ones in that object code, read from its disassembly).  Every operation is
// an IEEE-754 double add / multiply / fma or an integer operation, so gfx950's v_fma_f64 / v_mul_f64 / v_add_f64 return
// the same bits as the host's vfmadd / vmulsd / vaddsd.  Tables: refmath_tables.inc (tools/libm_tables.py).
//
// Checked, not assumed: tools/refmath_check.cpp compiles this header for the host and compares with the installed libm on
// EVERY argument the search can produce (all floats in [-17.5, 0] for expf, all floats in [1, 2] for logf, all floats
// p in [0, 1] for log(p + FLT_MIN)): 0 differences (tests/test_host.py runs it); tests/test_gpu_parity.py compares the
// device's evaluation of the same functions with the host libm on the GPU box.
//
// Argument ranges: only what the search needs -- no overflow / underflow / NaN / subnormal branches.
//   ref_expf: |x| < 88;  ref_logf: positive normal floats;  ref_log: positive normal doubles.
#pragma once
#include <cstdint>
#include <cstring>

#include "refmath_tables.inc"

#if defined(__HIPCC__)
#define REFMATH_HD __host__ __device__ __forceinline__
#else
#define REFMATH_HD inline
#endif

namespace refmath {

constexpr int kExp2fTabWords = 32;    // uint64
constexpr int kLogfTabWords = 32;     // double: {invc, logc} x 16
constexpr int kLogTabWords = 256;     // double: {invc, logc} x 128

REFMATH_HD double as_double(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __longlong_as_double((long long)u);
#else
  double d; std::memcpy(&d, &u, 8); return d;
#endif
}
REFMATH_HD uint64_t as_u64(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint64_t)__double_as_longlong(d);
#else
  uint64_t u; std::memcpy(&u, &d, 8); return u;
#endif
}
REFMATH_HD float as_float(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f; std::memcpy(&f, &u, 4); return f;
#endif
}
REFMATH_HD uint32_t as_u32(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(f);
#else
  uint32_t u; std::memcpy(&u, &f, 4); return u;
#endif
}

// expf (e_expf.c): x / ln 2 = k / 32 + r, exp(x) = 2^(k/32) * exp2-polynomial(r), one rounding to float at the end.
// T = REFMATH_EXP2F_TAB (T[i] = bits(2^(i/32)) - (i << 47), so that adding ki << 47 builds the exponent).
REFMATH_HD float ref_expf(float x, const uint64_t* T) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double xd = (double)x;
  double kd = __builtin_fma(REFMATH_EXP2F_INVLN2N, xd, 0x1.8p+52);   // round-to-nearest integer in the low bits
  const uint64_t ki = as_u64(kd);
  kd = kd - 0x1.8p+52;
  const double r = __builtin_fma(REFMATH_EXP2F_INVLN2N, xd, -kd);
  const double s = as_double(T[ki & 31] + (ki << 47));
  const double z = __builtin_fma(REFMATH_EXP2F_C0, r, REFMATH_EXP2F_C1);
  const double r2 = r * r;
  double y = __builtin_fma(REFMATH_EXP2F_C2, r, 1.0);
  y = __builtin_fma(z, r2, y);
  y = y * s;
  return (float)y;
}

// logf (e_logf.c): x = 2^k z, z in [0x1.66p-1, 0x1.66p0) split into 16 intervals with centre c: log x = k ln 2 + log c +
// log1p(z / c - 1).  T = REFMATH_LOGF_TAB ({1 / c, log c} pairs).
REFMATH_HD float ref_logf(float x, const double* T) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const uint32_t ix = as_u32(x);
  if (ix == 0x3f800000u) return 0.f;
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) & 15u);
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double invc = T[2 * i], logc = T[2 * i + 1];
  const double z = (double)as_float(iz);
  const double r = __builtin_fma(z, invc, -1.0);
  const double y0 = __builtin_fma((double)k, REFMATH_LOGF_LN2, logc);
  const double r2 = r * r;
  double y = __builtin_fma(REFMATH_LOGF_A1, r, REFMATH_LOGF_A2);
  y = __builtin_fma(REFMATH_LOGF_A0, r2, y);
  y = __builtin_fma(y, r2, y0 + r);
  return (float)y;
}

// log (e_log.c, double): the same scheme with 128 intervals and a degree-5 polynomial, hi / lo accumulation of the leading
// terms; arguments within [1 - 2^-4, 1 + 0x1.09p-4) take a degree-11 polynomial in x - 1 whose leading terms are formed
// exactly (r split into 27-bit halves).  T = REFMATH_LOG_TAB.
REFMATH_HD double ref_log(double x, const double* T) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const uint64_t ix = as_u64(x);
  if (ix - 0x3fee000000000000ull <= 0x308ffffffffffull) {
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double r = x - 1.0;
    double a = __builtin_fma(r, REFMATH_LOG_B2, REFMATH_LOG_B1);
    double b = __builtin_fma(r, REFMATH_LOG_B5, REFMATH_LOG_B4);
    const double r2 = r * r;
    double c = __builtin_fma(r, REFMATH_LOG_B8, REFMATH_LOG_B7);
    a = __builtin_fma(r2, REFMATH_LOG_B3, a);
    b = __builtin_fma(r2, REFMATH_LOG_B6, b);
    const double r3 = r * r2;
    c = __builtin_fma(r2, REFMATH_LOG_B9, c);
    c = __builtin_fma(r3, REFMATH_LOG_B10, c);
    b = __builtin_fma(c, r3, b);
    const double poly = __builtin_fma(b, r3, a);
    const double w = __builtin_fma(r, 0x1p27, r);          // "r + w - w" with w = r 2^27, as contracted by the library's build
    const double rhi = __builtin_fma(-0x1p27, r, w);
    const double rhi2 = rhi * rhi;
    const double rlo = r - rhi;
    const double hi = __builtin_fma(rhi2, REFMATH_LOG_B0, r);
    double lo = __builtin_fma(rhi2, REFMATH_LOG_B0, r - hi);
    lo = __builtin_fma(REFMATH_LOG_B0 * rlo, r + rhi, lo);
    const double y = __builtin_fma(poly, r3, lo);
    return hi + y;
  }
  const uint64_t tmp = ix - 0x3fe6000000000000ull;
  const int i = (int)((tmp >> 45) & 127u);
  const int k = (int)((int64_t)tmp >> 52);
  const uint64_t iz = ix - (tmp & (0xfffull << 52));
  const double invc = T[2 * i], logc = T[2 * i + 1];
  const double z = as_double(iz);
  const double kd = (double)k;
  const double r = __builtin_fma(z, invc, -1.0);
  const double w = __builtin_fma(kd, REFMATH_LOG_LN2HI, logc);
  const double p1 = __builtin_fma(r, REFMATH_LOG_A2, REFMATH_LOG_A1);
  const double hi = r + w;
  const double r2 = r * r;
  double lo = (w - hi) + r;
  lo = __builtin_fma(kd, REFMATH_LOG_LN2LO, lo);
  const double r3 = r * r2;
  const double p2 = __builtin_fma(r, REFMATH_LOG_A4, REFMATH_LOG_A3);
  const double q = __builtin_fma(r2, REFMATH_LOG_A0, lo);
  const double p = __builtin_fma(p2, r2, p1);
  const double y = __builtin_fma(r3, p, q);
  return y + hi;
}

}  // namespace refmath
