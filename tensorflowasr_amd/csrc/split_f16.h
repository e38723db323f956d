// Two-term fp16 operand scheme of the dmodel-144 block kernels (fused_pp.hip, fused_ns.hip): types, the operand split, the
// power-of-two scales, and the generated activation + split schedule (prep2_sched.inc).  Per translation unit (anonymous
// namespace): included once by each kernel file.  See the header of fused_pp.hip for the arithmetic.
#pragma once
#include "common.h"

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct Split8 { u32x4_t t[2]; };             // 8 k-slots x (hi, lo) fp16 terms

DEV unsigned pp_pk_f16(float a, float b) {   // v_cvt_pk_f16_f32: both halves round to nearest
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, f16x2_t));
}
DEV float pp_f16_lo(unsigned p) { return (float)__builtin_bit_cast(f16x2_t, p).x; }
DEV float pp_f16_hi(unsigned p) { return (float)__builtin_bit_cast(f16x2_t, p).y; }
DEV Split8 split8(f32x4 lo, f32x4 hi) {      // the values are already in the operand's unit (x * sx)
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  unsigned d0[4], d1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    d0[k] = pp_pk_f16(v[2 * k], v[2 * k + 1]);
    d1[k] = pp_pk_f16(v[2 * k] - pp_f16_lo(d0[k]), v[2 * k + 1] - pp_f16_hi(d0[k]));
  }
  Split8 f;
  f.t[0] = u32x4_t{d0[0], d0[1], d0[2], d0[3]};
  f.t[1] = u32x4_t{d1[0], d1[1], d1[2], d1[3]};
  return f;
}
// 2^k with bound * 2^k in [2^13, 2^14), k clamped to [-14, 15] (the scale itself is an fp16 operand: the bias slot), from the
// exponent field of `bound` (>= 0; 0 gives 2^15)
DEV float pp_pow2_scale(float bound) {
  const int e = (int)((__builtin_bit_cast(unsigned, bound) >> 23) & 255u);          // bound in [2^(e - 127), 2^(e - 126))
  const int k = min(15, max(-14, 140 - e));
  return __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
}
DEV float pp_recip_pow2(float s) {            // 1 / s for a power of two (exact; exponent field 1 .. 253)
  return __builtin_bit_cast(float, 0x7f000000u - __builtin_bit_cast(unsigned, s));
}
// activation + two-term split of two finished hidden tiles, in slots of <= 2 instructions (prep2_sched.inc).  k1 = -log2 e /
// (unit of the hidden accumulators), ik2 = that unit / (unit of the operand being built): per-lane values
struct Prep2Ctx {
  f32x4 &lo, &hi;
  Split8& out;
  float k1, ik2;
  float ta, tb, m0, m1;
  unsigned hp;
};
using PpPrep = Prep2Ctx;
#include "prep2_sched.inc"


}  // namespace
