// dp_f16.h - host-side fp16 helpers of the two-limb packers (dp_pack_conv_weights_node_f16, dp_pack_conv_weights with DP_PREC_F16X3):
// bit-exact round-to-nearest-even conversions without relying on the host compiler's _Float16 support.
#pragma once
#include <string.h>

namespace dp {
inline unsigned short f16_rne(float x) {                     // fp32 -> fp16 bits, round to nearest even (overflow -> inf, subnormals kept)
  unsigned int u;
  memcpy(&u, &x, 4);
  const unsigned int sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u >= 0x7f800000u) return (unsigned short)(sign | (u > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (u >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);    // >= 65520 rounds to inf
  if (u < 0x38800000u) {                                            // below 2^-14: fp16 subnormal (or zero)
    if (u < 0x33000000u) return (unsigned short)sign;               // < 2^-25 rounds to zero
    const int e = (int)(u >> 23);                                   // biased exponent, 102..112
    unsigned int m = (u & 0x7fffffu) | 0x800000u;                   // 24-bit significand
    const int shift = 126 - e;                                      // bits to drop so that the result counts units of 2^-24
    const unsigned int halfway = 1u << (shift - 1), rest = m & ((1u << shift) - 1u);
    m >>= shift;
    if (rest > halfway || (rest == halfway && (m & 1u))) ++m;
    return (unsigned short)(sign | m);
  }
  unsigned int r = u - 0x38000000u;                                 // re-bias the exponent (127 -> 15)
  const unsigned int rest = r & 0x1fffu;
  r >>= 13;
  if (rest > 0x1000u || (rest == 0x1000u && (r & 1u))) ++r;
  return (unsigned short)(sign | r);
}
inline float f16_to_f32(unsigned short h) {
  const unsigned int sign = ((unsigned int)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  float f;
  if (e == 0) {
    f = (float)m * 5.9604644775390625e-8f;                          // m * 2^-24
    unsigned int u;
    memcpy(&u, &f, 4);
    u |= sign;
    memcpy(&f, &u, 4);
    return f;
  }
  const unsigned int u = sign | (e == 31 ? 0x7f800000u : ((e + 112u) << 23)) | (m << 13);
  memcpy(&f, &u, 4);
  return f;
}
inline void f16_split(float v, unsigned short& hi, unsigned short& lo) {
  hi = f16_rne(v);
  lo = f16_rne(v - f16_to_f32(hi));
}

// the exponent w that puts the largest magnitude `mx` of an operand image into (1024, 2048] when the image is stored as v * 2^w
inline int f16_image_exponent(double mx) {
  if (!(mx > 0.0) || !(mx < 1e30)) return 0;
  int w = 11 - (int)__builtin_ceil(__builtin_log2(mx));
  return w > 40 ? 40 : (w < -40 ? -40 : w);
}
}  // namespace dp
