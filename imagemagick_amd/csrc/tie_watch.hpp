// Is a value of a ROUNDED intermediate closer to a rounding boundary than two summation orders can
// differ?  (resize_acc.hpp has the why; this header is plain C++ so that tests/cpu/tie_watch_test.cpp
// can put the bit arithmetic through its paces on the host.)
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define MH_TIE_HD __host__ __device__ __forceinline__
#else
#define MH_TIE_HD inline
#endif

namespace mh {

// the low word of a double
static MH_TIE_HD unsigned tie_low_word(double v)
{
  unsigned long long bits;
  memcpy(&bits,&v,sizeof(bits));
  return (unsigned) bits;
}

template<bool kFloat>
struct TieWatchBits
{
  // Half-widths, in the tail's units.  Q16: value + 0.5 + 2^28 has its last place at 2^-24; its low
  // word shifted left by 8 is the fraction of value + 0.5 as a 32-bit fixed-point number (rounded to
  // 2^-24): units of 2^-32 level; the plain window is 1.5 * 2^-24 = 9e-8 level (values the clamp
  // decides leave the binade: whatever their bits say is harmless).  float: (Quantum) value keeps
  // 24 of the 53 significant bits, a tie is a tail of 29 bits at one half: units of the double's
  // last place; the plain window is 2^8 of them = 5.7e-14 relative (sum|terms| up to ten times |sum|).
  static constexpr unsigned kPlain=kFloat ? 0x100u : 0x180u;
  static constexpr unsigned kWidest=kFloat ? 0x08000000u : 0x40000000u;
  unsigned bias,twice;
  MH_TIE_HD void set(unsigned half)
  {
    bias=kFloat ? half-0x10000000u : half;
    twice=2u*half;
  }
  MH_TIE_HD void plain() { set(kPlain); }
  // r = the reciprocal of the alpha sum: 4.123e-5 * |r| level = 177090 * |r| units (Q16: 177200);
  // 6.3e-10 * |r| relative <= 5.7e6 * |r| units (float) — on top of the plain window
  MH_TIE_HD void quotient(double r)
  {
    const double magnitude=r < 0.0 ? -r : r;
    const double wide=kFloat ? 5.7e6*magnitude+256.0 : 177200.0*magnitude+384.0;
    // (a conversion that saturates, as v_cvt_u32_f64 does)
    const unsigned half=!(wide < 4294967295.0) ? 0xffffffffu : (unsigned) wide;
    set(half);
    // a window as wide as the tail's whole range (an alpha sum of 1e-3 and less): every value is reported
    bias=half < kWidest ? bias : 0u;
    twice=half < kWidest ? twice : 0xffffffffu;
  }
  MH_TIE_HD bool near(double v) const
  {
    if (kFloat)
      return ((tie_low_word(v) & 0x1fffffffu)+bias) <= twice;
    return ((tie_low_word(v+268435456.5) << 8)+bias) <= twice;
  }
};

} // namespace mh
