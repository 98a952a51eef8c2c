// field29.h -- "lazy" BN254 prime-field arithmetic for the hot kernels: 9 signed limbs, radix 2^29.
//
// Why a second representation (measured, profiles/r01_instr_rates.txt): on gfx950 v_mad_u64_u32 /
// v_mad_i64_i32 issue at the SAME half rate as v_add_co_u32 / v_addc_co_u32 / v_lshl_add_u64, so
// the cost of a 254-bit Montgomery product is its instruction count, not its multiplier count.
// With saturated 32-bit limbs (field.h) every 32x32 product needs a carry chain: ~4 extra
// instructions per v_mad.  With 29-bit limbs in 32-bit registers
//   * a column of the schoolbook product is <= 18 terms of < 2^58: it accumulates in ONE 64-bit
//     register with one v_mad_i64_i32 per term and no carry handling at all,
//   * additions / subtractions are 9 independent 32-bit ops (no carries, no modular correction):
//     limbs and values are allowed to drift ("lazy") and are only brought back by the next
//     Montgomery product, whose 261 - 254 = 7 spare bits absorb values up to ~13 p.
// Limbs are SIGNED so that a - b needs no "add a multiple of p" correction.
//
// Representation.  value(a) = sum a.l[i] * 2^(29 i), an integer that is only defined mod p and may
// be negative.  Montgomery radix R' = 2^261: mul(a, b) = a b / R' mod p.  The storage form used
// everywhere else in the repo (and by the reference: ark_ff Fp256 Montgomery, R = 2^256, reference
// src/zkey.rs:320-332) is converted at the edges by from_mont256 / to_mont256.
//
// Contract of mul / sqr (checked by the emulator build, see F29_CHECK):
//   inputs : sum_i |a.l[i]| * |b.l[k-i]| < 2^62.9 for every column k   (e.g. all |limbs| <= 2^30 on
//            one side and <= 2^29.5 on the other), |value(a) * value(b)| < 169 p^2;
//   output : limbs 0..7 in [0, 2^29), |top limb| < 2^24, value in (-p, 2p).
// carry() renormalises limbs (not values) after a few additions: limbs 0..7 end in [-8, 2^29 + 8).
#pragma once
#include "field.h"

#if defined(G16_EMU) && !defined(F29_NO_CHECK)
#include <assert.h>
#include <math.h>
#define F29_CHECK 1
#endif

namespace g16 {

namespace f29 {

constexpr int N = 9;
constexpr uint32_t MASK = (1u << 29) - 1u;

// Identity the optimiser cannot see through (device code only).  Limbs that come out of a mask
// are KNOWN non-negative, and LLVM then lowers limb * signed-limb as zext x sext: two
// v_mad_u64_u32 plus a sign fix-up (v_ashrrev_i32 + 2 v_mov) instead of ONE v_mad_i64_i32.
// Hiding the known bits makes every limb product a single signed multiply-add.  Same-box A/B at
// 2^22 (profiles/r02_ab_field_variants.txt): the G2 accumulation gains 2.3 % with the points'
// limbs hidden (12.40 -> 12.13 ms), the G1 accumulation loses ~1.5 %, hiding product outputs as
// well gains nothing -- so only Lazy<Fq2>::load_packed uses it.
G16_HD int32_t opaque(int32_t x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(F29_NO_OPAQUE)
  asm("" : "+v"(x));
#endif
  return x;
}

// F29_CHAIN_MAD (round 6 experiment, variant builds only): every limb product of mul / mul2 / sqr as an inline-asm
// v_mad_i64_i32 whose addend is the running column sum, so that the carry of column k is the FIRST addend of column
// k + 1.  Left to itself LLVM sums every column from zero (17 independent chains) and adds the carry afterwards with a
// v_lshl_add_u64: 17 extra half-rate instructions per product, ~7 % of the accumulation kernels' issue slots -- at
// the price of making a product ONE dependent chain of 171 multiply-adds (scripts/ubench/fqmul_chain.hip).
#if defined(F29_CHAIN_MAD) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void mad_vv(int64_t& acc, int32_t a, int32_t b) {
  uint64_t sd;
  asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sd) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mad_vs(int64_t& acc, int32_t a, int32_t b) {  // b: a compile-time constant (modulus limb)
  uint64_t sd;
  asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sd) : "v"(a), "s"(b));
}
#define F29_MAD(acc, a, b) ::g16::f29::mad_vv(acc, a, b)
#define F29_MAD_C(acc, a, b) ::g16::f29::mad_vs(acc, a, (int32_t)(b))
#else
#define F29_MAD(acc, a, b) acc += (int64_t)(a) * (int64_t)(b)
#define F29_MAD_C(acc, a, b) acc += (int64_t)(a) * (int64_t)(b)
#endif

struct W8 {
  uint32_t v[8];
};
struct L9 {
  int32_t v[N];
};

constexpr W8 words_of(const uint32_t (&m)[8]) {
  W8 r{};
  for (int i = 0; i < 8; ++i) r.v[i] = m[i];
  return r;
}
// canonical 256-bit integer -> 9 limbs of 29 bits (limb 8 holds bits 232..255)
constexpr L9 split(const W8& w) {
  L9 r{};
  for (int i = 0; i < N; ++i) {
    const int bit = 29 * i, wd = bit >> 5, sh = bit & 31;
    uint64_t x = (uint64_t)w.v[wd] >> sh;
    if (sh + 29 > 32 && wd + 1 < 8) x |= (uint64_t)w.v[wd + 1] << (32 - sh);
    r.v[i] = (int32_t)(x & MASK);
  }
  return r;
}
constexpr bool geq(const W8& a, const W8& b) {
  for (int i = 7; i >= 0; --i) {
    if (a.v[i] != b.v[i]) return a.v[i] > b.v[i];
  }
  return true;
}
// 2a mod p for a < p < 2^255
constexpr W8 dbl_mod(const W8& a, const W8& p) {
  W8 d{};
  uint32_t c = 0;
  for (int i = 0; i < 8; ++i) {
    d.v[i] = (a.v[i] << 1) | c;
    c = a.v[i] >> 31;
  }
  if (geq(d, p)) {
    uint64_t br = 0;
    for (int i = 0; i < 8; ++i) {
      uint64_t t = (uint64_t)d.v[i] - p.v[i] - br;
      d.v[i] = (uint32_t)t;
      br = (t >> 32) & 1;
    }
  }
  return d;
}
// a * 2^k mod p
constexpr W8 shl_mod(W8 a, int k, const W8& p) {
  for (int i = 0; i < k; ++i) a = dbl_mod(a, p);
  return a;
}
// -p^-1 mod 2^29
constexpr uint32_t neg_inv(uint32_t p0) {
  uint32_t x = p0;  // correct to 3 bits
  for (int i = 0; i < 5; ++i) x *= 2u - p0 * x;
  return (0u - x) & MASK;
}
// s * p as limbs (0..7 normalised, top limb takes the rest)
constexpr L9 times(const L9& p, int s) {
  L9 r{};
  int64_t c = 0;
  for (int i = 0; i < N; ++i) {
    int64_t t = (int64_t)p.v[i] * s + c;
    if (i < N - 1) {
      r.v[i] = (int32_t)(t & MASK);
      c = t >> 29;
    } else {
      r.v[i] = (int32_t)t;
    }
  }
  return r;
}

template <class P>
struct Consts {
  static constexpr W8 MODW = words_of(P::MOD);
  static constexpr L9 MOD = split(MODW);
  static constexpr uint32_t NINV = neg_inv(P::MOD[0]);
  static constexpr W8 ONE256W = words_of(P::ONE);                 // 2^256 mod p
  static constexpr L9 ONE256 = split(ONE256W);
  static constexpr L9 ONE261 = split(shl_mod(ONE256W, 5, MODW));  // the internal "1"
  static constexpr L9 C266 = split(shl_mod(ONE256W, 10, MODW));   // Montgomery-256 -> 261
  static constexpr L9 MOD32 = times(MOD, 32);
  static constexpr L9 R3 = split(shl_mod(ONE256W, 3 * 261 - 256, MODW));  // 2^(3*261) mod p (inv_vartime)
};

}  // namespace f29

template <class P>
struct F29 {
  int32_t l[f29::N];
  using C = f29::Consts<P>;

  static G16_HD F29 zero() {
    F29 r;
#pragma unroll
    for (int i = 0; i < f29::N; ++i) r.l[i] = 0;
    return r;
  }
  static G16_HD F29 from_limbs(const f29::L9& c) {
    F29 r;
#pragma unroll
    for (int i = 0; i < f29::N; ++i) r.l[i] = c.v[i];
    return r;
  }
  static G16_HD F29 one() { return from_limbs(C::ONE261); }
  // exact all-limbs-zero test (NOT "== 0 mod p"): the encoding of the point at infinity
  G16_HD bool limbs_all_zero() const {
    int32_t o = 0;
#pragma unroll
    for (int i = 0; i < f29::N; ++i) o |= l[i];
    return o == 0;
  }

  // ---- cheap linear ops: no carries, no reduction -------------------------------------------
  friend G16_HD F29 operator+(const F29& a, const F29& b) {
    F29 r;
#pragma unroll
    for (int i = 0; i < f29::N; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
  }
  friend G16_HD F29 operator-(const F29& a, const F29& b) {
    F29 r;
#pragma unroll
    for (int i = 0; i < f29::N; ++i) r.l[i] = a.l[i] - b.l[i];
    return r;
  }
  G16_HD F29 neg() const {
    F29 r;
#pragma unroll
    for (int i = 0; i < f29::N; ++i) r.l[i] = -l[i];
    return r;
  }
  G16_HD F29 dbl() const {
    F29 r;
#pragma unroll
    for (int i = 0; i < f29::N; ++i) r.l[i] = l[i] * 2;
    return r;
  }
  // one parallel carry step: limbs 0..7 -> [-2^k, 2^29 + 2^k) for inputs below 2^(29+k)
  G16_HD F29 carry() const {
    F29 r;
    r.l[0] = l[0] & (int32_t)f29::MASK;
#pragma unroll
    for (int i = 1; i < f29::N - 1; ++i) r.l[i] = (l[i] & (int32_t)f29::MASK) + (l[i - 1] >> 29);
    r.l[f29::N - 1] = l[f29::N - 1] + (l[f29::N - 2] >> 29);
    return r;
  }

  // ---- Montgomery product a b / 2^261 ----------------------------------------------------------
  friend G16_HD F29 operator*(const F29& a, const F29& b) {
    constexpr int N = f29::N;
#ifdef F29_CHECK
    check_mul_inputs(a, b);
#endif
    int64_t acc = 0;
    int32_t m[N];
    F29 r;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
      for (int i = 0; i <= k; ++i) F29_MAD(acc, a.l[i], b.l[k - i]);
#pragma unroll
      for (int i = 0; i < k; ++i) F29_MAD_C(acc, m[i], C::MOD.v[k - i]);
      m[k] = (int32_t)(((uint32_t)acc * C::NINV) & f29::MASK);
      F29_MAD_C(acc, m[k], C::MOD.v[0]);
      acc >>= 29;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
      for (int i = k - N + 1; i < N; ++i) F29_MAD(acc, a.l[i], b.l[k - i]);
#pragma unroll
      for (int i = k - N + 1; i < N; ++i) F29_MAD_C(acc, m[i], C::MOD.v[k - i]);
      r.l[k - N] = (int32_t)((uint32_t)acc & f29::MASK);
      acc >>= 29;
    }
    r.l[N - 1] = (int32_t)acc;
    return r;
  }
  // (a b + c d) / 2^261 with ONE reduction (18 products per column: callers keep all four
  // operands' limbs within +-(2^29 + 2^4))
  static G16_HD F29 mul2(const F29& a, const F29& b, const F29& c, const F29& d) {
    constexpr int N = f29::N;
#ifdef F29_CHECK
    check_mul2_inputs(a, b, c, d);
#endif
    int64_t acc = 0;
    int32_t m[N];
    F29 r;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
      for (int i = 0; i <= k; ++i) {
        F29_MAD(acc, a.l[i], b.l[k - i]);
        F29_MAD(acc, c.l[i], d.l[k - i]);
      }
#pragma unroll
      for (int i = 0; i < k; ++i) F29_MAD_C(acc, m[i], C::MOD.v[k - i]);
      m[k] = (int32_t)(((uint32_t)acc * C::NINV) & f29::MASK);
      F29_MAD_C(acc, m[k], C::MOD.v[0]);
      acc >>= 29;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
      for (int i = k - N + 1; i < N; ++i) {
        F29_MAD(acc, a.l[i], b.l[k - i]);
        F29_MAD(acc, c.l[i], d.l[k - i]);
      }
#pragma unroll
      for (int i = k - N + 1; i < N; ++i) F29_MAD_C(acc, m[i], C::MOD.v[k - i]);
      r.l[k - N] = (int32_t)((uint32_t)acc & f29::MASK);
      acc >>= 29;
    }
    r.l[N - 1] = (int32_t)acc;
    return r;
  }
  // a b - c d, one reduction; result in (-p, 2p) with normalised limbs
  static G16_HD F29 mul_sub(const F29& a, const F29& b, const F29& c, const F29& d) {
    return mul2(a, b, c.neg(), d);
  }
  // a^2 / 2^261: the cross terms a_i a_j (i < j) are formed once against the doubled limb
  G16_HD F29 sqr() const {
    constexpr int N = f29::N;
#ifdef F29_CHECK
    check_mul_inputs(*this, *this);
#endif
    int32_t d[N];
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = l[i] * 2;
    int64_t acc = 0;
    int32_t m[N];
    F29 r;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; ++k) {
      const int lo = k < N ? 0 : k - N + 1;
      const int hi = k < N ? k : N - 1;
#pragma unroll
      for (int i = lo; i <= hi; ++i) {
        const int j = k - i;
        if (i < j) F29_MAD(acc, l[i], d[j]);
        else if (i == j) F29_MAD(acc, l[i], l[i]);
      }
      if (k < N) {
#pragma unroll
        for (int i = 0; i < k; ++i) F29_MAD_C(acc, m[i], C::MOD.v[k - i]);
        m[k] = (int32_t)(((uint32_t)acc * C::NINV) & f29::MASK);
        F29_MAD_C(acc, m[k], C::MOD.v[0]);
      } else {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) F29_MAD_C(acc, m[i], C::MOD.v[k - i]);
        r.l[k - N] = (int32_t)((uint32_t)acc & f29::MASK);
      }
      acc >>= 29;
    }
    r.l[N - 1] = (int32_t)acc;
    return r;
  }

  // ---- exact (slow) canonicalisation: the unique representative in [0, p), limbs normalised ----
  // valid for |value| < 32 p and |limbs| < 2^31 - 2^3
  G16_HD F29 canonical() const {
    constexpr int N = f29::N;
    int32_t t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = l[i];
    ripple(t);
    // + 32 p  -> (0, 64 p)
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] += C::MOD32.v[i];
    ripple(t);
#pragma unroll
    for (int s = 5; s >= 0; --s) {
      int32_t u[N];
      // u = t - 2^s p, limb-wise with the shifted modulus limbs recomputed exactly
      int64_t c = 0;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        int64_t x = (int64_t)t[i] - (((int64_t)C::MOD.v[i]) << s) + c;
        if (i < N - 1) {
          u[i] = (int32_t)(x & (int64_t)f29::MASK);
          c = x >> 29;
        } else {
          u[i] = (int32_t)x;
        }
      }
      const bool ge = u[N - 1] >= 0;
#pragma unroll
      for (int i = 0; i < N; ++i) t[i] = ge ? u[i] : t[i];
    }
    F29 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = t[i];
    return r;
  }
  // value == 0 (mod p)?  Exact.  Fast reject: if v = j p then j = v_0 * p_0^-1 (mod 2^29) must be
  // tiny; a random non-multiple passes that filter with probability ~2^-23 and then takes the
  // exact path.  Valid for |value| < 32 p.
  G16_HD bool is_zero_mod_p() const {
    if (!maybe_zero_mod_p()) return false;
    return canonical().limbs_all_zero();
  }
  // the cheap necessary condition alone (3 instructions); false positives ~2^-23
  G16_HD bool maybe_zero_mod_p() const {
    const uint32_t j = ((uint32_t)l[0] * (0u - C::NINV)) & f29::MASK;  // v_0 / p_0 mod 2^29
    const uint32_t dist = j < (f29::MASK + 1u - j) ? j : (f29::MASK + 1u - j);
    return dist <= 40u;
  }

  // ---- conversions -----------------------------------------------------------------------------
  // canonical integer < p given as 8 words -> limbs (no Montgomery change)
  static G16_HD F29 unpack(const uint32_t (&w)[8]) {
    F29 r;
    r.l[0] = (int32_t)(w[0] & f29::MASK);
    r.l[1] = (int32_t)(((w[0] >> 29) | (w[1] << 3)) & f29::MASK);
    r.l[2] = (int32_t)(((w[1] >> 26) | (w[2] << 6)) & f29::MASK);
    r.l[3] = (int32_t)(((w[2] >> 23) | (w[3] << 9)) & f29::MASK);
    r.l[4] = (int32_t)(((w[3] >> 20) | (w[4] << 12)) & f29::MASK);
    r.l[5] = (int32_t)(((w[4] >> 17) | (w[5] << 15)) & f29::MASK);
    r.l[6] = (int32_t)(((w[5] >> 14) | (w[6] << 18)) & f29::MASK);
    r.l[7] = (int32_t)(((w[6] >> 11) | (w[7] << 21)) & f29::MASK);
    r.l[8] = (int32_t)(w[7] >> 8);
    return r;
  }
  // the same with the limbs' known-zero top bits hidden from the optimiser (see f29::opaque)
  static G16_HD F29 unpack_opaque(const uint32_t (&w)[8]) {
    F29 r = unpack(w);
#pragma unroll
    for (int i = 0; i < f29::N; ++i) r.l[i] = f29::opaque(r.l[i]);
    return r;
  }
  // limbs of a canonical value (all in [0, 2^29), top < 2^24) -> 8 words
  G16_HD void pack(uint32_t (&w)[8]) const {
    const uint32_t* u = reinterpret_cast<const uint32_t*>(l);
    w[0] = u[0] | (u[1] << 29);
    w[1] = (u[1] >> 3) | (u[2] << 26);
    w[2] = (u[2] >> 6) | (u[3] << 23);
    w[3] = (u[3] >> 9) | (u[4] << 20);
    w[4] = (u[4] >> 12) | (u[5] << 17);
    w[5] = (u[5] >> 15) | (u[6] << 14);
    w[6] = (u[6] >> 18) | (u[7] << 11);
    w[7] = (u[7] >> 21) | (u[8] << 8);
  }
  // storage form (Montgomery, R = 2^256, canonical) <-> internal form (R' = 2^261, lazy)
  static G16_HD F29 from_mont256(const Fp<P>& a) { return unpack(a.v) * from_limbs(C::C266); }
  G16_HD Fp<P> to_mont256() const {
    F29 c = ((*this) * from_limbs(C::ONE256)).canonical();
    Fp<P> r;
    c.pack(r.v);
    return r;
  }
  // internal form with canonical limbs, packed into 8 words (the HBM format of precomputed points)
  G16_HD void pack_internal(uint32_t (&w)[8]) const { canonical().pack(w); }

 private:
  static G16_HD void ripple(int32_t (&t)[f29::N]) {
#pragma unroll
    for (int i = 0; i < f29::N - 1; ++i) {
      t[i + 1] += t[i] >> 29;
      t[i] &= (int32_t)f29::MASK;
    }
  }
#ifdef F29_CHECK
  static long double approx_over_p(const F29& a) {
    long double v = 0, pv = 0;
    for (int i = f29::N - 1; i >= 0; --i) {
      v = v * 536870912.0L + (long double)a.l[i];
      pv = pv * 536870912.0L + (long double)C::MOD.v[i];
    }
    return v / pv;
  }
  static void check_mul2_inputs(const F29& a, const F29& b, const F29& c, const F29& d) {
    for (int k = 0; k < 2 * f29::N - 1; ++k) {
      __int128 s = 0;
      for (int i = 0; i < f29::N; ++i) {
        const int j = k - i;
        if (j < 0 || j >= f29::N) continue;
        __int128 x = (__int128)a.l[i] * b.l[j];
        __int128 y = (__int128)c.l[i] * d.l[j];
        s += (x < 0 ? -x : x) + (y < 0 ? -y : y);
      }
      s += (__int128)9 << 58;
      s += (__int128)1 << 36;
      assert(s < ((__int128)1 << 63) && "F29 mul2: column overflow");
    }
    long double v = fabsl(approx_over_p(a) * approx_over_p(b)) + fabsl(approx_over_p(c) * approx_over_p(d));
    assert(v < 168.9L && "F29 mul2: value bound exceeded");
  }
  static void check_mul_inputs(const F29& a, const F29& b) {
    {
      long double v = fabsl(approx_over_p(a) * approx_over_p(b));
      assert(v < 168.9L && "F29 mul: value bound exceeded");
    }
    for (int k = 0; k < 2 * f29::N - 1; ++k) {
      __int128 s = 0;
      for (int i = 0; i < f29::N; ++i) {
        const int j = k - i;
        if (j < 0 || j >= f29::N) continue;
        __int128 x = (__int128)a.l[i] * b.l[j];
        s += x < 0 ? -x : x;
      }
      s += (__int128)9 << 58;
      s += (__int128)1 << 36;
      assert(s < ((__int128)1 << 63) && "F29 mul: column overflow");
    }
  }
#endif
};

// a^(p-2): inversion of O(1) elements per kernel (plane precomputation, debug); 0 -> 0
template <class P>
G16_HD F29<P> f29_inv(const F29<P>& a) {
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = P::MOD[i];
  e[0] -= 2;
  F29<P> r = F29<P>::one();
  bool started = false;
  for (int i = 7; i >= 0; --i) {
    for (int bit = 31; bit >= 0; --bit) {
      if (started) r = r.sqr();
      if ((e[i] >> bit) & 1) {
        r = r * a;
        started = true;
      }
    }
  }
  return r;
}

// a^-1 by the binary extended Euclidean algorithm on the canonical integer (Guide to ECC, Alg.
// 2.22): ~4x fewer instructions than the Fermat ladder but data-dependent control flow -- for the
// single-lane affine conversions of the finalisation only (a whole wave running it would diverge;
// the plane precomputation keeps the uniform f29_inv).  0 -> 0.
template <class P>
G16_HD F29<P> f29_inv_vartime(const F29<P>& a) {
  uint32_t u[8], v[8], x1[8], x2[8];
  a.canonical().pack(u);
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    nz |= u[i];
    v[i] = P::MOD[i];
    x1[i] = 0;
    x2[i] = 0;
  }
  if (!nz) return F29<P>::zero();
  x1[0] = 1;
  auto is_one = [](const uint32_t* t) {
    uint32_t o = t[0] ^ 1u;
    for (int i = 1; i < 8; ++i) o |= t[i];
    return o == 0;
  };
  auto shr1 = [](uint32_t* t, uint32_t top) {  // t = (top : t) >> 1
    for (int i = 0; i < 7; ++i) t[i] = (t[i] >> 1) | (t[i + 1] << 31);
    t[7] = (t[7] >> 1) | (top << 31);
  };
  auto halve_mod = [&](uint32_t* x) {  // x / 2 mod p
    uint32_t top = 0;
    if (x[0] & 1u) {
      uint64_t c = 0;
      for (int i = 0; i < 8; ++i) {
        c += (uint64_t)x[i] + P::MOD[i];
        x[i] = (uint32_t)c;
        c >>= 32;
      }
      top = (uint32_t)c;
    }
    shr1(x, top);
  };
  auto geq = [](const uint32_t* s, const uint32_t* t) {
    for (int i = 7; i >= 0; --i)
      if (s[i] != t[i]) return s[i] > t[i];
    return true;
  };
  auto sub = [](uint32_t* s, const uint32_t* t) {  // s -= t, returns the borrow
    uint64_t br = 0;
    for (int i = 0; i < 8; ++i) {
      const uint64_t d = (uint64_t)s[i] - t[i] - br;
      s[i] = (uint32_t)d;
      br = (d >> 32) & 1;
    }
    return (uint32_t)br;
  };
  auto sub_mod = [&](uint32_t* s, const uint32_t* t) {  // s = s - t mod p
    if (sub(s, t)) {
      uint64_t c = 0;
      for (int i = 0; i < 8; ++i) {
        c += (uint64_t)s[i] + P::MOD[i];
        s[i] = (uint32_t)c;
        c >>= 32;
      }
    }
  };
  for (int guard = 0; guard < 2048 && !is_one(u) && !is_one(v); ++guard) {
    while (!(u[0] & 1u)) {
      shr1(u, 0);
      halve_mod(x1);
    }
    while (!(v[0] & 1u)) {
      shr1(v, 0);
      halve_mod(x2);
    }
    if (geq(u, v)) {
      sub(u, v);
      sub_mod(x1, x2);
    } else {
      sub(v, u);
      sub_mod(x2, x1);
    }
  }
  // plain integer inverse I = a^-1 (of the INTERNAL value a = z 2^261), wanted: z^-1 2^261 = I 2^522
  const uint32_t(&res)[8] = is_one(u) ? x1 : x2;
  return F29<P>::unpack(res) * F29<P>::from_limbs(F29<P>::C::R3);
}

// Fq2 = Fq[i]/(i^2 + 1) over lazy limbs.  Same contracts as F29, per component, with the value
// bound |a0 b0| + |a1 b1| < 169 p^2 (both components of both operands below ~9 p).
template <class P>
struct F29x2 {
  F29<P> c0, c1;
  using B = F29<P>;
  static G16_HD F29x2 zero() { return F29x2{B::zero(), B::zero()}; }
  static G16_HD F29x2 one() { return F29x2{B::one(), B::zero()}; }
  G16_HD bool limbs_all_zero() const { return c0.limbs_all_zero() && c1.limbs_all_zero(); }
  friend G16_HD F29x2 operator+(const F29x2& a, const F29x2& b) { return F29x2{a.c0 + b.c0, a.c1 + b.c1}; }
  friend G16_HD F29x2 operator-(const F29x2& a, const F29x2& b) { return F29x2{a.c0 - b.c0, a.c1 - b.c1}; }
  G16_HD F29x2 neg() const { return F29x2{c0.neg(), c1.neg()}; }
  G16_HD F29x2 dbl() const { return F29x2{c0.dbl(), c1.dbl()}; }
  G16_HD F29x2 carry() const { return F29x2{c0.carry(), c1.carry()}; }
  friend G16_HD F29x2 operator*(const F29x2& a, const F29x2& b) {
    return F29x2{B::mul_sub(a.c0, b.c0, a.c1, b.c1), B::mul2(a.c0, b.c1, a.c1, b.c0)};
  }
  G16_HD F29x2 sqr() const { return F29x2{B::mul_sub(c0, c0, c1, c1), c0.dbl() * c1}; }
  // a b - c d; components carried (two separate reductions per component would overflow the
  // 64-bit columns if merged: 36 products)
  static G16_HD F29x2 mul_sub(const F29x2& a, const F29x2& b, const F29x2& c, const F29x2& d) {
    return (a * b - c * d).carry();
  }
  G16_HD bool is_zero_mod_p() const { return c0.is_zero_mod_p() && c1.is_zero_mod_p(); }
  G16_HD bool maybe_zero_mod_p() const { return c0.maybe_zero_mod_p() && c1.maybe_zero_mod_p(); }
  G16_HD F29x2 canonical() const { return F29x2{c0.canonical(), c1.canonical()}; }
  static G16_HD F29x2 from_mont256(const Fq2& a) {
    return F29x2{B::from_mont256(a.c0), B::from_mont256(a.c1)};
  }
  G16_HD Fq2 to_mont256() const { return Fq2{c0.to_mont256(), c1.to_mont256()}; }
};
template <class P>
G16_HD F29x2<P> f29_inv(const F29x2<P>& a) {
  // 1 / (c0 + c1 i) = (c0 - c1 i) / (c0^2 + c1^2)
  F29<P> n = f29_inv(F29<P>::mul2(a.c0, a.c0, a.c1, a.c1));
  return F29x2<P>{a.c0 * n, (a.c1 * n).neg()};
}
template <class P>
G16_HD F29x2<P> f29_inv_vartime(const F29x2<P>& a) {
  F29<P> n = f29_inv_vartime(F29<P>::mul2(a.c0, a.c0, a.c1, a.c1));
  return F29x2<P>{a.c0 * n, (a.c1 * n).neg()};
}

using Fq29 = F29<FqParams>;
using Fq2x29 = F29x2<FqParams>;

// legacy storage field <-> lazy compute field, and the raw "packed internal" HBM form: the
// canonical representative of the internal (R' = 2^261) value packed into the 8 words of an Fq.
template <class F> struct Lazy;
template <> struct Lazy<Fq> {
  using type = F29<FqParams>;
  static G16_HD type load_packed(const Fq& raw) { return type::unpack(raw.v); }
  static G16_HD Fq store_packed(const type& a) {
    Fq r;
    a.pack_internal(r.v);
    return r;
  }
};
template <> struct Lazy<Fq2> {
  using type = F29x2<FqParams>;
  static G16_HD type load_packed(const Fq2& raw) {
    return type{F29<FqParams>::unpack_opaque(raw.c0.v), F29<FqParams>::unpack_opaque(raw.c1.v)};
  }
  static G16_HD Fq2 store_packed(const type& a) {
    Fq2 r;
    a.c0.pack_internal(r.c0.v);
    a.c1.pack_internal(r.c1.v);
    return r;
  }
};

using Fr29 = F29<FrParams>;

}  // namespace g16
