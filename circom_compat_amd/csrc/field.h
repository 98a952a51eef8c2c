// field.h -- BN254 Fr / Fq / Fq2 arithmetic for gfx950 (and, for unit tests only, the host).
//
// Representation: 8 x 32-bit little-endian limbs in Montgomery form (R = 2^256), i.e. byte-for-byte
// the in-memory/on-disk form the reference uses for ark_bn254::{Fr,Fq} (4 x u64 LE Montgomery;
// reference src/zkey.rs:320-332 deserialize_field / deserialize_field_fr).  CDNA4 has no 64x64
// multiplier in the VALU: the natural unit is v_mad_u64_u32 (32x32+64 -> 64), so limbs are 32-bit
// and every inner step below is written as (u64)a*b + c so hipcc selects that instruction.
//
// Everything is __host__ __device__ so tests/emu can exercise the exact same source on the CPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define G16_HD __host__ __device__ __forceinline__
#define G16_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define G16_HD inline __attribute__((always_inline))
#define G16_NOINLINE __attribute__((noinline))
#endif

namespace g16 {

// ------------------------------------------------------------------------------------------------
// modulus parameter packs (values: SURVEY.md Appendix B, re-derived and checked in tests)
// ------------------------------------------------------------------------------------------------
struct FrParams {
  // r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
  static constexpr uint32_t MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t INV = 0xefffffffu;  // -r^-1 mod 2^32
  // R mod r
  static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                      0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  // R^2 mod r
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                     0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

struct FqParams {
  // q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
  static constexpr uint32_t MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t INV = 0xe4866389u;  // -q^-1 mod 2^32
  // R mod q  (== the snarkjs "Fq one" golden bytes, reference src/zkey.rs:398-402)
  static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                      0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  // R^2 mod q
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                     0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};

// 256-bit raw integer (canonical scalars for the MSM digit extraction)
struct alignas(16) U256 {
  uint32_t v[8];
};

// ------------------------------------------------------------------------------------------------
// prime field element
// ------------------------------------------------------------------------------------------------
template <class P>
struct alignas(16) Fp {
  uint32_t v[8];

  static G16_HD Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0;
    return r;
  }
  static G16_HD Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = P::ONE[i];
    return r;
  }
  static G16_HD Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = P::R2[i];
    return r;
  }
  G16_HD bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= v[i];
    return o == 0;
  }
  G16_HD bool operator==(const Fp& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= (v[i] ^ b.v[i]);
    return o == 0;
  }
  G16_HD bool operator!=(const Fp& b) const { return !(*this == b); }

  // r = a - MOD if a >= MOD else a    (a < 2*MOD, possibly with a 9th carry bit `top`)
  static G16_HD Fp reduce_once(const Fp& a, uint32_t top = 0) {
    Fp d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint64_t t = (uint64_t)a.v[i] - P::MOD[i] - br;
      d.v[i] = (uint32_t)t;
      br = (t >> 32) & 1;
    }
    // a >= MOD  <=>  no final borrow, or the carry bit absorbs it
    bool ge = (br == 0) || (top != 0);
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = ge ? d.v[i] : a.v[i];
    return r;
  }

  friend G16_HD Fp operator+(const Fp& a, const Fp& b) {
    Fp s;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      c += (uint64_t)a.v[i] + b.v[i];
      s.v[i] = (uint32_t)c;
      c >>= 32;
    }
    return reduce_once(s, (uint32_t)c);  // MOD < 2^254 so c is always 0; kept for generality
  }
  friend G16_HD Fp operator-(const Fp& a, const Fp& b) {
    Fp d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
      d.v[i] = (uint32_t)t;
      br = (t >> 32) & 1;
    }
    // add MOD back when the subtraction borrowed
    uint32_t mask = (uint32_t)0 - (uint32_t)br;
    uint64_t c = 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      c += (uint64_t)d.v[i] + (P::MOD[i] & mask);
      r.v[i] = (uint32_t)c;
      c >>= 32;
    }
    return r;
  }
  G16_HD Fp neg() const {
    if (is_zero()) return *this;
    Fp m;
#pragma unroll
    for (int i = 0; i < 8; ++i) m.v[i] = P::MOD[i];
    // MOD - a, a in (0, MOD)
    Fp d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint64_t t = (uint64_t)m.v[i] - v[i] - br;
      d.v[i] = (uint32_t)t;
      br = (t >> 32) & 1;
    }
    return d;
  }
  G16_HD Fp dbl() const { return *this + *this; }

  // Montgomery product a*b*R^-1 mod MOD (CIOS, 32-bit limbs).  MOD < 2^254 so the running value
  // stays below 2*MOD < 2^255: the 9th word never carries out ("no-carry" variant).
  friend G16_HD Fp operator*(const Fp& a, const Fp& b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint64_t c = 0;
      const uint32_t bi = b.v[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        c = (uint64_t)a.v[j] * bi + t[j] + c;
        t[j] = (uint32_t)c;
        c >>= 32;
      }
      t[8] = (uint32_t)c;  // t[8] was 0 on entry to this round (see shift below)
      const uint32_t m = t[0] * P::INV;
      c = ((uint64_t)m * P::MOD[0] + t[0]) >> 32;
#pragma unroll
      for (int j = 1; j < 8; ++j) {
        c = (uint64_t)m * P::MOD[j] + t[j] + c;
        t[j - 1] = (uint32_t)c;
        c >>= 32;
      }
      c += t[8];
      t[7] = (uint32_t)c;
      t[8] = 0;  // (c >> 32) == 0 because the value is < 2*MOD < 2^255
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = t[i];
    return reduce_once(r);
  }
  G16_HD Fp sqr() const { return (*this) * (*this); }

  // Montgomery -> canonical integer (ark-ff PrimeField::into_bigint): one reduction of (a, 0)
  G16_HD U256 to_canonical() const {
    Fp o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = 0;
    o.v[0] = 1;
    Fp r = (*this) * o;
    U256 u;
#pragma unroll
    for (int i = 0; i < 8; ++i) u.v[i] = r.v[i];
    return u;
  }
  // canonical integer (< MOD) -> Montgomery
  static G16_HD Fp from_canonical(const U256& u) {
    Fp a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a.v[i] = u.v[i];
    return a * r2();
  }
  static G16_HD Fp from_u32(uint32_t x) {
    U256 u;
#pragma unroll
    for (int i = 0; i < 8; ++i) u.v[i] = 0;
    u.v[0] = x;
    return from_canonical(u);
  }

  // a^e, e given as 8 x u32 LE (square and multiply, MSB first).  Used for inversion (e = MOD-2)
  // on O(1) elements per proof, so speed is irrelevant here.
  G16_HD Fp pow(const uint32_t (&e)[8]) const {
    Fp r = one();
    bool started = false;
    for (int i = 7; i >= 0; --i) {
      for (int bit = 31; bit >= 0; --bit) {
        if (started) r = r.sqr();
        if ((e[i] >> bit) & 1) {
          r = r * (*this);
          started = true;
        }
      }
    }
    return r;
  }
  G16_HD Fp inv() const {  // Fermat; 0 -> 0
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = P::MOD[i];
    e[0] -= 2;  // MOD is odd and MOD[0] >= 2 for both fields
    return pow(e);
  }
};

using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

// ------------------------------------------------------------------------------------------------
// Fq2 = Fq[i]/(i^2+1); memory order c0|c1 as in the zkey G2 encoding (reference src/zkey.rs:334-338)
// ------------------------------------------------------------------------------------------------
struct alignas(16) Fq2 {
  Fq c0, c1;
  static G16_HD Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
  static G16_HD Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
  G16_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  G16_HD bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
  G16_HD bool operator!=(const Fq2& b) const { return !(*this == b); }
  friend G16_HD Fq2 operator+(const Fq2& a, const Fq2& b) { return Fq2{a.c0 + b.c0, a.c1 + b.c1}; }
  friend G16_HD Fq2 operator-(const Fq2& a, const Fq2& b) { return Fq2{a.c0 - b.c0, a.c1 - b.c1}; }
  G16_HD Fq2 neg() const { return Fq2{c0.neg(), c1.neg()}; }
  G16_HD Fq2 dbl() const { return Fq2{c0.dbl(), c1.dbl()}; }
  friend G16_HD Fq2 operator*(const Fq2& a, const Fq2& b) {  // Karatsuba, 3 Fq mults
    Fq v0 = a.c0 * b.c0;
    Fq v1 = a.c1 * b.c1;
    Fq s = (a.c0 + a.c1) * (b.c0 + b.c1);
    return Fq2{v0 - v1, s - v0 - v1};
  }
  G16_HD Fq2 sqr() const {  // (c0+c1)(c0-c1), 2 c0 c1
    Fq p = c0 * c1;
    return Fq2{(c0 + c1) * (c0 - c1), p + p};
  }
  G16_HD Fq2 inv() const {
    Fq n = (c0.sqr() + c1.sqr()).inv();
    return Fq2{c0 * n, (c1 * n).neg()};
  }
};

}  // namespace g16
