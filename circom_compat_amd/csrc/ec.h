// ec.h -- BN254 G1 / G2 group arithmetic (short Weierstrass, a = 0) for the MSM kernels.
//
// Affine points use the packed on-disk form the reference's loader decodes: x|y, Montgomery LE
// limbs, all-zero = point at infinity (reference src/zkey.rs:340-360 deserialize_g1/g2).
// Accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// a mixed add costs 8M+2S and needs no inversion, the cheapest inversion-free form for a bucket
// that is filled by affine points.  Group results are unique, so the affine outputs are
// bit-identical to ark-ec's Jacobian arithmetic (VariableBaseMSM::msm_bigint) whatever the
// coordinate system.
#pragma once
#include "field.h"

namespace g16 {

template <class F>
struct alignas(16) Affine {
  F x, y;
  static G16_HD Affine infinity() { return Affine{F::zero(), F::zero()}; }
  G16_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  G16_HD Affine neg() const { return Affine{x, y.neg()}; }
};

template <class F>
struct alignas(16) XYZZ {
  F x, y, zz, zzz;

  static G16_HD XYZZ infinity() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
  G16_HD bool is_inf() const { return zz.is_zero(); }
  static G16_HD XYZZ from_affine(const Affine<F>& p) {
    if (p.is_inf()) return infinity();
    return XYZZ{p.x, p.y, F::one(), F::one()};
  }
  G16_HD XYZZ neg() const { return XYZZ{x, y.neg(), zz, zzz}; }

  // dbl-2008-s-1 (a = 0)
  G16_HD void dbl_in_place() {
    if (is_inf()) return;
    F U = y.dbl();
    F V = U.sqr();
    F W = U * V;
    F S = x * V;
    F X2 = x.sqr();
    F M = X2.dbl() + X2;
    F X3 = M.sqr() - S.dbl();
    F Y3 = M * (S - X3) - W * y;
    zz = V * zz;
    zzz = W * zzz;
    x = X3;
    y = Y3;
  }
  // mdbl-2008-s-1: doubling of an affine point
  static G16_HD XYZZ dbl_affine(const Affine<F>& p) {
    if (p.is_inf()) return infinity();
    F U = p.y.dbl();
    F V = U.sqr();
    F W = U * V;
    F S = p.x * V;
    F X2 = p.x.sqr();
    F M = X2.dbl() + X2;
    F X3 = M.sqr() - S.dbl();
    F Y3 = M * (S - X3) - W * p.y;
    return XYZZ{X3, Y3, V, W};
  }
  // madd-2008-s: this += affine p   (8M + 2S)
  G16_HD void madd(const Affine<F>& p) {
    if (p.is_inf()) return;
    if (is_inf()) {
      x = p.x;
      y = p.y;
      zz = F::one();
      zzz = F::one();
      return;
    }
    F U2 = p.x * zz;
    F S2 = p.y * zzz;
    F Pp = U2 - x;
    F R = S2 - y;
    if (Pp.is_zero()) {
      if (R.is_zero()) {
        *this = dbl_affine(p);
      } else {
        *this = infinity();
      }
      return;
    }
    F PP = Pp.sqr();
    F PPP = Pp * PP;
    F Q = x * PP;
    F X3 = R.sqr() - PPP - Q.dbl();
    F Y3 = R * (Q - X3) - y * PPP;
    zz = zz * PP;
    zzz = zzz * PPP;
    x = X3;
    y = Y3;
  }
  // add-2008-s: this += q   (12M + 2S)
  G16_HD void add(const XYZZ& q) {
    if (q.is_inf()) return;
    if (is_inf()) {
      *this = q;
      return;
    }
    F U1 = x * q.zz;
    F U2 = q.x * zz;
    F S1 = y * q.zzz;
    F S2 = q.y * zzz;
    F Pp = U2 - U1;
    F R = S2 - S1;
    if (Pp.is_zero()) {
      if (R.is_zero()) {
        dbl_in_place();
      } else {
        *this = infinity();
      }
      return;
    }
    F PP = Pp.sqr();
    F PPP = Pp * PP;
    F Q = U1 * PP;
    F X3 = R.sqr() - PPP - Q.dbl();
    F Y3 = R * (Q - X3) - S1 * PPP;
    zz = zz * q.zz * PP;
    zzz = zzz * q.zzz * PPP;
    x = X3;
    y = Y3;
  }
  // x = X/ZZ, y = Y/ZZZ.  One field inversion: 1/ZZZ, then 1/ZZ = ZZZ^-2 * ZZ^2 ... computed as
  // (1/ZZZ)^2 * ZZ^2 = ZZ^2/ZZZ^2 = 1/ZZ  (because ZZZ^2 = ZZ^3).
  G16_HD Affine<F> to_affine() const {
    if (is_inf()) return Affine<F>::infinity();
    F iz3 = zzz.inv();
    F iz2 = iz3.sqr() * zz.sqr();
    return Affine<F>{x * iz2, y * iz3};
  }
  // k * this for a canonical 256-bit scalar, MSB-first double-and-add (O(1) uses per proof)
  G16_HD XYZZ mul(const U256& k) const {
    XYZZ acc = infinity();
    bool started = false;
    for (int i = 7; i >= 0; --i) {
      for (int bit = 31; bit >= 0; --bit) {
        if (started) acc.dbl_in_place();
        if ((k.v[i] >> bit) & 1) {
          acc.add(*this);
          started = true;
        }
      }
    }
    return acc;
  }
  // small scalar (bucket index weights in the window reduction)
  G16_HD XYZZ mul_u32(uint32_t k) const {
    XYZZ acc = infinity();
    for (int bit = 31; bit >= 0; --bit) {
      acc.dbl_in_place();
      if ((k >> bit) & 1) acc.add(*this);
    }
    return acc;
  }
};

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;

}  // namespace g16
