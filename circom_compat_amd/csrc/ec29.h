// ec29.h -- BN254 G1 / G2 accumulator arithmetic over the lazy 9 x 29-bit limbs of field29.h.
//
// Same group law and coordinates as ec.h (extended Jacobian XYZZ, a = 0; the results are group
// elements, hence identical to ark-ec's whatever the representation), re-derived with explicit
// limb/value bookkeeping so that no operation needs a modular correction:
//
//   products ("M class")        :  limbs 0..7 in [0, 2^29), value in (-p, 2p)
//   stored x                    :  limbs 0..7 in [-8, 2^29 + 8), |value| < 8 p
//   stored y                    :  limbs 0..7 in [-8, 2^29 + 8), |value| < 3 p
//   stored zz, zzz              :  M class
//
// Every formula below lists the classes it produces; F29_CHECK (emulator build) asserts the
// column and value bounds on every product actually formed in the test-suite.
// The point at infinity is the exact all-zero limb pattern of zz (a valid zz is != 0 mod p, so it
// never has all-zero limbs).
#pragma once
#include "ec.h"
#include "field29.h"

namespace g16 {

template <class LF>
struct Aff29 {
  LF x, y;  // M class (loaded canonical)
  bool inf;
};

template <class LF>
struct XYZZ29 {
  LF x, y, zz, zzz;

  static G16_HD XYZZ29 infinity() { return XYZZ29{LF::zero(), LF::zero(), LF::zero(), LF::zero()}; }
  G16_HD bool is_inf() const { return zz.limbs_all_zero(); }
  static G16_HD XYZZ29 from_affine(const Aff29<LF>& p) {
    if (p.inf) return infinity();
    return XYZZ29{p.x, p.y, LF::one(), LF::one()};
  }
  G16_HD XYZZ29 neg() const { return XYZZ29{x, y.neg().carry(), zz, zzz}; }

  // mdbl-2008-s-1 on an affine point (rare path of madd)
  static G16_HD XYZZ29 dbl_affine(const Aff29<LF>& p) {
    if (p.inf) return infinity();
    LF U = p.y.dbl().carry();        // |v| < 4p
    LF V = U.sqr();                  // M
    LF W = U * V;                    // M
    LF S = p.x * V;                  // M
    LF X2 = p.x.sqr();               // M
    LF Mm = (X2.dbl() + X2).carry();  // |v| < 6p
    LF X3 = (Mm.sqr() - S.dbl()).carry();            // (-5p, 4p)
    LF Y3 = LF::mul_sub(Mm, S - X3, W, p.y);         // M (carried for Fq2)
    return XYZZ29{X3, Y3, V, W};
  }
  // dbl-2008-s-1
  G16_HD void dbl_in_place() {
    if (is_inf()) return;
    LF U = y.dbl().carry();          // |v| < 6p
    LF V = U.sqr();
    LF W = U * V;
    LF S = x * V;
    LF X2 = x.sqr();
    LF Mm = (X2.dbl() + X2).carry();
    LF X3 = (Mm.sqr() - S.dbl()).carry();
    LF Y3 = LF::mul_sub(Mm, S - X3, W, y);
    zz = V * zz;
    zzz = W * zzz;
    x = X3;
    y = Y3;
  }
  // rare tail of madd: x(P) == x(Q) exactly?  Kept out of line so that the hot loop's register
  // allocation and instruction footprint do not pay for it.
  static G16_NOINLINE XYZZ29 madd_rare(Aff29<LF> p, LF Pp, LF R, bool* handled) {
    if (!Pp.is_zero_mod_p()) {
      *handled = false;
      return infinity();
    }
    *handled = true;
    if (R.is_zero_mod_p()) return dbl_affine(p);
    return infinity();
  }
  // madd-2008-s: this += affine p.  8M + 2S, one merged reduction for Y3 in G1.
  G16_HD void madd(const Aff29<LF>& p) {
    if (p.inf) return;
    if (is_inf()) {
      x = p.x;
      y = p.y;
      zz = LF::one();
      zzz = LF::one();
      return;
    }
    LF U2 = p.x * zz;   // M
    LF S2 = p.y * zzz;  // M
    LF Pp = U2 - x;     // limbs within +-(2^29+8); |v| < 10p
    LF R = S2 - y;      // same; |v| < 10p
    if (Pp.maybe_zero_mod_p()) {  // ~2^-23 of the calls: exact test and the P = +-Q cases, out of line
      bool handled;
      XYZZ29 r = madd_rare(p, Pp, R, &handled);
      if (handled) {
        *this = r;
        return;
      }
    }
    LF PP = Pp.sqr();                              // M
    LF PPP = Pp * PP;                              // M
    LF Q = x * PP;                                 // M
    LF X3 = (R.sqr() - PPP - Q.dbl()).carry();     // S: (-7p, 5p)
    LF Y3 = LF::mul_sub(R, Q - X3, y, PPP);        // M (S for Fq2)
    zz = zz * PP;
    zzz = zzz * PPP;
    x = X3;
    y = Y3;
  }
  // Branch-free core of madd for software-pipelined callers (two independent additions per lane
  // in ONE basic block so the scheduler can interleave their dependent multiply chains):
  // returns acc + p by the general formula and reports through *special when the x-coordinates
  // may coincide (~2^-23 of calls: the caller then redoes this addition with madd()).  The
  // infinity cases are resolved by selects, not branches.
  static G16_HD XYZZ29 madd_select(const XYZZ29& acc, const Aff29<LF>& p, bool* special) {
    const bool acc_inf = acc.is_inf();
    // an infinite accumulator has all-zero coordinates: give the formulas something harmless
    LF U2 = p.x * acc.zz;
    LF S2 = p.y * acc.zzz;
    LF Pp = U2 - acc.x;
    LF R = S2 - acc.y;
    *special = !acc_inf && !p.inf && Pp.maybe_zero_mod_p();
    LF PP = Pp.sqr();
    LF PPP = Pp * PP;
    LF Q = acc.x * PP;
    LF X3 = (R.sqr() - PPP - Q.dbl()).carry();
    LF Y3 = LF::mul_sub(R, Q - X3, acc.y, PPP);
    XYZZ29 r{X3, Y3, acc.zz * PP, acc.zzz * PPP};
    if (acc_inf) r = XYZZ29{p.x, p.y, LF::one(), LF::one()};
    if (p.inf) r = acc;
    return r;
  }
  // add-2008-s: this += q.  12M + 2S
  G16_HD void add(const XYZZ29& q) {
    if (q.is_inf()) return;
    if (is_inf()) {
      *this = q;
      return;
    }
    LF U1 = x * q.zz;
    LF U2 = q.x * zz;
    LF S1 = y * q.zzz;
    LF S2 = q.y * zzz;
    LF Pp = U2 - U1;  // limbs in (-2^29, 2^29), |v| < 3p
    LF R = S2 - S1;
    if (Pp.is_zero_mod_p()) {
      if (R.is_zero_mod_p()) {
        dbl_in_place();
      } else {
        *this = infinity();
      }
      return;
    }
    LF PP = Pp.sqr();
    LF PPP = Pp * PP;
    LF Q = U1 * PP;
    LF X3 = (R.sqr() - PPP - Q.dbl()).carry();
    LF Y3 = LF::mul_sub(R, Q - X3, S1, PPP);
    zz = (zz * q.zz) * PP;
    zzz = (zzz * q.zzz) * PPP;
    x = X3;
    y = Y3;
  }
  // x = X/ZZ, y = Y/ZZZ (one inversion), result canonical in the INTERNAL Montgomery form.
  // VARTIME: binary-GCD inversion (single-lane callers only, see f29_inv_vartime)
  template <bool VARTIME = false>
  G16_HD Aff29<LF> to_affine() const {
    if (is_inf()) return Aff29<LF>{LF::zero(), LF::zero(), true};
    LF iz3 = VARTIME ? f29_inv_vartime(zzz) : f29_inv(zzz);
    LF iz2 = iz3.sqr() * zz.sqr();
    return Aff29<LF>{(x * iz2).canonical(), (y * iz3).canonical(), false};
  }
};

// ---- HBM forms ---------------------------------------------------------------------------------
// precomputed points: Affine<F> containers holding the PACKED INTERNAL form (see Lazy<F>);
// all-zero = infinity, as in the reference's encoding (src/zkey.rs:343-348).
template <class F>
G16_HD Aff29<typename Lazy<F>::type> load_packed_affine(const Affine<F>& raw) {
  using LF = typename Lazy<F>::type;
  Aff29<LF> p;
  p.inf = raw.is_inf();
  p.x = Lazy<F>::load_packed(raw.x);
  p.y = Lazy<F>::load_packed(raw.y);
  return p;
}
template <class F>
G16_HD Affine<F> store_packed_affine(const Aff29<typename Lazy<F>::type>& p) {
  if (p.inf) return Affine<F>::infinity();
  return Affine<F>{Lazy<F>::store_packed(p.x), Lazy<F>::store_packed(p.y)};
}
// storage form (Montgomery-256 affine, the zkey encoding) -> lazy registers
template <class F>
G16_HD Aff29<typename Lazy<F>::type> affine_from_mont256(const Affine<F>& a) {
  using LF = typename Lazy<F>::type;
  if (a.is_inf()) return Aff29<LF>{LF::zero(), LF::zero(), true};
  return Aff29<LF>{LF::from_mont256(a.x), LF::from_mont256(a.y), false};
}
// lazy accumulator -> legacy XYZZ in storage form (what finalize.hip consumes)
template <class F>
G16_HD XYZZ<F> xyzz_to_mont256(const XYZZ29<typename Lazy<F>::type>& a) {
  if (a.is_inf()) return XYZZ<F>::infinity();
  return XYZZ<F>{a.x.to_mont256(), a.y.to_mont256(), a.zz.to_mont256(), a.zzz.to_mont256()};
}

using G1XYZZ29 = XYZZ29<Fq29>;
using G2XYZZ29 = XYZZ29<Fq2x29>;

}  // namespace g16
