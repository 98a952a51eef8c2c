// emu_exports.cpp -- TEST INFRASTRUCTURE ONLY: C entry points that expose the host compilation of
// field.h / ec.h (the exact source the kernels use) so pytest can compare single operations with
// the oracle.  Linked only into tests/emu/libg16_emu.so.
#include <string.h>

#include "ec.h"

using namespace g16;

namespace {
template <class F>
void fp_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    F x, y, r;
    memcpy(x.v, a + 8 * i, 32);
    if (b) memcpy(y.v, b + 8 * i, 32);
    switch (op) {
      case 0: r = x * y; break;
      case 1: r = x + y; break;
      case 2: r = x - y; break;
      case 3: r = x.neg(); break;
      case 4: r = x.inv(); break;
      case 5: r = x.sqr(); break;
      case 6: { U256 u = x.to_canonical(); memcpy(r.v, u.v, 32); break; }
      case 7: { U256 u; memcpy(u.v, x.v, 32); r = F::from_canonical(u); break; }
      case 8: r = x.dbl(); break;
      default: r = F::zero();
    }
    memcpy(out + 8 * i, r.v, 32);
  }
}

template <class F>
XYZZ<F> scaled(const Affine<F>& p, const F& lam) {
  if (p.is_inf()) return XYZZ<F>::infinity();
  F l2 = lam.sqr(), l3 = l2 * lam;
  return XYZZ<F>{p.x * l2, p.y * l3, l2, l3};
}

// op 0: scaled(P,l1) + scaled(Q,l2) (add)   1: scaled(P,l1) madd Q   2: dbl scaled(P,l1)
// op 3: k * P (k = canonical 256-bit at `k`)  4: dbl_affine(P)         5: mul_u32(k[0])
template <class F>
void ec_op(int op, const uint8_t* P, const uint8_t* Q, const uint8_t* l1, const uint8_t* l2,
           const uint8_t* k, uint8_t* out, size_t n) {
  const size_t ps = sizeof(Affine<F>), fs = sizeof(F);
  for (size_t i = 0; i < n; ++i) {
    Affine<F> p, q;
    F a = F::one(), b = F::one();
    memcpy(&p, P + i * ps, ps);
    if (Q) memcpy(&q, Q + i * ps, ps);
    if (l1) memcpy(&a, l1 + i * fs, fs);
    if (l2) memcpy(&b, l2 + i * fs, fs);
    XYZZ<F> r;
    switch (op) {
      case 0: r = scaled(p, a); r.add(scaled(q, b)); break;
      case 1: r = scaled(p, a); r.madd(q); break;
      case 2: r = scaled(p, a); r.dbl_in_place(); break;
      case 3: { U256 u; memcpy(u.v, k + i * 32, 32); r = scaled(p, a).mul(u); break; }
      case 4: r = XYZZ<F>::dbl_affine(p); break;
      case 5: { uint32_t s; memcpy(&s, k + i * 32, 4); r = scaled(p, a).mul_u32(s); break; }
      default: r = XYZZ<F>::infinity();
    }
    Affine<F> o = r.to_affine();
    memcpy(out + i * ps, &o, ps);
  }
}
}  // namespace

extern "C" {
void emu_fp_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
  if (field == 0) fp_op<Fr>(op, a, b, out, n);
  else fp_op<Fq>(op, a, b, out, n);
}
void emu_fq2_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    Fq2 x, y, r;
    memcpy(&x, a + 16 * i, 64);
    if (b) memcpy(&y, b + 16 * i, 64);
    switch (op) {
      case 0: r = x * y; break;
      case 1: r = x + y; break;
      case 2: r = x - y; break;
      case 3: r = x.neg(); break;
      case 4: r = x.inv(); break;
      case 5: r = x.sqr(); break;
      default: r = Fq2::zero();
    }
    memcpy(out + 16 * i, &r, 64);
  }
}
void emu_g1_op(int op, const uint8_t* P, const uint8_t* Q, const uint8_t* l1, const uint8_t* l2,
               const uint8_t* k, uint8_t* out, size_t n) {
  ec_op<Fq>(op, P, Q, l1, l2, k, out, n);
}
void emu_g2_op(int op, const uint8_t* P, const uint8_t* Q, const uint8_t* l1, const uint8_t* l2,
               const uint8_t* k, uint8_t* out, size_t n) {
  ec_op<Fq2>(op, P, Q, l1, l2, k, out, n);
}
}
