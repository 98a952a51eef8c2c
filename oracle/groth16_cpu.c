/* oracle/groth16_cpu.c -- TEST INFRASTRUCTURE ONLY: the checker and the timed CPU baseline.
 * Never linked into, loaded by or called from the product library.
 *
 * C restatement of the CPU proving path of arkworks-rs/circom-compat, multithreaded the way the
 * reference is (rayon inside the upstream FFT / MSM only; SURVEY.md section 5 note):
 *   g16cpu_witness_map : CircomReduction::witness_map_from_matrices, reference src/circom/qap.rs:23-88
 *                        (serial row loops :37-58,:83-85 as in the crate, parallel radix-2 FFTs)
 *   g16cpu_msm_g1/g2   : ark-ec 0.5 VariableBaseMSM::msm_bigint (signed-digit windows,
 *                        c = ln_without_floats(n)+2, windows processed in parallel, serial
 *                        mixed-addition bucket fill per window, running-sum reduce, Horner combine)
 *   g16cpu_prove       : ark-groth16 0.5 create_proof_with_reduction_and_matrices ->
 *                        create_proof_with_assignment (call sites reference src/zkey.rs:903-911,
 *                        benches/groth16.rs:52-60; equations SURVEY.md section 3.1)
 * The upstream crates are not vendored in /root/reference (Cargo.toml:24-32) and cargo is not
 * available, so this is a restatement of their published algorithms; it is pinned against
 * oracle/bn254_ref.py, which itself reproduces the reference's golden vectors and KATs
 * (tests/test_oracle.py).  Field elements: 4 x u64 Montgomery, same packed formats as the C ABI.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct { u64 p[4]; u64 inv; u64 one[4]; u64 r2[4]; } Mod;

static const Mod FR = {
  {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  0xc2e1f593efffffffULL,
  {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
  {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};
static const Mod FQ = {
  {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  0x87d20782e4866389ULL,
  {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
  {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}};

/* ------------------------------------------------------------------ prime field ------------- */
static inline int ge4(const u64 a[4], const u64 b[4]) {
  for (int i = 3; i >= 0; --i) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
  return 1;
}
static inline void sub4(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 br = 0;
  for (int i = 0; i < 4; ++i) { u128 t = (u128)a[i] - b[i] - br; r[i] = (u64)t; br = (u64)(t >> 64) & 1; }
}
static inline void fp_add(u64 r[4], const u64 a[4], const u64 b[4], const Mod* M) {
  u64 c = 0, t[4];
  for (int i = 0; i < 4; ++i) { u128 s = (u128)a[i] + b[i] + c; t[i] = (u64)s; c = (u64)(s >> 64); }
  if (c || ge4(t, M->p)) sub4(r, t, M->p); else memcpy(r, t, 32);
}
static inline void fp_sub(u64 r[4], const u64 a[4], const u64 b[4], const Mod* M) {
  u64 br = 0, t[4];
  for (int i = 0; i < 4; ++i) { u128 s = (u128)a[i] - b[i] - br; t[i] = (u64)s; br = (u64)(s >> 64) & 1; }
  if (br) { u64 c = 0; for (int i = 0; i < 4; ++i) { u128 s = (u128)t[i] + M->p[i] + c; t[i] = (u64)s; c = (u64)(s >> 64); } }
  memcpy(r, t, 32);
}
static inline int fp_is_zero(const u64 a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
static inline int fp_eq(const u64 a[4], const u64 b[4]) { return !((a[0]^b[0]) | (a[1]^b[1]) | (a[2]^b[2]) | (a[3]^b[3])); }
static inline void fp_neg(u64 r[4], const u64 a[4], const Mod* M) {
  if (fp_is_zero(a)) { memcpy(r, a, 32); return; }
  sub4(r, M->p, a);
}
static inline void fp_mul(u64 r[4], const u64 a[4], const u64 b[4], const Mod* M) {
  u64 t[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)a[j] * b[i] + t[j]; t[j] = (u64)c; c >>= 64; }
    u64 t4 = t[4] + (u64)c;
    u64 m = t[0] * M->inv;
    c = ((u128)m * M->p[0] + t[0]) >> 64;
    for (int j = 1; j < 4; ++j) { c += (u128)m * M->p[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
    c += t4; t[3] = (u64)c; t[4] = (u64)(c >> 64);
  }
  if (t[4] || ge4(t, M->p)) sub4(r, t, M->p); else memcpy(r, t, 32);
}
static inline void fp_sqr(u64 r[4], const u64 a[4], const Mod* M) { fp_mul(r, a, a, M); }
static void fp_pow(u64 r[4], const u64 a[4], const u64 e[4], const Mod* M) {
  u64 acc[4]; memcpy(acc, M->one, 32);
  for (int i = 255; i >= 0; --i) {
    fp_sqr(acc, acc, M);
    if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(acc, acc, a, M);
  }
  memcpy(r, acc, 32);
}
static void fp_inv(u64 r[4], const u64 a[4], const Mod* M) {
  u64 e[4]; memcpy(e, M->p, 32); e[0] -= 2;
  fp_pow(r, a, e, M);
}
static inline void fp_from_mont(u64 r[4], const u64 a[4], const Mod* M) {
  static const u64 one[4] = {1, 0, 0, 0};
  fp_mul(r, a, one, M);
}
static inline void fp_to_mont(u64 r[4], const u64 a[4], const Mod* M) { fp_mul(r, a, M->r2, M); }

/* ------------------------------------------------------------------ Fq / Fq2 as "FE" -------- */
typedef struct { u64 v[4]; } fq;
typedef struct { fq c0, c1; } fq2;

static inline void fq_add(fq* r, const fq* a, const fq* b) { fp_add(r->v, a->v, b->v, &FQ); }
static inline void fq_sub(fq* r, const fq* a, const fq* b) { fp_sub(r->v, a->v, b->v, &FQ); }
static inline void fq_mul(fq* r, const fq* a, const fq* b) { fp_mul(r->v, a->v, b->v, &FQ); }
static inline void fq_sqr(fq* r, const fq* a) { fp_mul(r->v, a->v, a->v, &FQ); }
static inline void fq_neg(fq* r, const fq* a) { fp_neg(r->v, a->v, &FQ); }
static inline void fq_dbl(fq* r, const fq* a) { fp_add(r->v, a->v, a->v, &FQ); }
static inline int fq_is_zero(const fq* a) { return fp_is_zero(a->v); }
static inline int fq_eq(const fq* a, const fq* b) { return fp_eq(a->v, b->v); }
static inline void fq_one(fq* r) { memcpy(r->v, FQ.one, 32); }
static inline void fq_zero(fq* r) { memset(r, 0, sizeof *r); }
static void fq_inv(fq* r, const fq* a) { fp_inv(r->v, a->v, &FQ); }

static inline void fq2_add(fq2* r, const fq2* a, const fq2* b) { fq_add(&r->c0, &a->c0, &b->c0); fq_add(&r->c1, &a->c1, &b->c1); }
static inline void fq2_sub(fq2* r, const fq2* a, const fq2* b) { fq_sub(&r->c0, &a->c0, &b->c0); fq_sub(&r->c1, &a->c1, &b->c1); }
static inline void fq2_neg(fq2* r, const fq2* a) { fq_neg(&r->c0, &a->c0); fq_neg(&r->c1, &a->c1); }
static inline void fq2_dbl(fq2* r, const fq2* a) { fq_dbl(&r->c0, &a->c0); fq_dbl(&r->c1, &a->c1); }
static inline void fq2_mul(fq2* r, const fq2* a, const fq2* b) {
  fq v0, v1, s, t;
  fq_mul(&v0, &a->c0, &b->c0); fq_mul(&v1, &a->c1, &b->c1);
  fq_add(&s, &a->c0, &a->c1); fq_add(&t, &b->c0, &b->c1); fq_mul(&s, &s, &t);
  fq_sub(&s, &s, &v0); fq_sub(&s, &s, &v1);
  fq_sub(&r->c0, &v0, &v1); r->c1 = s;
}
static inline void fq2_sqr(fq2* r, const fq2* a) {
  fq p, s, d;
  fq_mul(&p, &a->c0, &a->c1); fq_add(&s, &a->c0, &a->c1); fq_sub(&d, &a->c0, &a->c1);
  fq_mul(&r->c0, &s, &d); fq_dbl(&r->c1, &p);
}
static inline int fq2_is_zero(const fq2* a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }
static inline int fq2_eq(const fq2* a, const fq2* b) { return fq_eq(&a->c0, &b->c0) && fq_eq(&a->c1, &b->c1); }
static inline void fq2_one(fq2* r) { fq_one(&r->c0); fq_zero(&r->c1); }
static inline void fq2_zero(fq2* r) { memset(r, 0, sizeof *r); }
static void fq2_inv(fq2* r, const fq2* a) {
  fq n, t; fq_sqr(&n, &a->c0); fq_sqr(&t, &a->c1); fq_add(&n, &n, &t); fq_inv(&n, &n);
  fq_mul(&r->c0, &a->c0, &n); fq_mul(&t, &a->c1, &n); fq_neg(&r->c1, &t);
}

/* ------------------------------------------------------------------ curves (Jacobian, a = 0) - */
#define DEFINE_CURVE(G, FE)                                                                        \
  typedef struct { FE x, y; } G##_aff;                                                             \
  typedef struct { FE x, y, z; } G##_jac;                                                          \
  static inline int G##_aff_is_inf(const G##_aff* p) { return FE##_is_zero(&p->x) && FE##_is_zero(&p->y); } \
  static inline void G##_set_inf(G##_jac* p) { FE##_one(&p->x); FE##_one(&p->y); FE##_zero(&p->z); } \
  static inline int G##_is_inf(const G##_jac* p) { return FE##_is_zero(&p->z); }                   \
  static inline void G##_from_aff(G##_jac* r, const G##_aff* p) {                                  \
    if (G##_aff_is_inf(p)) { G##_set_inf(r); return; }                                             \
    r->x = p->x; r->y = p->y; FE##_one(&r->z); }                                                   \
  /* dbl-2009-l, as ark-ec double_in_place for a = 0 */                                            \
  static void G##_dbl(G##_jac* p) {                                                                \
    if (G##_is_inf(p)) return;                                                                     \
    FE a, b, c, d, e, f, t;                                                                        \
    FE##_sqr(&a, &p->x); FE##_sqr(&b, &p->y); FE##_sqr(&c, &b);                                    \
    FE##_add(&t, &p->x, &b); FE##_sqr(&t, &t); FE##_sub(&t, &t, &a); FE##_sub(&t, &t, &c); FE##_dbl(&d, &t); \
    FE##_dbl(&e, &a); FE##_add(&e, &e, &a); FE##_sqr(&f, &e);                                      \
    FE##_mul(&p->z, &p->z, &p->y); FE##_dbl(&p->z, &p->z);                                         \
    FE##_dbl(&t, &d); FE##_sub(&p->x, &f, &t);                                                     \
    FE##_sub(&t, &d, &p->x); FE##_mul(&t, &e, &t);                                                 \
    FE##_dbl(&c, &c); FE##_dbl(&c, &c); FE##_dbl(&c, &c); FE##_sub(&p->y, &t, &c); }               \
  /* madd-2007-bl, as ark-ec add_assign_mixed */                                                   \
  static void G##_madd(G##_jac* p, const G##_aff* q) {                                             \
    if (G##_aff_is_inf(q)) return;                                                                 \
    if (G##_is_inf(p)) { G##_from_aff(p, q); return; }                                             \
    FE z1z1, u2, s2, h, hh, i, j, r, v, t;                                                         \
    FE##_sqr(&z1z1, &p->z); FE##_mul(&u2, &q->x, &z1z1);                                           \
    FE##_mul(&s2, &q->y, &p->z); FE##_mul(&s2, &s2, &z1z1);                                        \
    if (FE##_eq(&p->x, &u2)) { if (FE##_eq(&p->y, &s2)) { G##_dbl(p); } else { G##_set_inf(p); } return; } \
    FE##_sub(&h, &u2, &p->x); FE##_sqr(&hh, &h); FE##_dbl(&i, &hh); FE##_dbl(&i, &i);              \
    FE##_mul(&j, &h, &i); FE##_sub(&r, &s2, &p->y); FE##_dbl(&r, &r); FE##_mul(&v, &p->x, &i);     \
    FE##_add(&t, &p->z, &h); FE##_sqr(&t, &t); FE##_sub(&t, &t, &z1z1); FE##_sub(&p->z, &t, &hh);  \
    FE##_sqr(&t, &r); FE##_sub(&t, &t, &j); FE##_sub(&t, &t, &v); FE##_sub(&p->x, &t, &v);         \
    FE##_sub(&t, &v, &p->x); FE##_mul(&t, &r, &t); FE##_mul(&j, &p->y, &j); FE##_dbl(&j, &j);      \
    FE##_sub(&p->y, &t, &j); }                                                                     \
  /* add-2007-bl, as ark-ec add_assign */                                                          \
  static void G##_add(G##_jac* p, const G##_jac* q) {                                              \
    if (G##_is_inf(q)) return;                                                                     \
    if (G##_is_inf(p)) { *p = *q; return; }                                                        \
    FE z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t;                                               \
    FE##_sqr(&z1z1, &p->z); FE##_sqr(&z2z2, &q->z);                                                \
    FE##_mul(&u1, &p->x, &z2z2); FE##_mul(&u2, &q->x, &z1z1);                                      \
    FE##_mul(&s1, &p->y, &q->z); FE##_mul(&s1, &s1, &z2z2);                                        \
    FE##_mul(&s2, &q->y, &p->z); FE##_mul(&s2, &s2, &z1z1);                                        \
    if (FE##_eq(&u1, &u2)) { if (FE##_eq(&s1, &s2)) { G##_dbl(p); } else { G##_set_inf(p); } return; } \
    FE##_sub(&h, &u2, &u1); FE##_dbl(&i, &h); FE##_sqr(&i, &i); FE##_mul(&j, &h, &i);              \
    FE##_sub(&r, &s2, &s1); FE##_dbl(&r, &r); FE##_mul(&v, &u1, &i);                               \
    FE##_add(&t, &p->z, &q->z); FE##_sqr(&t, &t); FE##_sub(&t, &t, &z1z1); FE##_sub(&t, &t, &z2z2); \
    FE##_mul(&p->z, &t, &h);                                                                       \
    FE##_sqr(&t, &r); FE##_sub(&t, &t, &j); FE##_sub(&t, &t, &v); FE##_sub(&p->x, &t, &v);         \
    FE##_sub(&t, &v, &p->x); FE##_mul(&t, &r, &t); FE##_mul(&s1, &s1, &j); FE##_dbl(&s1, &s1);     \
    FE##_sub(&p->y, &t, &s1); }                                                                    \
  static void G##_neg_aff(G##_aff* r, const G##_aff* p) { r->x = p->x; FE##_neg(&r->y, &p->y); }   \
  static void G##_to_aff(G##_aff* r, const G##_jac* p) {                                           \
    if (G##_is_inf(p)) { memset(r, 0, sizeof *r); return; }                                        \
    FE zi, zi2; FE##_inv(&zi, &p->z); FE##_sqr(&zi2, &zi);                                         \
    FE##_mul(&r->x, &p->x, &zi2); FE##_mul(&zi2, &zi2, &zi); FE##_mul(&r->y, &p->y, &zi2); }       \
  /* k*P, k canonical 4 x u64 (mul_bigint: MSB-first double and add) */                            \
  static void G##_mul(G##_jac* r, const G##_jac* p, const u64 k[4]) {                              \
    G##_jac acc; G##_set_inf(&acc);                                                                \
    for (int i = 255; i >= 0; --i) { G##_dbl(&acc); if ((k[i >> 6] >> (i & 63)) & 1) G##_add(&acc, p); } \
    *r = acc; }                                                                                    \
  /* VariableBaseMSM::msm_bigint (ark-ec 0.5 msm_bigint_wnaf).  scalars canonical. */              \
  static void G##_msm(G##_jac* out, const G##_aff* bases, const u64* scalars, size_t n) {          \
    if (n == 0) { G##_set_inf(out); return; }                                                      \
    int c;                                                                                         \
    if (n < 32) c = 3; else { int lg = 0; while (((size_t)1 << lg) < n) ++lg; c = lg * 69 / 100 + 2; } \
    if (g_msm_window > 1 && g_msm_window < 31) c = g_msm_window;  /* "all cores" column only */      \
    const int num_bits = 254, W = (num_bits + c - 1) / c;                                          \
    int32_t* digits = (int32_t*)malloc((size_t)n * W * sizeof(int32_t));                           \
    _Pragma("omp parallel for schedule(static)")                                                   \
    for (size_t i = 0; i < n; ++i) make_digits(scalars + 4 * i, c, W, digits + i * W);             \
    /* g_msm_chunks == 1: one task per window, as ark-ec's msm_bigint runs them (rayon over the     \
     * windows).  > 1 (g16cpu_set_msm_chunks: bench.py's conservative "all cores" CPU column): every  \
     * window's bases are cut into that many contiguous chunks, one task per (window, chunk) with its \
     * own buckets, and the chunk sums of a window are added -- the same group element */             \
    const int CH = g_msm_chunks > 1 ? g_msm_chunks : 1;                                            \
    G##_jac* wsum = (G##_jac*)malloc((size_t)W * CH * sizeof(G##_jac));                            \
    _Pragma("omp parallel for schedule(dynamic, 1)")                                               \
    for (int t = 0; t < W * CH; ++t) {                                                             \
      const int w = t / CH, ch = t % CH;                                                           \
      const size_t i_lo = n * (size_t)ch / CH, i_hi = n * (size_t)(ch + 1) / CH;                   \
      const size_t nb = (size_t)1 << (c - 1);                                                      \
      G##_jac* buckets = (G##_jac*)malloc(nb * sizeof(G##_jac));                                   \
      for (size_t b = 0; b < nb; ++b) G##_set_inf(&buckets[b]);                                    \
      for (size_t i = i_lo; i < i_hi; ++i) {                                                       \
        const int32_t d = digits[i * W + w];                                                       \
        if (d > 0) G##_madd(&buckets[d - 1], &bases[i]);                                           \
        else if (d < 0) { G##_aff nq; G##_neg_aff(&nq, &bases[i]); G##_madd(&buckets[-d - 1], &nq); } \
      }                                                                                            \
      G##_jac run, res; G##_set_inf(&run); G##_set_inf(&res);                                      \
      for (size_t b = nb; b-- > 0;) { G##_add(&run, &buckets[b]); G##_add(&res, &run); }           \
      wsum[t] = res; free(buckets);                                                                \
    }                                                                                              \
    for (int w = 0; w < W; ++w) {                                                                  \
      for (int ch = 1; ch < CH; ++ch) G##_add(&wsum[w * CH], &wsum[w * CH + ch]);                  \
      wsum[w] = wsum[w * CH];                                                                      \
    }                                                                                              \
    G##_jac total; G##_set_inf(&total);                                                            \
    for (int w = W - 1; w >= 1; --w) { G##_add(&total, &wsum[w]); for (int k = 0; k < c; ++k) G##_dbl(&total); } \
    G##_add(&total, &wsum[0]);                                                                     \
    *out = total; free(wsum); free(digits); }

static int g_msm_chunks = 1;  /* G##_msm: tasks per window (1 = ark-ec's shape) */
static int g_msm_window = 0;  /* G##_msm: window bits (0 = ark-ec's ln-based choice) */

/* ark-ec make_digits: signed radix-2^c digits, carry folded into the next window */
static void make_digits(const u64 s[4], int c, int W, int32_t* out) {
  const u64 radix = (u64)1 << c, window_mask = radix - 1;
  u64 carry = 0;
  for (int i = 0; i < W; ++i) {
    const int bit = i * c, u = bit >> 6, sh = bit & 63;
    u64 bits = 0;
    if (u < 4) {
      bits = s[u] >> sh;
      if (sh + c > 64 && u + 1 < 4) bits |= s[u + 1] << (64 - sh);
    }
    u64 coef = carry + (bits & window_mask);
    carry = (coef + radix / 2) >> c;
    int64_t d = (int64_t)coef - (int64_t)(carry << c);
    if (i == W - 1) d += (int64_t)(carry << c);   /* the last window absorbs its own carry */
    out[i] = (int32_t)d;
  }
}

DEFINE_CURVE(g1, fq)
DEFINE_CURVE(g2, fq2)

/* ------------------------------------------------------------------ radix-2 FFT over Fr ----- */
static void fr_root(u64 w[4], int log_n) {
  /* 5^((r-1)/2^28), squared down to order 2^log_n */
  u64 e[4]; memcpy(e, FR.p, 32); e[0] -= 1;
  u64 s[4];
  for (int i = 0; i < 4; ++i) s[i] = (e[i] >> 28) | (i + 1 < 4 ? e[i + 1] << 36 : 0);
  u64 five[4] = {5, 0, 0, 0}; fp_to_mont(five, five, &FR);
  fp_pow(w, five, s, &FR);
  for (int i = log_n; i < 28; ++i) fp_sqr(w, w, &FR);
}

/* in-place, natural order in and out (ark-poly fft_in_place / ifft_in_place semantics) */
static void fft(u64* a, int log_n, int inverse) {
  const size_t n = (size_t)1 << log_n;
  if (n == 1) return;
  u64 w[4]; fr_root(w, log_n);
  if (inverse) fp_inv(w, w, &FR);
  for (size_t i = 0; i < n; ++i) {   /* bit reversal */
    size_t j = 0;
    for (int b = 0; b < log_n; ++b) j |= ((i >> b) & 1) << (log_n - 1 - b);
    if (i < j) { u64 t[4]; memcpy(t, a + 4 * i, 32); memcpy(a + 4 * i, a + 4 * j, 32); memcpy(a + 4 * j, t, 32); }
  }
  u64* tw = (u64*)malloc((n / 2) * 32);   /* omega^i, i < n/2 */
  memcpy(tw, FR.one, 32);
  /* two-level fill so it parallelises: blocks of 1024 */
  {
    const size_t blk = 1024;
    for (size_t i = 1; i < (n / 2 < blk ? n / 2 : blk); ++i) fp_mul(tw + 4 * i, tw + 4 * (i - 1), w, &FR);
    if (n / 2 > blk) {
      u64 wb[4]; fp_mul(wb, tw + 4 * (blk - 1), w, &FR);   /* omega^blk */
      const size_t nblk = (n / 2) / blk;
      u64* heads = (u64*)malloc(nblk * 32);
      memcpy(heads, FR.one, 32);
      for (size_t b = 1; b < nblk; ++b) fp_mul(heads + 4 * b, heads + 4 * (b - 1), wb, &FR);
      _Pragma("omp parallel for schedule(static)")
      for (size_t b = 1; b < nblk; ++b)
        for (size_t i = 0; i < blk; ++i) fp_mul(tw + 4 * (b * blk + i), tw + 4 * i, heads + 4 * b, &FR);
      free(heads);
    }
  }
  for (int s = 1; s <= log_n; ++s) {
    const size_t len = (size_t)1 << s, half = len >> 1, step = n / len;
    _Pragma("omp parallel for schedule(static)")
    for (size_t k = 0; k < n / 2; ++k) {
      const size_t blk = k / half, i = k % half;
      u64* u = a + 4 * (blk * len + i);
      u64* v = u + 4 * half;
      u64 t[4], x[4];
      fp_mul(t, v, tw + 4 * (i * step), &FR);
      memcpy(x, u, 32);
      fp_add(u, x, t, &FR);
      fp_sub(v, x, t, &FR);
    }
  }
  free(tw);
  if (inverse) {
    u64 ni[4] = {(u64)n, 0, 0, 0}; fp_to_mont(ni, ni, &FR); fp_inv(ni, ni, &FR);
    _Pragma("omp parallel for schedule(static)")
    for (size_t i = 0; i < n; ++i) fp_mul(a + 4 * i, a + 4 * i, ni, &FR);
  }
}

/* distribute_powers_and_mul_by_const(v, g, 1): v[i] *= g^i (parallel in ark-poly) */
static void distribute_powers(u64* a, size_t n, const u64 g[4]) {
  const size_t blk = 4096;
  const size_t nblk = (n + blk - 1) / blk;
  u64* heads = (u64*)malloc(nblk * 32);
  u64 gb[4]; memcpy(gb, FR.one, 32);
  for (size_t i = 0; i < blk; ++i) fp_mul(gb, gb, g, &FR);
  memcpy(heads, FR.one, 32);
  for (size_t b = 1; b < nblk; ++b) fp_mul(heads + 4 * b, heads + 4 * (b - 1), gb, &FR);
  _Pragma("omp parallel for schedule(static)")
  for (size_t b = 0; b < nblk; ++b) {
    u64 p[4]; memcpy(p, heads + 4 * b, 32);
    const size_t e = (b + 1) * blk < n ? (b + 1) * blk : n;
    for (size_t i = b * blk; i < e; ++i) { fp_mul(a + 4 * i, a + 4 * i, p, &FR); fp_mul(p, p, g, &FR); }
  }
  free(heads);
}

/* ------------------------------------------------------------------ public API -------------- */
typedef struct { const uint32_t* row_ptr; const uint32_t* col; const u64* coeff; u64 nnz; } csr_t;
typedef struct {
  uint32_t n_vars, n_public, domain_size;
  const uint8_t *a_query, *b_g1_query, *b_g2_query, *l_query, *h_query;
  uint8_t alpha_g1[64], beta_g1[64], delta_g1[64], beta_g2[128], delta_g2[128];
} pkey_t;

void g16cpu_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
/* chunks of bases per MSM window (see G##_msm); 1 restores the ark-ec shape */
void g16cpu_set_msm_chunks(int n) { g_msm_chunks = n > 1 ? n : 1; }
/* window bits of every MSM (0 restores ark-ec's choice): smaller windows keep a task's buckets in its core's
 * cache when all host threads run MSM tasks at once (2^16 buckets x 96 B x 128 threads does not) */
void g16cpu_set_msm_window(int c) { g_msm_window = c; }
int g16cpu_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* evaluate_constraint (qap.rs:42-43): sum coeff * w[idx]; multiply skipped when coeff == 1 */
static void eval_row(u64 out[4], const csr_t* m, uint32_t i, const u64* w) {
  u64 acc[4] = {0, 0, 0, 0};
  for (uint32_t j = m->row_ptr[i]; j < m->row_ptr[i + 1]; ++j) {
    u64 t[4];
    const u64* cf = m->coeff + 4 * (size_t)j;
    const u64* x = w + 4 * (size_t)m->col[j];
    if (fp_eq(cf, FR.one)) memcpy(t, x, 32); else fp_mul(t, x, cf, &FR);
    fp_add(acc, acc, t, &FR);
  }
  memcpy(out, acc, 32);
}

/* qap.rs:23-88.  Returns 0, or 2 (PolynomialDegreeTooLarge). h_out: domain_size x 4 u64. */
int g16cpu_witness_map(const csr_t* A, const csr_t* B, uint32_t num_inputs, uint32_t m,
                       const u64* w, u64* h_out, uint32_t* domain_out) {
  int k = 0;
  while (((u64)1 << k) < (u64)m + num_inputs) ++k;
  if (k + 1 > 28) return 2;
  const size_t n = (size_t)1 << k;
  if (domain_out) *domain_out = (uint32_t)n;
  u64* a = (u64*)calloc(n, 32);
  u64* b = (u64*)calloc(n, 32);
  u64* c = (u64*)calloc(n, 32);
  for (uint32_t i = 0; i < m; ++i) { eval_row(a + 4 * (size_t)i, A, i, w); eval_row(b + 4 * (size_t)i, B, i, w); }
  memcpy(a + 4 * (size_t)m, w, (size_t)num_inputs * 32);
  for (uint32_t i = 0; i < m; ++i) fp_mul(c + 4 * (size_t)i, a + 4 * (size_t)i, b + 4 * (size_t)i, &FR);
  fft(a, k, 1); fft(b, k, 1);
  u64 g[4]; fr_root(g, k + 1);
  distribute_powers(a, n, g); distribute_powers(b, n, g);
  fft(a, k, 0); fft(b, k, 0);
  _Pragma("omp parallel for schedule(static)")
  for (size_t i = 0; i < n; ++i) fp_mul(a + 4 * i, a + 4 * i, b + 4 * i, &FR);
  fft(c, k, 1); distribute_powers(c, n, g); fft(c, k, 0);
  for (size_t i = 0; i < n; ++i) fp_sub(h_out + 4 * i, a + 4 * i, c + 4 * i, &FR);
  free(a); free(b); free(c);
  return 0;
}

/* ark-groth16 0.5 LibsnarkReduction::witness_map_from_matrices (un-vendored crate: restated from its
 * published algorithm, as oracle/bn254_ref.py:witness_map_libsnark, which pins this function at
 * small sizes -- tests/test_oracle.py).  The reference reaches it as the default QAP of
 * `Groth16<Bn254>` (tests/groth16.rs:9,25-35).  a, b, c on the size-n domain -> coefficients ->
 * coset g H with g = Fr::GENERATOR = 5 (distribute_powers + fft) -> (a b - c) / Z_H(g) pointwise
 * -> coset ifft (ifft, then powers of 1/g).  c_i = a_i b_i (the matrices hold A and B only).
 * h_out: n coefficients, the top one is 0.  Returns 0, or 2 (PolynomialDegreeTooLarge).          */
int g16cpu_witness_map_libsnark(const csr_t* A, const csr_t* B, uint32_t num_inputs, uint32_t m,
                                const u64* w, u64* h_out, uint32_t* domain_out) {
  int k = 0;
  while (((u64)1 << k) < (u64)m + num_inputs) ++k;
  if (k > 28) return 2;
  const size_t n = (size_t)1 << k;
  if (domain_out) *domain_out = (uint32_t)n;
  u64* a = (u64*)calloc(n, 32);
  u64* b = (u64*)calloc(n, 32);
  u64* c = (u64*)calloc(n, 32);
  _Pragma("omp parallel for schedule(static)")
  for (uint32_t i = 0; i < m; ++i) {
    eval_row(a + 4 * (size_t)i, A, i, w);
    eval_row(b + 4 * (size_t)i, B, i, w);
    fp_mul(c + 4 * (size_t)i, a + 4 * (size_t)i, b + 4 * (size_t)i, &FR);
  }
  memcpy(a + 4 * (size_t)m, w, (size_t)num_inputs * 32);
  u64 g[4] = {5, 0, 0, 0}; fp_to_mont(g, g, &FR);
  fft(a, k, 1); distribute_powers(a, n, g); fft(a, k, 0);
  fft(b, k, 1); distribute_powers(b, n, g); fft(b, k, 0);
  fft(c, k, 1); distribute_powers(c, n, g); fft(c, k, 0);
  /* Z_H(g w^i) = g^n - 1 for every i */
  u64 zi[4]; memcpy(zi, g, 32);
  for (int i = 0; i < k; ++i) fp_sqr(zi, zi, &FR);
  fp_sub(zi, zi, FR.one, &FR); fp_inv(zi, zi, &FR);
  _Pragma("omp parallel for schedule(static)")
  for (size_t i = 0; i < n; ++i) {
    u64 t[4];
    fp_mul(t, a + 4 * i, b + 4 * i, &FR);
    fp_sub(t, t, c + 4 * i, &FR);
    fp_mul(h_out + 4 * i, t, zi, &FR);
  }
  u64 gi[4]; fp_inv(gi, g, &FR);
  fft(h_out, k, 1); distribute_powers(h_out, n, gi);
  free(a); free(b); free(c);
  return 0;
}

static u64* to_canonical(const u64* in, size_t n) {
  u64* out = (u64*)malloc((n ? n : 1) * 32);
  _Pragma("omp parallel for schedule(static)")
  for (size_t i = 0; i < n; ++i) fp_from_mont(out + 4 * i, in + 4 * i, &FR);
  return out;
}

/* scalars: Montgomery Fr (into_bigint is part of the timed path upstream as well) */
void g16cpu_msm_g1(const uint8_t* bases, const u64* scalars_mont, size_t n, uint8_t out[64]) {
  u64* s = to_canonical(scalars_mont, n);
  g1_jac r; g1_msm(&r, (const g1_aff*)bases, s, n);
  g1_aff a; g1_to_aff(&a, &r); memcpy(out, &a, 64); free(s);
}
void g16cpu_msm_g2(const uint8_t* bases, const u64* scalars_mont, size_t n, uint8_t out[128]) {
  u64* s = to_canonical(scalars_mont, n);
  g2_jac r; g2_msm(&r, (const g2_aff*)bases, s, n);
  g2_aff a; g2_to_aff(&a, &r); memcpy(out, &a, 128); free(s);
}

/* create_proof_with_reduction_and_matrices.  r, s Montgomery.  h_opt: if non-NULL receives h.
 * reduction 0 = CircomReduction, 1 = LibsnarkReduction (H query of domain_size - 1 points; the
 * caller's array may carry a padding entry, which msm_bigint's min(bases, scalars) never reads). */
int g16cpu_prove_ex(const pkey_t* key, const csr_t* A, const csr_t* B, uint32_t m, const u64 r[4],
                    const u64 s_[4], const u64* w, uint8_t proof[256], u64* h_opt, int reduction) {
  const uint32_t N = key->n_vars, p = key->n_public, ni = p + 1;
  uint32_t n = 0;
  u64* h = (u64*)malloc((size_t)key->domain_size * 32);
  int st = reduction == 1 ? g16cpu_witness_map_libsnark(A, B, ni, m, w, h, &n)
                          : g16cpu_witness_map(A, B, ni, m, w, h, &n);
  if (st) { free(h); return st; }
  if (n != key->domain_size) { free(h); return 1; }
  if (h_opt) memcpy(h_opt, h, (size_t)n * 32);
  if (reduction == 1) n -= 1;   /* h_query holds n - 1 points */
  u64 rc[4], sc[4], rs[4];
  fp_from_mont(rc, r, &FR); fp_from_mont(sc, s_, &FR);
  fp_mul(rs, r, s_, &FR); fp_from_mont(rs, rs, &FR);

  u64* hc = to_canonical(h, n);
  g1_jac h_acc; g1_msm(&h_acc, (const g1_aff*)key->h_query, hc, n);
  free(hc); free(h);
  u64* wc = to_canonical(w, N);
  g1_jac l_acc; g1_msm(&l_acc, (const g1_aff*)key->l_query, wc + 4 * (size_t)ni, N - ni);

  g1_jac delta1; g1_from_aff(&delta1, (const g1_aff*)key->delta_g1);
  g1_jac g_a, g1_b, t1;
  g1_mul(&g_a, &delta1, rc);                                   /* r * delta */
  g1_madd(&g_a, (const g1_aff*)key->a_query);                  /* + a_query[0] */
  g1_msm(&t1, (const g1_aff*)key->a_query + 1, wc + 4, N - 1); g1_add(&g_a, &t1);
  g1_madd(&g_a, (const g1_aff*)key->alpha_g1);
  g1_mul(&g1_b, &delta1, sc);
  g1_madd(&g1_b, (const g1_aff*)key->b_g1_query);
  g1_msm(&t1, (const g1_aff*)key->b_g1_query + 1, wc + 4, N - 1); g1_add(&g1_b, &t1);
  g1_madd(&g1_b, (const g1_aff*)key->beta_g1);
  g2_jac delta2, g2_b, t2; g2_from_aff(&delta2, (const g2_aff*)key->delta_g2);
  g2_mul(&g2_b, &delta2, sc);
  g2_madd(&g2_b, (const g2_aff*)key->b_g2_query);
  g2_msm(&t2, (const g2_aff*)key->b_g2_query + 1, wc + 4, N - 1); g2_add(&g2_b, &t2);
  g2_madd(&g2_b, (const g2_aff*)key->beta_g2);
  free(wc);

  g1_jac g_c, tmp;
  g1_mul(&g_c, &g_a, sc);                    /* s * g_a */
  g1_mul(&tmp, &g1_b, rc); g1_add(&g_c, &tmp); /* + r * g1_b */
  g1_mul(&tmp, &delta1, rs);                 /* - rs * delta */
  { g1_aff ta, tn; g1_to_aff(&ta, &tmp); g1_neg_aff(&tn, &ta); g1_madd(&g_c, &tn); }
  g1_add(&g_c, &l_acc); g1_add(&g_c, &h_acc);
  g1_aff pa, pc; g2_aff pb;
  g1_to_aff(&pa, &g_a); g2_to_aff(&pb, &g2_b); g1_to_aff(&pc, &g_c);
  memcpy(proof, &pa, 64); memcpy(proof + 64, &pb, 128); memcpy(proof + 192, &pc, 64);
  return 0;
}

int g16cpu_prove(const pkey_t* key, const csr_t* A, const csr_t* B, uint32_t m, const u64 r[4],
                 const u64 s_[4], const u64* w, uint8_t proof[256], u64* h_opt) {
  return g16cpu_prove_ex(key, A, B, m, r, s_, w, proof, h_opt, 0);
}

/* k * P for tests (P affine bytes, k canonical) */
void g16cpu_g1_mul(const uint8_t P[64], const u64 k[4], uint8_t out[64]) {
  g1_jac j, r; g1_from_aff(&j, (const g1_aff*)P); g1_mul(&r, &j, k);
  g1_aff a; g1_to_aff(&a, &r); memcpy(out, &a, 64);
}
void g16cpu_g2_mul(const uint8_t P[128], const u64 k[4], uint8_t out[128]) {
  g2_jac j, r; g2_from_aff(&j, (const g2_aff*)P); g2_mul(&r, &j, k);
  g2_aff a; g2_to_aff(&a, &r); memcpy(out, &a, 128);
}
/* out[i] = k_i * P for n canonical scalars: how the tests turn the oracle's trapdoor SCALARS into the
 * key points they compare a key generator's output with (plain MSB-first double-and-add, G##_mul) */
void g16cpu_g1_mul_batch(const uint8_t P[64], const u64* k, size_t n, uint8_t* out) {
  _Pragma("omp parallel for schedule(dynamic, 16)")
  for (size_t i = 0; i < n; ++i) g16cpu_g1_mul(P, k + 4 * i, out + 64 * i);
}
void g16cpu_g2_mul_batch(const uint8_t P[128], const u64* k, size_t n, uint8_t* out) {
  _Pragma("omp parallel for schedule(dynamic, 16)")
  for (size_t i = 0; i < n; ++i) g16cpu_g2_mul(P, k + 4 * i, out + 128 * i);
}
void g16cpu_fft(u64* data, int log_n, int inverse) { fft(data, log_n, inverse); }
