"""oracle/bn254_ref.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Pure-Python big-integer restatement of the Groth16 proving path that
arkworks-rs/circom-compat exposes, used as the *checker* for the HIP kernels:

  * CircomReduction::witness_map_from_matrices  -> reference src/circom/qap.rs:23-88
  * CircomReduction::h_query_scalars            -> reference src/circom/qap.rs:90-105
  * read_zkey / BinFile                         -> reference src/zkey.rs:53-60,73-133,151-196,288-368
  * R1CSFile::new / R1CS::from                  -> reference src/circom/r1cs_reader.rs:26-39,54-249
  * CircomCircuit::get_public_inputs            -> reference src/circom/circuit.rs:18-26
  * Groth16::create_proof_with_reduction_and_matrices, create_proof_with_assignment,
    process_vk / verify_with_processed_vk       -> call sites reference src/zkey.rs:866-870,903-916,
                                                   benches/groth16.rs:52-67

The prover/verifier/MSM/FFT arithmetic itself lives in crates that are NOT vendored in
/root/reference (ark-groth16, ark-ec, ark-poly, ark-ff, ark-bn254, all "0.5.0", reference
Cargo.toml:24-32, no Cargo.lock pin).  Their published algorithms are restated here; parity is
anchored on (a) every golden byte vector the reference's own tests hold for the loaders
(src/zkey.rs:398-432,465-517,545-779, src/circom/r1cs_reader.rs:257-338), (b) the
reference's proof predicate `verify_with_processed_vk == true/false` evaluated by the
pairing verifier below on the reference's own test.zkey, and (c) SURVEY.md Appendix C KATs.
Proof *bytes* are never pinned by the reference (r,s come from thread_rng, src/zkey.rs:865);
they are mathematically unique given (pk, r, s, witness), which is what bit-exact parity means.
"""
from __future__ import annotations

import struct

# ----------------------------------------------------------------------------------------------
# constants (r = witness_calculator.rs:330, r1cs_reader.rs:181)
# ----------------------------------------------------------------------------------------------
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # Fr
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # Fq
MONT_R = 1 << 256
FR_GENERATOR = 5
FR_TWO_ADICITY = 28
FR_TWO_ADIC_ROOT = pow(FR_GENERATOR, (R_MOD - 1) >> FR_TWO_ADICITY, R_MOD)
BN_X = 4965661367192848881
ATE_LOOP = 6 * BN_X + 2
G1_B = 3
G1_GEN = (1, 2)
XI = (9, 1)  # Fq2 non-residue 9+i


def fr_inv(a):
    return pow(a, R_MOD - 2, R_MOD)


def fq_inv(a):
    return pow(a, Q_MOD - 2, Q_MOD)


# ----------------------------------------------------------------------------------------------
# Fq2 = Fq[i]/(i^2+1)
# ----------------------------------------------------------------------------------------------
def f2_add(a, b):
    return ((a[0] + b[0]) % Q_MOD, (a[1] + b[1]) % Q_MOD)


def f2_sub(a, b):
    return ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD)


def f2_neg(a):
    return ((-a[0]) % Q_MOD, (-a[1]) % Q_MOD)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)


def f2_sqr(a):
    return f2_mul(a, a)


def f2_inv(a):
    n = fq_inv((a[0] * a[0] + a[1] * a[1]) % Q_MOD)
    return (a[0] * n % Q_MOD, (-a[1]) * n % Q_MOD)


def f2_conj(a):
    return (a[0], (-a[1]) % Q_MOD)


def f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


G2_B = f2_mul((3, 0), f2_inv(XI))
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)  # reference src/zkey.rs:443-463


# ----------------------------------------------------------------------------------------------
# generic short-Weierstrass arithmetic (a = 0), Jacobian coordinates, field ops passed in
# ----------------------------------------------------------------------------------------------
class _Fld:
    pass


class _FqOps(_Fld):
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % Q_MOD)
    sub = staticmethod(lambda a, b: (a - b) % Q_MOD)
    mul = staticmethod(lambda a, b: (a * b) % Q_MOD)
    neg = staticmethod(lambda a: (-a) % Q_MOD)
    inv = staticmethod(fq_inv)
    b = G1_B


class _Fq2Ops(_Fld):
    zero, one = (0, 0), (1, 0)
    add = staticmethod(f2_add)
    sub = staticmethod(f2_sub)
    mul = staticmethod(f2_mul)
    neg = staticmethod(f2_neg)
    inv = staticmethod(f2_inv)
    b = G2_B


class Curve:
    """Points are affine tuples (x, y) or None for infinity.  Internals use Jacobian (X,Y,Z)."""

    def __init__(self, F):
        self.F = F

    def on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.mul(y, y) == F.add(F.mul(F.mul(x, x), x), F.b)

    # -- jacobian helpers
    def _jdbl(self, P):
        F = self.F
        X, Y, Z = P
        if Z == F.zero:
            return P
        A = F.mul(X, X)
        B = F.mul(Y, Y)
        C = F.mul(B, B)
        t = F.add(X, B)
        D = F.sub(F.sub(F.mul(t, t), A), C)
        D = F.add(D, D)
        E = F.add(F.add(A, A), A)
        Fq_ = F.mul(E, E)
        X3 = F.sub(Fq_, F.add(D, D))
        C8 = F.add(C, C)
        C8 = F.add(C8, C8)
        C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        Z3 = F.mul(F.add(Y, Y), Z)
        return (X3, Y3, Z3)

    def _jadd(self, P, Q):
        F = self.F
        if P[2] == F.zero:
            return Q
        if Q[2] == F.zero:
            return P
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Q
        Z1Z1 = F.mul(Z1, Z1)
        Z2Z2 = F.mul(Z2, Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(F.mul(Y1, Z2), Z2Z2)
        S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
        if U1 == U2:
            if S1 == S2:
                return self._jdbl(P)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, U1)
        Rr = F.sub(S2, S1)
        HH = F.mul(H, H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.mul(Rr, Rr), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def _to_j(self, P):
        F = self.F
        return (F.one, F.one, F.zero) if P is None else (P[0], P[1], F.one)

    def _to_a(self, P):
        F = self.F
        if P[2] == F.zero:
            return None
        zi = F.inv(P[2])
        zi2 = F.mul(zi, zi)
        return (F.mul(P[0], zi2), F.mul(P[1], F.mul(zi2, zi)))

    # -- affine API
    def add(self, P, Q):
        return self._to_a(self._jadd(self._to_j(P), self._to_j(Q)))

    def neg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    def sub(self, P, Q):
        return self.add(P, self.neg(Q))

    def mul(self, P, k):
        F = self.F
        if k < 0:
            return self.mul(self.neg(P), -k)
        acc = (F.one, F.one, F.zero)
        base = self._to_j(P)
        while k:
            if k & 1:
                acc = self._jadd(acc, base)
            base = self._jdbl(base)
            k >>= 1
        return self._to_a(acc)

    def sum(self, pts):
        acc = (self.F.one, self.F.one, self.F.zero)
        for P in pts:
            acc = self._jadd(acc, self._to_j(P))
        return self._to_a(acc)

    def msm(self, bases, scalars):
        """sum_i scalars[i]*bases[i] (scalars canonical ints).  Bucket method; the result is the
        unique group element, so it equals ark-ec VariableBaseMSM::msm_bigint for any window."""
        n = min(len(bases), len(scalars))
        if n == 0:
            return None
        if n >= 256:
            # sum_i s_i B_i is linear in the scalars: scalars of EQUAL bases are added first.  Changes
            # nothing for a real key (all bases distinct: one dict pass); the keys of the large C-vs-Python
            # pins cycle through 64 points (tests/test_oracle.py) and fold to 64 terms.
            first, ub, us = {}, [], []
            for b, k in zip(bases[:n], scalars[:n]):
                if b is None:
                    continue
                j = first.get(b)
                if j is None:
                    first[b] = len(ub)
                    ub.append(b)
                    us.append(k % R_MOD)
                else:
                    us[j] = (us[j] + k) % R_MOD
            if 2 * len(ub) <= n:
                return self.msm(ub, us) if ub else None
        F = self.F
        inf = (F.one, F.one, F.zero)
        if n < 8:
            acc = inf
            for b, s in zip(bases[:n], scalars[:n]):
                acc = self._jadd(acc, self._to_j(self.mul(b, s % R_MOD)))
            return self._to_a(acc)
        c = max(2, min(12, n.bit_length() - 2))
        nwin = (254 + c - 1) // c
        jb = [self._to_j(b) for b in bases[:n]]
        total = inf
        for w in reversed(range(nwin)):
            for _ in range(c):
                total = self._jdbl(total)
            buckets = [inf] * ((1 << c) - 1)
            sh = w * c
            mask = (1 << c) - 1
            for b, s in zip(jb, scalars[:n]):
                d = ((s % R_MOD) >> sh) & mask
                if d:
                    buckets[d - 1] = self._jadd(buckets[d - 1], b)
            run = inf
            acc = inf
            for b in reversed(buckets):
                run = self._jadd(run, b)
                acc = self._jadd(acc, run)
            total = self._jadd(total, acc)
        return self._to_a(total)


G1 = Curve(_FqOps)
G2 = Curve(_Fq2Ops)


# ----------------------------------------------------------------------------------------------
# radix-2 evaluation domain over Fr (ark-poly Radix2EvaluationDomain semantics: natural order in
# and out, inverse scales by 1/n, element(i) = omega_n^i)
# ----------------------------------------------------------------------------------------------
def domain_size_for(k):
    """EvaluationDomain::new(k): smallest power of two >= k (None if > 2^28)."""
    n = 1
    lg = 0
    while n < k:
        n <<= 1
        lg += 1
    if lg > FR_TWO_ADICITY:
        return None
    return n


def root_of_unity(n):
    lg = n.bit_length() - 1
    assert 1 << lg == n and lg <= FR_TWO_ADICITY
    return pow(FR_TWO_ADIC_ROOT, 1 << (FR_TWO_ADICITY - lg), R_MOD)


def _bitrev_permute(a):
    n = len(a)
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]


def ntt(vals, inverse=False):
    a = list(vals)
    n = len(a)
    if n == 1:
        return a
    w = root_of_unity(n)
    if inverse:
        w = fr_inv(w)
    _bitrev_permute(a)
    length = 2
    while length <= n:
        wl = pow(w, n // length, R_MOD)
        half = length >> 1
        tw = [1] * half
        for i in range(1, half):
            tw[i] = tw[i - 1] * wl % R_MOD
        for s in range(0, n, length):
            for i in range(half):
                u = a[s + i]
                v = a[s + i + half] * tw[i] % R_MOD
                a[s + i] = (u + v) % R_MOD
                a[s + i + half] = (u - v) % R_MOD
        length <<= 1
    if inverse:
        ni = fr_inv(n)
        a = [x * ni % R_MOD for x in a]
    return a


def dft_naive(vals, inverse=False):
    n = len(vals)
    w = root_of_unity(n)
    if inverse:
        w = fr_inv(w)
    out = []
    for i in range(n):
        acc = 0
        wi = pow(w, i, R_MOD)
        x = 1
        for v in vals:
            acc = (acc + v * x) % R_MOD
            x = x * wi % R_MOD
        out.append(acc)
    if inverse:
        ni = fr_inv(n)
        out = [x * ni % R_MOD for x in out]
    return out


# ----------------------------------------------------------------------------------------------
# CircomReduction  (reference src/circom/qap.rs)
# ----------------------------------------------------------------------------------------------
def evaluate_constraint(row, assignment):
    """ark-groth16 r1cs_to_qap::evaluate_constraint (imported qap.rs:2, called qap.rs:42-43):
    sum coeff*w[idx] over one sparse row of (coeff, index)."""
    acc = 0
    for coeff, idx in row:
        acc += coeff * assignment[idx]
    return acc % R_MOD


def witness_map_from_matrices(a_rows, b_rows, num_inputs, num_constraints, full_assignment):
    """Line-by-line restatement of reference src/circom/qap.rs:23-88.  Returns h (len n) or
    raises ValueError('PolynomialDegreeTooLarge') (qap.rs:31,66)."""
    n = domain_size_for(num_constraints + num_inputs)          # qap.rs:30-32
    if n is None or domain_size_for(2 * n) is None:
        raise ValueError("PolynomialDegreeTooLarge")
    a = [0] * n
    b = [0] * n
    for i in range(num_constraints):                           # qap.rs:37-44
        a[i] = evaluate_constraint(a_rows[i], full_assignment)
        b[i] = evaluate_constraint(b_rows[i], full_assignment)
    for i in range(num_inputs):                                # qap.rs:46-50
        a[num_constraints + i] = full_assignment[i] % R_MOD
    c = [0] * n
    for i in range(num_constraints):                           # qap.rs:52-58
        c[i] = a[i] * b[i] % R_MOD
    a = ntt(a, inverse=True)                                   # qap.rs:60-61
    b = ntt(b, inverse=True)
    w2n = root_of_unity(2 * n)                                 # qap.rs:63-68
    pw = 1
    for i in range(n):                                         # qap.rs:69-70
        a[i] = a[i] * pw % R_MOD
        b[i] = b[i] * pw % R_MOD
        pw = pw * w2n % R_MOD
    a = ntt(a)                                                 # qap.rs:72-73
    b = ntt(b)
    ab = [x * y % R_MOD for x, y in zip(a, b)]                 # qap.rs:75
    c = ntt(c, inverse=True)                                   # qap.rs:79-81
    pw = 1
    for i in range(n):
        c[i] = c[i] * pw % R_MOD
        pw = pw * w2n % R_MOD
    c = ntt(c)
    return [(x - y) % R_MOD for x, y in zip(ab, c)]            # qap.rs:83-87


def witness_map_libsnark(a_rows, b_rows, num_inputs, num_constraints, full_assignment, c_rows=None):
    """ark-groth16 0.5 LibsnarkReduction::witness_map_from_matrices (un-vendored crate, restated from
    its published algorithm; the reference reaches it as the default QAP of `Groth16<Bn254>`,
    tests/groth16.rs:9,25-35): a, b (and c) evaluated on the size-n domain, moved to the coset
    g * H (g = Fr::GENERATOR = 5), h = (a b - c) / Z_H on the coset, inverse coset FFT.
    Returns the n coefficients of h (the top one is 0).  c_rows=None: c_i = a_i b_i, which is what
    C . w equals for a satisfying assignment."""
    n = domain_size_for(num_constraints + num_inputs)
    if n is None:
        raise ValueError("PolynomialDegreeTooLarge")
    a = [0] * n
    b = [0] * n
    c = [0] * n
    for i in range(num_constraints):
        a[i] = evaluate_constraint(a_rows[i], full_assignment)
        b[i] = evaluate_constraint(b_rows[i], full_assignment)
        c[i] = a[i] * b[i] % R_MOD if c_rows is None else evaluate_constraint(c_rows[i], full_assignment)
    for i in range(num_inputs):
        a[num_constraints + i] = full_assignment[i] % R_MOD
    g = FR_GENERATOR

    def to_coset(v):
        v = ntt(v, inverse=True)
        pw = 1
        for i in range(n):
            v[i] = v[i] * pw % R_MOD
            pw = pw * g % R_MOD
        return ntt(v)

    a, b, c = to_coset(a), to_coset(b), to_coset(c)
    z_inv = fr_inv((pow(g, n, R_MOD) - 1) % R_MOD)             # 1 / Z_H(g w^i), constant on the coset
    h = [(x * y - z) * z_inv % R_MOD for x, y, z in zip(a, b, c)]
    h = ntt(h, inverse=True)
    gi = fr_inv(g)
    pw = 1
    for i in range(n):
        h[i] = h[i] * pw % R_MOD
        pw = pw * gi % R_MOD
    return h


def h_query_scalars_libsnark(max_power, t, zt, delta_inverse):
    """ark-groth16 LibsnarkReduction::h_query_scalars: zt / delta * t^i for i < max_power."""
    return [zt * delta_inverse % R_MOD * pow(t, i, R_MOD) % R_MOD for i in range(max_power)]


def h_query_scalars(max_power, t, delta_inverse, ntt_fn=None):
    """reference src/circom/qap.rs:90-105.  ntt_fn: another (pinned) implementation of ntt() for the
    one transform of size 2n (tests at 2^16 .. 2^20 pass the C restatement's FFT)."""
    scalars = [delta_inverse * pow(t, i, R_MOD) % R_MOD for i in range(2 * max_power + 1)]
    size = domain_size_for(len(scalars))
    if size is None:
        raise ValueError("PolynomialDegreeTooLarge")
    scalars = scalars + [0] * (size - len(scalars))            # ifft_in_place resizes to the domain
    scalars = (ntt_fn or ntt)(scalars, inverse=True)
    return scalars[1::2]


# ----------------------------------------------------------------------------------------------
# file formats
# ----------------------------------------------------------------------------------------------
R1CS_PRIME_LE = bytes.fromhex("010000f093f5e1439170b97948e833285d588181b64550b829a031e1724e6430")


def _le(b):
    return int.from_bytes(b, "little")


def read_r1cs(data: bytes):
    """reference src/circom/r1cs_reader.rs:54-249 (+ R1CS::from :26-39).  Returns dict."""
    if data[:4] != b"r1cs":
        raise ValueError("Invalid magic number")
    version, nsec = struct.unpack_from("<II", data, 4)
    if version != 1:
        raise ValueError("Unsupported version")
    off = 12
    secs = {}
    for _ in range(nsec):
        typ, size = struct.unpack_from("<IQ", data, off)
        off += 12
        secs[typ] = (off, size)          # later duplicates overwrite, like the HashMap insert
        off += size
    for t, what in ((1, "header"), (2, "constraint"), (3, "wire2label")):
        if t not in secs:
            raise ValueError(f"No section offset for {what} type found")
    o, size = secs[1]
    (field_size,) = struct.unpack_from("<I", data, o)
    if field_size != 32:
        raise ValueError("This parser only supports 32-byte fields")
    if size != 32 + field_size:
        raise ValueError("Invalid header section size")
    prime = data[o + 4:o + 36]
    if prime != R1CS_PRIME_LE:
        raise ValueError("This parser only supports bn256")
    n_wires, n_pub_out, n_pub_in, n_prv_in, n_labels, n_constraints = struct.unpack_from(
        "<IIIIQI", data, o + 36)
    o, _ = secs[2]
    constraints = []
    for _ in range(n_constraints):
        lcs = []
        for _k in range(3):
            (cnt,) = struct.unpack_from("<I", data, o)
            o += 4
            lc = []
            for _j in range(cnt):
                (wire,) = struct.unpack_from("<I", data, o)
                val = _le(data[o + 4:o + 36])
                if val >= R_MOD:
                    raise ValueError("non-canonical field element")
                lc.append((wire, val))
                o += 36
            lcs.append(lc)
        constraints.append(tuple(lcs))
    o, size = secs[3]
    if size != n_wires * 8:
        raise ValueError("Invalid map section size")
    wire_mapping = list(struct.unpack_from(f"<{n_wires}Q", data, o))
    if wire_mapping[0] != 0:
        raise ValueError("Wire 0 should always be mapped to 0")
    num_inputs = 1 + n_pub_in + n_pub_out
    return dict(version=version, field_size=field_size, prime=prime, n_wires=n_wires,
                n_pub_out=n_pub_out, n_pub_in=n_pub_in, n_prv_in=n_prv_in, n_labels=n_labels,
                n_constraints=n_constraints, constraints=constraints, wire_mapping=wire_mapping,
                num_inputs=num_inputs, num_variables=n_wires, num_aux=n_wires - num_inputs)


def read_wtns(data: bytes):
    """snarkjs .wtns (not parsed by the reference)."""
    if data[:4] != b"wtns":
        raise ValueError("bad wtns magic")
    _ver, nsec = struct.unpack_from("<II", data, 4)
    off = 12
    secs = {}
    for _ in range(nsec):
        typ, size = struct.unpack_from("<IQ", data, off)
        off += 12
        secs[typ] = (off, size)
        off += size
    o, _ = secs[1]
    (n8,) = struct.unpack_from("<I", data, o)
    (nw,) = struct.unpack_from("<I", data, o + 4 + n8)
    o, _ = secs[2]
    return [_le(data[o + i * n8:o + (i + 1) * n8]) for i in range(nw)]


def _fq_from_mont(b):
    """deserialize_field (zkey.rs:328-332): bytes are the Montgomery representation."""
    return _le(b) * fq_inv(MONT_R % Q_MOD) % Q_MOD


def _g1_from(b):
    """deserialize_g1 (zkey.rs:340-349): (0,0) is infinity."""
    x, y = _fq_from_mont(b[:32]), _fq_from_mont(b[32:64])
    return None if (x == 0 and y == 0) else (x, y)


def _g2_from(b):
    """deserialize_g2 (zkey.rs:351-360): x.c0|x.c1|y.c0|y.c1."""
    x = (_fq_from_mont(b[:32]), _fq_from_mont(b[32:64]))
    y = (_fq_from_mont(b[64:96]), _fq_from_mont(b[96:128]))
    return None if (x == (0, 0) and y == (0, 0)) else (x, y)


def read_zkey(data: bytes):
    """reference src/zkey.rs:53-60 (BinFile::new :73-101, proving_key :103-133, matrices :151-196,
    HeaderGroth::read :288-317).  Returns (pk dict, matrices dict)."""
    _ver, nsec = struct.unpack_from("<II", data, 4)
    off = 12
    secs = {}
    for _ in range(nsec):
        sid, size = struct.unpack_from("<IQ", data, off)
        off += 12
        secs.setdefault(sid, (off, size))          # get_section takes the first (zkey.rs:135-137)
        off += size
    o, _ = secs[2]
    (n8q,) = struct.unpack_from("<I", data, o)
    o += 4
    q = _le(data[o:o + n8q])
    o += n8q
    (n8r,) = struct.unpack_from("<I", data, o)
    o += 4
    r = _le(data[o:o + n8r])
    o += n8r
    n_vars, n_public, domain_size = struct.unpack_from("<III", data, o)
    o += 12
    alpha_g1 = _g1_from(data[o:o + 64]); o += 64
    beta_g1 = _g1_from(data[o:o + 64]); o += 64
    beta_g2 = _g2_from(data[o:o + 128]); o += 128
    gamma_g2 = _g2_from(data[o:o + 128]); o += 128
    delta_g1 = _g1_from(data[o:o + 64]); o += 64
    delta_g2 = _g2_from(data[o:o + 128]); o += 128

    def g1_section(num, sid):
        p, _ = secs[sid]
        return [_g1_from(data[p + 64 * i:p + 64 * i + 64]) for i in range(num)]

    def g2_section(num, sid):
        p, _ = secs[sid]
        return [_g2_from(data[p + 128 * i:p + 128 * i + 128]) for i in range(num)]

    pk = dict(
        q=q, r=r, n_vars=n_vars, n_public=n_public, domain_size=domain_size,
        alpha_g1=alpha_g1, beta_g1=beta_g1, beta_g2=beta_g2, gamma_g2=gamma_g2,
        delta_g1=delta_g1, delta_g2=delta_g2,
        ic=g1_section(n_public + 1, 3),
        a_query=g1_section(n_vars, 5), b_g1_query=g1_section(n_vars, 6),
        b_g2_query=g2_section(n_vars, 7), l_query=g1_section(n_vars - n_public - 1, 8),
        h_query=g1_section(domain_size, 9),
    )
    # matrices(): Coefs section 4 (zkey.rs:151-196)
    p, _ = secs[4]
    (ncoef,) = struct.unpack_from("<I", data, p)
    p += 4
    mats = [[[] for _ in range(domain_size)] for _ in range(2)]
    max_c = 0
    rinv2 = pow(fr_inv(MONT_R % R_MOD), 2, R_MOD)
    for _ in range(ncoef):
        m, c, s = struct.unpack_from("<III", data, p)
        # deserialize_field_fr (zkey.rs:322-325): value on disk is v*R^2
        v = _le(data[p + 12:p + 44]) * rinv2 % R_MOD
        p += 44
        max_c = max(max_c, c)
        mats[m][c].append((v, s))
    num_constraints = max_c - n_public
    a = mats[0][:num_constraints]
    b = mats[1][:num_constraints]
    matrices = dict(num_instance_variables=n_public + 1, num_witness_variables=n_vars - n_public,
                    num_constraints=num_constraints, a=a, b=b,
                    a_num_non_zero=sum(map(len, a)), b_num_non_zero=sum(map(len, b)))
    return pk, matrices


def get_public_inputs(witness, num_inputs, wire_mapping=None):
    """reference src/circom/circuit.rs:18-26."""
    if wire_mapping is None:
        return list(witness[1:num_inputs])
    return [witness[i] for i in wire_mapping[1:num_inputs]]


# ----------------------------------------------------------------------------------------------
# Groth16 prover (ark-groth16 0.5 create_proof_with_assignment; equations SURVEY.md section 3.1)
# ----------------------------------------------------------------------------------------------
def create_proof_with_assignment(pk, r, s, h, input_assignment, aux_assignment):
    h_acc = G1.msm(pk["h_query"], h)
    l_acc = G1.msm(pk["l_query"], aux_assignment)
    assignment = list(input_assignment) + list(aux_assignment)

    def calc(initial, query, vk_param, asg, C):
        acc = C.msm(query[1:], asg)
        res = C.add(initial, query[0])
        res = C.add(res, acc)
        return C.add(res, vk_param)

    r_delta = G1.mul(pk["delta_g1"], r)
    g_a = calc(r_delta, pk["a_query"], pk["alpha_g1"], assignment, G1)
    s_g_a = G1.mul(g_a, s)
    s_delta = G1.mul(pk["delta_g1"], s)
    if r != 0:
        g1_b = calc(s_delta, pk["b_g1_query"], pk["beta_g1"], assignment, G1)
    else:
        g1_b = None
    s_g2 = G2.mul(pk["delta_g2"], s)
    g2_b = calc(s_g2, pk["b_g2_query"], pk["beta_g2"], assignment, G2)
    r_g1_b = G1.mul(g1_b, r)
    rs_delta = G1.mul(pk["delta_g1"], r * s % R_MOD)
    g_c = G1.add(s_g_a, r_g1_b)
    g_c = G1.sub(g_c, rs_delta)
    g_c = G1.add(g_c, l_acc)
    g_c = G1.add(g_c, h_acc)
    return dict(a=g_a, b=g2_b, c=g_c)


def create_proof_with_reduction_and_matrices(pk, r, s, matrices, num_inputs, num_constraints,
                                             full_assignment, reduction="circom"):
    """Groth16::<Bn254,QAP>::create_proof_with_reduction_and_matrices, argument order as at reference
    benches/groth16.rs:52-60 / src/zkey.rs:903-911.  QAP = CircomReduction (default) or
    LibsnarkReduction (reduction="libsnark": the `Groth16<Bn254>` of reference tests/groth16.rs:9)."""
    if reduction == "libsnark":
        h = witness_map_libsnark(matrices["a"], matrices["b"], num_inputs, num_constraints,
                                 full_assignment, matrices.get("c"))
    else:
        h = witness_map_from_matrices(matrices["a"], matrices["b"], num_inputs, num_constraints,
                                      full_assignment)
    return create_proof_with_assignment(pk, r, s, h, full_assignment[1:num_inputs],
                                        full_assignment[num_inputs:])


# ----------------------------------------------------------------------------------------------
# pairing verifier = stand-in for Groth16::process_vk + verify_with_processed_vk
# (SURVEY.md Appendix C.3).  Fq12 = Fq[w]/(w^12 - 18 w^6 + 82), w^6 = 9 + i.
# ----------------------------------------------------------------------------------------------
def _f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                if bj:
                    t[i + j] += ai * bj
    for k in range(22, 11, -1):
        v = t[k]
        if v:
            t[k - 6] += 18 * v
            t[k - 12] -= 82 * v
    return [x % Q_MOD for x in t[:12]]


_F12_ONE = [1] + [0] * 11


def _f12_pow(a, e):
    r = _F12_ONE
    while e:
        if e & 1:
            r = _f12_mul(r, a)
        a = _f12_mul(a, a)
        e >>= 1
    return r


def _embed(z, shift, out, sign=1):
    """add sign * (z in Fq2) * w^shift into the coefficient list out.  a+bi -> (a-9b) + b w^6."""
    a, b = z
    out[shift] = (out[shift] + sign * (a - 9 * b)) % Q_MOD
    out[shift + 6] = (out[shift + 6] + sign * b) % Q_MOD


def _line(T, Q, P):
    """Line through twisted points T,Q (affine Fq2) evaluated at P in G1; returns (f12, T+Q)."""
    xP, yP = P
    xT, yT = T
    xQ, yQ = Q
    l = [0] * 12
    if xT == xQ and yT != yQ:                    # vertical
        l[0] = xP % Q_MOD
        _embed(xT, 2, l, -1)
        return l, None
    if T == Q:
        m = f2_mul(f2_mul((3, 0), f2_sqr(xT)), f2_inv(f2_add(yT, yT)))
    else:
        m = f2_mul(f2_sub(yQ, yT), f2_inv(f2_sub(xQ, xT)))
    x3 = f2_sub(f2_sub(f2_sqr(m), xT), xQ)
    y3 = f2_sub(f2_mul(m, f2_sub(xT, x3)), yT)
    l[0] = (-yP) % Q_MOD
    _embed(f2_mul(m, (xP, 0)), 1, l)
    _embed(f2_sub(yT, f2_mul(m, xT)), 3, l)
    return l, (x3, y3)


_FROB_X = f2_pow(XI, (Q_MOD - 1) // 3)
_FROB_Y = f2_pow(XI, (Q_MOD - 1) // 2)


def _frob_g2(Qp):
    return (f2_mul(f2_conj(Qp[0]), _FROB_X), f2_mul(f2_conj(Qp[1]), _FROB_Y))


def miller_loop(Qp, P):
    """ML(Q in G2, P in G1) for the optimal ate pairing; infinity on either side gives 1."""
    if Qp is None or P is None:
        return list(_F12_ONE)
    f = list(_F12_ONE)
    T = Qp
    for i in range(ATE_LOOP.bit_length() - 2, -1, -1):
        l, T2 = _line(T, T, P)
        f = _f12_mul(_f12_mul(f, f), l)
        T = T2
        if (ATE_LOOP >> i) & 1:
            l, T2 = _line(T, Qp, P)
            f = _f12_mul(f, l)
            T = T2
    Q1 = _frob_g2(Qp)
    Q2 = _frob_g2(Q1)
    l, T2 = _line(T, Q1, P)
    f = _f12_mul(f, l)
    T = T2
    l, _ = _line(T, G2.neg(Q2), P)
    f = _f12_mul(f, l)
    return f


_FINAL_EXP = (Q_MOD ** 12 - 1) // R_MOD


def final_exponentiation(f):
    return _f12_pow(f, _FINAL_EXP)


def pairing(Qp, P):
    return final_exponentiation(miller_loop(Qp, P))


def verify_proof(vk, public_inputs, proof):
    """Groth16 check e(A,B) = e(alpha,beta) e(IC0 + sum pub_i IC_{i+1}, gamma) e(C,delta), as one
    product (stand-in for process_vk + verify_with_processed_vk, zkey.rs:868-870,914-916)."""
    ic = vk["ic"]
    if len(public_inputs) + 1 != len(ic):
        raise ValueError("MalformedVerifyingKey")
    acc = ic[0]
    for x, P in zip(public_inputs, ic[1:]):
        acc = G1.add(acc, G1.mul(P, x % R_MOD))
    if not (G1.on_curve(proof["a"]) and G1.on_curve(proof["c"]) and G2.on_curve(proof["b"])):
        return False
    f = miller_loop(proof["b"], proof["a"])
    f = _f12_mul(f, miller_loop(vk["beta_g2"], G1.neg(vk["alpha_g1"])))
    f = _f12_mul(f, miller_loop(vk["gamma_g2"], G1.neg(acc)))
    f = _f12_mul(f, miller_loop(vk["delta_g2"], G1.neg(proof["c"])))
    return final_exponentiation(f) == _F12_ONE


# ----------------------------------------------------------------------------------------------
# trapdoor (known-tau) circom/snarkjs-style setup -- used to mint synthetic keys for tests
# (SURVEY.md Appendix C.2).  Not part of the reference's proving path.
# ----------------------------------------------------------------------------------------------
def lagrange_at_tau(n, tau, ntt_fn=None):
    """L_j(tau) for the size-n domain = inverse DFT of the powers of tau."""
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * tau % R_MOD
    return (ntt_fn or ntt)(pw, inverse=True)


def trapdoor_scalars(constraints, n_vars, n_public, tau, alpha, beta, gamma, delta, reduction="circom",
                     ntt_fn=None):
    """The scalar side of the trapdoor setup: every query point of the key is k * G for the k returned
    here (u -> a_query, v -> b_g1_query / b_g2_query, k_l -> l_query, k_h -> h_query, k_ic -> vk.ic).
    O(n log n) field operations and no group operation, so tests can pin a key generator at sizes
    where forming the points in Python would take minutes (ntt_fn: see h_query_scalars)."""
    m = len(constraints)
    num_inputs = n_public + 1
    n = domain_size_for(m + num_inputs)
    L = lagrange_at_tau(n, tau, ntt_fn)
    u = [0] * n_vars
    v = [0] * n_vars
    w = [0] * n_vars
    for j, (A, B, C) in enumerate(constraints):
        for wire, coeff in A:
            u[wire] = (u[wire] + coeff * L[j]) % R_MOD
        for wire, coeff in B:
            v[wire] = (v[wire] + coeff * L[j]) % R_MOD
        for wire, coeff in C:
            w[wire] = (w[wire] + coeff * L[j]) % R_MOD
    for i in range(num_inputs):
        u[i] = (u[i] + L[m + i]) % R_MOD
    gi, di = fr_inv(gamma), fr_inv(delta)
    k_ic = [(beta * u[i] + alpha * v[i] + w[i]) * gi % R_MOD for i in range(num_inputs)]
    k_l = [(beta * u[i] + alpha * v[i] + w[i]) * di % R_MOD for i in range(num_inputs, n_vars)]
    if reduction == "libsnark":
        zt = (pow(tau, n, R_MOD) - 1) % R_MOD
        k_h = h_query_scalars_libsnark(n - 1, tau, zt, di) + [0]
    else:
        k_h = h_query_scalars(n - 1, tau, di, ntt_fn)
    return dict(u=u, v=v, w=w, k_l=k_l, k_h=k_h, k_ic=k_ic, tau=tau, alpha=alpha, beta=beta, gamma=gamma,
                delta=delta, domain_size=n)


def trapdoor_setup(constraints, n_vars, n_public, tau, alpha, beta, gamma, delta, reduction="circom"):
    """constraints: list of (A,B,C) rows, each a list of (wire, coeff) as in the .r1cs file.
    Returns a pk dict shaped like read_zkey's (plus the scalar-side trapdoor data under 'td').
    reduction: "circom" (CircomReduction::h_query_scalars, qap.rs:90-105) or "libsnark" (arkworks'
    default QAP, reference tests/groth16.rs:25); the libsnark H query has n - 1 entries and is
    padded with the point at infinity to n."""
    td = trapdoor_scalars(constraints, n_vars, n_public, tau, alpha, beta, gamma, delta, reduction)
    n = td.pop("domain_size")
    g1m = lambda k: G1.mul(G1_GEN, k)
    g2m = lambda k: G2.mul(G2_GEN, k)
    pk = dict(
        q=Q_MOD, r=R_MOD, n_vars=n_vars, n_public=n_public, domain_size=n,
        alpha_g1=g1m(alpha), beta_g1=g1m(beta), beta_g2=g2m(beta), gamma_g2=g2m(gamma),
        delta_g1=g1m(delta), delta_g2=g2m(delta),
        ic=[g1m(k) for k in td["k_ic"]],
        a_query=[g1m(k) for k in td["u"]], b_g1_query=[g1m(k) for k in td["v"]],
        b_g2_query=[g2m(k) for k in td["v"]], l_query=[g1m(k) for k in td["k_l"]],
        h_query=[g1m(k) for k in td["k_h"]],
        td=td,
    )
    return pk


def matrices_from_r1cs(constraints):
    """(coeff, index) rows of A and B in the arkworks ConstraintMatrices orientation
    (zkey.rs:168; r1cs side is (index, coeff), src/circom/mod.rs:14)."""
    a = [[(c, wdx) for wdx, c in A] for A, _B, _C in constraints]
    b = [[(c, wdx) for wdx, c in B] for _A, B, _C in constraints]
    return a, b


# ----------------------------------------------------------------------------------------------
# serialisation helpers shared by tests (packed on-disk / C-ABI forms)
# ----------------------------------------------------------------------------------------------
def fr_to_mont_bytes(x):
    return ((x % R_MOD) * MONT_R % R_MOD).to_bytes(32, "little")


_FR_MONT_R_INV = pow(MONT_R % R_MOD, R_MOD - 2, R_MOD)


def fr_from_mont_bytes(b):
    return _le(b) * _FR_MONT_R_INV % R_MOD


def fq_to_mont_bytes(x):
    return ((x % Q_MOD) * MONT_R % Q_MOD).to_bytes(32, "little")


def g1_to_bytes(P):
    return bytes(64) if P is None else fq_to_mont_bytes(P[0]) + fq_to_mont_bytes(P[1])


def g2_to_bytes(P):
    if P is None:
        return bytes(128)
    (x0, x1), (y0, y1) = P
    return b"".join(fq_to_mont_bytes(v) for v in (x0, x1, y0, y1))


g1_from_bytes = _g1_from
g2_from_bytes = _g2_from


def proof_to_bytes(proof):
    """A(64)|B(128)|C(64), affine, Montgomery LE, (0,0)=infinity -- the g16_prove output layout."""
    return g1_to_bytes(proof["a"]) + g2_to_bytes(proof["b"]) + g1_to_bytes(proof["c"])


def write_zkey(pk, coefs, path=None):
    """snarkjs-format .zkey writer.  coefs: list of
    (matrix, constraint, signal, value) with canonical values, INCLUDING the n_public+1 extra rows
    snarkjs appends (row m+i has A coefficient 1 on signal i)."""
    def sec(sid, payload):
        return struct.pack("<IQ", sid, len(payload)) + payload
    hdr = struct.pack("<I", 32) + Q_MOD.to_bytes(32, "little") + struct.pack("<I", 32) + \
        R_MOD.to_bytes(32, "little") + struct.pack("<III", pk["n_vars"], pk["n_public"],
                                                    pk["domain_size"])
    hdr += g1_to_bytes(pk["alpha_g1"]) + g1_to_bytes(pk["beta_g1"]) + g2_to_bytes(pk["beta_g2"])
    hdr += g2_to_bytes(pk["gamma_g2"]) + g1_to_bytes(pk["delta_g1"]) + g2_to_bytes(pk["delta_g2"])
    r2 = MONT_R * MONT_R % R_MOD
    cf = struct.pack("<I", len(coefs)) + b"".join(
        struct.pack("<III", m, c, s) + (v * r2 % R_MOD).to_bytes(32, "little")
        for m, c, s, v in coefs)
    body = [sec(1, struct.pack("<I", 1)), sec(2, hdr),
            sec(3, b"".join(map(g1_to_bytes, pk["ic"]))), sec(4, cf),
            sec(5, b"".join(map(g1_to_bytes, pk["a_query"]))),
            sec(6, b"".join(map(g1_to_bytes, pk["b_g1_query"]))),
            sec(7, b"".join(map(g2_to_bytes, pk["b_g2_query"]))),
            sec(8, b"".join(map(g1_to_bytes, pk["l_query"]))),
            sec(9, b"".join(map(g1_to_bytes, pk["h_query"]))),
            sec(10, struct.pack("<I", 0) + bytes(64))]
    out = b"zkey" + struct.pack("<II", 1, len(body)) + b"".join(body)
    if path:
        with open(path, "wb") as f:
            f.write(out)
    return out


def coefs_from_r1cs(constraints, n_public):
    """Coefs(4) content snarkjs writes for an r1cs: A and B entries plus the public-input rows."""
    out = []
    for j, (A, B, _C) in enumerate(constraints):
        out += [(0, j, wdx, c) for wdx, c in A]
        out += [(1, j, wdx, c) for wdx, c in B]
    m = len(constraints)
    out += [(0, m + i, i, 1) for i in range(n_public + 1)]
    return out
