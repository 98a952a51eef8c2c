"""shared helpers for the parity tests (oracle <-> packed C-ABI forms)"""
import ctypes as C
import random

import numpy as np

import bn254_ref as o


def rand_fr(rng, n):
    return [rng.randrange(o.R_MOD) for _ in range(n)]


def fr_mont_arr(vals):
    """canonical ints -> (n,4) uint64 Montgomery via the ORACLE (independent of the library)"""
    raw = b"".join(o.fr_to_mont_bytes(v) for v in vals)
    return np.frombuffer(raw, dtype=np.uint64).reshape(-1, 4).copy()


def fr_from_mont_arr(arr):
    b = np.ascontiguousarray(arr, dtype=np.uint64).tobytes()
    return [o.fr_from_mont_bytes(b[i:i + 32]) for i in range(0, len(b), 32)]


def g1_arr(points):
    return np.frombuffer(b"".join(o.g1_to_bytes(p) for p in points), dtype=np.uint8).reshape(-1, 64).copy()


def g2_arr(points):
    return np.frombuffer(b"".join(o.g2_to_bytes(p) for p in points), dtype=np.uint8).reshape(-1, 128).copy()


def rand_g1(rng, n):
    """n pseudo-random G1 points (cheap chain: P_{i+1} = P_i + k*G style would correlate; use muls)"""
    return [o.G1.mul(o.G1_GEN, rng.randrange(1, o.R_MOD)) for _ in range(n)]


def rand_g2(rng, n):
    return [o.G2.mul(o.G2_GEN, rng.randrange(1, o.R_MOD)) for _ in range(n)]


def pk_from_oracle(pk):
    """oracle pk dict (affine int tuples) -> circom_compat_amd.ProvingKey (packed bytes)"""
    import circom_compat_amd as cc
    vk = cc.VerifyingKey(o.g1_to_bytes(pk["alpha_g1"]), o.g2_to_bytes(pk["beta_g2"]),
                         o.g2_to_bytes(pk["gamma_g2"]), o.g2_to_bytes(pk["delta_g2"]),
                         g1_arr(pk["ic"]))
    return cc.ProvingKey(pk["n_vars"], pk["n_public"], pk["domain_size"], vk,
                         o.g1_to_bytes(pk["beta_g1"]), o.g1_to_bytes(pk["delta_g1"]),
                         g1_arr(pk["a_query"]), g1_arr(pk["b_g1_query"]), g2_arr(pk["b_g2_query"]),
                         g1_arr(pk["l_query"]) if pk["l_query"] else np.zeros((0, 64), np.uint8),
                         g1_arr(pk["h_query"]))


def matrices_from_rows(a_rows, b_rows, num_inputs, n_vars, lib):
    import circom_compat_amd as cc
    return cc.ConstraintMatrices(num_inputs, n_vars - num_inputs + 1, len(a_rows),
                                 cc.Csr.from_rows(a_rows, lib), cc.Csr.from_rows(b_rows, lib))


def proof_from_bytes(raw):
    return dict(a=o.g1_from_bytes(raw[:64]), b=o.g2_from_bytes(raw[64:192]), c=o.g1_from_bytes(raw[192:256]))


def squaring_chain(k_or_m, x0=3, m=None):
    """SURVEY section 8(d) synthetic circuit: wires [1, out, x0, x1, ...], row i: (-x_i)*(x_i) = (-x_{i+1});
    p = 1, m = 2^k - 2 so that m + num_inputs = 2^k.  Returns (constraints, witness, n_vars, n_public)."""
    if m is None:
        m = (1 << k_or_m) - 2
    P = o.R_MOD
    xs = [x0 % P]
    for _ in range(m):
        xs.append(xs[-1] * xs[-1] % P)
    # wires: 0 = one, 1 = out (= x_m), 2.. = x_0 .. x_{m-1}; x_m is the output wire
    n_vars = m + 2
    wire = lambda j: 1 if j == m else 2 + j
    cons = []
    for i in range(m):
        cons.append(([(wire(i), P - 1)], [(wire(i), 1)], [(wire(i + 1), P - 1)]))
    w = [0] * n_vars
    w[0] = 1
    for j, x in enumerate(xs):
        w[wire(j)] = x
    return cons, w, n_vars, 1


def dense_skewed_circuit(m, seed=0, n_inputs=8, long_rows=()):
    """SURVEY 8(d) config-5 substitute: seeded R1CS with 3-term A rows / 2-term B rows and small +-
    coefficients (the shape of circuit2.r1cs: 2.95 / 1.96 nnz per row), a few very long rows, and
    a skewed witness (>= 50 % of the scalars in {0, 1}, as bit-decomposition circuits have),
    satisfiable by construction: row i defines a fresh wire out_i = (A_i.w)(B_i.w).
    Returns (constraints [(A, B, C)] with [(wire, coeff)] lists, witness ints, n_vars, n_public)."""
    import random
    rng = random.Random(seed)
    R = o.R_MOD
    w = [1, 0]                                    # wire 0 = 1, wire 1 = public output (set at the end)
    w += [rng.randrange(2) for _ in range(n_inputs)]
    bits = list(range(2, 2 + n_inputs))
    cons = []
    coeffs = [1, R - 1, 2, R - 2, 3]

    def lc(terms):
        return sum(c * w[i] for i, c in terms) % R

    def distinct(k, pool):
        return rng.sample(pool, k) if len(pool) >= k else [rng.choice(pool) for _ in range(k)]

    for i in range(m):
        hi = len(w)
        if i in long_rows:
            A = [(wi, rng.choice(coeffs)) for wi in distinct(min(65, hi - 2), range(2, hi))]
            B = [(wi, rng.choice(coeffs)) for wi in distinct(min(64, hi - 2), range(2, hi))]
        elif rng.random() < 0.55:
            xi, xj, xk = distinct(3, bits)
            A = [(xi, 1), (xj, 1), (xk, R - 1)]                # in {-1, 0, 1, 2}
            B = [(rng.choice(bits), 1)]
            if rng.random() < 0.5:                             # 2-term B: bit * (1 - bit2) style
                B = [(0, 1), (rng.choice(bits), R - 1)]
        else:
            A = [(wi, rng.choice(coeffs)) for wi in distinct(3, range(2, hi))]  # a range, not a list: O(m) overall, same draws
            B = [(wi, rng.choice(coeffs)) for wi in distinct(2, range(2, hi))]
        val = lc(A) * lc(B) % R
        w.append(val)
        if val in (0, 1):
            bits.append(len(w) - 1)
        cons.append((A, B, [(len(w) - 1, 1)]))
    w[1] = w[-1]                                   # public output: the last defined wire
    cons.append(([(len(w) - 1, 1)], [(0, 1)], [(1, 1)]))
    return cons, w, len(w), 1
