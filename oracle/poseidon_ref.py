"""poseidon_ref.py -- TEST INFRASTRUCTURE ONLY (checker side; never imported by the product).

The Poseidon permutation over BN254 Fr with circomlib's parameter set, restated from the published
algorithm -- none of it lives in /root/reference (BASELINE.json configs[4] names a "Circom-generated
2^20 Poseidon-hash-chain .zkey"; circom, circomlib, snarkjs and a ptau file are not available offline):

* parameters: the Poseidon paper's reference generator (Grassi, Khovratovich, Rechberger, Roy,
  Schofnegger, "Poseidon: a new hash function for zero-knowledge proof systems", USENIX Security 2021,
  `generate_parameters_grain.sage`): an 80-bit Grain LFSR seeded with
  (field = 1 | sbox = 0 | n = 254 | t | R_F | R_P | 30 ones), 160 warm-up clocks, self-shrinking output
  (a pair of bits is used only when its first bit is 1), round constants by rejection sampling below
  the modulus, then a Cauchy MDS matrix M[i][j] = 1 / (x_i + y_j) from the next 2t field elements.
  circomlibjs' `poseidon_constants` were generated with exactly that call
  (`sage generate_parameters_grain.sage 1 0 254 t 8 R_P 0x30644e72...0001`, R_P = 56, 57, 56, 60 ... for
  t = 2, 3, 4, 5 ...);
* the permutation: state (0, in_1 .. in_{t-1}); per round add the round constants, x^5 on every lane
  (the first and last R_F / 2 rounds) or on lane 0 only (R_P partial rounds), multiply by M; output
  lane 0 -- circomlibjs `poseidon_reference.js` (the textbook form; circomlib's circuit uses an
  equivalent sparse factorisation of the partial rounds).

PINNED by the hash known-answer tests of circomlibjs (test/poseidon.js), checked in
tests/test_oracle.py::test_poseidon_reference_matches_circomlibjs_kats:

    poseidon([1, 2]) = 0x115cc0f5e7d690413df64c6b9662e9cf2a3617f2743245519e19607a4417189a
    poseidon([1])    = 18586133768512220936620570745912940619677854269274689475585506675881198879027

Both come out of the generator below with no constant typed in by hand.
"""
from functools import lru_cache

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
N_ROUNDS_F = 8
N_ROUNDS_P = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]   # t = 2 .. 17 (circomlibjs)

KATS = {
    (1, 2): 7853200120776062878684798364095072458815029376092732009249414926327459813530,
    (1,): 18586133768512220936620570745912940619677854269274689475585506675881198879027,
}


class _Grain:
    """80-bit LFSR b[i+80] = b[i+62] ^ b[i+51] ^ b[i+38] ^ b[i+23] ^ b[i+13] ^ b[i], self-shrinking."""

    def __init__(self, n, t, r_f, r_p):
        seed = "01" + "0000" + format(n, "012b") + format(t, "012b") + format(r_f, "010b") + format(r_p, "010b") + "1" * 30
        self.s = [int(c) for c in seed]
        assert len(self.s) == 80
        for _ in range(160):
            self._clock()

    def _clock(self):
        s = self.s
        b = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        del s[0]
        s.append(b)
        return b

    def bit(self):
        while self._clock() == 0:      # first bit of the pair is 0: discard the pair
            self._clock()
        return self._clock()

    def bits(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v


@lru_cache(maxsize=None)
def parameters(t):
    """(round constants as a flat list of (R_F + R_P) * t values, MDS matrix as t rows) for width t"""
    r_p = N_ROUNDS_P[t - 2]
    g = _Grain(254, t, N_ROUNDS_F, r_p)
    rc = []
    while len(rc) < (N_ROUNDS_F + r_p) * t:
        v = g.bits(254)
        if v < R_MOD:
            rc.append(v)
    while True:
        pts = [g.bits(254) % R_MOD for _ in range(2 * t)]
        if len(set(pts)) != 2 * t:
            continue
        xs, ys = pts[:t], pts[t:]
        if all((x + y) % R_MOD for x in xs for y in ys):
            break
    mds = [[pow((x + y) % R_MOD, R_MOD - 2, R_MOD) for y in ys] for x in xs]
    return rc, mds


def permute(state):
    t = len(state)
    rc, mds = parameters(t)
    r_p = N_ROUNDS_P[t - 2]
    st = [x % R_MOD for x in state]
    for r in range(N_ROUNDS_F + r_p):
        st = [(x + rc[r * t + i]) % R_MOD for i, x in enumerate(st)]
        if r < N_ROUNDS_F // 2 or r >= N_ROUNDS_F // 2 + r_p:
            st = [pow(x, 5, R_MOD) for x in st]
        else:
            st[0] = pow(st[0], 5, R_MOD)
        st = [sum(mds[i][j] * st[j] for j in range(t)) % R_MOD for i in range(t)]
    return st


def poseidon(inputs):
    """circomlibjs poseidon(inputs): capacity lane 0 starts at 0, output = lane 0"""
    return permute([0] + list(inputs))[0]


def hash_chain(h0, xs):
    """h_{i+1} = poseidon([h_i, x_i]): the chain bench.poseidon_chain_circuit constrains"""
    out = [h0]
    for x in xs:
        out.append(poseidon([out[-1], x]))
    return out
