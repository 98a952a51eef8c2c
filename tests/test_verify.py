"""GPU batch verification (g16_verify_batch: Groth16::process_vk + verify_with_processed_vk, reference
src/zkey.rs:868-870,914-916) against the oracle's pairing check, which tests/test_oracle.py pins to
the reference's own predicate (proofs of the reference's test.zkey verify, wrong inputs do not)."""
import os
import random

import numpy as np
import pytest

import bn254_ref as o
import helpers as H


def _vk(cc, opk):
    return cc.VerifyingKey(o.g1_to_bytes(opk["alpha_g1"]), o.g2_to_bytes(opk["beta_g2"]),
                           o.g2_to_bytes(opk["gamma_g2"]), o.g2_to_bytes(opk["delta_g2"]), H.g1_arr(opk["ic"]))


def test_verify_batch_on_the_reference_zkey(lib, golden):
    """valid proofs (two different (r, s)), a wrong public input, a proof whose A is another curve
    point, a proof with an off-curve coordinate, the all-infinity proof: same verdicts as the oracle"""
    import circom_compat_amd as cc
    data = open(os.path.join(golden, "test.zkey"), "rb").read()
    opk, omats = o.read_zkey(data)
    w = [1, 33, 3, 11]
    vk = _vk(cc, opk)
    proofs, pubs, want = [], [], []
    for r, s in ((0, 0), (3413513218498352040262653353725127729454431939539290118844322056224532443637,
                          6077776500692565155461894309070795882353485867345896979329447163197530625403)):
        p = o.create_proof_with_reduction_and_matrices(opk, r, s, omats, 2, 1, w)
        raw = o.proof_to_bytes(p)
        proofs += [raw, raw]
        pubs += [[33], [34]]
        want += [True, False]
    good = o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, 5, 7, omats, 2, 1, w))
    swapped = o.g1_to_bytes(o.G1_GEN) + good[64:]                       # on the curve, wrong point
    off = bytearray(good)
    off[0] ^= 1                                                          # off the curve
    proofs += [swapped, bytes(off), bytes(256)]
    pubs += [[33], [33], [33]]
    for raw, pi in zip(proofs[4:], pubs[4:]):
        want.append(bool(o.verify_proof(opk, pi, H.proof_from_bytes(raw))))
    assert want == [True, False, True, False, False, False, False]
    got = cc.verify_batch(vk, proofs, pubs, lib=lib)
    assert got == want
    # the single-proof mirror of verify_with_processed_vk
    assert cc.Groth16.verify(vk, [33], cc.Proof(good), lib=lib) is True
    assert cc.Groth16.verify(vk, [32], cc.Proof(good), lib=lib) is False
    with pytest.raises(cc.G16Error):
        cc.verify_batch(vk, [good], [[33, 1]], lib=lib)                  # MalformedVerifyingKey


@pytest.mark.parametrize("n_pub", [0, 3])
def test_verify_batch_public_input_counts(lib, n_pub):
    """no public inputs (prepared inputs = IC_0) and several: proofs made by the product prover on a
    trapdoor key verify on the GPU, each wrong input is rejected"""
    import circom_compat_amd as cc
    P = o.R_MOD
    m = 3
    base = 1 + n_pub
    n_vars = base + m + 1
    cons = [([(base + i, 1)], [(base + i, 1)], [(base + i + 1, 1)]) for i in range(m)]
    w = [1] + [rng_v for rng_v in (7, 8, 9)][:n_pub] + [3]
    for _ in range(m):
        w.append(w[-1] * w[-1] % P)
    rng = random.Random(n_pub)
    tox = [rng.randrange(1, P) for _ in range(5)]
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *tox)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, n_pub + 1, n_vars, lib)
    pr = cc.Prover(H.pk_from_oracle(opk), mats, lib=lib)
    proof = pr.prove(rng.randrange(P), rng.randrange(P), w)
    vk = _vk(cc, opk)
    pub = w[1:1 + n_pub]
    batch, pubs = [proof], [pub]
    for j in range(n_pub):
        bad = list(pub)
        bad[j] = (bad[j] + 1) % P
        batch.append(proof)
        pubs.append(bad)
    assert cc.verify_batch(vk, batch, pubs, lib=lib) == [True] + [False] * n_pub
    assert o.verify_proof(opk, pub, H.proof_from_bytes(proof.raw))


def _twist_point_outside_g2(seed):
    """a point of E'(Fq2): y^2 = x^3 + 3/(9+i) that is NOT in the r-torsion (the twist's cofactor is
    2q - r: a random point is outside G2 with overwhelming probability; checked with [r]P != inf)"""
    q = o.Q_MOD
    rng = random.Random(seed)

    def fq_sqrt(a):
        s = pow(a, (q + 1) // 4, q)
        return s if s * s % q == a % q else None

    while True:
        x = (rng.randrange(q), rng.randrange(q))
        rhs = o.f2_add(o.f2_mul(o.f2_sqr(x), x), o.G2_B)
        a0, a1 = rhs
        n = fq_sqrt((a0 * a0 + a1 * a1) % q)                   # sqrt of the norm
        if n is None:
            continue
        for sgn in (n, q - n):
            half = (a0 + sgn) * pow(2, q - 2, q) % q
            y0 = fq_sqrt(half)
            if y0 is None or y0 == 0:
                continue
            y = (y0, a1 * pow(2 * y0, q - 2, q) % q)
            if o.f2_sqr(y) == (rhs[0] % q, rhs[1] % q):
                P = (x, y)
                if o.G2.mul(P, o.R_MOD) is not None:
                    return P


def test_verify_batch_rejects_what_deserialisation_rejects(lib, golden):
    """ark-groth16 only ever pairs a deserialised Proof, and that deserialisation (Validate::Yes)
    rejects (i) a B that is on the twist but outside the prime-order subgroup and (ii) coordinates
    that are not canonical (>= q: a second encoding of the same point, i.e. a malleable proof).
    g16_verify_batch takes raw bytes, so it must refuse both itself -- a valid proof beside them
    still verifies."""
    import circom_compat_amd as cc
    data = open(os.path.join(golden, "test.zkey"), "rb").read()
    opk, omats = o.read_zkey(data)
    vk = _vk(cc, opk)
    good = o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, 5, 7, omats, 2, 1, [1, 33, 3, 11]))
    T = _twist_point_outside_g2(1)
    assert o.G2.mul(T, o.R_MOD) is not None
    cof = good[:64] + o.g2_to_bytes(T) + good[192:]                      # on the curve, wrong subgroup
    B = H.proof_from_bytes(good)["b"]
    mixed = good[:64] + o.g2_to_bytes(o.G2.add(B, o.G2.mul(T, o.R_MOD))) + good[192:]   # B + a cofactor-torsion point
    assert o.G2.mul(o.G2.mul(T, o.R_MOD), 2 * o.Q_MOD - o.R_MOD) is None  # [r]T has order dividing the cofactor

    def plus_q(raw, off):                                                # stored value m -> m + q (< 2^256)
        v = int.from_bytes(raw[off:off + 32], "little") + o.Q_MOD
        assert v < 1 << 256
        return raw[:off] + v.to_bytes(32, "little") + raw[off + 32:]
    noncanon = [plus_q(good, off) for off in (0, 32, 64, 160, 192, 224)]  # A.x, A.y, B.x.c0, B.y.c1, C.x, C.y
    batch = [good, cof, mixed] + noncanon + [good]
    got = cc.verify_batch(vk, batch, [[33]] * len(batch), lib=lib)
    assert got == [True, False, False] + [False] * len(noncanon) + [True]
