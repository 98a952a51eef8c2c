"""Pins the oracle (oracle/bn254_ref.py, oracle/groth16_cpu.c) against every golden vector,
known-answer test and fixture the reference's own tests hold for the proving path
(tests/golden/reference_vectors.json is extracted from the reference test sources by
tests/golden/make_golden.py; the binary fixtures are the reference's test-vectors).  CPU only."""
import json
import os
import random

import numpy as np
import pytest

import bn254_ref as o
import helpers as H


@pytest.fixture(scope="module")
def vec(golden):
    return json.load(open(os.path.join(golden, "reference_vectors.json")))


def test_constants():
    assert o.R_MOD == int.from_bytes(o.R1CS_PRIME_LE, "little")          # r1cs_reader.rs:181
    assert pow(o.FR_TWO_ADIC_ROOT, 1 << 28, o.R_MOD) == 1 and pow(o.FR_TWO_ADIC_ROOT, 1 << 27, o.R_MOD) != 1
    assert o.root_of_unity(4) == 21888242871839275217838484774961031246007050428528088939761107053157389710902
    assert o.root_of_unity(8) == 19540430494807482326159819597004422086093766032135589407132600596362845576832
    assert o.G1.on_curve(o.G1_GEN) and o.G2.on_curve(o.G2_GEN)
    assert o.G1.mul(o.G1_GEN, o.R_MOD) is None and o.G2.mul(o.G2_GEN, o.R_MOD) is None


def test_snarkjs_generator_dumps(vec):
    """reference src/zkey.rs:465-517 (can_deser_fq / g1 / g2)"""
    assert o.fq_to_mont_bytes(1) == bytes(vec["fq_one_mont"])
    assert o.g1_from_bytes(bytes(vec["g1_generator"])) == (1, 2)
    assert o.g1_to_bytes((1, 2)) == bytes(vec["g1_generator"])
    assert o.g2_from_bytes(bytes(vec["g2_generator"])) == o.G2_GEN
    assert o.g2_to_bytes(o.G2_GEN) == bytes(vec["g2_generator"])


def test_zkey_header_and_every_point(vec, golden):
    """reference src/zkey.rs:519-543 (header) and :545-763 (deser_key)"""
    pk, mats = o.read_zkey(open(os.path.join(golden, "test.zkey"), "rb").read())
    hd = vec["test_zkey_header"]
    assert (pk["n_vars"], pk["n_public"], pk["domain_size"]) == (hd["n_vars"], hd["n_public"], hd["domain_size"])
    assert pk["q"] == o.Q_MOD and pk["r"] == o.R_MOD
    tz = vec["test_zkey"]
    for name, dec in (("ic", o.g1_from_bytes), ("a_query", o.g1_from_bytes), ("b_g1_query", o.g1_from_bytes),
                      ("b_g2_query", o.g2_from_bytes), ("l_query", o.g1_from_bytes), ("h_query", o.g1_from_bytes)):
        assert pk[name] == [dec(bytes(b)) for b in tz[name]], name
    assert (mats["num_instance_variables"], mats["num_witness_variables"], mats["num_constraints"]) == (2, 3, 1)
    assert mats["a"] == [[(o.R_MOD - 1, 2)]] and mats["b"] == [[(1, 3)]]


def test_zkey_vk_matches_snarkjs_json(golden):
    """reference src/zkey.rs:765-779 (deser_vk)"""
    pk, _ = o.read_zkey(open(os.path.join(golden, "test.zkey"), "rb").read())
    vk = json.load(open(os.path.join(golden, "verification_key.json")))
    g1 = lambda p: (int(p[0]), int(p[1]))
    g2 = lambda p: ((int(p[0][0]), int(p[0][1])), (int(p[1][0]), int(p[1][1])))
    assert pk["alpha_g1"] == g1(vk["vk_alpha_1"]) and pk["beta_g2"] == g2(vk["vk_beta_2"])
    assert pk["gamma_g2"] == g2(vk["vk_gamma_2"]) and pk["delta_g2"] == g2(vk["vk_delta_2"])
    assert pk["ic"] == [g1(p) for p in vk["IC"]]


def test_r1cs_sample(vec):
    """reference src/circom/r1cs_reader.rs:257-338"""
    f = o.read_r1cs(bytes.fromhex(vec["r1cs_sample_hex"]))
    e = vec["r1cs_sample_expect"]
    for k in ("version", "field_size", "n_wires", "n_pub_out", "n_pub_in", "n_prv_in", "n_labels", "n_constraints"):
        assert f[k] == e[k], k
    assert f["prime"] == o.R1CS_PRIME_LE
    c = f["constraints"]
    assert len(c) == 3 and len(c[0][0]) == e["c0_a_len"] and list(c[0][0][0]) == e["c0_a0"]
    assert list(c[2][1][0]) == e["c2_b0"] and len(c[1][2]) == e["c1_c_len"]
    assert len(f["wire_mapping"]) == e["wire_mapping_len"] and f["wire_mapping"][1] == e["wire_mapping_1"]


def test_r1cs_fixtures(golden):
    a = o.read_r1cs(open(os.path.join(golden, "mycircuit.r1cs"), "rb").read())
    assert (a["n_constraints"], a["n_wires"], a["num_inputs"]) == (1, 4, 2)
    b = o.read_r1cs(open(os.path.join(golden, "circuit2.r1cs"), "rb").read())
    assert (b["n_constraints"], b["n_wires"], b["num_inputs"]) == (131, 132, 2)
    nnz = [sum(len(c[k]) for c in b["constraints"]) for k in range(3)]
    assert nnz == [387, 257, 3]
    w = o.read_wtns(open(os.path.join(golden, "circuit2.wtns"), "rb").read())
    assert len(w) == 132 and w[0] == 1
    for A, B, C in b["constraints"]:                                  # the witness satisfies the r1cs
        ev = lambda lc: sum(cf * w[i] for i, cf in lc) % o.R_MOD
        assert ev(A) * ev(B) % o.R_MOD == ev(C)


def test_r1cs_error_paths(vec):
    """error behaviour of R1CSFile::new (r1cs_reader.rs:57-69,163-189,232-247)"""
    good = bytearray(bytes.fromhex(vec["r1cs_sample_hex"]))
    for mutate, msg in ((lambda d: d.__setitem__(0, 0x73), "Invalid magic number"),
                        (lambda d: d.__setitem__(4, 2), "Unsupported version"),
                        (lambda d: d.__setitem__(24, 31), "32-byte fields"),
                        (lambda d: d.__setitem__(28, 2), "bn256")):
        d = bytearray(good)
        mutate(d)
        with pytest.raises(ValueError, match=msg):
            o.read_r1cs(bytes(d))


def test_witness_map_kat_and_proof_predicate(golden):
    """reference src/zkey.rs:846-919: prove on test.zkey verifies; SURVEY Appendix C.1 KATs"""
    pk, mats = o.read_zkey(open(os.path.join(golden, "test.zkey"), "rb").read())
    w = [int(x) for x in json.load(open(os.path.join(golden, "mycircuit-witness.json")))]
    assert w == [1, 33, 3, 11]
    h = o.witness_map_from_matrices(mats["a"], mats["b"], 2, 1, w)
    assert h[0] == 190042957931705914545745448213290365653268903107554486158945885339918534040
    assert h[3] == 9524701523582778197584879850388312504848302205747610345200903552183809682204
    pr = o.create_proof_with_reduction_and_matrices(pk, 0, 0, mats, 2, 1, w)
    assert pr["a"] == (21820242516822140966541162377276968686232843738113587401096982992192344668894,
                       11813319305207505272935972628809616933491158390363636948064175237855995334902)
    assert pr["c"] == (16517915790659730733697074034691836627766957980661041574367499432757153082558,
                       14547307060850695604573432667032315680903080725809615957085938618335278860351)
    assert o.verify_proof(pk, [33], pr) and not o.verify_proof(pk, [34], pr)
    rng = random.Random(0)
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    pr = o.create_proof_with_reduction_and_matrices(pk, r, s, mats, 2, 1, w)
    assert o.verify_proof(pk, o.get_public_inputs(w, 2), pr)


def test_pairing_bilinearity():
    e1 = o.pairing(o.G2.mul(o.G2_GEN, 3), o.G1.mul(o.G1_GEN, 6))
    e2 = o.pairing(o.G2.mul(o.G2_GEN, 9), o.G1.mul(o.G1_GEN, 2))
    assert e1 == e2 and e1 != [1] + [0] * 11


def test_trapdoor_setup_circuit2_verifies(golden):
    """known-tau setup on circuit2 -> proofs verify, wrong input rejected, and the
    two scalar-side identities hold (MSM(H,h) and the QAP identity)"""
    c2 = o.read_r1cs(open(os.path.join(golden, "circuit2.r1cs"), "rb").read())
    w = o.read_wtns(open(os.path.join(golden, "circuit2.wtns"), "rb").read())
    rng = random.Random(5)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    tau, alpha, beta, gamma, delta = tox
    # scalar side only (fast): k_h and the QAP identity
    n = o.domain_size_for(c2["n_constraints"] + 2)
    a_rows, b_rows = o.matrices_from_r1cs(c2["constraints"])
    h = o.witness_map_from_matrices(a_rows, b_rows, 2, c2["n_constraints"], w)
    k_h = o.h_query_scalars(n - 1, tau, o.fr_inv(delta))
    L = o.lagrange_at_tau(n, tau)
    U = V = W = 0
    for j, (A, B, Cc) in enumerate(c2["constraints"]):
        U += sum(cf * w[i] for i, cf in A) * L[j]
        V += sum(cf * w[i] for i, cf in B) * L[j]
        W += sum(cf * w[i] for i, cf in Cc) * L[j]
    for i in range(2):
        U += w[i] * L[c2["n_constraints"] + i]
    lhs = sum(x * y for x, y in zip(h, k_h)) % o.R_MOD
    assert lhs == (U * V - W) * o.fr_inv(delta) % o.R_MOD


def test_c_oracle_matches_python_oracle(golden):
    """oracle/groth16_cpu.c (the timed CPU baseline) == oracle/bn254_ref.py on FFT, MSM and prove"""
    import cpu_ref
    rng = random.Random(1)
    x = H.rand_fr(rng, 128)
    a = H.fr_mont_arr(x)
    assert H.fr_from_mont_arr(cpu_ref.fft(a, 7)) == o.ntt(x)
    assert H.fr_from_mont_arr(cpu_ref.fft(a, 7, True)) == o.ntt(x, inverse=True)
    pts = H.rand_g1(rng, 8)
    for n in (1, 5, 31, 33, 200):
        bases = [pts[i % 8] if i % 11 else None for i in range(n)]
        sc = H.rand_fr(rng, n)
        sc[0] = o.R_MOD - 1
        assert cpu_ref.msm_g1(H.g1_arr(bases), H.fr_mont_arr(sc)) == o.g1_to_bytes(o.G1.msm(bases, sc)), n
    p2 = H.rand_g2(rng, 4)
    b2 = [p2[i % 4] for i in range(40)]
    sc = H.rand_fr(rng, 40)
    assert cpu_ref.msm_g2(H.g2_arr(b2), H.fr_mont_arr(sc)) == o.g2_to_bytes(o.G2.msm(b2, sc))
    # the chunk-parallel shape of the "all cores" CPU column (bench.py cpu_baseline.value_all_cores): same elements
    for chunks in (2, 7, 64):
        cpu_ref.set_msm_chunks(chunks)
        try:
            assert cpu_ref.msm_g1(H.g1_arr(bases), H.fr_mont_arr(H.rand_fr(random.Random(5), 200))) == \
                o.g1_to_bytes(o.G1.msm(bases, H.rand_fr(random.Random(5), 200))), chunks
            assert cpu_ref.msm_g2(H.g2_arr(b2), H.fr_mont_arr(sc)) == o.g2_to_bytes(o.G2.msm(b2, sc)), chunks
        finally:
            cpu_ref.set_msm_chunks(1)
    cons, wit, nv, npub = H.squaring_chain(5)
    opk = o.trapdoor_setup(cons, nv, npub, 11, 22, 33, 44, 55)
    ar, br = o.matrices_from_r1cs(cons)

    class M:  # minimal ConstraintMatrices look-alike built without the product library
        pass
    def csr(rows):
        m = M()
        m.row_ptr = np.array([0] + list(np.cumsum([len(r) for r in rows])), dtype=np.uint32)
        m.col = np.array([i for r in rows for _c, i in r], dtype=np.uint32)
        m.coeff = H.fr_mont_arr([c for r in rows for c, _i in r])
        return m
    mats = M()
    mats.a, mats.b, mats.num_constraints = csr(ar), csr(br), len(cons)
    pk = M()
    pk.n_vars, pk.n_public, pk.domain_size = nv, npub, opk["domain_size"]
    pk.a_query, pk.b_g1_query, pk.b_g2_query = H.g1_arr(opk["a_query"]), H.g1_arr(opk["b_g1_query"]), H.g2_arr(opk["b_g2_query"])
    pk.l_query, pk.h_query = H.g1_arr(opk["l_query"]), H.g1_arr(opk["h_query"])
    pk.vk = M()
    pk.vk.alpha_g1, pk.vk.beta_g2, pk.vk.delta_g2 = o.g1_to_bytes(opk["alpha_g1"]), o.g2_to_bytes(opk["beta_g2"]), o.g2_to_bytes(opk["delta_g2"])
    pk.beta_g1, pk.delta_g1 = o.g1_to_bytes(opk["beta_g1"]), o.g1_to_bytes(opk["delta_g1"])
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    got, h = cpu_ref.prove(pk, mats, H.fr_mont_arr([r]), H.fr_mont_arr([s]), H.fr_mont_arr(wit), want_h=True)
    want = o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=ar, b=br), 2, len(cons), wit)
    assert got == o.proof_to_bytes(want)
    assert H.fr_from_mont_arr(h) == o.witness_map_from_matrices(ar, br, 2, len(cons), wit)
    assert o.verify_proof(opk, wit[1:2], want)


@pytest.mark.parametrize("logm,kind", [(5, "chain"), (8, "chain"), (10, "chain"), (7, "dense")])
def test_c_oracle_libsnark_reduction_matches_python_oracle(logm, kind):
    """oracle/groth16_cpu.c's LibsnarkReduction witness map (coset g = 5, division by Z_H, inverse coset
    transform) and the prove that uses it == oracle/bn254_ref.py's restatement at 2^5 .. 2^10 (chain
    and uneven rows): the large-size Libsnark checker of tests/test_gpu_large.py stands on this pin;
    the Python restatement is itself anchored by the pairing predicate (tests above)."""
    import cpu_ref
    if kind == "dense":
        cons, wit, nv, _ = H.dense_skewed_circuit((1 << logm) - 5, seed=5, long_rows=(9,))
        npub = 1
    else:
        cons, wit, nv, npub = H.squaring_chain(logm)
    rng = random.Random(logm)
    opk = o.trapdoor_setup(cons, nv, npub, *[rng.randrange(1, o.R_MOD) for _ in range(5)], reduction="libsnark")
    ar, br = o.matrices_from_r1cs(cons)

    class M:
        pass

    def csr(rows):
        m = M()
        m.row_ptr = np.array([0] + list(np.cumsum([len(r) for r in rows])), dtype=np.uint32)
        m.col = np.array([i for r in rows for _c, i in r], dtype=np.uint32)
        m.coeff = H.fr_mont_arr([c for r in rows for c, _i in r])
        return m
    mats = M()
    mats.a, mats.b, mats.num_constraints = csr(ar), csr(br), len(cons)
    ni = npub + 1
    h_c = cpu_ref.witness_map(mats.a, mats.b, ni, len(cons), H.fr_mont_arr(wit), reduction="libsnark")
    h_py = o.witness_map_libsnark(ar, br, ni, len(cons), wit)
    assert H.fr_from_mont_arr(h_c) == h_py and h_py[-1] == 0
    pk = M()
    pk.n_vars, pk.n_public, pk.domain_size = nv, npub, opk["domain_size"]
    pk.a_query, pk.b_g1_query, pk.b_g2_query = H.g1_arr(opk["a_query"]), H.g1_arr(opk["b_g1_query"]), H.g2_arr(opk["b_g2_query"])
    pk.l_query, pk.h_query = H.g1_arr(opk["l_query"]), H.g1_arr(opk["h_query"])
    assert len(opk["h_query"]) == opk["domain_size"] and opk["h_query"][-1] is None   # padded with infinity
    pk.vk = M()
    pk.vk.alpha_g1, pk.vk.beta_g2, pk.vk.delta_g2 = o.g1_to_bytes(opk["alpha_g1"]), o.g2_to_bytes(opk["beta_g2"]), o.g2_to_bytes(opk["delta_g2"])
    pk.beta_g1, pk.delta_g1 = o.g1_to_bytes(opk["beta_g1"]), o.g1_to_bytes(opk["delta_g1"])
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    got = cpu_ref.prove(pk, mats, H.fr_mont_arr([r]), H.fr_mont_arr([s]), H.fr_mont_arr(wit), reduction="libsnark")
    want = o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=ar, b=br), ni, len(cons), wit,
                                                      reduction="libsnark")
    assert got == o.proof_to_bytes(want)
    if logm <= 8:
        assert o.verify_proof(opk, wit[1:ni], want)


def test_c_oracle_scalar_mul_matches_python_oracle():
    """g16cpu_g1_mul_batch / g16cpu_g2_mul_batch (k * P by plain double-and-add) == bn254_ref's G1.mul /
    G2.mul: the helper the GPU suite uses to turn the oracle's trapdoor SCALARS (trapdoor_scalars, no
    group operations) into the points a key generator's output is compared with at 2^12 .. 2^20."""
    import cpu_ref
    rng = random.Random(12)
    ks = [0, 1, 2, o.R_MOD - 1] + [rng.randrange(o.R_MOD) for _ in range(28)]
    g1 = cpu_ref.g1_mul_batch(o.g1_to_bytes(o.G1_GEN), ks)
    g2 = cpu_ref.g2_mul_batch(o.g2_to_bytes(o.G2_GEN), ks)
    for i, k in enumerate(ks):
        assert bytes(g1[i]) == o.g1_to_bytes(o.G1.mul(o.G1_GEN, k)), k
        assert bytes(g2[i]) == o.g2_to_bytes(o.G2.mul(o.G2_GEN, k)), k
    p = o.G1.mul(o.G1_GEN, 77)
    got = cpu_ref.g1_mul_batch(o.g1_to_bytes(p), ks[:8])
    assert [bytes(x) for x in got] == [o.g1_to_bytes(o.G1.mul(p, k)) for k in ks[:8]]


@pytest.mark.slow
@pytest.mark.parametrize("logm", [12, 14, 16] + [pytest.param(k, marks=pytest.mark.skipif(
    not os.environ.get("G16_SLOW_PINS"), reason="opt-in (G16_SLOW_PINS=1): 1 / 5 min of pure-Python NTTs; "
    "run once per round, log under profiles/")) for k in (18, 20)] + [pytest.param(22, marks=pytest.mark.skipif(
    os.environ.get("G16_SLOW_PINS") != "22", reason="opt-in (G16_SLOW_PINS=22): the headline size, ~20 min of pure-Python NTTs"))])
def test_c_oracle_prove_matches_python_oracle_2p12_2p14(logm):
    """oracle/groth16_cpu.c's CircomReduction prove == oracle/bn254_ref.py at 2^12, 2^14 and 2^16
    constraints (default suite) and at 2^18 and 2^20 -- BASELINE configs[1]'s size -- (opt-in), with uneven
    rows (dense-skewed family + one 9-term row): h element for element and the 256 proof bytes.  This is
    the byte-level pin of the C restatement, the checker of every GPU proof at 2^14 .. 2^27.  The key
    cycles through 64 random points (prove is linear in the key: no trapdoor structure is needed); since
    round 5 the Python MSM adds the scalars of equal bases first (bn254_ref.Group.msm), so its cost at
    these sizes is the six pure-Python NTTs: 12 s at 2^16, 1 min at 2^18, ~5 min at 2^20 (round 4: 11 min
    at 2^18, most of it a quadratic list build in the circuit generator and a modular inversion per
    converted element)."""
    import cpu_ref
    cons, wit, nv, _ = H.dense_skewed_circuit((1 << logm) - 5, seed=12, long_rows=(9,))
    npub, ni = 1, 2
    rng = random.Random(logm)
    K = 64
    b1 = [o.G1.mul(o.G1_GEN, rng.randrange(1, o.R_MOD)) for _ in range(K)]
    b2 = [o.G2.mul(o.G2_GEN, rng.randrange(1, o.R_MOD)) for _ in range(K)]
    n = o.domain_size_for(len(cons) + ni)
    assert n == 1 << logm
    opk = dict(n_vars=nv, n_public=npub, domain_size=n, alpha_g1=b1[0], beta_g1=b1[1], beta_g2=b2[0], gamma_g2=b2[1],
               delta_g1=b1[2], delta_g2=b2[2], ic=b1[:2],
               a_query=[b1[i % K] for i in range(nv)], b_g1_query=[b1[(i * 7 + 1) % K] for i in range(nv)],
               b_g2_query=[b2[(i * 5 + 2) % K] if i % 13 else None for i in range(nv)],
               l_query=[b1[(i * 3 + 5) % K] for i in range(nv - ni)], h_query=[b1[(i * 11 + 3) % K] for i in range(n)])
    ar, br = o.matrices_from_r1cs(cons)

    class M:
        pass

    def csr(rows):
        m = M()
        m.row_ptr = np.array([0] + list(np.cumsum([len(r) for r in rows])), dtype=np.uint32)
        m.col = np.array([i for r in rows for _c, i in r], dtype=np.uint32)
        m.coeff = H.fr_mont_arr([c for r in rows for c, _i in r])
        return m
    mats = M()
    mats.a, mats.b, mats.num_constraints = csr(ar), csr(br), len(cons)
    pk = M()
    pk.n_vars, pk.n_public, pk.domain_size = nv, npub, n
    pk.a_query, pk.b_g1_query, pk.b_g2_query = H.g1_arr(opk["a_query"]), H.g1_arr(opk["b_g1_query"]), H.g2_arr(opk["b_g2_query"])
    pk.l_query, pk.h_query = H.g1_arr(opk["l_query"]), H.g1_arr(opk["h_query"])
    pk.vk = M()
    pk.vk.alpha_g1, pk.vk.beta_g2, pk.vk.delta_g2 = o.g1_to_bytes(opk["alpha_g1"]), o.g2_to_bytes(opk["beta_g2"]), o.g2_to_bytes(opk["delta_g2"])
    pk.beta_g1, pk.delta_g1 = o.g1_to_bytes(opk["beta_g1"]), o.g1_to_bytes(opk["delta_g1"])
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    got, h = cpu_ref.prove(pk, mats, H.fr_mont_arr([r]), H.fr_mont_arr([s]), H.fr_mont_arr(wit), want_h=True)
    assert H.fr_from_mont_arr(h) == o.witness_map_from_matrices(ar, br, ni, len(cons), wit)
    want = o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=ar, b=br), ni, len(cons), wit)
    assert got == o.proof_to_bytes(want)


def test_poseidon_reference_matches_circomlibjs_kats(emu):
    """BASELINE configs[4] checker: oracle/poseidon_ref.py (Grain-LFSR parameters of the Poseidon paper's
    reference generator, circomlib's t / R_F / R_P) reproduces circomlibjs' hash known answers with no
    constant typed in; bench.py's own generator (an independent implementation of the same LFSR) yields
    the same 195 round constants and 3 x 3 MDS matrix; the witness of bench.poseidon_chain_circuit walks
    the oracle's hash chain (h_1 = the KAT, public output = h_H) and satisfies every row (the
    satisfiability kernel on the emulator)."""
    import sys
    import poseidon_ref as pr
    import circom_compat_amd as cc
    from circom_compat_amd import _binding
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert pr.poseidon([1, 2]) == 0x115cc0f5e7d690413df64c6b9662e9cf2a3617f2743245519e19607a4417189a == pr.KATS[(1, 2)]
    assert pr.poseidon([1]) == 18586133768512220936620570745912940619677854269274689475585506675881198879027
    rc, mds = pr.parameters(3)
    assert len(rc) == 65 * 3 and all(0 < c < o.R_MOD for c in rc)
    assert bench.poseidon_parameters(3, 8, 57) == (rc, mds)
    assert bench.poseidon_parameters(2, 8, 56) == pr.parameters(2)
    saved, _binding._default = _binding._default, emu
    try:
        mats, (A, B, Cm), w, n_vars = bench.poseidon_chain_circuit(cc, 11)
    finally:
        _binding._default = saved
    n_hashes = mats.num_constraints // 240
    assert n_hashes == 8 and mats.num_constraints == 240 * 8
    chain = pr.hash_chain(1, [i + 2 for i in range(n_hashes)])
    assert chain[1] == pr.KATS[(1, 2)] and w[1] == chain[-1] and w[2] == 1 and w[3:3 + n_hashes] == list(range(2, 10))
    # the x5 wire of lane 0's last S-box of hash 0 and the two beside it mix into h_1
    s_base = 3 + n_hashes
    x5 = [w[s_base + q] for q in (239, 236, 233)]
    assert sum(m * x for m, x in zip(mds[0], x5)) % o.R_MOD == chain[1]
    circ = cc.CircomCircuit(type("R", (), dict(a=A, b=B, c=Cm, num_constraints=mats.num_constraints,
                                               wire_mapping=None, num_inputs=2, num_variables=n_vars)), w)
    assert circ.first_unsatisfied(emu) == -1
    bad = list(w)
    bad[s_base + 700] ^= 1
    circ_bad = cc.CircomCircuit(type("R", (), dict(a=A, b=B, c=Cm, num_constraints=mats.num_constraints,
                                                   wire_mapping=None, num_inputs=2, num_variables=n_vars)), bad)
    assert circ_bad.first_unsatisfied(emu) >= 0


def test_checker_thread_pool_follows_the_cgroup_cpu_grant(tmp_path):
    """oracle/cpu_ref.py sizes the C restatement's OpenMP pool to the cores the host GRANTS: a gpurun box shows
    256 logical CPUs under cgroup v2 cpu.max = 1600000 100000 (16 cores), and 128 throttled threads were slower
    than 32 (profiles/r06_cpu_threads_sweep.txt).  The parser: v2 quota, v2 'max', v1 quota, v1 unlimited, nothing."""
    import cpu_ref
    v2 = tmp_path / "v2"
    v2.mkdir()
    (v2 / "cpu.max").write_text("1600000 100000\n")
    assert cpu_ref.host_cpu_grant(str(v2), aff=256)[0] == 16
    assert cpu_ref.host_cpu_grant(str(v2), aff=8)[0] == 8            # the affinity mask is the tighter bound
    (v2 / "cpu.max").write_text("150000 100000\n")
    assert cpu_ref.host_cpu_grant(str(v2), aff=256)[0] == 2          # 1.5 cores: rounded up
    (v2 / "cpu.max").write_text("max 100000\n")
    assert cpu_ref.host_cpu_grant(str(v2), aff=64) == (64, "64 CPUs in the affinity mask, no cgroup CPU quota")
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("400000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert cpu_ref.host_cpu_grant(str(v1), aff=64)[0] == 4
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert cpu_ref.host_cpu_grant(str(v1), aff=64)[0] == 64
    assert cpu_ref.host_cpu_grant(str(tmp_path / "none"), aff=12)[0] == 12
    # the loaded library: never more threads than OpenMP's own default, never fewer than one
    assert 1 <= cpu_ref.max_threads() <= cpu_ref.omp_default_threads()
