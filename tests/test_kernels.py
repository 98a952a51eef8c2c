"""Parity of every kernel stage with the oracle: NTT passes, CircomReduction witness map, MSM
(sort / accumulate / combine / reduce), proof assembly, sharded partial/finish.

Each case runs twice through the SAME C ABI:
  [emu]  kernel sources stepped through on the CPU SIMT emulator (tests/emu) -- catches indexing and
         algorithm bugs without a GPU; not a parity claim;
  [gpu]  the product library on a real MI355X (`-m gpu`) -- THE parity tests: bit-exact against the
         oracle and the reference's golden fixtures."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import bn254_ref as o
import helpers as H


def _ntt(lib, vals, inverse, algo):
    arr = H.fr_mont_arr(vals)
    k = len(vals).bit_length() - 1
    st = lib.g16_fft_in_place(0, arr.ctypes.data, k, 1 if inverse else 0, algo)
    lib.check(st)
    return H.fr_from_mont_arr(arr)


@pytest.mark.parametrize("k", [1, 2, 3, 5, 8, 10, 11, 12])
def test_ntt_vs_oracle(lib, k):
    rng = random.Random(k)
    x = H.rand_fr(rng, 1 << k)
    assert _ntt(lib, x, False, 0) == o.ntt(x)
    assert _ntt(lib, x, True, 0) == o.ntt(x, inverse=True)
    assert _ntt(lib, x, False, 1) == o.ntt(x)
    # the witness map's own kernels (lazy 29-bit limbs, limb planes): algo 2 = DIF, 3 = DIT
    assert _ntt(lib, x, False, 2) == o.ntt(x)
    assert _ntt(lib, x, True, 2) == o.ntt(x, inverse=True)
    assert _ntt(lib, x, False, 3) == o.ntt(x)


@pytest.mark.parametrize("two_level", [False, True])
@pytest.mark.parametrize("k", [13, 16, 17])
def test_ntt29_multi_pass_single_level_and_two_level_tables(lib, monkeypatch, k, two_level):
    """The witness map's NTT with strided passes of 3 .. 7 stages (2^16: two passes of 6 stages run
    without the mid-pass value renormalisation since round 5; 2^17: a 7-stage pass keeps it), with the
    round-5 single-level twiddle tables (one product per inter-pass twiddle, the default up to 2^24) and
    with the two-level tables every larger plan uses (G16_NTT_TWO_LEVEL=1): DIF forward / inverse and DIT
    == the oracle's plain radix-2 transform; the emulator build asserts the limb / value bounds."""
    if k == 17 and lib.path.endswith("libg16_emu.so"):
        pytest.skip("GPU suite only (the CPU suite keeps 2^13 and 2^16)")
    if two_level:
        monkeypatch.setenv("G16_NTT_TWO_LEVEL", "1")
    rng = random.Random(100 + k)
    x = H.rand_fr(rng, 1 << k)
    want_f, want_i = o.ntt(x), o.ntt(x, inverse=True)
    assert _ntt(lib, x, False, 2) == want_f
    assert _ntt(lib, x, True, 2) == want_i
    assert _ntt(lib, x, False, 3) == want_f
    if k == 16 and not two_level:       # the value bookkeeping at its extreme: every sum doubles
        ones = [o.R_MOD - 1] * (1 << k)
        assert _ntt(lib, ones, True, 2) == o.ntt(ones, inverse=True)
        assert _ntt(lib, ones, False, 3) == o.ntt(ones)


@pytest.mark.parametrize("two_level", [False, True])
def test_witness_map_multi_pass_2p12_both_table_modes(lib, monkeypatch, two_level):
    """CircomReduction::witness_map_from_matrices (qap.rs:23-88) at 2^12 rows -- a strided pass, the
    contiguous pass and the coset twist fused into the last inverse pass -- element for element == the
    oracle, with the single-level tables (twist read at the stored position) and the two-level ones."""
    import circom_compat_amd as cc
    if two_level:
        monkeypatch.setenv("G16_NTT_TWO_LEVEL", "1")
    cons, w, n_vars, _ = H.squaring_chain(12)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, 2, n_vars, lib)
    h = cc.CircomReduction.witness_map_from_matrices(mats, 2, len(cons), w, lib=lib)
    assert H.fr_from_mont_arr(h) == o.witness_map_from_matrices(a_rows, b_rows, 2, len(cons), w)


def test_ntt_lazy_limbs_edge_values(lib):
    """all-zero, all (r-1) and a single spike: the value-drift bookkeeping of ntt29.hip is exercised at
    its extremes (the emulator build asserts the limb / value bounds on every product)"""
    k = 11
    n = 1 << k
    for x in ([0] * n, [o.R_MOD - 1] * n, [1] + [0] * (n - 1), [o.R_MOD - 1 if i % 2 else 1 for i in range(n)]):
        assert _ntt(lib, x, False, 2) == o.ntt(x)
        assert _ntt(lib, x, True, 2) == o.ntt(x, inverse=True)
        assert _ntt(lib, x, False, 3) == o.ntt(x)


def test_ntt_small_matches_naive_dft(lib):
    rng = random.Random(3)
    x = H.rand_fr(rng, 16)
    assert _ntt(lib, x, False, 0) == o.dft_naive(x)


def _prover(lib, pk, mats, **kw):
    import circom_compat_amd as cc
    return cc.Prover(pk, mats, lib=lib, **kw)


def test_witness_map_test_zkey(lib, golden):
    """reference fixture test.zkey, w = [1,33,3,11] -> SURVEY Appendix C.1 h vector"""
    import circom_compat_amd as cc
    pk, mats = cc.read_zkey(os.path.join(golden, "test.zkey"), lib=lib)
    assert (mats.num_instance_variables, mats.num_constraints) == (2, 1)
    pr = _prover(lib, pk, mats)
    h = H.fr_from_mont_arr(pr.witness_map([1, 33, 3, 11]))
    assert h == [190042957931705914545745448213290365653268903107554486158945885339918534040,
                 1419419912336859849922499081568465451015963211367001816420814749527554252505,
                 21698199913907568871316484237715844311305012280401884867766642092812430274913,
                 9524701523582778197584879850388312504848302205747610345200903552183809682204]
    # CircomReduction surface (stateless call, witness-map-only ctx)
    h2 = cc.CircomReduction.witness_map_from_matrices(mats, 2, 1, [1, 33, 3, 11], lib=lib)
    assert H.fr_from_mont_arr(h2) == h


def test_prove_test_zkey(lib, golden):
    """prove on the reference's own zkey; bit-exact vs oracle and accepted by the pairing check
    (= reference tests src/zkey.rs:875-919 with the rng pinned)"""
    import circom_compat_amd as cc
    data = open(os.path.join(golden, "test.zkey"), "rb").read()
    pk, mats = cc.read_zkey(data, lib=lib)
    opk, omats = o.read_zkey(data)
    w = [1, 33, 3, 11]
    for r, s in [(0, 0),
                 (3413513218498352040262653353725127729454431939539290118844322056224532443637,
                  6077776500692565155461894309070795882353485867345896979329447163197530625403)]:
        proof = cc.Groth16.create_proof_with_reduction_and_matrices(pk, r, s, mats, 2, 1, w, lib=lib)
        want = o.create_proof_with_reduction_and_matrices(opk, r, s, omats, 2, 1, w)
        assert proof.raw == o.proof_to_bytes(want)
        assert o.verify_proof(opk, [33], H.proof_from_bytes(proof.raw))
        assert not o.verify_proof(opk, [34], H.proof_from_bytes(proof.raw))


def test_separate_a_b1_arrays_knob_gives_the_same_proof(lib, golden):
    """G16_NO_PAIR_AB=1 (read once per process: A and B1 planes in separate arrays, two launches) is the
    A/B partner of the interleaved A|B1 pair launch -- same bytes either way"""
    import subprocess
    import sys
    import circom_compat_amd as cc
    data = open(os.path.join(golden, "test.zkey"), "rb").read()
    pk, mats = cc.read_zkey(data, lib=lib)
    r, s = 12345678901234567890, 98765432109876543210
    here = cc.Groth16.create_proof_with_reduction_and_matrices(pk, r, s, mats, 2, 1, [1, 33, 3, 11], lib=lib).raw
    code = ("import sys, circom_compat_amd as cc\n"
            "from circom_compat_amd import _binding\n"
            f"lib = _binding.Library({lib.path!r})\n"
            f"pk, mats = cc.read_zkey(open({os.path.join(golden, 'test.zkey')!r}, 'rb').read(), lib=lib)\n"
            f"p = cc.Groth16.create_proof_with_reduction_and_matrices(pk, {r}, {s}, mats, 2, 1, [1, 33, 3, 11], lib=lib)\n"
            "print(bytes(p.raw).hex())\n")
    env = dict(os.environ, G16_NO_PAIR_AB="1", PYTHONPATH=os.pathsep.join(sys.path))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == bytes(here).hex()


def test_witness_map_circuit2(lib, golden):
    """131 constraints with very uneven rows (65-term rows), n = 256; witness from the shipped .wtns"""
    import circom_compat_amd as cc
    r1 = cc.R1CS.from_file(os.path.join(golden, "circuit2.r1cs"), lib=lib)
    w = cc.read_wtns(os.path.join(golden, "circuit2.wtns"), lib=lib)
    wi = H.fr_from_mont_arr(w)
    oc = o.read_r1cs(open(os.path.join(golden, "circuit2.r1cs"), "rb").read())
    assert wi == o.read_wtns(open(os.path.join(golden, "circuit2.wtns"), "rb").read())
    a_rows, b_rows = o.matrices_from_r1cs(oc["constraints"])
    mats = r1.matrices()
    h = cc.CircomReduction.witness_map_from_matrices(mats, r1.num_inputs, r1.num_constraints, w, lib=lib)
    want = o.witness_map_from_matrices(a_rows, b_rows, oc["num_inputs"], oc["n_constraints"], wi)
    assert H.fr_from_mont_arr(h) == want


def _row_class_circuit(rng, lens):
    """rows of the given (A terms, B terms) lengths over fresh random wires, unit and non-unit coefficients mixed,
    satisfiable by construction (C row = one fresh wire); wire 0 (the constant) is a term of every long row"""
    P = o.R_MOD
    w = [1, 0] + [rng.randrange(P) for _ in range(40)]
    cons = []

    def lc(k):
        terms = []
        for j in range(k):
            wi = 0 if (j == 0 and k > 4) else rng.randrange(2, len(w))
            cf = 1 if rng.random() < 0.4 else (P - 1 if rng.random() < 0.2 else rng.randrange(P))
            terms.append((wi, cf))
        return terms
    for la, lb in lens:
        A, B = lc(la), lc(lb)
        val = sum(c * w[i] for i, c in A) * sum(c * w[i] for i, c in B) % P
        w.append(val)
        cons.append((A, B, [(len(w) - 1, 1)]))
    cons.append(([(len(w) - 1, 1)], [(0, 1)], [(1, 1)]))
    w[1] = w[-1]
    return cons, w


@pytest.mark.parametrize("huge", [3000, pytest.param(100000, marks=pytest.mark.gpu)])
def test_witness_map_row_classes_vs_oracle(lib, huge):
    """Row-length-adaptive evaluate_constraint (csrc/spmv.h; reference qap.rs:37-44): one-term rows with
    coefficient 1 (the skipped multiplication), rows at and across the short / medium / huge class boundaries
    (4 | 5, 64 | 65 terms), rows that are long in A only or in B only, an empty A row, and one row of `huge`
    terms (10^5 on the GPU: a Num2Bits / long linear sum of a real circom circuit) -- h == oracle."""
    import circom_compat_amd as cc
    rng = random.Random(4242 + huge)
    lens = [(1, 1), (4, 4), (5, 1), (1, 5), (4, 5), (16, 17), (61, 9), (64, 64), (65, 1), (1, 65), (0, 3),
            (128, 129), (257, 130), (huge, 7), (1, 1), (2, 3)]
    cons, w = _row_class_circuit(rng, lens)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, 2, len(w), lib)
    h = cc.CircomReduction.witness_map_from_matrices(mats, 2, len(cons), H.fr_mont_arr(w), lib=lib)
    want = o.witness_map_from_matrices(a_rows, b_rows, 2, len(cons), w)
    assert H.fr_from_mont_arr(h) == want


def test_trapdoor_setup_with_a_huge_column_vs_oracle(lib):
    """The key generator's column sums (csrc/keygen.hip k_spmv over the TRANSPOSED matrices): the constant
    wire is a term of 150 rows here, i.e. one transposed row of the huge class next to one-term rows -- the
    shape that cost one thread 0.9 s on the 2^20 Poseidon chain.  Every query point == oracle trapdoor_setup."""
    import circom_compat_amd as cc
    rng = random.Random(99)
    cons, w = _row_class_circuit(rng, [(5, 1)] * 150 + [(2, 2)] * 4)
    n_vars = len(w)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    opk = o.trapdoor_setup(cons, n_vars, 1, *tox)
    rows = lambda k: [[(c, wdx) for wdx, c in con[k]] for con in cons]
    a, b, c = (cc.Csr.from_rows(rows(k), lib) for k in range(3))
    pk = cc.trapdoor_setup(a, b, c, n_vars, 1, tox, lib=lib)
    want = H.pk_from_oracle(opk)
    for name in ("a_query", "b_g1_query", "b_g2_query", "l_query", "h_query"):
        assert np.array_equal(getattr(pk, name), getattr(want, name)), name
    assert np.array_equal(pk.vk.gamma_abc_g1, want.vk.gamma_abc_g1)


@pytest.mark.parametrize("n,c,planes", [(5, 3, 0), (40, 4, 1), (40, 4, 3), (300, 7, 0), (300, 6, 2),
                                        (60, 17, 0), (60, 19, 5)])  # large windows: long offsets in the bucket reduction
def test_msm_vs_oracle(lib, n, c, planes):
    """G1 and G2 MSM through the resident-query entry points, several window/plane layouts"""
    import circom_compat_amd as cc
    if c > 17 and lib.path.endswith("libg16_emu.so"):
        pytest.skip("2^18 buckets stepped through the emulator take half a minute; c = 17 covers the path there")
    rng = random.Random(n * 31 + c)
    N = n + 1
    pts1 = H.rand_g1(rng, 6)
    pts2 = H.rand_g2(rng, 4)
    A = [pts1[rng.randrange(6)] if rng.random() > 0.1 else None for _ in range(N)]
    for i in range(2, N, 7):           # distinct multiples so buckets see many different points
        A[i] = o.G1.mul(pts1[i % 6], i + 3)
    B2 = [o.G2.mul(pts2[i % 4], i + 1) if i % 9 else None for i in range(N)]
    pk = dict(n_vars=N, n_public=1, domain_size=0, alpha_g1=pts1[0], beta_g1=pts1[1],
              beta_g2=pts2[0], gamma_g2=pts2[1], delta_g1=pts1[2], delta_g2=pts2[2], ic=pts1[:2],
              a_query=A, b_g1_query=list(reversed(A)), b_g2_query=B2, l_query=A[2:], h_query=None)
    # a trivial circuit with the right shape: m = 1 row, domain from m + num_inputs
    m = 1
    dom = o.domain_size_for(m + 2)
    pk["domain_size"] = dom
    pk["h_query"] = [A[(3 * i + 1) % N] for i in range(dom)]
    mats = H.matrices_from_rows([[(1, 1)]], [[(1, 0)]], 2, N, lib)
    pr = cc.Prover(H.pk_from_oracle(pk), mats, lib=lib, window_bits=c, planes=planes)
    scal = H.rand_fr(rng, n)
    scal[0] = 0
    scal[1] = 1
    scal[2] = o.R_MOD - 1
    if n > 20:
        for i in range(5, 20):
            scal[i] = 1               # hot bucket
        scal[20] = (1 << (c - 1))     # digit exactly half -> stays positive
        scal[21] = (1 << (c - 1)) + 1 # first negative digit with carry
    assert pr.msm_g1(0, scal) == o.g1_to_bytes(o.G1.msm(A[1:], scal))
    assert pr.msm_g1(1, scal) == o.g1_to_bytes(o.G1.msm(pk["b_g1_query"][1:], scal))
    assert pr.msm_g2(scal) == o.g2_to_bytes(o.G2.msm(B2[1:], scal))
    assert pr.msm_g1(2, scal[:n - 1]) == o.g1_to_bytes(o.G1.msm(pk["l_query"], scal[:n - 1]))
    hs = H.rand_fr(rng, dom)
    assert pr.msm_g1(3, hs) == o.g1_to_bytes(o.G1.msm(pk["h_query"], hs))


def test_msm_hot_bucket_split(lib):
    """more than MSM_CHUNK * MSM_SMALL_MULTI equal scalars: exercises task splitting and the
    workgroup-per-bucket combine"""
    import circom_compat_amd as cc
    rng = random.Random(5)
    emu = lib.path.endswith("libg16_emu.so")
    n = 600 if emu else 256 * 33 + 50      # one bucket spanning 75 / 1062 lane segments (> MSM_SMALL_MULTI)
    N = n + 1
    base = H.rand_g1(rng, 3)
    A = [base[i % 3] for i in range(N)]
    g2 = H.rand_g2(rng, 1)
    pk = dict(n_vars=N, n_public=1, domain_size=4, alpha_g1=base[0], beta_g1=base[1], beta_g2=g2[0],
              gamma_g2=g2[0], delta_g1=base[2], delta_g2=g2[0], ic=base[:2], a_query=A, b_g1_query=A,
              b_g2_query=[g2[0]] * N, l_query=A[2:], h_query=base + base[:1])
    mats = H.matrices_from_rows([[(1, 1)]], [[(1, 0)]], 2, N, lib)
    # full planes (51 windows) on the GPU; the emulator precomputes 3 planes (17 bucket sets folded by
    # k_horner) -- the plane precomputation of the unused G2 query would otherwise take minutes there
    pr = cc.Prover(H.pk_from_oracle(pk), mats, lib=lib, window_bits=5, planes=3 if emu else 0)
    scal = [1] * n
    for i in range(0, n, 97):
        scal[i] = rng.randrange(o.R_MOD)
    # closed form: sum of scalars per distinct base point
    sums = [0, 0, 0]
    for i, s in enumerate(scal):
        sums[(i + 1) % 3] = (sums[(i + 1) % 3] + s) % o.R_MOD
    want = o.G1.sum([o.G1.mul(base[j], sums[j]) for j in range(3)])
    assert pr.msm_g1(0, scal) == o.g1_to_bytes(want)


def test_optimistic_accumulation_overflow_falls_back_to_the_exact_kernel(lib):
    """A key whose query points are ALL the same point and a witness of ones: every mixed addition after a
    lane's first is acc + P with acc = P, i.e. the x-coordinates coincide, the optimistic G1 kernel sets
    every one of them aside and its list overflows (segments of 8 entries x 7 deferrals each) -- k_acc_fixup
    raises `overflow` and the exact kernel behind it redoes the launch.  Deterministic, unlike a hot
    bucket of several points; covers the interleaved A|B1 pair launch (prove) and the single-query
    launches (L, H, msm_g1).  The G2 launch runs the exact kernel anyway.  Bytes == oracle."""
    import circom_compat_amd as cc
    rng = random.Random(812)
    # the list holds 512 entries in the emulator build, 2048 in the product (msm.h, MSM_FIX_CAP): 100 / 300
    # segments of 8 entries defer 7 additions each
    n = 800 if lib.path.endswith("libg16_emu.so") else 2400
    N = n + 1
    P = o.G1.mul(o.G1_GEN, 12345)
    Q = o.G2.mul(o.G2_GEN, 54321)
    base = H.rand_g1(rng, 3)
    g2 = H.rand_g2(rng, 3)
    pk = dict(n_vars=N, n_public=1, domain_size=4, alpha_g1=base[0], beta_g1=base[1], beta_g2=g2[0], gamma_g2=g2[1],
              delta_g1=base[2], delta_g2=g2[2], ic=base[:2], a_query=[P] * N, b_g1_query=[P] * N,
              b_g2_query=[Q] * N, l_query=[P] * (N - 2), h_query=[P] * 4)
    a_rows, b_rows = [[(1, 1)]], [[(1, 0)]]
    mats = H.matrices_from_rows(a_rows, b_rows, 2, N, lib)
    pr = cc.Prover(H.pk_from_oracle(pk), mats, lib=lib, window_bits=16)
    ones = [1] * n
    assert pr.msm_g1(0, ones) == o.g1_to_bytes(o.G1.mul(P, n))
    assert pr.msm_g1(2, ones[:n - 1]) == o.g1_to_bytes(o.G1.mul(P, n - 1))
    assert pr.msm_g2(ones) == o.g2_to_bytes(o.G2.mul(Q, n))
    w = [1] * N
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    want = o.create_proof_with_reduction_and_matrices(pk, r, s, dict(a=a_rows, b=b_rows), 2, 1, w)
    assert pr.prove(r, s, w).raw == o.proof_to_bytes(want)
    # a mixed case right behind it on the same ctx: the list and its overflow flag are reset per launch
    w2 = [1] + H.rand_fr(rng, n)
    want2 = o.create_proof_with_reduction_and_matrices(pk, r, s, dict(a=a_rows, b=b_rows), 2, 1, w2)
    assert pr.prove(r, s, w2).raw == o.proof_to_bytes(want2)


def test_prove_synthetic_circuit_vs_oracle(lib):
    """squaring-chain circuit with a trapdoor key: proof bytes == oracle, proof verifies,
    wrong public input is rejected; also the world = 2 partial/finish path gives the same bytes"""
    import circom_compat_amd as cc
    cons, w, n_vars, n_pub = H.squaring_chain(4)       # m = 14, n = 16
    rng = random.Random(42)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *tox)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    omats = dict(a=a_rows, b=b_rows)
    mats = H.matrices_from_rows(a_rows, b_rows, 2, n_vars, lib)
    pk = H.pk_from_oracle(opk)
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    want = o.create_proof_with_reduction_and_matrices(opk, r, s, omats, 2, len(cons), w)
    assert o.verify_proof(opk, w[1:2], want)
    pr = cc.Prover(pk, mats, lib=lib)
    proof = pr.prove(r, s, w)
    assert proof.raw == o.proof_to_bytes(want)
    assert not o.verify_proof(opk, [(w[1] + 1) % o.R_MOD], H.proof_from_bytes(proof.raw))
    # sharded: two ranks, all-gather emulated by concatenation
    parts = b""
    for rank in range(2):
        p2 = cc.Prover(pk, mats, lib=lib, rank=rank, world=2)
        parts += p2.prove_partial(r, s, w)
    assert p2.prove_finish(r, s, parts).raw == proof.raw


def test_memory_plan_falls_back_to_fewer_planes_instead_of_refusing(emu, monkeypatch):
    """g16_ctx_create plans both MSM configurations against the free device memory (api.hip,
    plan_msm_configs): full planes when they fit, otherwise the pair of plane counts that fits with the
    fewest bucket sets to reduce per proof (5 D_w + D_h; D > 1 sets are folded by k_horner), and only a
    key for which no plane count fits is refused, with the reason.  The emulator reports whatever
    G16_EMU_FREE_BYTES says, so the test walks the free memory up from the refusal threshold: every
    configuration on the way proves the oracle's bytes and the reduction cost never goes up with more
    memory.  (At this toy size every extra bucket set costs more workspace than a plane of 126 points
    saves, so the walk starts at 16 planes; at 2^26 a plane is 21 GB and a bucket set 1.5 GB.)"""
    import circom_compat_amd as cc
    cons, w, n_vars, n_pub = H.squaring_chain(7)
    rng = random.Random(77)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *tox)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, 2, n_vars, emu)
    pk = H.pk_from_oracle(opk)
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    want = o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), 2, len(cons), w))
    WB = 4                       # W = 64 windows -> 32 planes at most, 40 KB per plane of the witness queries

    def create(extra):           # the plan keeps min(2 GiB + 2 %, a quarter) of the free memory out of its budget:
        monkeypatch.setenv("G16_EMU_FREE_BYTES", str(int(extra / 0.75) + 1))   # a budget of `extra` bytes
        try:
            return cc.Prover(pk, mats, lib=emu, window_bits=WB)
        except Exception as e:
            assert "does not fit" in str(e), e
            return None

    lo, hi = 0, 16 << 10         # smallest `extra` (KiB) that is not refused
    while lo < hi:
        mid = (lo + hi) // 2
        pr = create(mid << 10)
        if pr is not None:
            pr.close()
        lo, hi = (lo, mid) if pr is not None else (mid + 1, hi)
    t_min = lo << 10
    assert t_min > 0 and create(t_min - 1024) is None, "a key that cannot fit is refused, with the reason"
    seen = {}
    cost = []
    for j in range(16):
        pr = create(t_min + j * (96 << 10))
        info = pr.info()
        key = (info["planes_w"], info["planes_h"])
        cost.append(5 * info["D_w"] + info["D_h"])
        if key not in seen:
            seen[key] = info
            assert info["D_w"] == -(-info["W_w"] // info["planes_w"]) and info["D_h"] == -(-info["W_h"] // info["planes_h"])
            assert pr.prove(r, s, w).raw == want, "planes %r: proof differs from the oracle" % (key,)
        pr.close()
        if key == (32, 32):
            break
    monkeypatch.delenv("G16_EMU_FREE_BYTES")
    assert (32, 32) in seen and len(seen) >= 3, sorted(seen)
    assert all(x >= y for x, y in zip(cost, cost[1:])), "more memory must never cost more bucket sets: %r" % (cost,)


def test_trapdoor_setup_vs_oracle(lib):
    """GPU key generator (g16_setup_create) == oracle trapdoor setup, point for point; the key then
    proves and the proof verifies.  Mirrors reference tests/groth16.rs:11-40 (setup -> prove ->
    verify) with the rng pinned and CircomReduction as the QAP."""
    import circom_compat_amd as cc
    cons, w, n_vars, n_pub = H.squaring_chain(3)       # m = 6, n = 8
    rng = random.Random(77)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *tox)
    rows = lambda k: [[(c, wdx) for wdx, c in con[k]] for con in cons]
    a, b, c = (cc.Csr.from_rows(rows(k), lib) for k in range(3))
    pk = cc.trapdoor_setup(a, b, c, n_vars, n_pub, tox, lib=lib)
    want = H.pk_from_oracle(opk)
    for name in ("a_query", "b_g1_query", "b_g2_query", "l_query", "h_query"):
        assert np.array_equal(getattr(pk, name), getattr(want, name)), name
    assert np.array_equal(pk.vk.gamma_abc_g1, want.vk.gamma_abc_g1)
    for name in ("alpha_g1", "beta_g2", "gamma_g2", "delta_g2"):
        assert bytes(getattr(pk.vk, name)) == bytes(getattr(want.vk, name)), name
    assert bytes(pk.beta_g1) == bytes(want.beta_g1) and bytes(pk.delta_g1) == bytes(want.delta_g1)
    mats = cc.ConstraintMatrices(2, n_vars - 1, len(cons), a, b)
    proof = cc.Prover(pk, mats, lib=lib).prove(123, 456, w)
    assert o.verify_proof(opk, w[1:2], H.proof_from_bytes(proof.raw))


def test_constraint_satisfaction_kernel(lib, golden):
    """reference src/circom/circuit.rs:92-107 (cs.is_satisfied()) and the debug-build check of
    CircomBuilder::build (src/circom/builder.rs:101-114), on the reference's own fixtures"""
    import circom_compat_amd as cc
    import json
    # circuits come from the builder, as in the reference's test: wire mapping disabled (builder.rs:84-85)
    b1 = cc.CircomBuilder(cc.R1CS.from_file(os.path.join(golden, "mycircuit.r1cs"), lib))
    assert b1.build([1, 33, 3, 11]).first_unsatisfied(lib) == -1
    assert b1.build([1, 34, 3, 11]).first_unsatisfied(lib) == 0
    with pytest.raises(cc.G16Error):
        b1.build([1, 34, 3, 11], sanity_check=True, lib=lib)
    # a circuit that keeps R1CS::from's mapping reads w[m[i]] (circuit.rs:35-58): mycircuit's labels
    # permute the wires, so the wire-order witness no longer satisfies it and the public input moves
    raw = cc.CircomCircuit(b1.r1cs, [1, 33, 3, 11])
    m = b1.r1cs.wire_mapping
    assert raw.full_assignment() == [[1, 33, 3, 11][m[i]] for i in range(4)]
    assert raw.get_public_inputs() == [[1, 33, 3, 11][m[1]]]
    assert b1.setup().r1cs.wire_mapping is None and b1.r1cs.wire_mapping is not None
    b2 = cc.CircomBuilder(cc.R1CS.from_file(os.path.join(golden, "circuit2.r1cs"), lib))
    w2 = [int(x) for x in json.load(open(os.path.join(golden, "safe-circuit-witness.json")))]
    c2 = b2.build(w2)
    assert c2.is_satisfied(lib)
    w_bad = list(w2)
    w_bad[70] = (w_bad[70] + 1) % o.R_MOD
    bad = b2.build(w_bad).first_unsatisfied(lib)
    # the oracle names the same first failing row
    cons = o.read_r1cs(open(os.path.join(golden, "circuit2.r1cs"), "rb").read())["constraints"]
    lc = lambda terms: sum(c * w_bad[j] for j, c in terms) % o.R_MOD
    want = next(i for i, (A, B, Cc) in enumerate(cons) if lc(A) * lc(B) % o.R_MOD != lc(Cc))
    assert bad == want


def test_dense_skewed_circuit_through_zkey_writer_and_loader(lib, tmp_path):
    """SURVEY 8(d) config 5 (substitute family): uneven rows (3/2-term and two 65/64-term rows), a
    witness with most scalars in {0,1}; the key is minted on the device, written by the product's
    zkey writer, re-read by the product's read_zkey (Coefs(4) path, src/zkey.rs:151-196), and the
    proof must equal the oracle's byte for byte."""
    import circom_compat_amd as cc
    cons, w, n_vars, n_pub = H.dense_skewed_circuit(60, seed=5, long_rows=(17, 41))
    assert sum(1 for x in w if x in (0, 1)) * 2 >= len(w) - 10
    rows = lambda k: [[(c, wdx) for wdx, c in con[k]] for con in cons]
    a, b, c = (cc.Csr.from_rows(rows(k), lib) for k in range(3))
    r1cs_like = type("R", (), dict(a=a, b=b, c=c, num_constraints=len(cons), wire_mapping=None, num_inputs=2))
    assert cc.CircomCircuit(r1cs_like, w).first_unsatisfied(lib) == -1
    rng = random.Random(9)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    pk = cc.trapdoor_setup(a, b, c, n_vars, n_pub, tox, lib=lib)
    mats = cc.ConstraintMatrices(2, n_vars - 1, len(cons), a, b)
    path = str(tmp_path / "dense.zkey")
    cc.write_zkey(path, pk, mats, lib=lib)
    pk2, mats2 = cc.read_zkey(path, lib)
    assert (pk2.n_vars, pk2.n_public, pk2.domain_size) == (pk.n_vars, pk.n_public, pk.domain_size)
    assert np.array_equal(pk2.a_query, pk.a_query) and np.array_equal(pk2.h_query, pk.h_query)
    assert mats2.num_constraints == len(cons)
    assert np.array_equal(mats2.a.row_ptr, a.row_ptr) and np.array_equal(mats2.b.coeff, b.coeff)
    # the oracle parses the same file and proves with it
    opk, omats = o.read_zkey(open(path, "rb").read())
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    want = o.create_proof_with_reduction_and_matrices(opk, r, s, omats, 2, len(cons), w)
    assert o.verify_proof(opk, w[1:2], want)
    proof = cc.Groth16.create_proof_with_reduction_and_matrices(pk2, r, s, mats2, 2, len(cons), w, lib=lib)
    assert proof.raw == o.proof_to_bytes(want)


@pytest.mark.parametrize("n_pub", [4, 7])
def test_several_public_inputs(lib, tmp_path, n_pub):
    """num_inputs = p + 1 > 2 (every reference fixture has p = 1): the witness map copies p + 1
    witness rows after the constraints (qap.rs:46-48), the L query starts at wire p + 1
    (src/zkey.rs:118-121), the verifier folds p inputs into IC.  Wires 1..p of the dense circuit are
    declared public; key through the zkey writer and loader; proof bytes equal the oracle's and the
    proof verifies against exactly those p inputs."""
    import circom_compat_amd as cc
    cons, w, n_vars, _ = H.dense_skewed_circuit(45, seed=11 + n_pub, long_rows=(9,))
    ni = n_pub + 1
    rows = lambda k: [[(c, wdx) for wdx, c in con[k]] for con in cons]
    a, b, c = (cc.Csr.from_rows(rows(k), lib) for k in range(3))
    rng = random.Random(21 + n_pub)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    pk = cc.trapdoor_setup(a, b, c, n_vars, n_pub, tox, lib=lib)
    assert len(pk.vk.gamma_abc_g1) == ni and pk.l_query.shape[0] == n_vars - ni
    mats = cc.ConstraintMatrices(ni, n_vars - n_pub, len(cons), a, b)
    path = str(tmp_path / "pub.zkey")
    cc.write_zkey(path, pk, mats, lib=lib)
    pk2, mats2 = cc.read_zkey(path, lib)
    assert pk2.n_public == n_pub and mats2.num_instance_variables == ni
    opk, omats = o.read_zkey(open(path, "rb").read())
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    want = o.create_proof_with_reduction_and_matrices(opk, r, s, omats, ni, len(cons), w)
    assert o.verify_proof(opk, w[1:ni], want)
    bad = list(w[1:ni])
    bad[-1] = (bad[-1] + 1) % o.R_MOD
    assert not o.verify_proof(opk, bad, want)
    proof = cc.Groth16.create_proof_with_reduction_and_matrices(pk2, r, s, mats2, ni, len(cons), w, lib=lib)
    assert proof.raw == o.proof_to_bytes(want)
    h = cc.CircomReduction.witness_map_from_matrices(mats2, ni, len(cons), w, lib=lib)
    assert cc.fr_to_ints(h) == o.witness_map_from_matrices(omats["a"], omats["b"], ni, len(cons), w)


# ---- LibsnarkReduction (arkworks-generated keys): reference tests/groth16.rs ---------------------
def _fixture(golden, name):
    import json
    r1 = o.read_r1cs(open(os.path.join(golden, name + ".r1cs"), "rb").read())
    if name == "mycircuit":
        w = [1, 33, 3, 11]
    else:
        w = [int(x) for x in json.load(open(os.path.join(golden, "safe-circuit-witness.json")))]
    return r1, w


@pytest.mark.parametrize("name", ["mycircuit", "circuit2"])
def test_libsnark_witness_map_vs_oracle(lib, golden, name):
    """LibsnarkReduction::witness_map_from_matrices on the reference's circuits: h coefficients equal
    the oracle's restatement, value for value"""
    import circom_compat_amd as cc
    r1, w = _fixture(golden, name)
    cons = r1["constraints"]
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    want = o.witness_map_libsnark(a_rows, b_rows, r1["num_inputs"], len(cons), w)
    mats = H.matrices_from_rows(a_rows, b_rows, r1["num_inputs"], r1["n_wires"], lib)
    got = cc.LibsnarkReduction.witness_map_from_matrices(mats, r1["num_inputs"], len(cons), w, lib=lib)
    assert H.fr_from_mont_arr(got) == want
    # and CircomReduction on the same matrices is a different vector (README.md:69-74: do not mix)
    got_c = cc.CircomReduction.witness_map_from_matrices(mats, r1["num_inputs"], len(cons), w, lib=lib)
    assert H.fr_from_mont_arr(got_c) == o.witness_map_from_matrices(a_rows, b_rows, r1["num_inputs"], len(cons), w)
    assert H.fr_from_mont_arr(got_c) != want


@pytest.mark.parametrize("name", ["mycircuit", "circuit2"])
def test_groth16_libsnark_setup_prove_verify(lib, golden, name):
    """reference tests/groth16.rs:11-40 (mycircuit) and :75-104 (circuit2): generate parameters with
    the default (Libsnark) reduction, prove, verify; :42-73: a wrong public input is rejected.
    Key and proof are also compared with the oracle's byte for byte."""
    import circom_compat_amd as cc
    r1, w = _fixture(golden, name)
    cons = r1["constraints"]
    r1cs = cc.R1CS.from_file(os.path.join(golden, name + ".r1cs"), lib)
    rng = random.Random(2024)
    pk = cc.Groth16.generate_random_parameters_with_reduction(r1cs, rng, lib=lib)       # libsnark
    rng2 = random.Random(2024)
    tox = [rng2.randrange(1, o.R_MOD) for _ in range(5)]
    opk = o.trapdoor_setup(cons, r1["n_wires"], r1["num_inputs"] - 1, *tox, reduction="libsnark")
    assert bytes(pk.h_query.tobytes()) == b"".join(o.g1_to_bytes(p) for p in opk["h_query"])
    assert bytes(pk.a_query.tobytes()) == b"".join(o.g1_to_bytes(p) for p in opk["a_query"])
    mats = r1cs.matrices()
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    proof = cc.Groth16.create_proof_with_reduction_and_matrices(
        pk, r, s, mats, mats.num_instance_variables, mats.num_constraints, w, lib=lib)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    want = o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), r1["num_inputs"],
                                                      len(cons), w, reduction="libsnark")
    assert proof.raw == o.proof_to_bytes(want)
    pub = w[1:r1["num_inputs"]]
    assert o.verify_proof(opk, pub, H.proof_from_bytes(proof.raw))
    assert not o.verify_proof(opk, [(pub[0] + 1) % o.R_MOD] + pub[1:], H.proof_from_bytes(proof.raw))
    # a circom-reduction proof does not verify under a libsnark key
    bad = cc.Prover(pk, mats, lib=lib, reduction="circom").prove(r, s, w)
    assert not o.verify_proof(opk, pub, H.proof_from_bytes(bad.raw))


def test_error_paths_through_the_abi(lib, golden):
    """errors are status codes + messages, never exceptions across the boundary: wrong witness
    length, unknown reduction, distributed witness map on a non-power-of-two world, partial-only
    entry points on the wrong kind of ctx"""
    import circom_compat_amd as cc
    pk, mats = cc.read_zkey(os.path.join(golden, "test.zkey"), lib)
    pr = cc.Prover(pk, mats, lib=lib)
    with pytest.raises(cc.G16Error) as e:
        pr.prove(1, 2, [1, 33, 3])                      # n_vars is 4
    assert "witness length" in str(e.value)
    with pytest.raises(cc.G16Error):
        pr.prove_finish(1, 2, bytes(2 * 1024))          # world mismatch (ctx world = 1)
    assert pr.exchange_bytes() == 0                     # not a dist_wm ctx
    with pytest.raises(cc.G16Error):
        pr.dist_phase2(0, 0)
    opt_bad = dict(lib=lib, rank=0, world=3, dist_wm=True)
    with pytest.raises(cc.G16Error) as e:
        cc.Prover(pk, mats, **opt_bad)
    assert "power-of-two" in str(e.value)
    with pytest.raises(KeyError):
        cc.Prover(pk, mats, lib=lib, reduction="pinocchio")
    # a proving ctx refuses more scalars than it has points for
    with pytest.raises(cc.G16Error):
        pr.msm_g1(0, [1] * 10)
    # and the prover still works after the failed calls
    assert len(pr.prove(1, 2, [1, 33, 3, 11]).raw) == 256


@pytest.mark.parametrize("corrupt,devices,word", [("0:1:2", [0] * 4, "exchange 0"), ("1:3:0", [0] * 4, "exchange 1"),
                                                    ("2:1:0", [0] * 4, "gather"), ("2:2:0", [0] * 3, "gather"),
                                                    ("0:0:0", [0] * 2, "exchange 0")])
def test_multi_device_create_self_test_catches_a_misrouted_chunk(lib, monkeypatch, corrupt, devices, word):
    """g16_ctx_create_multi runs a 4 KiB-per-pair all-to-all echo and a gather of the partial records
    through the very code a proof's exchanges use (csrc/multi.hip: push_chunks / await_chunks behind
    ev_send on the aux stream, the peer copies behind ev_part), with known patterns.  With one copy
    deliberately reading the wrong source (G16_DEBUG_SELFTEST_CORRUPT = exchange:src:dst, a test-only
    hook) creation must FAIL and name the pair -- a broken peer path between two devices is an error
    at create, never a wrong proof; without the hook the same ctx is created and proves (the test
    above).  Three ranks: no distributed witness map, so only the gather exists and is checked."""
    import circom_compat_amd as cc
    cons, w, n_vars, n_pub = H.squaring_chain(4)
    rng = random.Random(4)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *tox)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, 2, n_vars, lib)
    pk = H.pk_from_oracle(opk)
    x, src, dst = (int(t) for t in corrupt.split(":"))
    monkeypatch.setenv("G16_DEBUG_SELFTEST_CORRUPT", corrupt)
    with pytest.raises(cc.G16Error) as e:
        cc.Prover(pk, mats, lib=lib, devices=devices)
    msg = str(e.value)
    assert "self-test failed" in msg and word in msg and "rank %d " % src in msg and "-> rank %d " % dst in msg, msg
    monkeypatch.delenv("G16_DEBUG_SELFTEST_CORRUPT")
    pr = cc.Prover(pk, mats, lib=lib, devices=devices)      # and the healthy path is created and proves
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    want = o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), 2, len(cons), w)
    assert pr.prove(r, s, w).raw == o.proof_to_bytes(want)
    pr.close()


@pytest.mark.parametrize("shard", ["points", "buckets"])
@pytest.mark.parametrize("logm,devices", [(4, [0, 0]), (5, [0, 0, 0, 0]), (4, [0, 0, 0])])
def test_in_library_multi_device_prover(lib, logm, devices, shard):
    """g16_ctx_create_multi: one ctx, the ranks (MSMs sharded by point range or by bucket range +
    distributed witness map for power-of-two device counts, replicated witness map otherwise) and
    both all-to-all exchanges and the gather live inside the library -- no host framework, no collective library.  Every listed device is ordinal 0
    here (ranks time-sharing one device; bucket-sharded ranks borrow rank 0's planes); the proof must
    equal the oracle's, and two consecutive proofs with different (r, s) cover buffer / event reuse."""
    import circom_compat_amd as cc
    cons, w, n_vars, n_pub = H.squaring_chain(logm)
    rng = random.Random(logm)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *tox)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, 2, n_vars, lib)
    pk = H.pk_from_oracle(opk)
    pr = cc.Prover(pk, mats, lib=lib, devices=devices, shard=shard)
    assert pr.info()["devices"] == len(devices)
    assert pr.info()["shard_mode"] == shard and pr.info()["peer_access"] == 1
    assert pr.info()["shard_w"] == (n_vars - 1 if shard == "buckets" else (n_vars - 1) // len(devices))
    for _ in range(2):
        r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
        want = o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), 2, len(cons), w)
        assert pr.prove(r, s, w).raw == o.proof_to_bytes(want)
    assert o.verify_proof(opk, w[1:2], want)
    # device-resident witness (peer-broadcast from the first device): same bytes
    wm = H.fr_mont_arr(w)
    if lib.path.endswith("libg16_emu.so"):
        ptr, keep = wm.ctypes.data, wm          # the emulator's device memory is host memory
    else:
        import torch
        keep = torch.from_numpy(wm.view(np.int64)).cuda()
        torch.cuda.synchronize()
        ptr = keep.data_ptr()
    assert pr.prove_dev(r, s, ptr).raw == o.proof_to_bytes(want)
    # witness made resident on every device once (g16_witness_upload), then proofs that move nothing
    res = pr.upload_witness(w)
    assert res == pr.witness_buffer()
    assert pr.prove_dev(r, s, res).raw == o.proof_to_bytes(want)
    if not lib.path.endswith("libg16_emu.so") or len(devices) == 2:  # (the emulator steps through every rank)
        assert pr.prove_dev(r, s, res).raw == o.proof_to_bytes(want)
        assert pr.prove(r, s, w).raw == o.proof_to_bytes(want)          # host path again (resets residency)
        assert pr.prove_dev(r, s, ptr).raw == o.proof_to_bytes(want)    # broadcast path again
    # the sharded ctx proves only
    with pytest.raises(cc.G16Error):
        pr.witness_map(w)
    with pytest.raises(cc.G16Error):
        pr.prove(1, 2, w[:-1])
    pr.close()


def test_groth16_prove_surface_and_wire_mapping(lib, golden):
    """SNARK::prove(&pk, circuit, rng) (reference src/zkey.rs:866, tests/groth16.rs:31): r, s from
    the rng, circuit from the builder (wire mapping disabled, builder.rs:84-85) -> the proof
    verifies for get_public_inputs(); same seed -> same (r, s) -> same bytes as the matrices entry."""
    import circom_compat_amd as cc
    pk, mats = cc.read_zkey(os.path.join(golden, "test.zkey"), lib)
    opk, _ = o.read_zkey(open(os.path.join(golden, "test.zkey"), "rb").read())
    builder = cc.CircomBuilder(cc.R1CS.from_file(os.path.join(golden, "mycircuit.r1cs"), lib))
    circuit = builder.build([1, 33, 3, 11])
    proof = cc.Groth16.prove(pk, mats, circuit, random.Random(7), lib=lib)
    pub = circuit.get_public_inputs()
    assert pub == [33]
    assert o.verify_proof(opk, pub, H.proof_from_bytes(proof.raw))
    assert not o.verify_proof(opk, [34], H.proof_from_bytes(proof.raw))
    rng = random.Random(7)
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    same = cc.Groth16.create_proof_with_reduction_and_matrices(pk, r, s, mats, 2, 1, [1, 33, 3, 11], lib=lib)
    assert same.raw == proof.raw
    # a circuit that keeps a (non-identity) wire mapping is proved over w[m[i]] (circuit.rs:35-58),
    # consistently with its own get_public_inputs(): a label-ordered witness through the mapping
    # gives the same assignment, hence the same bytes; the wire-order witness through it does not
    import copy
    r1 = copy.copy(builder.r1cs)
    r1.wire_mapping = [0, 3, 1, 2]
    mapped = cc.CircomCircuit(r1, [1, 3, 11, 33])          # w'[m[i]] = w[i]
    assert mapped.full_assignment() == [1, 33, 3, 11] and mapped.get_public_inputs() == [33]
    assert mapped.first_unsatisfied(lib) == -1
    assert cc.Groth16.prove(pk, mats, mapped, random.Random(7), lib=lib).raw == proof.raw
    wrong = cc.CircomCircuit(r1, [1, 33, 3, 11])
    assert wrong.first_unsatisfied(lib) == 0
    assert cc.Groth16.prove(pk, mats, wrong, random.Random(7), lib=lib).raw != proof.raw


def test_domain_too_large_and_bad_matrices(lib):
    """SynthesisError::PolynomialDegreeTooLarge (reference src/circom/qap.rs:31,66: both the size-n
    and the size-2n domain must exist, Fr two-adicity 28) surfaces as G16_ERR_DOMAIN_TOO_LARGE;
    out-of-range wire indices / inconsistent row pointers are rejected at create time (the
    reference panics on the out-of-bounds index)."""
    import circom_compat_amd as cc
    from circom_compat_amd import _binding as B
    one = cc.fr_from_ints([1], lib)
    m = (1 << 28) - 1                                   # m + num_inputs > 2^27: no 2n domain
    # zero-copy all-zero row pointers: the domain check fires before anything is uploaded
    rp = np.zeros(m + 1, dtype=np.uint32)
    empty = cc.Csr.__new__(cc.Csr)
    empty.row_ptr, empty.col, empty.coeff = rp, np.zeros(0, np.uint32), np.zeros((0, 4), np.uint64)
    mats = cc.ConstraintMatrices(2, 3, m, empty, empty)
    with pytest.raises(cc.SynthesisError) as e:
        cc.Prover(None, mats, lib=lib, n_vars=4)
    assert e.value.status == B.G16_ERR_DOMAIN_TOO_LARGE
    # wire index beyond the witness
    bad = cc.Csr([0, 1], [7], one)
    ok = cc.Csr([0, 1], [1], one)
    with pytest.raises(cc.G16Error) as e:
        cc.Prover(None, cc.ConstraintMatrices(2, 2, 1, bad, ok), lib=lib, n_vars=4)
    assert "wire index" in str(e.value)
    # row_ptr that does not end at nnz
    broken = cc.Csr.__new__(cc.Csr)
    broken.row_ptr, broken.col, broken.coeff = np.array([0, 2], np.uint32), np.array([1], np.uint32), one
    with pytest.raises(cc.G16Error) as e:
        cc.Prover(None, cc.ConstraintMatrices(2, 2, 1, broken, ok), lib=lib, n_vars=4)
    assert "row_ptr" in str(e.value)
    with pytest.raises(cc.G16Error):
        cc.Prover(None, cc.ConstraintMatrices(2, 2, 1, ok, ok), lib=lib)     # n_vars is required


@pytest.mark.parametrize("shard", ["points", "buckets"])
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0, 0, 0, 0]])
def test_multi_device_prover_on_the_reference_zkey(lib, golden, devices, shard):
    """test.zkey (4 wires, domain 4) through read_zkey -- zero-copy, UNALIGNED views of the file's point
    sections -- on 2 / 3 / 4 ranks: shards of one or zero points, the smallest four-step split
    (n1 = n2 = 2), bytes == oracle (SURVEY C.1 inputs)."""
    import circom_compat_amd as cc
    data = open(os.path.join(golden, "test.zkey"), "rb").read()
    pk, mats = cc.read_zkey(data, lib)
    opk, omats = o.read_zkey(data)
    w = [1, 33, 3, 11]
    r = 3413513218498352040262653353725127729454431939539290118844322056224532443637
    s = 6077776500692565155461894309070795882353485867345896979329447163197530625403
    want = o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, r, s, omats, 2, 1, w))
    pr = cc.Prover(pk, mats, lib=lib, devices=devices, shard=shard)
    assert pr.prove(r, s, w).raw == want
    assert pr.prove(0, 0, w).raw == o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, 0, 0, omats, 2, 1, w))
    pr.close()


@pytest.mark.parametrize("n_pub,m", [(0, 5), (0, 1), (2, 1)])
def test_no_public_inputs_and_single_constraint(lib, n_pub, m):
    """edge shapes of the key: n_public = 0 (IC has one point, every wire but the constant has an L
    point) and a single constraint (domain 2 / 4), on one device and on two ranks: bytes == oracle"""
    import circom_compat_amd as cc
    P = o.R_MOD
    base = 1 + n_pub
    n_vars = base + m + 1
    cons = [([(base + i, 1)], [(base + i, 1)], [(base + i + 1, 1)]) for i in range(m)]
    w = [1] + [7] * n_pub + [3]
    for _ in range(m):
        w.append(w[-1] * w[-1] % P)
    rng = random.Random(1)
    tox = [rng.randrange(1, P) for _ in range(5)]
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *tox)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, n_pub + 1, n_vars, lib)
    pk = H.pk_from_oracle(opk)
    r, s = rng.randrange(P), rng.randrange(P)
    want = o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), n_pub + 1, m, w)
    assert o.verify_proof(opk, w[1:1 + n_pub], want)
    for devices, shard in ((None, "auto"), ([0, 0], "points"), ([0, 0], "buckets")):
        pr = cc.Prover(pk, mats, lib=lib, devices=devices, shard=shard)
        assert pr.prove(r, s, w).raw == o.proof_to_bytes(want)
        pr.close()


@pytest.mark.parametrize("logm,world,wb,planes,kind", [(5, 4, 13, 0, "dense"), (6, 8, 14, 3, "chain"),
                                                     (4, 3, 12, 0, "dense"), (5, 2, 0, 1, "chain")])
def test_bucket_range_sharding_windows_planes_and_skew(lib, logm, world, wb, planes, kind):
    """MSMs sharded by BUCKET range (g16_options.shard = G16_SHARD_BUCKETS): every rank sorts all
    scalars but keeps its run of sort partitions, accumulates and reduces that run only.  Covered:
    windows large enough for multi-bucket partitions (c = 13, 14: the reduction's chunk grid anchored
    at the run start), fewer planes than windows (several bucket sets, D = 7, D = W: a run spans
    sets), a skewed 0/1-heavy witness (one partition holds most entries: a rank may own one partition
    or none), world = 3 (replicated witness map), 8 ranks on 64 points.  bytes == oracle, two
    proofs."""
    import circom_compat_amd as cc
    if lib.path.endswith("libg16_emu.so") and wb >= 14 and world >= 8:
        world = 4                                    # the emulator steps through every rank's 2^13-bucket sets
    if kind == "dense":
        cons, w, n_vars, _ = H.dense_skewed_circuit((1 << logm) - 5, seed=3, long_rows=(7,))
        n_pub = 1
    else:
        cons, w, n_vars, n_pub = H.squaring_chain(logm)
    rng = random.Random(100 * logm + world)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *tox)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, n_pub + 1, n_vars, lib)
    pr = cc.Prover(H.pk_from_oracle(opk), mats, lib=lib, devices=[0] * world, shard="buckets",
                   window_bits=wb, planes=planes)
    info = pr.info()
    assert info["shard_mode"] == "buckets" and info["shard_w"] == n_vars - 1
    assert info["shard_h"] == info["domain_size"] // world      # H stays cut by point range
    if wb:
        assert info["c_w"] == wb
    for _ in range(2):
        r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
        want = o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), n_pub + 1, len(cons), w)
        assert pr.prove(r, s, w).raw == o.proof_to_bytes(want)
    pr.close()


def test_poseidon_shaped_generator_full_width_rows_vs_oracle(lib):
    """bench.poseidon_shaped_circuit (the config-5 substitute with full-width coefficients on both sides of
    the product rows) at 2^7 rows: satisfiable, and the proof over a trapdoor key == the Python oracle's
    -- on one device and on 4 bucket-sharded ranks."""
    import sys
    import circom_compat_amd as cc
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from circom_compat_amd import _binding
    saved, _binding._default = _binding._default, lib       # the generator converts through the default library
    try:
        mats, (A, B, Cm), w, n_vars = bench.poseidon_shaped_circuit(cc, 7)
    finally:
        _binding._default = saved
    circ = cc.CircomCircuit(type("R", (), dict(a=A, b=B, c=Cm, num_constraints=mats.num_constraints,
                                               wire_mapping=None, num_inputs=2, num_variables=n_vars)), w)
    assert circ.first_unsatisfied(lib) == -1

    def rows(m):
        cf = cc.fr_to_ints(m.coeff, lib)
        return [[(int(m.col[j]), cf[j]) for j in range(m.row_ptr[i], m.row_ptr[i + 1])] for i in range(m.num_rows)]
    cons = list(zip(rows(A), rows(B), rows(Cm)))
    assert max(len(a) for a, _b, _c in cons) == 4 and max(len(b) for _a, b, _c in cons) == 4
    rng = random.Random(77)
    opk = o.trapdoor_setup(cons, n_vars, 1, *[rng.randrange(1, o.R_MOD) for _ in range(5)])
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    want = o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), 2,
                                                                       len(cons), w))
    pk = H.pk_from_oracle(opk)
    assert cc.Prover(pk, mats, lib=lib).prove(r, s, w).raw == want
    pr = cc.Prover(pk, mats, lib=lib, devices=[0] * 4, shard="buckets")
    assert pr.prove(r, s, w).raw == want
    pr.close()


def test_poseidon_chain_one_hash_public_output_is_the_circomlibjs_kat(lib):
    """bench.poseidon_chain_circuit (BASELINE configs[4]: a REAL Poseidon(2) instance with circomlib's
    Grain-LFSR parameters) at one hash = 240 rows (circomlib's constraint count for Poseidon(2): the
    capacity lane's first S-box is a constant and folds away): the circuit is satisfiable, its PUBLIC output is
    circomlibjs' known answer poseidon([1, 2]) = 0x115cc0f5...189a, rows carry up to 61 full-width terms,
    and the proof over a trapdoor key == the Python oracle's -- on one device and on 4 bucket-sharded
    ranks; the oracle's pairing check accepts it for the KAT and rejects it for KAT + 1."""
    import sys
    import circom_compat_amd as cc
    import poseidon_ref
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from circom_compat_amd import _binding
    saved, _binding._default = _binding._default, lib       # the generator converts through the default library
    try:
        mats, (A, B, Cm), w, n_vars = bench.poseidon_chain_circuit(cc, 8)
    finally:
        _binding._default = saved
    kat = 0x115cc0f5e7d690413df64c6b9662e9cf2a3617f2743245519e19607a4417189a
    assert mats.num_constraints == 240 and w[1] == kat == poseidon_ref.poseidon([1, 2])
    circ = cc.CircomCircuit(type("R", (), dict(a=A, b=B, c=Cm, num_constraints=mats.num_constraints,
                                               wire_mapping=None, num_inputs=2, num_variables=n_vars)), w)
    assert circ.first_unsatisfied(lib) == -1

    def rows(m):
        cf = cc.fr_to_ints(m.coeff, lib)
        return [[(int(m.col[j]), cf[j]) for j in range(m.row_ptr[i], m.row_ptr[i + 1])] for i in range(m.num_rows)]
    cons = list(zip(rows(A), rows(B), rows(Cm)))
    assert max(len(a) for a, _b, _c in cons) == 61 and len(cons[-1][2]) == 3
    rng = random.Random(78)
    opk = o.trapdoor_setup(cons, n_vars, 1, *[rng.randrange(1, o.R_MOD) for _ in range(5)])
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    want = o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), 2,
                                                                       len(cons), w))
    pk = H.pk_from_oracle(opk)
    proof = cc.Prover(pk, mats, lib=lib).prove(r, s, w)
    assert proof.raw == want
    pr = cc.Prover(pk, mats, lib=lib, devices=[0] * 4, shard="buckets")
    assert pr.prove(r, s, w).raw == want
    pr.close()
    assert o.verify_proof(opk, [kat], H.proof_from_bytes(proof.raw))
    assert not o.verify_proof(opk, [kat + 1], H.proof_from_bytes(proof.raw))


def test_donor_destroyed_before_its_sibling_keeps_the_planes_alive(lib):
    """g16_ctx_destroy on a ctx that still lends its point planes retires the handle only: the sibling
    keeps proving (oracle bytes) and frees the donor's state with its own destroy (round 5; rounds 3-4
    warned and freed under the borrower: dangling plane pointers).  A second sibling created and closed
    in between leaves the count right.  Under scripts/asan_emu.sh a use-after-free is a report."""
    import gc
    import circom_compat_amd as cc
    cons, w, n_vars, n_pub = H.squaring_chain(6)
    rng = random.Random(56)
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *[rng.randrange(1, o.R_MOD) for _ in range(5)])
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, 2, n_vars, lib)
    pk = H.pk_from_oracle(opk)
    donor = cc.Prover(pk, mats, lib=lib)
    sib = cc.Prover(pk, mats, lib=lib, sibling_of=donor)
    sib2 = cc.Prover(pk, mats, lib=lib, sibling_of=donor)
    sib2.close()
    sib._donor = None                      # drop the Python-side keep-alive: the C library has to cope
    donor.close()
    del donor
    gc.collect()
    for _ in range(2):
        r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
        want = o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), 2,
                                                                           len(cons), w))
        assert sib.prove(r, s, w).raw == want
    sib.close()                            # frees the sibling, then the retired donor
    # a fresh donor / sibling pair after all that still works (nothing was left half-freed)
    d2 = cc.Prover(pk, mats, lib=lib)
    s2 = cc.Prover(pk, mats, lib=lib, sibling_of=d2)
    assert s2.prove(r, s, w).raw == want and d2.prove(r, s, w).raw == want
    s2.close()
    d2.close()


def test_fixed_base_tables_prover_test_zkey(lib, golden):
    """Small keys prove through fixed-base tables (g16_options.fixed_tables, csrc/msm_table.hip): every
    MSM of create_proof_with_assignment as table lookups + a tree sum, s*A and r*B1 as two more table
    MSMs.  On the reference's own zkey: proof bytes == the oracle's for the KAT (r, s), for r = s = 0
    (no blinding: the fixed-base sums are all empty) and r != 0 = s; `tables=0` (automatic) selects the
    path for a key this small, `tables=-1` never does and gives the same bytes; a sibling borrows the
    tables, and survives its donor."""
    import circom_compat_amd as cc
    pk, mats = cc.read_zkey(os.path.join(golden, "test.zkey"), lib=lib)
    opk, omats = o.read_zkey(open(os.path.join(golden, "test.zkey"), "rb").read())
    w = [1, 33, 3, 11]
    pr = cc.Prover(pk, mats, lib=lib, tables=1)
    assert pr.info()["fixed_tables"] == 1
    auto = cc.Prover(pk, mats, lib=lib, tables=0)
    assert auto.info()["fixed_tables"] == 1
    never = cc.Prover(pk, mats, lib=lib, tables=-1)
    assert never.info()["fixed_tables"] == 0
    sib = cc.Prover(pk, mats, lib=lib, sibling_of=pr, tables=0)
    assert sib.info()["fixed_tables"] == 1
    rs = [(3413513218498352040262653353725127729454431939539290118844322056224532443637,
           6077776500692565155461894309070795882353485867345896979329447163197530625403), (0, 0), (5, 0),
          (o.R_MOD - 1, o.R_MOD - 1)]
    for r, s in rs:
        want = o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, r, s, omats, 2, 1, w))
        assert pr.prove(r, s, w).raw == want
        assert never.prove(r, s, w).raw == want
        assert sib.prove(r, s, w).raw == want
    assert auto.prove(*rs[0], w).raw == pr.prove(*rs[0], w).raw
    sib._donor = None
    pr.close()
    want = o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, 7, 9, omats, 2, 1, w))
    assert sib.prove(7, 9, w).raw == want
    # a wrong witness length is still an error on this path, and the proof of a wrong witness does not verify
    with pytest.raises(cc.G16Error):
        auto.prove(1, 2, w + [5])
    bad = auto.prove(1, 2, [1, 34, 3, 11])
    assert not o.verify_proof(opk, [34], H.proof_from_bytes(bad.raw))
    assert o.verify_proof(opk, [33], H.proof_from_bytes(auto.prove(1, 2, w).raw))


@pytest.mark.parametrize("case", ["chain", "public-inputs", "infinity-points", "libsnark"])
def test_fixed_base_tables_prover_circuits(lib, golden, case):
    """The table path on circuits with more than one row: a squaring chain (multi-element NTT domain),
    several public inputs (the L query starts at wire p + 1), a dense circuit whose witness is full of
    0 / 1 scalars and whose B queries hold points at infinity (table rows of zeros), and
    LibsnarkReduction on the reference's mycircuit.r1cs (H query padded with infinity): bytes == the
    Python oracle's, scalars 0, 1, r - 1 and full-width among them."""
    import circom_compat_amd as cc
    if case in ("public-inputs", "infinity-points") and lib.path.endswith("libg16_emu.so"):
        pytest.skip("GPU suite only: 45 s each of table construction on the emulator (the CPU suite keeps two cases)")
    rng = random.Random(len(case) * 7919)
    red = "circom"
    if case == "chain":
        cons, w, n_vars, n_pub = H.squaring_chain(3)
    elif case == "public-inputs":
        cons, w, n_vars, _ = H.dense_skewed_circuit(7, seed=3, long_rows=())
        n_pub = 3
    elif case == "infinity-points":
        cons, w, n_vars, n_pub = H.dense_skewed_circuit(10, seed=4, long_rows=(5,))
        w = list(w)
    else:
        r1, w = _fixture(golden, "mycircuit")
        cons, n_vars, n_pub, red = r1["constraints"], r1["n_wires"], r1["num_inputs"] - 1, "libsnark"
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *tox, reduction=red)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, n_pub + 1, n_vars, lib)
    pk = H.pk_from_oracle(opk)
    pr = cc.Prover(pk, mats, lib=lib, tables=1, reduction=red)
    assert pr.info()["fixed_tables"] == 1
    for r, s in ((rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)), (1, o.R_MOD - 1)):
        want = o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), n_pub + 1, len(cons), w,
                                                          reduction=red)
        assert pr.prove(r, s, w).raw == o.proof_to_bytes(want)
    pr.close()


def test_sibling_ctx_shares_planes_two_proofs_in_flight(lib, golden):
    """g16_ctx_create_sibling: a second prover over the same key that borrows the donor's point planes.
    Both give the oracle's bytes -- alternately and from two host threads at once (the throughput mode
    of bench.py); closing the sibling leaves the donor working; a multi-device ctx cannot lend."""
    import threading
    import circom_compat_amd as cc
    cons, w, n_vars, n_pub = H.squaring_chain(5)
    rng = random.Random(55)
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *[rng.randrange(1, o.R_MOD) for _ in range(5)])
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, 2, n_vars, lib)
    pk = H.pk_from_oracle(opk)
    donor = cc.Prover(pk, mats, lib=lib)
    sib = cc.Prover(pk, mats, lib=lib, sibling_of=donor)
    rs = [(rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)) for _ in range(2)]
    want = [o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), 2,
                                                                        len(cons), w)) for r, s in rs]
    assert donor.prove(*rs[0], w).raw == want[0] and sib.prove(*rs[1], w).raw == want[1]
    assert sib.prove(*rs[0], w).raw == want[0] and donor.prove(*rs[1], w).raw == want[1]
    if not lib.path.endswith("libg16_emu.so"):            # the emulator is single-threaded by construction
        got = [None, None]

        def run(i, p):
            for _ in range(3):
                got[i] = p.prove(*rs[i], w).raw
        ths = [threading.Thread(target=run, args=(i, p)) for i, p in enumerate((donor, sib))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert got == want
    # what a sibling may not do: ask for another window / plane count than the donor's,
    # borrow from a borrower, bring other matrices of the same sizes
    with pytest.raises(cc.G16Error):
        cc.Prover(pk, mats, lib=lib, sibling_of=donor, window_bits=donor.info()["c_w"] + 1)
    with pytest.raises(cc.G16Error):
        cc.Prover(pk, mats, lib=lib, sibling_of=donor, planes=1)
    with pytest.raises(cc.G16Error):
        cc.Prover(pk, mats, lib=lib, sibling_of=sib)
    other = H.matrices_from_rows(a_rows[:-1] + [a_rows[-1] + a_rows[-1]], b_rows, 2, n_vars, lib)
    with pytest.raises(cc.G16Error):
        cc.Prover(pk, other, lib=lib, sibling_of=donor)
    sib.close()
    assert donor.prove(*rs[0], w).raw == want[0]
    multi = cc.Prover(pk, mats, lib=lib, devices=[0, 0])
    with pytest.raises(cc.G16Error):
        cc.Prover(pk, mats, lib=lib, sibling_of=multi)
    multi.close()
    donor.close()


@pytest.mark.parametrize("n_rows", [10, 29])
def test_more_wires_than_the_domain(lib, n_rows):
    """A circuit whose wire count exceeds its evaluation domain (every row brings four fresh wires:
    n_vars = 4 m + 2 > 2^ceil(log2(m + 2))): the A / B1 / B2 / L queries are longer than the H query and
    than the NTT size -- what the real Poseidon chain of configs[4] has at 2^20 (1 052 931 wires, domain
    2^20).  Proof bytes == the Python oracle's on one device and on 4 ranks, both cuts."""
    import circom_compat_amd as cc
    rng = random.Random(1000 + n_rows)
    P = o.R_MOD
    w = [1, 0]
    cons = []
    for i in range(n_rows):
        a, b, c = (rng.randrange(P) for _ in range(3))
        base = len(w)
        w.extend([a, b, c, (a + 5 * b) * c % P])
        cons.append(([(base, 1), (base + 1, 5)], [(base + 2, 1)], [(base + 3, 1)]))
    cons.append(([(len(w) - 1, 1)], [(0, 1)], [(1, 1)]))
    w[1] = w[-1]
    n_vars = len(w)
    dom = 1 << (len(cons) + 2 - 1).bit_length()
    assert n_vars > 2 * dom
    opk = o.trapdoor_setup(cons, n_vars, 1, *[rng.randrange(1, P) for _ in range(5)])
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, 2, n_vars, lib)
    pk = H.pk_from_oracle(opk)
    r, s = rng.randrange(P), rng.randrange(P)
    want = o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), 2,
                                                                       len(cons), w))
    assert cc.Prover(pk, mats, lib=lib).prove(r, s, w).raw == want
    for shard in ("points", "buckets"):
        pr = cc.Prover(pk, mats, lib=lib, devices=[0] * 4, shard=shard)
        assert pr.prove(r, s, w).raw == want
        pr.close()


@pytest.mark.parametrize("wb", [0, 6])
def test_sparse_b_queries_use_a_filtered_view_of_the_witness_sort(lib, monkeypatch, wb):
    """Real circom keys hold the point at infinity in b_g1_query / b_g2_query for every wire that appears
    in no B row.  With G16_SPARSE_B=1 (the automatic rule needs >= 2^15 wires: GPU suite) the B2 (G2) MSM
    accumulates and reduces over MsmSort::run_view -- level 2 of the witness sort re-run without those
    points; B1 stays in the A | B1 pair launch.  A circuit whose B rows touch a quarter of the wires (and a
    0/1-heavy witness: hot buckets in both views): proof bytes == the oracle's == the unfiltered path's,
    with the default window and with c = 6 (several buckets per sort partition); a sibling inherits the view."""
    import circom_compat_amd as cc
    rng = random.Random(4242 + wb)
    P = o.R_MOD
    w = [1, 0]
    cons = []
    for i in range(60):
        a, b = rng.randrange(P), rng.choice([0, 1, 1, rng.randrange(P)])
        base = len(w)
        w.extend([a, rng.randrange(2), rng.randrange(P), 0])
        if i % 4 == 0:       # one row in four has its own B wire; the others multiply by wire `base + 1` of row 0 or by one
            w[base + 1] = b
            cons.append(([(base, 1), (base + 2, 3)], [(base + 1, 1)], [(base + 3, 1)]))
            w[base + 3] = (a + 3 * w[base + 2]) * b % P
        else:
            cons.append(([(base, 1), (base + 1, 2)], [(0, 1)], [(base + 3, 1)]))
            w[base + 3] = (a + 2 * w[base + 1]) % P
    cons.append(([(len(w) - 1, 1)], [(0, 1)], [(1, 1)]))
    w[1] = w[-1]
    n_vars = len(w)
    opk = o.trapdoor_setup(cons, n_vars, 1, *[rng.randrange(1, P) for _ in range(5)])
    n_inf = sum(1 for p in opk["b_g1_query"][1:] if p is None)
    assert n_inf * 8 >= n_vars and all((p is None) == (q is None) for p, q in zip(opk["b_g1_query"], opk["b_g2_query"]))
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, 2, n_vars, lib)
    pk = H.pk_from_oracle(opk)
    plain = cc.Prover(pk, mats, lib=lib, window_bits=wb)
    assert plain.info()["sparse_b"] == 0
    monkeypatch.setenv("G16_SPARSE_B", "1")
    view = cc.Prover(pk, mats, lib=lib, window_bits=wb)
    assert view.info()["sparse_b"] == 1
    sib = cc.Prover(pk, mats, lib=lib, sibling_of=view, window_bits=wb)
    assert sib.info()["sparse_b"] == 1
    monkeypatch.delenv("G16_SPARSE_B")
    for _ in range(2):
        r, s = rng.randrange(P), rng.randrange(P)
        want = o.proof_to_bytes(o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), 2,
                                                                           len(cons), w))
        assert view.prove(r, s, w).raw == want
        assert plain.prove(r, s, w).raw == want
        assert sib.prove(r, s, w).raw == want


def test_multi_device_ctx_reports_its_link_probe(lib):
    """g16_multi_links: the probe (run by the first call, not at create) of every ordered (source,
    destination) pair of distinct ranks of a multi-device ctx with a distributed witness map -- one table
    entry per pair (the diagonal reads 0), the probed copy size, finite non-negative figures (on the GPU:
    positive off the diagonal; the emulator has no clock); a second call returns the same table; a ctx
    without exchange buffers (3 ranks: replicated witness map) reports zeros; a single-device ctx is an
    error."""
    import circom_compat_amd as cc
    cons, w, n_vars, n_pub = H.squaring_chain(8)      # exchange buffers of 13.5 KiB: the probe needs >= 4 KiB
    rng = random.Random(57)
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *[rng.randrange(1, o.R_MOD) for _ in range(5)])
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, 2, n_vars, lib)
    pk = H.pk_from_oracle(opk)
    pr = cc.Prover(pk, mats, lib=lib, devices=[0, 0])
    lk = pr.links()
    assert len(lk["gbps"]) == 2 and all(len(r) == 2 for r in lk["gbps"]) and lk["probe_bytes"] >= 4096
    flat = [x for r in lk["gbps"] for x in r] + [x for r in lk["echo_us"] for x in r]
    assert all(x >= 0.0 and x == x and x < 1e9 for x in flat)
    if not lib.path.endswith("libg16_emu.so"):
        for t in (lk["gbps"], lk["echo_us"]):
            assert all((t[a][b] > 0.0) == (a != b) for a in range(2) for b in range(2)), t
    assert pr.links() == lk
    pr.close()
    p3 = cc.Prover(pk, mats, lib=lib, devices=[0, 0, 0])
    assert all(x == 0.0 for r in p3.links()["gbps"] for x in r)
    p3.close()
    one = cc.Prover(pk, mats, lib=lib)
    with pytest.raises(cc.G16Error):
        one.links()
