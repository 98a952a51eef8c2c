"""GPU-only parity at sizes the emulator cannot reach: multi-pass NTTs, big MSMs with closed-form
expectations, skewed (0/1-heavy) witnesses, size-independent properties."""
import os
import random
import sys

import numpy as np
import pytest

import bn254_ref as o
import helpers as H

pytestmark = pytest.mark.gpu

# The driver gives the whole `-m gpu` suite 1200 s; round 5's took 665 s, 280 s of it in the two legs below
# (2^24 on one GPU: 127 s; 2^25 capacity point: 155 s -- mostly the CPU restatement proving beside them).
# They are opt-in (G16_TEST_LARGE=1, run by scripts/r6/final_a.sh and recorded under profiles/): the same sizes
# are covered by the bench lines profiles/r0N_bench_k24_single_gpu.json / _chain25.json, which byte-compare too.
LARGE = bool(os.environ.get("G16_TEST_LARGE"))


def _ntt(lib, arr, k, inverse, algo):
    a = arr.copy()
    lib.check(lib.g16_fft_in_place(0, a.ctypes.data, k, 1 if inverse else 0, algo))
    return a


@pytest.mark.parametrize("k", [13, 16])
def test_ntt_vs_oracle_large(gpulib, k):
    rng = random.Random(k)
    x = H.rand_fr(rng, 1 << k)
    arr = H.fr_mont_arr(x)
    assert H.fr_from_mont_arr(_ntt(gpulib, arr, k, False, 0)) == o.ntt(x)
    assert H.fr_from_mont_arr(_ntt(gpulib, arr, k, True, 0)) == o.ntt(x, inverse=True)
    assert H.fr_from_mont_arr(_ntt(gpulib, arr, k, False, 1)) == o.ntt(x)
    assert H.fr_from_mont_arr(_ntt(gpulib, arr, k, False, 2)) == o.ntt(x)
    assert H.fr_from_mont_arr(_ntt(gpulib, arr, k, True, 2)) == o.ntt(x, inverse=True)
    assert H.fr_from_mont_arr(_ntt(gpulib, arr, k, False, 3)) == o.ntt(x)


@pytest.mark.parametrize("k", [20, 22])
def test_ntt_properties_full_size(gpulib, k):
    """size-independent properties at BASELINE sizes: DIF and DIT kernels agree, inverse(forward) = id,
    and X[0] = sum x (checked on the host with numpy big-int free arithmetic on a sparse input)."""
    n = 1 << k
    rng = np.random.default_rng(k)
    arr = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    arr[:, 3] &= np.uint64((1 << 60) - 1)          # < modulus, arbitrary Montgomery residues
    f0 = _ntt(gpulib, arr, k, False, 0)
    f1 = _ntt(gpulib, arr, k, False, 1)
    assert np.array_equal(f0, f1)
    assert np.array_equal(f0, _ntt(gpulib, arr, k, False, 2))     # lazy-limb DIF
    assert np.array_equal(f0, _ntt(gpulib, arr, k, False, 3))     # lazy-limb DIT
    assert np.array_equal(arr, _ntt(gpulib, f0, k, True, 2))
    back = _ntt(gpulib, f0, k, True, 0)
    assert np.array_equal(back, arr)
    # delta at position j -> X[i] = omega^(i*j): check a few entries against the oracle
    j = 12345 % n
    d = np.zeros((n, 4), dtype=np.uint64)
    d[j] = H.fr_mont_arr([1])[0]
    X = _ntt(gpulib, d, k, False, 0)
    w = o.root_of_unity(n)
    for i in (0, 1, 2, n // 2 + 7, n - 1):
        assert H.fr_from_mont_arr(X[i:i + 1])[0] == pow(w, i * j, o.R_MOD)


def _cycled_key(rng, N, dom, K=64):
    base1 = [o.G1.mul(o.G1_GEN, rng.randrange(1, o.R_MOD)) for _ in range(K)]
    base2 = [o.G2.mul(o.G2_GEN, rng.randrange(1, o.R_MOD)) for _ in range(K)]
    b1 = H.g1_arr(base1)
    b2 = H.g2_arr(base2)
    idx = np.arange(N) % K
    import circom_compat_amd as cc
    vk = cc.VerifyingKey(o.g1_to_bytes(base1[0]), o.g2_to_bytes(base2[0]), o.g2_to_bytes(base2[1]),
                         o.g2_to_bytes(base2[2]), b1[:2].copy())
    pk = cc.ProvingKey(N, 1, dom, vk, o.g1_to_bytes(base1[1]), o.g1_to_bytes(base1[2]),
                       b1[idx].copy(), b1[(idx + 1) % K].copy(), b2[idx].copy(),
                       b1[(np.arange(N - 2) + 5) % K].copy(), b1[(np.arange(dom) * 3) % K].copy())
    return pk, base1, base2, K


def _closed_form(C, base, K, scal, shift):
    sums = [0] * K
    for i, s in enumerate(scal):
        sums[(i + shift) % K] = (sums[(i + shift) % K] + s) % o.R_MOD
    return C.sum([C.mul(base[j], sums[j]) for j in range(K)])


@pytest.mark.parametrize("logn,skew", [(14, False), (16, True), (18, False)])
def test_msm_closed_form_large(gpulib, logn, skew):
    """points cycle through K distinct bases, so sum s_i P_i has an O(n) scalar-side closed form"""
    import circom_compat_amd as cc
    rng = random.Random(logn)
    n = (1 << logn) - 3
    N = n + 1
    dom = 1 << logn
    pk, base1, base2, K = _cycled_key(rng, N, dom)
    mats = H.matrices_from_rows([[(1, 1)]] * (dom - 2), [[(1, 0)]] * (dom - 2), 2, N, gpulib)
    pr = cc.Prover(pk, mats, lib=gpulib)
    if skew:   # circom-like witness: mostly bits and small values
        scal = [rng.choice((0, 1, 1, 1, 2, 255, rng.randrange(o.R_MOD))) for _ in range(n)]
    else:
        scal = H.rand_fr(rng, n)
    sm = H.fr_mont_arr(scal)
    assert pr.msm_g1(0, sm) == o.g1_to_bytes(_closed_form(o.G1, base1, K, scal, 1))
    assert pr.msm_g1(1, sm) == o.g1_to_bytes(_closed_form(o.G1, base1, K, scal, 2))
    assert pr.msm_g2(sm) == o.g2_to_bytes(_closed_form(o.G2, base2, K, scal, 1))
    hs = H.rand_fr(rng, dom)
    want = [0] * K
    for i, s in enumerate(hs):
        want[(i * 3) % K] = (want[(i * 3) % K] + s) % o.R_MOD
    assert pr.msm_g1(3, H.fr_mont_arr(hs)) == o.g1_to_bytes(o.G1.sum([o.G1.mul(base1[j], want[j]) for j in range(K)]))


def test_witness_map_vs_oracle_2p14(gpulib):
    """squaring chain with m = 2^14 - 2 (three NTT passes): h element-for-element vs the oracle"""
    import circom_compat_amd as cc
    cons, w, n_vars, _ = H.squaring_chain(14)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, 2, n_vars, gpulib)
    h = cc.CircomReduction.witness_map_from_matrices(mats, 2, len(cons), w, lib=gpulib)
    want = o.witness_map_from_matrices(a_rows, b_rows, 2, len(cons), w)
    assert H.fr_from_mont_arr(h) == want


def test_determinism(gpulib, golden):
    """same inputs twice -> identical bytes (atomics only permute bucket order; group sums are exact)"""
    import os
    import circom_compat_amd as cc
    rng = random.Random(9)
    N = 5000
    pk, *_ = _cycled_key(rng, N, 8192)
    mats = H.matrices_from_rows([[(1, 1)]] * 8190, [[(1, 0)]] * 8190, 2, N, gpulib)
    pr = cc.Prover(pk, mats, lib=gpulib)
    w = H.fr_mont_arr([1] + H.rand_fr(rng, N - 1))
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    assert pr.prove(r, s, w).raw == pr.prove(r, s, w).raw


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["points", "buckets"])
@pytest.mark.parametrize("logm,world", [(10, 2), (14, 4), (16, 8)])
def test_fully_sharded_prover_emulated_ranks_one_gpu(gpulib, logm, world, shard):
    """The fully sharded prover through the per-process API (distributed witness map + MSMs sharded by
    point range or -- the witness-scalar queries -- by bucket range) with all `world` ranks living on
    ONE GPU: the two all-to-all exchanges are done by hand on device tensors.  The proof must be
    bit-identical to the CPU restatement's proof of the same (pk, r, s, w)."""
    import torch
    import circom_compat_amd as cc
    import cpu_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, logm)
    rng = random.Random(logm)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
    rs = cc.fr_from_ints([rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)])
    r, s = rs[0], rs[1]
    w = cc.fr_from_ints(w_ints)
    want = cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
    w_dev = torch.from_numpy(w.view(np.int64)).cuda()
    provers = [cc.Prover(pk, mats, rank=g, world=world, dist_wm=True, shard=shard) for g in range(world)]
    assert all(p.info()["shard_mode"] == shard for p in provers)
    nbytes = provers[0].exchange_bytes()
    send = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(world)]
    recv = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(world)]

    def all_to_all():
        chunk = nbytes // world
        for dst in range(world):
            for src in range(world):
                recv[dst][src * chunk:(src + 1) * chunk] = send[src][dst * chunk:(dst + 1) * chunk]
        torch.cuda.synchronize()

    for g, p in enumerate(provers):
        p.dist_phase1(r, s, w_dev.data_ptr(), send[g].data_ptr())
    all_to_all()
    for g, p in enumerate(provers):
        p.dist_phase2(recv[g].data_ptr(), send[g].data_ptr())
    all_to_all()
    parts = b"".join(p.dist_phase3(recv[g].data_ptr()) for g, p in enumerate(provers))
    proof = provers[0].prove_finish(r, s, parts)
    assert proof.raw == want
    assert o.verify_proof(_vk_dict(pk), [w_ints[1]], H.proof_from_bytes(proof.raw))


@pytest.mark.gpu
def test_dense_skewed_circuit_2p14_vs_cpu_restatement(gpulib, tmp_path):
    """config-5 substitute at 2^14 rows: uneven rows, hot buckets from a 0/1-heavy witness; key minted
    on the GPU, written and re-read through the zkey path; GPU proof == C restatement's proof."""
    import circom_compat_amd as cc
    import cpu_ref
    cons, w, n_vars, n_pub = H.dense_skewed_circuit((1 << 14) - 3, seed=14, n_inputs=64, long_rows=(100, 9000))
    frac01 = sum(1 for x in w if x in (0, 1)) / len(w)
    assert frac01 >= 0.5
    rows = lambda k: [[(c, wdx) for wdx, c in con[k]] for con in cons]
    a, b, c = (cc.Csr.from_rows(rows(k)) for k in range(3))
    r1cs_like = type("R", (), dict(a=a, b=b, c=c, num_constraints=len(cons), wire_mapping=None, num_inputs=2))
    assert cc.CircomCircuit(r1cs_like, w).first_unsatisfied() == -1
    rng = random.Random(14)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    pk = cc.trapdoor_setup(a, b, c, n_vars, n_pub, tox)
    mats = cc.ConstraintMatrices(2, n_vars - 1, len(cons), a, b)
    path = str(tmp_path / "dense14.zkey")
    cc.write_zkey(path, pk, mats)
    pk2, mats2 = cc.read_zkey(path)
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    rs = cc.fr_from_ints([r, s])
    wm = cc.fr_from_ints(w)
    proof = cc.Prover(pk2, mats2).prove(rs[0], rs[1], wm)
    want = cpu_ref.prove(pk2, mats2, rs[0:1].copy(), rs[1:2].copy(), wm)
    assert proof.raw == want


@pytest.mark.gpu
@pytest.mark.parametrize("logm", [16, 20])
def test_libsnark_reduction_large_bytes_and_pairing(gpulib, logm):
    """arkworks-style key (LibsnarkReduction, reference tests/groth16.rs:11-40 path) at 2^16 and 2^20:
    multi-pass coset NTTs + the 7th inverse transform.  Proof BYTES == the C restatement's Libsnark
    prove (pinned to the Python oracle at <= 2^10, tests/test_oracle.py), h element for element at
    2^16; the proof verifies, a wrong public input is rejected, and the key's H query really is
    Z(tau)/delta * tau^i (spot checks against the oracle)."""
    import circom_compat_amd as cc
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, logm)
    rng = random.Random(logm + 7)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox, reduction="libsnark")
    n = pk.domain_size
    tau, delta = tox[0], tox[4]
    zt = (pow(tau, n, o.R_MOD) - 1) % o.R_MOD
    for i in (0, 1, 12345, n - 2):
        k = zt * o.fr_inv(delta) % o.R_MOD * pow(tau, i, o.R_MOD) % o.R_MOD
        assert bytes(pk.h_query[i]) == o.g1_to_bytes(o.G1.mul(o.G1_GEN, k))
    assert not pk.h_query[n - 1].any()                      # padding: the point at infinity
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    pr = cc.Prover(pk, mats, reduction="libsnark")
    proof = pr.prove(r, s, w_ints)
    import cpu_ref
    rs = cc.fr_from_ints([r, s])
    w = cc.fr_from_ints(w_ints)
    want, h_c = cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w, want_h=True, reduction="libsnark")
    assert proof.raw == want, "GPU Libsnark proof bytes differ from the C restatement at 2^%d" % logm
    if logm <= 16:
        assert np.array_equal(pr.witness_map(w), h_c)
    vk = dict(alpha_g1=o.g1_from_bytes(bytes(pk.vk.alpha_g1)), beta_g2=o.g2_from_bytes(bytes(pk.vk.beta_g2)),
              gamma_g2=o.g2_from_bytes(bytes(pk.vk.gamma_g2)), delta_g2=o.g2_from_bytes(bytes(pk.vk.delta_g2)),
              ic=[o.g1_from_bytes(bytes(x)) for x in pk.vk.gamma_abc_g1])
    assert o.verify_proof(vk, [w_ints[1]], H.proof_from_bytes(proof.raw))
    assert not o.verify_proof(vk, [(w_ints[1] + 1) % o.R_MOD], H.proof_from_bytes(proof.raw))


def _vk_dict(pk):
    return dict(alpha_g1=o.g1_from_bytes(bytes(pk.vk.alpha_g1)), beta_g2=o.g2_from_bytes(bytes(pk.vk.beta_g2)),
                gamma_g2=o.g2_from_bytes(bytes(pk.vk.gamma_g2)), delta_g2=o.g2_from_bytes(bytes(pk.vk.delta_g2)),
                ic=[o.g1_from_bytes(bytes(x)) for x in pk.vk.gamma_abc_g1])


@pytest.mark.parametrize("k", [20, 22, pytest.param(24, marks=pytest.mark.skipif(not LARGE, reason="opt-in: G16_TEST_LARGE=1"))])
def test_full_prove_headline_sizes_bytes_pairing_and_qap_identities(gpulib, k):
    """BASELINE configs 2 / 3 / 4 (the bench circuit at 2^20, the headline 2^22, and configs[3]'s 2^24
    circuit on ONE GPU: 84 GiB of point planes): ONE full prove through the C ABI is
      * byte-identical to the CPU restatement's proof of the same (pk, r, s, w)   [SURVEY 8(d)],
      * accepted by the pairing check, a wrong public input rejected            [zkey.rs:868-870],
    and the two O(n) scalar-side identities of SURVEY Appendix C.2 hold on the GPU's OWN h:
      (ii) sum_i h_i k_h_i == (U(tau) V(tau) - W(tau)) / delta   -- the whole witness map, no CPU FFT of h
      (i)  MSM(H, h) == (sum_i h_i k_h_i) G1                     -- the H MSM at full size
    (k_h = CircomReduction::h_query_scalars, qap.rs:90-105; its two big transforms run on the C
    oracle's FFT, which tests/test_oracle.py pins to the Python oracle)."""
    import circom_compat_amd as cc
    import cpu_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    R = o.R_MOD
    mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
    m, n = mats.num_constraints, 1 << k
    rng = random.Random(k)
    tox = [rng.randrange(1, R) for _ in range(5)]
    tau, delta = tox[0], tox[4]
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
    rs_rng = random.Random(1000 + k)
    r, s = rs_rng.randrange(R), rs_rng.randrange(R)
    rs = cc.fr_from_ints([r, s])
    w = cc.fr_from_ints(w_ints)
    pr = cc.Prover(pk, mats)
    proof = pr.prove(rs[0], rs[1], w)
    # ---- bytes == CPU proof
    want = cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
    assert proof.raw == want, "GPU proof bytes differ from the CPU restatement at 2^%d" % k
    # ---- pairing predicate
    vk = _vk_dict(pk)
    assert o.verify_proof(vk, [w_ints[1]], H.proof_from_bytes(proof.raw))
    assert not o.verify_proof(vk, [(w_ints[1] + 1) % R], H.proof_from_bytes(proof.raw))
    # ---- C.2 identities on the GPU's h
    h = cc.fr_to_ints(pr.witness_map(w))
    di = o.fr_inv(delta)
    pw = [1] * (2 * n)
    for i in range(1, 2 * n - 1):
        pw[i] = pw[i - 1] * tau % R
    pw[2 * n - 1] = 0                                   # h_query_scalars: 2(n-1)+1 powers, zero padded
    kh_all = cpu_ref.fft(cc.fr_from_ints(pw), k + 1, inverse=True)
    k_h = [x * di % R for x in cc.fr_to_ints(kh_all[1::2])]
    lhs = sum(x * y for x, y in zip(h, k_h)) % R
    L = cc.fr_to_ints(cpu_ref.fft(cc.fr_from_ints(pw[:n]), k, inverse=True))   # L_j(tau)
    xs = w_ints[2:] + [w_ints[1]]                       # x_0 .. x_m (x_m is the public output wire)
    V = sum(xs[j] * L[j] for j in range(m)) % R
    U = (-V + w_ints[0] * L[m] + w_ints[1] * L[m + 1]) % R          # rows m, m+1: the copied inputs (qap.rs:46-50)
    W = -sum(xs[j + 1] * L[j] for j in range(m)) % R
    assert lhs == (U * V - W) * di % R, "QAP identity fails on the GPU witness map"
    assert pr.msm_g1(3, cc.fr_from_ints(h)) == o.g1_to_bytes(o.G1.mul(o.G1_GEN, lhs))
    # ---- the filtered B view (ctx.h sort_b; what a key with >= 1/8 points at infinity in its B queries
    # selects) in this size's schedule branch (2^20: reductions on the `red` stream, 2^22: everything in
    # order): forced here on a key without such points, so the view holds every pair -- same bytes
    if k <= 22:
        os.environ["G16_SPARSE_B"] = "1"
        try:
            sp = cc.Prover(pk, mats)
        finally:
            del os.environ["G16_SPARSE_B"]
        assert sp.info()["sparse_b"] == 1 and pr.info()["sparse_b"] == 0
        assert sp.prove(rs[0], rs[1], w).raw == want, "filtered-B-view proof differs at 2^%d" % k
        sp.close()
    # ---- BASELINE configs[3] as far as one GPU allows: the SAME circuit sharded over 8 ranks
    # (g16_ctx_create_multi, every rank on this GPU, MSMs cut by bucket range, distributed witness
    # map): bytes == the CPU proof above; a second proof with other (r, s) verifies
    if k >= 22:
        pr.close()
        del pr
        multi = cc.Prover(pk, mats, devices=[0] * 8, shard="buckets")
        info = multi.info()
        assert (info["devices"], info["shard_mode"], info["shard_w"]) == (8, "buckets", n_vars - 1)
        wptr = multi.upload_witness(w)
        assert multi.prove_dev(rs[0], rs[1], wptr).raw == want, "8-rank bucket-sharded proof differs at 2^%d" % k
        rs2 = cc.fr_from_ints([rs_rng.randrange(R), rs_rng.randrange(R)])
        p2 = multi.prove_dev(rs2[0], rs2[1], wptr)
        assert p2.raw != want and o.verify_proof(vk, [w_ints[1]], H.proof_from_bytes(p2.raw))
        multi.close()


@pytest.mark.skipif(not LARGE, reason="opt-in: G16_TEST_LARGE=1")
def test_capacity_point_2p25_on_one_gpu(gpulib):
    """One size above BASELINE's largest circuit: 2^25 constraints on ONE GPU -- the last size whose full
    point planes fit 288 GB (12 planes x 384 B x 2^25 = 154 GB), window c = 22.
    The proof passes the pairing check, a wrong public input is rejected (size-independent
    properties) and -- since round 4 in the default suite -- the 256 bytes equal the CPU restatement's
    (about two minutes of host time; G16_TEST_2P25_NO_BYTES=1 skips that leg)."""
    import torch
    import circom_compat_amd as cc
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    import psutil
    if torch.cuda.get_device_properties(0).total_memory < 250 * 2**30:
        pytest.skip("needs the 288 GB of an MI355X")
    if psutil.virtual_memory().available < 96 * 2**30:
        pytest.skip("needs ~50 GB of host memory for the key and the witness")
    k = 25
    R = o.R_MOD
    mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
    rng = random.Random(k)
    tox = [rng.randrange(1, R) for _ in range(5)]
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
    rs = cc.fr_from_ints([rng.randrange(R), rng.randrange(R)])
    w = cc.fr_from_ints(w_ints)
    pr = cc.Prover(pk, mats)
    info = pr.info()
    assert info["planes_w"] == info["W_w"] and info["D_w"] == 1, "full planes must fit at 2^25: %r" % (info,)
    proof = pr.prove(rs[0], rs[1], w)
    assert pr.prove(rs[0], rs[1], w).raw == proof.raw
    pr.close()
    vk = _vk_dict(pk)
    assert o.verify_proof(vk, [w_ints[1]], H.proof_from_bytes(proof.raw))
    assert not o.verify_proof(vk, [(w_ints[1] + 1) % R], H.proof_from_bytes(proof.raw))
    if not os.environ.get("G16_TEST_2P25_NO_BYTES"):
        import cpu_ref
        assert proof.raw == cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)


@pytest.mark.parametrize("kind,k", [("dense", 12), ("chain", 12), ("chain", 14), ("chain", 16), ("chain", 20),
                                    ("chain", 22)])     # 22: the headline key itself
def test_key_generator_pinned_to_the_oracles_trapdoor_scalars(gpulib, kind, k):
    """Every large test proves under a key minted by the product's own g16_setup_create, and a
    self-consistent wrong key would still give GPU bytes == CPU bytes.  So the key generator is pinned
    one level up: the oracle's trapdoor SCALARS (bn254_ref.trapdoor_scalars: Lagrange values at tau,
    the A / B / C column sums, CircomReduction::h_query_scalars of qap.rs:90-105 -- field operations
    only) say that every query point is k_i * G; 1000 random indices per query (+ the ends) of the
    GPU-made A, B1, B2, L and H arrays are compared with k_i * G formed by the C restatement's plain
    double-and-add (pinned to bn254_ref in tests/test_oracle.py).  At 2^16 / 2^20 / 2^22 the two big inverse
    transforms of the scalar side run on the C restatement's FFT (pinned likewise)."""
    import circom_compat_amd as cc
    import cpu_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    R = o.R_MOD
    if kind == "dense":
        mats, (A, B, Cm), w_ints, n_vars = bench.dense_skewed_circuit(cc, k)
    else:
        mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
    rng = random.Random(k * 3 + len(kind))
    tox = [rng.randrange(1, R) for _ in range(5)]
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)

    def rows(M):
        co = cc.fr_to_ints(M.coeff)
        col = M.col.tolist()
        rp = M.row_ptr.tolist()
        return [list(zip(col[rp[i]:rp[i + 1]], co[rp[i]:rp[i + 1]])) for i in range(len(rp) - 1)]
    cons = list(zip(rows(A), rows(B), rows(Cm)))

    def c_ntt(x, inverse=False):
        lg = (len(x) - 1).bit_length()
        return cc.fr_to_ints(cpu_ref.fft(cc.fr_from_ints(x), lg, inverse=inverse))
    td = o.trapdoor_scalars(cons, n_vars, 1, *tox, ntt_fn=c_ntt if k >= 16 else None)
    assert td["domain_size"] == pk.domain_size == 1 << k
    g1, g2 = o.g1_to_bytes(o.G1_GEN), o.g2_to_bytes(o.G2_GEN)

    def check(name, arr, scal, mul, base):
        n = len(scal)
        assert arr.shape[0] == n, (name, arr.shape, n)
        idx = sorted(set([0, 1, n - 2, n - 1] + [rng.randrange(n) for _ in range(1000)]))
        want = mul(base, [scal[i] for i in idx])
        bad = [i for j, i in enumerate(idx) if bytes(arr[i]) != bytes(want[j])]
        assert not bad, "%s: %d of %d sampled points differ from k * G, first at index %d" % (name, len(bad), len(idx), bad[0])
    check("a_query", pk.a_query, td["u"], cpu_ref.g1_mul_batch, g1)
    check("b_g1_query", pk.b_g1_query, td["v"], cpu_ref.g1_mul_batch, g1)
    check("b_g2_query", pk.b_g2_query, td["v"], cpu_ref.g2_mul_batch, g2)
    check("l_query", pk.l_query, td["k_l"], cpu_ref.g1_mul_batch, g1)
    check("h_query", pk.h_query, td["k_h"], cpu_ref.g1_mul_batch, g1)
    ic = cpu_ref.g1_mul_batch(g1, td["k_ic"])
    assert [bytes(x) for x in pk.vk.gamma_abc_g1] == [bytes(x) for x in ic]
    for name, kk in (("alpha_g1", tox[1]), ("beta_g1", tox[2]), ("delta_g1", tox[4])):
        got = pk.vk.alpha_g1 if name == "alpha_g1" else getattr(pk, name)
        assert bytes(got) == bytes(cpu_ref.g1_mul_batch(g1, [kk])[0]), name
    for name, kk in (("beta_g2", tox[2]), ("gamma_g2", tox[3]), ("delta_g2", tox[4])):
        assert bytes(getattr(pk.vk, name)) == bytes(cpu_ref.g2_mul_batch(g2, [kk])[0]), name


def test_fewer_planes_than_windows_2p22_bytes(gpulib):
    """The fallback every domain above the full-plane capacity point (2^25) runs on -- planes < W, D > 1
    bucket sets per MSM folded by k_horner with c doublings in between -- on the REAL kernels at the
    headline size: 2^22 with planes = 4 (W = 13 -> D = 4: 2^21 buckets, four reductions folded per MSM),
    and planes = 1 on the H query's own sort (D = W).  Bytes == the CPU restatement's proof."""
    import circom_compat_amd as cc
    import cpu_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    k = 22
    R = o.R_MOD
    mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
    rng = random.Random(k + 400)
    tox = [rng.randrange(1, R) for _ in range(5)]
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
    rs = cc.fr_from_ints([rng.randrange(R), rng.randrange(R)])
    w = cc.fr_from_ints(w_ints)
    want = cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
    for planes in (4, 1):
        pr = cc.Prover(pk, mats, planes=planes)
        info = pr.info()
        assert info["planes_w"] == planes and info["D_w"] == -(-info["W_w"] // planes) and info["D_w"] > 1, info
        assert info["planes_h"] == planes and info["D_h"] > 1, info
        got = pr.prove(rs[0], rs[1], w)
        pr.close()
        assert got.raw == want, "planes=%d (D=%d) proof differs from the CPU restatement" % (planes, info["D_w"])
    assert o.verify_proof(_vk_dict(pk), [w_ints[1]], H.proof_from_bytes(got.raw))


def test_domain_2p26_on_one_gpu_not_refused(gpulib):
    """A domain the reference accepts (qap.rs:30-32,63-68: any n with a 2n-th root of unity, n <= 2^27)
    must not be refused: 2^26 constraints on ONE GPU -- past the size whose full point planes fit 288 GB,
    so plan_msm_configs (api.hip) picks planes < W for the witness queries.  Opt-in (G16_TEST_2P26=1:
    ~35 GB of host memory for the key, several minutes; the round's record is the bench line
    profiles/r04_bench_chain26.json from scripts/r4_run1.sh, and scripts/r4_chain27.sh does the same at 2^27,
    the reference's own limit): pairing check, wrong input rejected, and with
    G16_TEST_2P26_BYTES=1 the 256 bytes against the CPU restatement."""
    if not os.environ.get("G16_TEST_2P26"):
        pytest.skip("opt-in: G16_TEST_2P26=1")
    import torch
    import circom_compat_amd as cc
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    import psutil
    if torch.cuda.get_device_properties(0).total_memory < 250 * 2**30:
        pytest.skip("needs the 288 GB of an MI355X")
    if psutil.virtual_memory().available < 160 * 2**30:
        pytest.skip("needs ~100 GB of host memory for the key, the matrices and the witness")
    k = 26
    R = o.R_MOD
    mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
    rng = random.Random(k)
    tox = [rng.randrange(1, R) for _ in range(5)]
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
    rs = cc.fr_from_ints([rng.randrange(R), rng.randrange(R)])
    w = cc.fr_from_ints(w_ints)
    pr = cc.Prover(pk, mats)
    info = pr.info()
    print("2^26 configuration:", info)
    assert info["domain_size"] == 1 << k and info["D_w"] > 1, info
    proof = pr.prove(rs[0], rs[1], w)
    assert pr.prove(rs[0], rs[1], w).raw == proof.raw
    pr.close()
    vk = _vk_dict(pk)
    assert o.verify_proof(vk, [w_ints[1]], H.proof_from_bytes(proof.raw))
    assert not o.verify_proof(vk, [(w_ints[1] + 1) % R], H.proof_from_bytes(proof.raw))
    if os.environ.get("G16_TEST_2P26_BYTES"):
        import cpu_ref
        assert proof.raw == cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)


@pytest.mark.parametrize("logm,world,shard", [(14, 4, "points"), (14, 4, "buckets"), (17, 8, "buckets"),
                                              (17, 3, "buckets"), (22, 8, "points"),
                                              (22, 2, "points")])   # 2^19 buckets per rank: round 4's reductions-off-main schedule at that size
def test_in_library_multi_device_prover_large(gpulib, logm, world, shard):
    """g16_ctx_create_multi with every rank on this one GPU (device_ids = [0] * world): the exchanges,
    events and per-device host threads of csrc/multi.hip are the real ones (only the peer copies
    degenerate to device-local copies).  Two consecutive proofs with different (r, s): bytes == the
    CPU restatement's."""
    import circom_compat_amd as cc
    import cpu_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, logm)
    rng = random.Random(logm)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
    w = cc.fr_from_ints(w_ints)
    pr = cc.Prover(pk, mats, devices=[0] * world, shard=shard)
    assert pr.info()["devices"] == world and pr.info()["shard_mode"] == shard
    host = pr.witness_host_buffer()                     # pinned staging buffer owned by the ctx
    host[:] = w
    for _ in range(2 if logm < 20 else 1):              # headline size: one CPU proof (~16 s) is enough
        rs = cc.fr_from_ints([rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)])
        got = pr.prove(rs[0], rs[1], host)
        assert got.raw == cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
    assert o.verify_proof(_vk_dict(pk), [w_ints[1]], H.proof_from_bytes(got.raw))
    pr.close()


@pytest.mark.parametrize("shard", ["points", "buckets"])
def test_per_rank_api_with_exchange_stream_no_host_syncs(gpulib, shard):
    """The per-process API driven the way bench.py drives it under torchrun, with the collectives'
    stream registered (g16_dist_set_exchange_stream): the phase calls never block the host, every
    hand-off is an event.  world = 4 ranks on this one GPU, the all-to-all / all-gather done by hand
    with device copies ENQUEUED on the registered stream."""
    import torch
    import circom_compat_amd as cc
    import cpu_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    logm, world = 15, 4
    mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, logm)
    rng = random.Random(99)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
    w = cc.fr_from_ints(w_ints)
    w_dev = torch.from_numpy(w.view(np.int64)).cuda()
    torch.cuda.synchronize()
    xs = torch.cuda.Stream(priority=-1)   # high priority: shares a hardware queue with the aux stream, not with the MSM streams
    provers = [cc.Prover(pk, mats, rank=g, world=world, dist_wm=True, shard=shard) for g in range(world)]
    for p in provers:
        p.set_exchange_stream(xs.cuda_stream)
    nbytes = provers[0].exchange_bytes()
    chunk = nbytes // world
    send = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(world)]
    recv = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(world)]

    def all_to_all():                                   # on the registered stream, no host sync
        with torch.cuda.stream(xs):
            for dst in range(world):
                for src in range(world):
                    recv[dst][src * chunk:(src + 1) * chunk].copy_(send[src][dst * chunk:(dst + 1) * chunk],
                                                                     non_blocking=True)

    for _ in range(2):
        rs = cc.fr_from_ints([rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)])
        for g, p in enumerate(provers):
            p.dist_phase1(rs[0], rs[1], w_dev.data_ptr(), send[g].data_ptr())
        all_to_all()
        for g, p in enumerate(provers):
            p.dist_phase2(recv[g].data_ptr(), send[g].data_ptr())
        all_to_all()
        for g, p in enumerate(provers):
            p.dist_phase3_dev(recv[g].data_ptr())
        # all-gather of the 1 KiB records, device to device on the registered stream
        with torch.cuda.stream(xs):
            for dst in provers:
                out = cc.device_tensor(dst.gather_buffer(), world * 1024)
                for g, src in enumerate(provers):
                    out[g * 1024:(g + 1) * 1024].copy_(cc.device_tensor(src.partial_buffer(), 1024),
                                                       non_blocking=True)
        proofs = [p.prove_finish_dev(rs[0], rs[1]).raw for p in provers]
        want = cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
        assert all(x == want for x in proofs)


def test_reference_bench_workload_complex_circuit(gpulib, tmp_path):
    """The reference's own bench case (benches/groth16.rs:13-85 on complex-circuit-10000-10000):
    r1cs loader -> witness -> key (trapdoor, since the snapshot has no .zkey) written and re-read
    through the snarkjs format -> Groth16::create_proof_with_reduction_and_matrices with the
    bench's argument order -> verify, as the bench asserts; bytes == the CPU restatement."""
    import circom_compat_amd as cc
    import cpu_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    mats_r, (A, B, Cm), w_ints, n_vars = bench.complex_circuit(cc)
    rng = random.Random(10000)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    pk0 = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
    path = str(tmp_path / "complex-circuit-10000-10000.zkey")
    cc.write_zkey(path, pk0, mats_r)
    params, matrices = cc.read_zkey(path)                       # let (params, matrices) = read_zkey(&mut file)
    num_inputs, num_constraints = matrices.num_instance_variables, matrices.num_constraints
    assert (num_inputs, num_constraints, params.domain_size) == (2, 10000, 1 << 14)
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    proof = cc.Groth16.create_proof_with_reduction_and_matrices(params, r, s, matrices, num_inputs,
                                                                num_constraints, w_ints)
    inputs = w_ints[1:num_inputs]
    vk = _vk_dict(params)
    assert o.verify_proof(vk, inputs, H.proof_from_bytes(proof.raw))
    assert not o.verify_proof(vk, [inputs[0] ^ 1], H.proof_from_bytes(proof.raw))
    rs = cc.fr_from_ints([r, s])
    assert proof.raw == cpu_ref.prove(params, matrices, rs[0:1].copy(), rs[1:2].copy(), cc.fr_from_ints(w_ints))


def test_dense_skewed_2p20_through_zkey_vs_cpu_restatement(gpulib, tmp_path):
    """BASELINE config 5 substitute AT SIZE (2^20 rows, ~80 % of the witness in {0, 1}: a handful of
    enormous buckets -> k_combine_large and the hot LDS bins of the sort): key through the zkey
    writer + read_zkey, proof bytes == the CPU restatement's, pairing accepted."""
    import circom_compat_amd as cc
    import cpu_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    k = 20
    mats0, (A, B, Cm), w_ints, n_vars = bench.dense_skewed_circuit(cc, k)
    assert sum(1 for x in w_ints if x in (0, 1)) >= 0.5 * len(w_ints)
    circ = cc.CircomCircuit(type("R", (), dict(a=A, b=B, c=Cm, num_constraints=mats0.num_constraints,
                                               wire_mapping=None, num_inputs=2, num_variables=n_vars)), w_ints)
    assert circ.first_unsatisfied() == -1
    rng = random.Random(k)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    pk0 = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
    path = str(tmp_path / "dense20.zkey")
    cc.write_zkey(path, pk0, mats0)
    pk, mats = cc.read_zkey(path)
    rs = cc.fr_from_ints([rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)])
    w = cc.fr_from_ints(w_ints)
    proof = cc.Prover(pk, mats).prove(rs[0], rs[1], w)
    assert proof.raw == cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
    assert o.verify_proof(_vk_dict(pk), [w_ints[1]], H.proof_from_bytes(proof.raw))


def _through_zkey(cc, tmp_path, name, A, B, Cm, n_vars, mats0, seed):
    rng = random.Random(seed)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    pk0 = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
    path = str(tmp_path / name)
    cc.write_zkey(path, pk0, mats0)
    pk, mats = cc.read_zkey(path)
    return pk, mats, rng


def test_poseidon_shaped_2p20_through_zkey_vs_cpu_restatement(gpulib, tmp_path):
    """BASELINE configs[4] substitute with the shape of a circom Poseidon hash chain AT SIZE (2^20 rows:
    x^5 S-boxes as three rows, 4-term linear combinations with full-width MDS / round constants in A
    AND B, uniform 254-bit witness -- bench.poseidon_shaped_circuit): satisfiable (GPU constraint check), key
    through g16_zkey_write -> read_zkey (the Coefs path, src/zkey.rs:151-196, with full-width values),
    proof bytes == the CPU restatement's, pairing accepted, wrong public input rejected."""
    import circom_compat_amd as cc
    import cpu_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    k = 20
    mats0, (A, B, Cm), w_ints, n_vars = bench.poseidon_shaped_circuit(cc, k)
    assert sum(1 for x in w_ints if x in (0, 1)) <= 4           # uniform witness
    wide = sum(1 for x in cc.fr_to_ints(A.coeff[:20000]) if x.bit_length() > 200)
    assert wide > 10000                                          # full-width coefficients dominate
    circ = cc.CircomCircuit(type("R", (), dict(a=A, b=B, c=Cm, num_constraints=mats0.num_constraints,
                                               wire_mapping=None, num_inputs=2, num_variables=n_vars)), w_ints)
    assert circ.first_unsatisfied() == -1
    pk, mats, rng = _through_zkey(cc, tmp_path, "poseidon20.zkey", A, B, Cm, n_vars, mats0, k)
    rs = cc.fr_from_ints([rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)])
    w = cc.fr_from_ints(w_ints)
    proof = cc.Prover(pk, mats).prove(rs[0], rs[1], w)
    assert proof.raw == cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
    assert o.verify_proof(_vk_dict(pk), [w_ints[1]], H.proof_from_bytes(proof.raw))
    assert not o.verify_proof(_vk_dict(pk), [(w_ints[1] + 1) % o.R_MOD], H.proof_from_bytes(proof.raw))


def test_real_poseidon_chain_2p20_through_zkey_one_gpu_and_8_ranks(gpulib, tmp_path):
    """BASELINE configs[4] as far as it can be built offline: a REAL Poseidon(2) hash chain with
    circomlib's Grain-LFSR parameters (bench.poseidon_chain_circuit; the checker oracle/poseidon_ref.py is
    pinned to circomlibjs' KATs) AT SIZE: 4369 hashes x 240 rows = 1 048 560 rows, 1 052 931 wires (more wires than
    the 2^20 domain), 9.8 M / 18.5 M full-width coefficients in A / B (rows of up to 61 terms).  The
    public output is the oracle chain's h_4369 and wire values include circomlibjs' poseidon([1, 2]);
    satisfiable (GPU constraint check); key through g16_zkey_write -> read_zkey (the Coefs path,
    src/zkey.rs:151-196); proof bytes == the CPU restatement's on ONE GPU and on 8 emulated ranks under
    BOTH cuts; pairing accepted for h_H, rejected for h_H + 1."""
    import circom_compat_amd as cc
    import cpu_ref
    import poseidon_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    k = 20
    mats0, (A, B, Cm), w_ints, n_vars = bench.poseidon_chain_circuit(cc, k)
    n_hashes = mats0.num_constraints // 240
    assert n_hashes == 4369 and n_vars > (1 << k) and int(np.diff(A.row_ptr).max()) == 61
    chain = poseidon_ref.hash_chain(1, [i + 2 for i in range(n_hashes)])
    assert chain[1] == poseidon_ref.KATS[(1, 2)] and w_ints[1] == chain[-1]
    circ = cc.CircomCircuit(type("R", (), dict(a=A, b=B, c=Cm, num_constraints=mats0.num_constraints,
                                               wire_mapping=None, num_inputs=2, num_variables=n_vars)), w_ints)
    assert circ.first_unsatisfied() == -1
    pk, mats, rng = _through_zkey(cc, tmp_path, "poseidon_real20.zkey", A, B, Cm, n_vars, mats0, k)
    assert np.array_equal(mats.b.col, B.col) and np.array_equal(mats.a.coeff, A.coeff)
    rs = cc.fr_from_ints([rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)])
    w = cc.fr_from_ints(w_ints)
    want = cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
    one = cc.Prover(pk, mats)
    # every x4 wire of an S-box appears in A only: a third of b_g1_query / b_g2_query is the point at
    # infinity, and the B2 (G2) MSM runs over the filtered view of the witness sort
    assert one.info()["sparse_b"] == 1
    proof = one.prove(rs[0], rs[1], w)
    one.close()
    assert proof.raw == want
    os.environ["G16_SPARSE_B"] = "0"
    try:
        full = cc.Prover(pk, mats)
    finally:
        del os.environ["G16_SPARSE_B"]
    assert full.info()["sparse_b"] == 0 and full.prove(rs[0], rs[1], w).raw == want
    full.close()
    assert o.verify_proof(_vk_dict(pk), [chain[-1]], H.proof_from_bytes(proof.raw))
    assert not o.verify_proof(_vk_dict(pk), [(chain[-1] + 1) % o.R_MOD], H.proof_from_bytes(proof.raw))
    for shard in ("points", "buckets"):
        pr = cc.Prover(pk, mats, devices=[0] * 8, shard=shard)
        assert pr.info()["shard_mode"] == shard
        assert pr.prove(rs[0], rs[1], w).raw == want, shard
        pr.close()


@pytest.mark.parametrize("shard", ["points", "buckets"])
@pytest.mark.parametrize("workload", ["dense-skewed", "poseidon-shaped", "poseidon"])
def test_config5_substitutes_sharded_over_8_ranks_2p17(gpulib, tmp_path, workload, shard):
    """The two config-5 substitutes through g16_ctx_create_multi([0] * 8) at 2^17 rows, both ways of
    cutting the MSMs: `dense-skewed` puts ~45 % of all sort entries into ONE bucket (the value-1 wires)
    -- split over point-range shards it spans many lanes of every rank, under bucket ranges it makes
    one partition larger than a rank's fair share (that rank owns it alone); `poseidon-shaped` is the
    uniform full-width corner with narrow rows, `poseidon` the real hash chain (rows of up to 61 terms).  bytes == the CPU restatement's, two proofs."""
    import circom_compat_amd as cc
    import cpu_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    k = 17
    gen = {"dense-skewed": bench.dense_skewed_circuit, "poseidon-shaped": bench.poseidon_shaped_circuit,
           "poseidon": bench.poseidon_chain_circuit}[workload]
    mats0, (A, B, Cm), w_ints, n_vars = gen(cc, k)
    pk, mats, rng = _through_zkey(cc, tmp_path, workload + "17.zkey", A, B, Cm, n_vars, mats0, k)
    w = cc.fr_from_ints(w_ints)
    pr = cc.Prover(pk, mats, devices=[0] * 8, shard=shard)
    assert pr.info()["shard_mode"] == shard
    for _ in range(2):
        rs = cc.fr_from_ints([rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)])
        assert pr.prove(rs[0], rs[1], w).raw == cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
    pr.close()


RCCL_WORKER = r'''
import os, sys, random
import numpy as np
import torch, torch.distributed as dist
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import bench, cpu_ref
import circom_compat_amd as cc
logm = int(sys.argv[2])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, logm)
rng = random.Random(logm)
tox = [rng.randrange(1, bench.R_MOD) for _ in range(5)]
pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
w = cc.fr_from_ints(w_ints)
w_dev = torch.from_numpy(w.view(np.int64)).cuda()
torch.cuda.synchronize()
xs = torch.cuda.Stream(priority=-1)      # the collectives' stream, registered with the library
for shard in ("points", "buckets"):
    p = cc.Prover(pk, mats, rank=0, world=1, dist_wm=True, shard=shard)
    p.set_exchange_stream(xs.cuda_stream)
    nbytes = p.exchange_bytes()
    send = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    recv = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    part_t = cc.device_tensor(p.partial_buffer(), 1024)
    gath_t = cc.device_tensor(p.gather_buffer(), 1024)
    for it in range(2):
        rs = cc.fr_from_ints([rng.randrange(bench.R_MOD), rng.randrange(bench.R_MOD)])
        p.dist_phase1(rs[0], rs[1], w_dev.data_ptr(), send.data_ptr())
        with torch.cuda.stream(xs):
            dist.all_to_all_single(recv, send)
        p.dist_phase2(recv.data_ptr(), send.data_ptr())
        with torch.cuda.stream(xs):
            dist.all_to_all_single(recv, send)
        p.dist_phase3_dev(recv.data_ptr())
        with torch.cuda.stream(xs):
            dist.all_gather_into_tensor(gath_t, part_t)
        proof = p.prove_finish_dev(rs[0], rs[1])
        want = cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
        assert proof.raw == want, f"{shard}: proof {it} differs from the CPU restatement"
    p.close()
torch.cuda.synchronize()
print("rccl backend:", dist.get_backend(), "ok")
dist.destroy_process_group()
'''


def test_rccl_executes_next_to_the_library_world1(gpulib, tmp_path):
    """RCCL (torch backend "nccl") loaded and executing in the same process as libg16_amd.so: a
    one-rank process group drives the per-process phase API -- all_to_all_single twice and
    all_gather_into_tensor of the 1 KiB records -- on the registered
    high-priority stream, with the library's event hand-offs on both sides of every collective.
    Degenerate exchanges (one rank), real code path: the first 8-GPU run is not the first time the
    two libraries meet.  bytes == CPU restatement, two consecutive proofs per shard mode."""
    import socket
    import subprocess
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
               TORCH_NCCL_HIGH_PRIORITY="1")
    r = subprocess.run([sys.executable, str(script), root, "15"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "rccl backend: nccl ok" in r.stdout


RCCL_INLIB_WORKER = r'''
import ctypes as C, os, sys, random
import numpy as np
import torch
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import bench, cpu_ref
import circom_compat_amd as cc
logm = int(sys.argv[2])
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
# the host's OWN RCCL (here: the one PyTorch ships; a Rust host links /opt/rocm's): a one-rank communicator
rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
uid = (C.c_char * 128)()
assert rccl.ncclGetUniqueId(uid) == 0
class Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
u = Uid.from_buffer_copy(bytes(uid))
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, u, 0) == 0
mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, logm)
rng = random.Random(logm)
tox = [rng.randrange(1, bench.R_MOD) for _ in range(5)]
pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
w = cc.fr_from_ints(w_ints)
w_dev = torch.from_numpy(w.view(np.int64)).cuda()
torch.cuda.synchronize()
for shard in ("points", "buckets"):
    p = cc.Prover(pk, mats, rank=0, world=1, dist_wm=True, shard=shard)
    assert p.rccl_ranks() == 0
    p.attach_rccl(comm.value)
    assert p.rccl_ranks() == 1
    for it in range(2):
        rs = cc.fr_from_ints([rng.randrange(bench.R_MOD), rng.randrange(bench.R_MOD)])
        proof = p.prove_dist(rs[0], rs[1], w_dev.data_ptr())
        want = cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
        assert proof.raw == want, f"{shard}: proof {it} differs from the CPU restatement"
    p.close()
# a ctx without a distributed witness map cannot attach; a communicator of another size is refused by rank / world
plain = cc.Prover(pk, mats)
try:
    plain.attach_rccl(comm.value)
    raise SystemExit("attach on a single-device ctx must fail")
except cc.G16Error:
    pass
plain.close()
torch.cuda.synchronize()
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
print("in-library rccl ok")
'''


def test_rccl_collectives_issued_by_the_library_world1(gpulib, tmp_path):
    """g16_dist_attach_rccl + g16_prove_dist: a host WITHOUT torch.distributed -- here a
    bare ctypes binding of RCCL standing in for the Rust shim -- creates the communicator and hands it over;
    ncclAllToAll (twice) and ncclAllGather are then issued by libg16_amd.so itself, which resolves them from
    the RCCL already in the process (it does not link RCCL).  One rank: degenerate exchanges, real call path.
    bytes == CPU restatement, both shard modes, two proofs each; attach on a ctx without dist_wm is an error."""
    import subprocess
    script = tmp_path / "rccl_inlib_worker.py"
    script.write_text(RCCL_INLIB_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), root, "14"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "in-library rccl ok" in r.stdout


@pytest.mark.parametrize("logm", [10, 13, 14])
def test_fixed_base_tables_prover_vs_cpu_restatement(gpulib, logm):
    """Small keys through the fixed-base tables (g16_options.fixed_tables; csrc/msm_table.hip) on the real
    kernels: squaring chains of 2^10 / 2^13 / 2^14 rows (the last one = the automatic rule's limit: 4 GiB
    per G1 query, 8 GiB for B2), key minted on the GPU.  `tables=0` (the library's default) selects the
    path; proof bytes == the CPU restatement's == the bucket path's (`tables=-1`), two (r, s) pairs;
    pairing accepted, wrong input rejected."""
    import circom_compat_amd as cc
    import cpu_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, logm)
    rng = random.Random(600 + logm)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
    w = cc.fr_from_ints(w_ints)
    tab = cc.Prover(pk, mats, tables=0)
    assert tab.info()["fixed_tables"] == 1
    buck = cc.Prover(pk, mats, tables=-1)
    assert buck.info()["fixed_tables"] == 0
    for _ in range(2):
        rs = cc.fr_from_ints([rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)])
        want = cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
        proof = tab.prove(rs[0], rs[1], w)
        assert proof.raw == want
        assert buck.prove(rs[0], rs[1], w).raw == want
    assert o.verify_proof(_vk_dict(pk), [w_ints[1]], H.proof_from_bytes(proof.raw))
    assert not o.verify_proof(_vk_dict(pk), [(w_ints[1] + 1) % o.R_MOD], H.proof_from_bytes(proof.raw))
    tab.close()
    buck.close()


def test_fixed_base_tables_reference_bench_circuit_and_limits(gpulib):
    """The reference bench's own circuit (complex-circuit-10000-10000.r1cs, benches/groth16.rs:106) takes the
    table path by default: bytes == the CPU restatement's, the proof verifies.  One size above the
    automatic limit (2^15 wires) the default stays on the bucket path; `tables=1` on a sharded ctx is an
    error that says why."""
    import circom_compat_amd as cc
    import cpu_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    mats, (A, B, Cm), w_ints, n_vars = bench.complex_circuit(cc)
    rng = random.Random(77)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
    w = cc.fr_from_ints(w_ints)
    pr = cc.Prover(pk, mats, tables=0)
    assert pr.info()["fixed_tables"] == 1
    rs = cc.fr_from_ints([rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)])
    proof = pr.prove(rs[0], rs[1], w)
    assert proof.raw == cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
    assert o.verify_proof(_vk_dict(pk), [w_ints[1]], H.proof_from_bytes(proof.raw))
    with pytest.raises(cc.G16Error) as e:
        cc.Prover(pk, mats, devices=[0, 0], tables=1)
    assert "fixed_tables" in str(e.value)
    pr.close()
    mats2, (A2, B2, C2), w2, nv2 = bench.chain_circuit(cc, 15)
    pk2 = cc.trapdoor_setup(A2, B2, C2, nv2, 1, tox)
    big = cc.Prover(pk2, mats2, tables=0)
    assert big.info()["fixed_tables"] == 0
    # ... and `tables=1` still forces the path there (51 GB of tables, within a third of 288 GB): same bytes
    forced = cc.Prover(pk2, mats2, tables=1)
    assert forced.info()["fixed_tables"] == 1
    wv = cc.fr_from_ints(w2)
    want2 = cpu_ref.prove(pk2, mats2, rs[0:1].copy(), rs[1:2].copy(), wv)
    assert forced.prove(rs[0], rs[1], wv).raw == want2 and big.prove(rs[0], rs[1], wv).raw == want2
    forced.close()
    big.close()
