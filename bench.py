#!/usr/bin/env python
"""bench.py -- Groth16 constraints/sec (BN254) on MI355X: BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full proof (CircomReduction witness map + 4 G1 MSMs + 1 G2 MSM + finalisation) through
the C ABI (Groth16::create_proof_with_reduction_and_matrices, reference benches/groth16.rs:52-60).
Default workload = BASELINE.json configs[2]: the synthetic squaring-chain circuit of SURVEY.md
section 8(d) with m = 2^22 - 2 constraints, a real trapdoor key minted on the GPU.

`value` follows the bench contract: the witness is resident in HBM when the timed region starts.
The SURVEY 8(d) step (host witness -> H2D -> ... -> 256 B D2H, witness in the ctx's page-locked
staging buffer) is timed right after it and reported as `value_pcie_inclusive`.

N > 1 (strong scaling of ONE proof): MSMs sharded by point range (--shard points / auto: every GPU
holds 1/N of the key) or, for the witness-scalar queries, by bucket range (--shard buckets: every GPU
holds all A/B1/B2/L points and the single-GPU window and keeps 1/N of the sorted bucket list; the
only MSM traffic either way is a 1 KiB record per rank; same rank time),
witness map distributed (four-step NTTs, two all-to-all exchanges), records gathered and summed.
  mode "in-library"  (default when this process can see N devices): rank 0 drives all N GPUs through
                     ONE g16_ctx_create_multi ctx -- exchanges are peer copies over xGMI inside the
                     library, no Python between phases; the other ranks only keep the barriers;
  mode "ranks"       (G16_BENCH_MODE=ranks, or fewer visible devices than ranks): one ctx per
                     process, RCCL all_to_all / all_gather on a torch stream the library orders
                     itself against with events (g16_dist_set_exchange_stream): no host syncs.
                     Under torch.distributed.run it is also the fallback when the in-library ctx
                     cannot be built or its first proof does not verify (`fallback_reason`).

Other workloads / modes (BASELINE configs 2 and 5, the reference's own bench circuit):
  --workload dense-skewed   3-term A rows / 2-term B rows, >= 50 % of the witness in {0, 1}, key
                            written and re-read through the snarkjs .zkey format (read_zkey path)
  --workload poseidon       BASELINE configs[4]: a REAL Poseidon(2) hash chain, circomlib parameters (Grain LFSR,
                            t = 3, R_F = 8, R_P = 57; pinned to circomlibjs' KATs by oracle/poseidon_ref.py),
                            240 S-box rows per hash with the linear layers folded into rows of up to 61
                            full-width terms (what circom --O2 leaves); R1CS not circom-compiled; key
                            through the .zkey format
  --workload poseidon-shaped  rounds 3-4's substitute: seeded constants, every lane materialised, 4-term rows
  --workload complex-circuit  tests/golden/complex-circuit-10000-10000.r1cs (benches/groth16.rs:87-108)
  --mode parts              witness map and each MSM timed separately (device-resident operands)

Rank 0 prints ONE JSON line: metric/value/unit/... plus `roofline` (dominant kernel, live HIP-event
timing on the kernel's own stream), `cpu_baseline` (the oracle's multithreaded C restatement of the
reference's CPU prover timed on this box's host cores on the SAME inputs when it fits the time
bound) and `parity`.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


# ------------------------------------------------------------------------------------------------
# workloads (synthetic data; no oracle involved)
# ------------------------------------------------------------------------------------------------
def chain_circuit(cc, k):
    """squaring chain: wires [1, out, x0, x1, ...]; row i: (-x_i) * (x_i) = (-x_{i+1})"""
    m = (1 << k) - 2
    n_vars = m + 2
    one = cc.fr_from_ints([1])[0]
    minus1 = cc.fr_from_ints([R_MOD - 1])[0]
    rp = np.arange(m + 1, dtype=np.uint32)
    wire = np.arange(m, dtype=np.uint32) + 2
    cwire = wire + 1
    cwire[m - 1] = 1                         # x_m is the public output wire
    A = cc.Csr(rp, wire, np.tile(minus1, (m, 1)))
    B = cc.Csr(rp, wire, np.tile(one, (m, 1)))
    Cm = cc.Csr(rp, cwire, np.tile(minus1, (m, 1)))
    xs = [3]
    for _ in range(m):
        xs.append(xs[-1] * xs[-1] % R_MOD)
    w = [1, xs[m]] + xs[:m]
    mats = cc.ConstraintMatrices(2, n_vars - 1, m, A, B)
    return mats, (A, B, Cm), w, n_vars


def dense_skewed_circuit(cc, k, seed=5, n_bits=4096, n_wide=64, p_bit=0.64):
    """SURVEY 8(d) config-5 substitute (the shape of circuit2.r1cs: ~3 / ~2 nnz per row): m = 2^k - 2
    rows, row i defines a fresh wire out_i = (A_i.w)(B_i.w).  p_bit of the rows are bit logic
    ((b_i + b_j - b_k) * b or (..) * (1 - b): results in {0, 1, 2, -1}), the rest mix earlier
    full-width wires with small +- coefficients, so that roughly 60 % of the witness is in {0, 1}, a
    few % are 2 / r - 1, and the rest is uniform 254-bit -- hot MSM buckets, as circom witnesses
    (bit decompositions next to hash state) have."""
    rng = random.Random(seed)
    m = (1 << k) - 2
    w = [1, 0] + [rng.randrange(2) for _ in range(n_bits)] + [rng.randrange(R_MOD) for _ in range(n_wide)]
    bits = list(range(2, 2 + n_bits))
    wide = list(range(2 + n_bits, 2 + n_bits + n_wide))
    coeffs = [1, R_MOD - 1, 2, R_MOD - 2, 3]
    a_rp, b_rp, c_rp = [0], [0], [0]
    a_col, a_val, b_col, b_val, c_col = [], [], [], [], []
    for i in range(m - 1):
        hi = len(w)
        if rng.random() < p_bit:
            ta = [(rng.choice(bits), 1), (rng.choice(bits), 1), (rng.choice(bits), R_MOD - 1)]
            tb = [(rng.choice(bits), 1)] if rng.random() < 0.5 else [(0, 1), (rng.choice(bits), R_MOD - 1)]
        else:
            ta = [(rng.choice(wide), rng.choice(coeffs)) for _ in range(3)]
            tb = [(rng.choice(wide), rng.choice(coeffs)) for _ in range(2)]
        va = sum(c * w[j] for j, c in ta) % R_MOD
        vb = sum(c * w[j] for j, c in tb) % R_MOD
        val = va * vb % R_MOD
        w.append(val)
        if val in (0, 1):
            bits.append(hi)
        elif val.bit_length() > 200:
            wide.append(hi)
        for j, c in ta:
            a_col.append(j)
            a_val.append(c)
        for j, c in tb:
            b_col.append(j)
            b_val.append(c)
        c_col.append(hi)
        a_rp.append(len(a_col))
        b_rp.append(len(b_col))
        c_rp.append(len(c_col))
    # last row: the public output copies the last wire
    w[1] = w[-1]
    a_col.append(len(w) - 1)
    a_val.append(1)
    b_col.append(0)
    b_val.append(1)
    c_col.append(1)
    a_rp.append(len(a_col))
    b_rp.append(len(b_col))
    c_rp.append(len(c_col))
    n_vars = len(w)
    one = cc.fr_from_ints([1])
    A = cc.Csr(a_rp, a_col, cc.fr_from_ints(a_val))
    B = cc.Csr(b_rp, b_col, cc.fr_from_ints(b_val))
    Cm = cc.Csr(c_rp, c_col, np.tile(one, (len(c_col), 1)))
    mats = cc.ConstraintMatrices(2, n_vars - 1, m, A, B)
    return mats, (A, B, Cm), w, n_vars


def poseidon_shaped_circuit(cc, k, seed=7, t=3, full_rounds=8, partial_rounds=57):
    """BASELINE configs[4] substitute with the SHAPE of a circom Poseidon hash chain (the real artefact
    needs circom + snarkjs + a ptau file: not generable offline): m = 2^k - 2 rows of chained width-3
    permutations, 8 full + 57 partial rounds, x^5 S-box.  As circom emits it without the sparse-matrix
    optimisation: the linear layer is folded into the consumer's linear combination, so an S-box is

        (sum_j M[i][j] s_j + c_r,i) * (same) = x2      4-term A and 4-term B, FULL-WIDTH coefficients
        x2 * x2 = x4
        x4 * (sum_j M[i][j] s_j + c_r,i) = x5          4-term B

    and a lane that skips the S-box in a partial round is materialised by a linear row
    (sum_j M[i][j] s_j + c) * 1 = s'.  M is a Cauchy matrix 1 / (x_i + y_j) (the Poseidon
    construction); the round constants are seeded uniform field elements, NOT the Grain-LFSR constants
    of a real instance -- this is a benchmark shape, not a hash.  Each permutation absorbs one fresh
    private input; every wire but the constant is a uniform 254-bit value (no 0/1 wires: the opposite
    corner of the witness space from `dense-skewed`).  2.6 / 2.4 nnz per A / B row, most of them
    full-width."""
    R = R_MOD
    rng = random.Random(seed)
    m = (1 << k) - 2
    rounds = full_rounds + partial_rounds
    rc = [[rng.randrange(1, R) for _ in range(t)] for _ in range(rounds)]
    xs = [rng.randrange(1, R) for _ in range(t)]
    ys = [rng.randrange(1, R) for _ in range(t)]
    M = [[pow((xs[i] + ys[j]) % R, R - 2, R) for j in range(t)] for i in range(t)]
    table, tindex = [1], {1: 0}                       # distinct coefficient values -> converted once

    def cid(v):
        if v not in tindex:
            tindex[v] = len(table)
            table.append(v)
        return tindex[v]

    w = [1, 0] + [rng.randrange(R) for _ in range(t)]
    state = list(range(2, 2 + t))
    a_rp, b_rp, c_rp = [0], [0], [0]
    a_col, a_cf, b_col, b_cf, c_col = [], [], [], [], []

    def row(A, B, cw):
        for j, c in A:
            a_col.append(j)
            a_cf.append(c)
        for j, c in B:
            b_col.append(j)
            b_cf.append(c)
        c_col.append(cw)
        a_rp.append(len(a_col))
        b_rp.append(len(b_col))
        c_rp.append(len(c_col))

    one = [(0, 0)]                                     # the constant wire with coefficient table[0] = 1
    rows, r_idx = 0, 0
    budget = m - 1
    while rows < budget:
        r = r_idx % rounds
        if r == 0 and r_idx:                           # next permutation: chain lane 0 and 2, absorb a fresh input
            w.append(rng.randrange(R))
            state = [state[0], len(w) - 1, state[1]]
        full = r < full_rounds // 2 or r >= rounds - full_rounds // 2
        sv = [w[j] for j in state]
        nxt = []
        for i in range(t):
            if rows >= budget:
                nxt.append(state[i])
                continue
            lc = [(state[j], cid(M[i][j])) for j in range(t)] + [(0, cid(rc[r][i]))]
            val = (sum(M[i][j] * sv[j] for j in range(t)) + rc[r][i]) % R
            if (full or i == 0) and rows + 3 <= budget:
                x2 = val * val % R
                x4 = x2 * x2 % R
                x5 = x4 * val % R
                base = len(w)
                w.extend((x2, x4, x5))
                row(lc, lc, base)
                row([(base, 0)], [(base, 0)], base + 1)
                row([(base + 1, 0)], lc, base + 2)
                rows += 3
                nxt.append(base + 2)
            else:
                w.append(val)
                row(lc, one, len(w) - 1)
                rows += 1
                nxt.append(len(w) - 1)
        state = nxt
        r_idx += 1
    w[1] = w[state[0]]                                 # public output: lane 0 of the last state
    row([(state[0], 0)], one, 1)
    n_vars = len(w)
    tab = cc.fr_from_ints(table)
    A = cc.Csr(a_rp, a_col, tab[np.asarray(a_cf, dtype=np.int64)])
    B = cc.Csr(b_rp, b_col, tab[np.asarray(b_cf, dtype=np.int64)])
    Cm = cc.Csr(c_rp, c_col, np.tile(tab[0], (len(c_col), 1)))
    mats = cc.ConstraintMatrices(2, n_vars - 1, m, A, B)
    return mats, (A, B, Cm), w, n_vars


def poseidon_parameters(t=3, r_f=8, r_p=57, n_bits=254):
    """circomlib's Poseidon parameter set for width t, generated the way the Poseidon paper's reference
    script does (Grain LFSR in self-shrinking mode: round constants by rejection sampling, then the
    Cauchy matrix 1 / (x_i + y_j) from the next 2t elements).  Workload data; the independent checker
    is oracle/poseidon_ref.py, pinned to circomlibjs' hash KATs (tests/test_oracle.py compares both)."""
    reg = int("01" + "0000" + format(n_bits, "012b") + format(t, "012b") + format(r_f, "010b")
              + format(r_p, "010b") + "1" * 30, 2)             # bit 79 = the oldest bit b[0]
    mask = (1 << 80) - 1

    def clock():
        nonlocal reg
        b = ((reg >> 17) ^ (reg >> 28) ^ (reg >> 41) ^ (reg >> 56) ^ (reg >> 66) ^ (reg >> 79)) & 1
        reg = ((reg << 1) | b) & mask
        return b

    for _ in range(160):
        clock()

    def field_bits():
        v, got = 0, 0
        while got < n_bits:
            first, second = clock(), clock()
            if first:
                v, got = (v << 1) | second, got + 1
        return v

    rc = []
    while len(rc) < (r_f + r_p) * t:
        v = field_bits()
        if v < R_MOD:
            rc.append(v)
    while True:
        pts = [field_bits() % R_MOD for _ in range(2 * t)]
        if len(set(pts)) == 2 * t and all((x + y) % R_MOD for x in pts[:t] for y in pts[t:]):
            break
    mds = [[pow((x + y) % R_MOD, R_MOD - 2, R_MOD) for y in pts[t:]] for x in pts[:t]]
    return rc, mds


def poseidon_chain_circuit(cc, k, n_hashes=None):
    """BASELINE configs[4] as far as it can be built offline: a REAL Poseidon hash chain
    h_{i+1} = Poseidon(2)([h_i, x_i]) over BN254 Fr with circomlib's parameters (t = 3, R_F = 8, R_P = 57,
    x^5; `poseidon_parameters`), h_0 = 1, x_i = i + 2 -- so h_1 is circomlibjs' known answer
    poseidon([1, 2]) = 0x115cc0f5...189a and the public output h_H is checked against the oracle's chain.

    The R1CS is NOT circom-compiled (no circom / snarkjs offline).  It is what a linear-substitution
    optimiser (circom --O2) leaves of the round function: ONLY the S-box rows -- 80 S-boxes x 3 rows
    (x2 = in*in, x4 = x2*x2, x5 = x4*in: circomlib's Sigma template) = 240 rows per hash, the constraint
    count circomlib's Poseidon(2) is known for: the capacity lane enters round 0 as the constant 0 + c_0, so
    its first S-box is a constant and folds away (81 S-boxes - 1) -- with every linear layer (round constants, MDS products) folded into the linear combination that feeds the next
    S-box.  The two lanes that skip the S-box in a partial round are never materialised, so the S-box
    input of partial round r is a combination of ~r + 4 wires with full-width coefficients: A rows carry
    up to 61 terms (9.3 per row on average), B rows 17.5 per row -- the width circomlib's Poseidon has
    after --O2 (the sparse-matrix factorisation of circomlib changes the coefficients, not this growth).
    The chain output is folded into the C side of the last S-box row (3 terms) as circom does for
    `out <== lc`.  Wires: 0 = one, 1 = h_H (public), 2 = h_0, 3.. = x_i, then 240 wires per hash.

    k = log2 of the domain: H = (2^k - 2) // 240 hashes unless n_hashes is given (m = 240 H rows)."""
    R = R_MOD
    ROWS = 240
    t, r_f, r_p = 3, 8, 57
    rounds = r_f + r_p
    rc, M = poseidon_parameters(t, r_f, r_p)
    H = n_hashes if n_hashes is not None else ((1 << k) - 2) // ROWS
    assert H >= 1, "one Poseidon(2) permutation needs 240 rows: k >= 8"

    # ---- one hash as a template over symbolic wires ('s', q) own S-box wires, ('p', j) the three
    # last-round wires of the previous hash, ('h',) h_0, ('x',) the absorbed input, ('1',) the constant
    def lc_scale_add(dst, src, c):
        for key, v in src.items():
            dst[key] = (dst.get(key, 0) + c * v) % R

    def template(first):
        h_in = {('h',): 1} if first else {('p', j): M[0][j] for j in range(t)}
        lanes = [{}, h_in, {('x',): 1}]
        rows = []                                    # (A lc, B lc, C wire)
        q = 0
        for r in range(rounds):
            for i in range(t):
                lanes[i] = dict(lanes[i])
                lanes[i][('1',)] = (lanes[i].get(('1',), 0) + rc[r * t + i]) % R
            full = r < r_f // 2 or r >= r_f // 2 + r_p
            order = ([2, 1, 0] if r == rounds - 1 else [0, 1, 2]) if full else [0]
            for i in order:
                lc = {key: v for key, v in lanes[i].items() if v}
                if set(lc) <= {('1',)}:              # constant input (lane 0 of round 0): the S-box folds away
                    lanes[i] = {('1',): pow(lc.get(('1',), 0), 5, R)}
                    continue
                rows.append((lc, lc, ('s', q)))
                rows.append(({('s', q): 1}, {('s', q): 1}, ('s', q + 1)))
                rows.append(({('s', q + 1): 1}, lc, ('s', q + 2)))
                lanes[i] = {('s', q + 2): 1}
                q += 3
            if r < rounds - 1:                       # the last mix is the consumer's business (('p', j) above)
                mixed = []
                for i in range(t):
                    acc = {}
                    for j in range(t):
                        lc_scale_add(acc, lanes[j], M[i][j])
                    mixed.append(acc)
                lanes = mixed
        assert q == ROWS and len(rows) == ROWS
        # last round ran lanes 2, 1, 0: wires s231..233 = lane 2, s234..236 = lane 1, s237..239 = lane 0
        return rows

    last_x5 = {0: 239, 1: 236, 2: 233}                # lane -> own wire index of its last-round x5
    x_base = 3
    s_base = 3 + H                                    # hash i owns wires s_base + ROWS i + q

    table, tindex = [1], {1: 0}

    def cid(v):
        if v not in tindex:
            tindex[v] = len(table)
            table.append(v)
        return tindex[v]

    def compile_rows(rows):
        """rows -> per side (row_ptr, kind, off, coeff id): col = off + kind-specific base of hash i"""
        sides = []
        for side in (0, 1):
            rp, kind, off, cf = [0], [], [], []
            for row in rows:
                for key, v in sorted(row[side].items(), key=lambda kv: (kv[0][0] != '1', kv[0])):
                    kd = key[0]
                    kind.append({'1': 0, 'h': 1, 'x': 2, 's': 3, 'p': 4}[kd])
                    off.append(key[1] if kd == 's' else last_x5[key[1]] if kd == 'p' else 0)
                    cf.append(cid(v))
                rp.append(len(kind))
            sides.append((np.asarray(rp, dtype=np.int64), np.asarray(kind, dtype=np.int64),
                          np.asarray(off, dtype=np.int64), np.asarray(cf, dtype=np.int64)))
        return sides

    def tile(sides, hashes):
        """CSR pieces of `sides` instantiated for the hash indices in `hashes`"""
        out = []
        hs = np.asarray(hashes, dtype=np.int64)
        for rp, kind, off, cf in sides:
            nnz = len(kind)
            base = np.zeros((len(hs), 5), dtype=np.int64)
            base[:, 1] = 2
            base[:, 2] = x_base + hs
            base[:, 3] = s_base + ROWS * hs
            base[:, 4] = s_base + ROWS * (hs - 1)
            col = off[None, :] + base[:, kind]
            out.append((rp, col.reshape(-1), np.tile(cf, len(hs)), nnz))
        return out

    pieces = []
    first_sides = compile_rows(template(True))
    pieces.append((tile(first_sides, [0]), 1))
    if H > 1:
        gen_sides = compile_rows(template(False))
        pieces.append((tile(gen_sides, list(range(1, H))), H - 1))

    def assemble(side):
        rps, cols, cfs, at = [np.zeros(1, dtype=np.int64)], [], [], 0
        for tiles, count in pieces:
            rp, col, cf, nnz = tiles[side]
            per = rp[1:]
            rps.append((at + (np.arange(count, dtype=np.int64) * nnz)[:, None] + per[None, :]).reshape(-1))
            cols.append(col)
            cfs.append(cf)
            at += count * nnz
        return np.concatenate(rps), np.concatenate(cols), np.concatenate(cfs)

    a_rp, a_col, a_cf = assemble(0)
    b_rp, b_col, b_cf = assemble(1)
    m = ROWS * H
    out_own = s_base + ROWS * (H - 1) + ROWS - 1       # the wire the public output replaces
    # C: row q of hash i defines wire s_base + ROWS i + q; the very last row defines
    # x5 = (out - M01 b - M02 c) / M00 with b, c the last-round x5 wires of lanes 1, 2
    c_col = s_base + np.arange(m, dtype=np.int64)
    inv00 = pow(M[0][0], R - 2, R)
    c_cf = np.zeros(m, dtype=np.int64)
    c_rp = np.arange(m + 1, dtype=np.int64)
    lb = s_base + ROWS * (H - 1) + last_x5[1]
    lc_ = s_base + ROWS * (H - 1) + last_x5[2]
    c_col = np.concatenate([c_col[:m - 1], np.asarray([1, lb, lc_], dtype=np.int64)])
    c_cf = np.concatenate([c_cf[:m - 1], np.asarray([cid(inv00), cid((R - M[0][1]) * inv00 % R),
                                                      cid((R - M[0][2]) * inv00 % R)], dtype=np.int64)])
    c_rp[m] = m + 2
    n_vars = out_own                                  # wires 0 .. out_own - 1 (the last own wire is wire 1)
    assert int(a_col.max()) < n_vars and int(b_col.max()) < n_vars

    # ---- witness: the textbook permutation, wire by wire
    w = [0] * n_vars
    w[0], w[2] = 1, 1
    h = 1
    for i in range(H):
        x = i + 2
        w[x_base + i] = x
        st = [0, h, x]
        base = s_base + ROWS * i
        q = 0
        for r in range(rounds):
            st = [(st[j] + rc[r * t + j]) % R for j in range(t)]
            full = r < r_f // 2 or r >= r_f // 2 + r_p
            order = ([2, 1, 0] if r == rounds - 1 else [0, 1, 2]) if full else [0]
            for j in order:
                x2 = st[j] * st[j] % R
                x4 = x2 * x2 % R
                x5 = x4 * st[j] % R
                if r == 0 and j == 0:                # (0 + c_0)^5: a constant, no wires
                    st[j] = x5
                    continue
                if base + q + 2 < n_vars:
                    w[base + q], w[base + q + 1], w[base + q + 2] = x2, x4, x5
                else:
                    w[base + q], w[base + q + 1] = x2, x4
                st[j] = x5
                q += 3
            st = [(M[j][0] * st[0] + M[j][1] * st[1] + M[j][2] * st[2]) % R for j in range(t)]
        h = st[0]
    w[1] = h
    tab = cc.fr_from_ints(table)
    u32 = lambda a: np.ascontiguousarray(a, dtype=np.uint32)
    A = cc.Csr(u32(a_rp), u32(a_col), tab[a_cf])
    B = cc.Csr(u32(b_rp), u32(b_col), tab[b_cf])
    Cm = cc.Csr(u32(c_rp), u32(c_col), tab[c_cf])
    mats = cc.ConstraintMatrices(2, n_vars - 1, m, A, B)
    return mats, (A, B, Cm), w, n_vars


def solve_r1cs_forward(r1cs, known):
    """witness of a circuit whose rows each define ONE new wire linearly in C (what circom emits
    for `x <== a*b`): propagate (A.w)(B.w) = C.w row by row.  Stands in for the WASM witness
    calculator (out of scope) on the reference's bench circuit."""
    import circom_compat_amd as cc
    n = r1cs.num_variables
    w = [None] * n
    for i, v in known.items():
        w[i] = v % R_MOD
    mats = []
    for m in (r1cs.a, r1cs.b, r1cs.c):
        mats.append((m.row_ptr.tolist(), m.col.tolist(), cc.fr_to_ints(m.coeff)))

    def lc(M, i):
        rp, col, val = M
        acc, unk = 0, None
        for j in range(rp[i], rp[i + 1]):
            x = w[col[j]]
            if x is None:
                if unk is not None:
                    return None, None, None
                unk = (col[j], val[j])
            else:
                acc = (acc + val[j] * x) % R_MOD
        return acc, unk, True

    for i in range(r1cs.num_constraints):
        a, ua, _ = lc(mats[0], i)
        b, ub, _ = lc(mats[1], i)
        c, uc, ok = lc(mats[2], i)
        if a is None or b is None or ua or ub or not ok:
            raise ValueError(f"row {i}: A or B has an unknown wire")
        if uc is None:
            if a * b % R_MOD != c:
                raise ValueError(f"row {i} is not satisfied")
            continue
        wire, cf = uc
        w[wire] = (a * b - c) * pow(cf, R_MOD - 2, R_MOD) % R_MOD
    if any(x is None for x in w):
        raise ValueError("unsolved wires remain")
    return w


def complex_shape_circuit(cc, num_variables, num_constraints):
    """the reference bench's circuit family (test-vectors/complex-circuit/complex-circuit.circom.template,
    swept by benches/groth16.rs:87-104 under feature bench-complex-all): b[0] = a*a, b[i] = b[i-1]^2 for
    i < NUM_VARIABLES, then NUM_CONSTRAINTS - NUM_VARIABLES repetitions of the last product constraint,
    c <== b[last] folded into the output wire.  Wire order and signs as circom emits them and as the
    shipped complex-circuit-10000-10000.r1cs has them: 0 = one, 1 = c, 2 = a, 3.. = b[]; row i is
    (-x_i) * (x_i) = (-x_{i+1}).  Input a = 3 (input.json)."""
    V, Cn = int(num_variables), int(num_constraints)
    assert 1 <= V <= Cn
    n_vars = V + 2
    one = cc.fr_from_ints([1])[0]
    minus1 = cc.fr_from_ints([R_MOD - 1])[0]
    src = np.concatenate([np.arange(V, dtype=np.uint32), np.full(Cn - V, V - 1, dtype=np.uint32)])  # x_i of row i
    wire = src + 2
    cwire = np.where(src + 1 == V, 1, src + 3).astype(np.uint32)                                     # x_{i+1}
    rp = np.arange(Cn + 1, dtype=np.uint32)
    A = cc.Csr(rp, wire, np.tile(minus1, (Cn, 1)))
    B = cc.Csr(rp, wire, np.tile(one, (Cn, 1)))
    Cm = cc.Csr(rp, cwire, np.tile(minus1, (Cn, 1)))
    xs = [3]
    for _ in range(V):
        xs.append(xs[-1] * xs[-1] % R_MOD)
    w = [1, xs[V]] + xs[:V]
    mats = cc.ConstraintMatrices(2, n_vars - 1, Cn, A, B)
    return mats, (A, B, Cm), w, n_vars


def complex_circuit(cc):
    """the reference bench's default circuit (benches/groth16.rs:87-108), input a = 3"""
    r1cs = cc.R1CS.from_file(os.path.join(ROOT, "tests", "golden", "complex-circuit-10000-10000.r1cs"))
    # circom wire order: 0 = one, 1 = output c, 2 = private input a, 3.. = b[]
    w = solve_r1cs_forward(r1cs, {0: 1, 2: 3})
    mats = r1cs.matrices()
    return mats, (r1cs.a, r1cs.b, r1cs.c), w, r1cs.num_variables


# ------------------------------------------------------------------------------------------------
def measure_pmc_traffic(args, k, budget_s=150.0):
    """roofline.traffic measured BY THIS RUN: bench.py re-executes itself for two
    proofs under `rocprofv3 --pmc` -- one pass for the L2's read requests to HBM by size, one for its
    write requests (separate passes and no trace domain, as MI355X_MICROARCH.md prescribes; sized by
    the TCC_EA0_*REQ_{32,64,128}B counters, which sidesteps the FETCH_SIZE x 2 correction of gfx950) --
    and parses the per-dispatch counters of the accumulation kernels (scripts/pmc_traffic.py).  Returns
    (record or None, note)."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 is not on PATH"
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import pmc_traffic
    out = tempfile.mkdtemp(prefix="g16_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", G16_NO_OVERLAP="1", G16_BENCH_NO_PIPELINE="1")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--no-pmc", "--log2", str(k), "--steps", "2",
             "--warmup", "0", "--cpu-log2", "0", "--workload", args.workload, "--window-bits", str(args.window_bits),
             "--planes", str(args.planes)]
    t0 = time.time()
    groups = ["TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B", "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"]
    try:
        for i, grp in enumerate(groups):
            left = budget_s - (time.time() - t0)
            if left < 20:
                return None, f"PMC passes stopped after {time.time() - t0:.0f} s (budget {budget_s:.0f} s)"
            cmd = (["rocprofv3", "--pmc"] + grp.split() + ["--kernel-include-regex", "k_bucket_accumulate", "-f", "csv",
                                                         "-d", os.path.join(out, f"p{i}"), "-o", f"p{i}", "--"] + child)
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=left)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc pass {i} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"
        rec = pmc_traffic.collect(out, k)
        if not rec.get("launches_averaged"):
            return None, "the PMC passes produced no counter rows for k_bucket_accumulate"
        rec["passes_s"] = round(time.time() - t0, 1)
        return rec, (f"THIS RUN: two rocprofv3 --pmc passes of `bench.py --steps 2` ({rec['passes_s']} s) after the timed "
                     f"region; read requests sized by TCC_EA0_RDREQ_32B/_64B/_128B, writes by TCC_EA0_WRREQ[_64B]; "
                     f"{rec['launches_averaged']} single-query launches averaged")
    except subprocess.TimeoutExpired:
        return None, f"PMC passes timed out (budget {budget_s:.0f} s)"
    except Exception as e:  # noqa: BLE001 -- the traffic figure is an extra: fall back to the stamped record
        return None, f"PMC passes failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(out, ignore_errors=True)

# ------------------------------------------------------------------------------------------------
def secondary_lines(args):
    """BASELINE configs[1] (2^20, G1 MSMs + NTT alone) and configs[4] (Poseidon chain 2^20) as part of the
    default line: one child run of this script each, after every timing leg of the parent
    (the parent's ctx is closed: the child has the device to itself), their lines condensed."""
    import subprocess
    out = {}

    def child(name, extra, pick, timeout):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-pmc", "--no-secondary"] + extra
        env = dict(os.environ)
        env["G16_BENCH_NO_PIPELINE"] = "1"
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
            line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"error": f"rc={r.returncode}: {r.stderr.strip()[-300:]}"}
                return
            d = json.loads(line[-1])
            rec = pick(d)
            rec["run_s"] = round(time.time() - t0, 1)
            rec["command"] = "bench.py " + " ".join(extra)
            out[name] = rec
        except Exception as e:  # noqa: BLE001 -- an extra, never the headline
            out[name] = {"error": f"{type(e).__name__}: {e}"}

    def bytes_equal(d, kk):
        return d["parity"].get("bit_identical_to_cpu_at_2^%d" % kk)

    child("parts_k20", ["--mode", "parts", "--log2", "20", "--steps", "3", "--warmup", "1", "--cpu-log2", "18",
                        "--cpu-budget", "20"],
          lambda d: {"what": "BASELINE configs[1]: synthetic 2^20 chain, every stage alone on one stream (G16_NO_OVERLAP)",
                     "witness_map_ms": d["parts_ms"]["witness_map_ms"],
                     "g1_msm_ms": {q: d["parts_ms"]["msm_%s_ms" % q] for q in ("A", "B1", "L", "H")},
                     "g2_msm_ms": d["parts_ms"]["msm_B2_ms"], "ms_per_step_one_stream": d["ms_per_step"],
                     "proof_verifies": d["parity"]["proof_verifies"], "bytes_equal": bytes_equal(d, 20)}, 240)
    child("poseidon20", ["--workload", "poseidon", "--log2", "20", "--steps", "10", "--warmup", "2", "--cpu-log2", "18",
                         "--cpu-budget", "20"],
          lambda d: {"what": "BASELINE configs[4]: Poseidon(2) hash chain, circomlib parameters, R1CS not circom-compiled",
                     "ms_per_step": d["ms_per_step"], "value": d["value"],
                     "ms_per_step_pcie_inclusive": d["ms_per_step_pcie_inclusive"],
                     "num_constraints": d["config"]["num_constraints"], "n_vars": d["config"]["n_vars"],
                     "sparse_b": d["config"]["msm"].get("sparse_b"),
                     "verifies": d["parity"]["proof_verifies"], "bytes_equal": bytes_equal(d, 20),
                     "cpu_constraints_per_s": (d.get("cpu_baseline") or {}).get("value")}, 300)
    return out


class ClockSampler:
    """sclk / package power of one GPU sampled on a host thread while proofs run (a slow box
    must be visible in the record).  sysfs hwmon (freq1_input in Hz, power1_average / power1_input in
    microwatts: one read each, ~20 us) when the box exposes it, `rocm-smi --showclocks --showpower
    --json` (~0.3 s per sample) otherwise.  mark(label) starts a new labelled interval."""

    def __init__(self, ordinal, period=0.004):
        import glob
        import threading
        self.samples, self.label, self.source = [], "warmup", None
        self._stop = threading.Event()
        self._th = None
        if ordinal is None:
            return
        # The host may expose more GPUs in sysfs than this container can open (one per tenant): the card
        # that IS HIP device `ordinal` is the one whose PCI address hipDeviceGetPCIBusId reports.
        hwmon = None
        try:
            want = None
            try:  # the runtime torch already holds: no second copy of libamdhip64 in the process
                import torch
                pr = torch.cuda.get_device_properties(int(ordinal))
                want = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            except Exception:  # noqa: BLE001 -- older torch: ask the HIP runtime the process has loaded
                import ctypes
                buf = ctypes.create_string_buffer(64)
                hip = ctypes.CDLL("libamdhip64.so")
                if hip.hipDeviceGetPCIBusId(buf, 64, int(ordinal)) == 0:
                    want = buf.value.decode().lower()
            for c in glob.glob("/sys/class/drm/card[0-9]*") if want else []:
                if os.path.basename(os.path.realpath(os.path.join(c, "device"))).lower() == want:
                    hw = sorted(glob.glob(os.path.join(c, "device", "hwmon", "hwmon*")))
                    if hw and os.path.exists(os.path.join(hw[0], "freq1_input")):
                        hwmon = hw[0]
                    break
        except Exception:  # noqa: BLE001 -- no sysfs view of the device: rocm-smi below
            hwmon = None
        self._freq = self._pow = None
        if hwmon:
            self._freq = os.path.join(hwmon, "freq1_input")
            for name in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(hwmon, name)):
                    self._pow = os.path.join(hwmon, name)
                    break
            self.source = f"sysfs {self._freq}" + (f" + {os.path.basename(self._pow)}" if self._pow else "")
        else:
            import shutil
            if not shutil.which("rocm-smi"):
                return
            self._ordinal, period = ordinal, 0.05
            self.source = "rocm-smi --showclocks --showpower --json"
        self._period = period
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()

    def _read(self):
        if self._freq:
            try:
                f = int(open(self._freq).read()) / 1e6
                pw = int(open(self._pow).read()) / 1e6 if self._pow else None
                return f, pw
            except (OSError, ValueError):
                return None
        import subprocess
        try:
            r = subprocess.run(["rocm-smi", "-d", str(self._ordinal), "--showclocks", "--showpower", "--json"],
                               capture_output=True, text=True, timeout=5)
            card = next(iter(json.loads(r.stdout).values()))
            f = pw = None
            for key, v in card.items():
                if key.startswith("sclk clock speed"):
                    f = float(str(v).strip("()Mhz "))
                elif "Power (W)" in key and pw is None:
                    pw = float(v)
            return (f, pw) if f else None
        except Exception:  # noqa: BLE001 -- a sampler never fails the bench
            return None

    def _run(self):
        while not self._stop.is_set():
            x = self._read()
            if x:
                self.samples.append((self.label, x[0], x[1]))
            self._stop.wait(self._period)

    def mark(self, label):
        self.label = label

    def stop(self):
        self._stop.set()
        if self._th:
            self._th.join(timeout=6)
            self._th = None

    def summary(self):
        def med(v):
            v = sorted(v)
            return v[len(v) // 2] if v else None
        out = {"source": self.source}
        for label in ("timed", "after"):
            fs = [f for l_, f, _ in self.samples if l_ == label]
            ps = [pw for l_, _, pw in self.samples if l_ == label and pw is not None]
            out[label] = {"samples": len(fs), "sclk_mhz_median": med(fs), "sclk_mhz_min": min(fs) if fs else None,
                          "sclk_mhz_max": max(fs) if fs else None, "power_w_median": med(ps)}
        return out


# ------------------------------------------------------------------------------------------------
def setup_prover(cc, torch, args, mode, rank, local_rank, n_gpus, world, one_gpu, A, B, Cm, n_vars, tox,
                 mats, w, rs, w_ints):
    """Key + ctx + resident witness for one launch mode.  Returns (active, pk, mats, m, prover, w_dev,
    w_ptr, zkey_path); pk is None on ranks that do not prove (in-library mode: every rank but 0)."""
    active = mode != "inlib" or rank == 0     # ranks that prove (idle ranks never touch a GPU)
    if not active:
        return False, None, None, None, None, None, None, None
    torch.cuda.set_device(local_rank if mode != "inlib" else (0 if not one_gpu else local_rank))
    dev0 = local_rank if mode != "inlib" else (local_rank if one_gpu else 0)
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox, device=dev0)
    zkey_path = None
    if args.workload != "chain":
        # through the file format: snarkjs-layout .zkey written, mapped and parsed back (read_zkey)
        import tempfile
        zkey_path = os.path.join(tempfile.gettempdir(), f"g16_bench_{os.getpid()}.zkey")
        cc.write_zkey(zkey_path, pk, mats)
        pk, mats = cc.read_zkey(zkey_path)
    kw = dict(window_bits=args.window_bits, planes=args.planes, shard=args.shard)
    if mode == "single":
        kw["tables"] = {"auto": 0, "on": 1, "off": -1}[args.tables]
    if mode == "single":
        prover = cc.Prover(pk, mats, device=local_rank, **kw)
    elif mode == "inlib":
        if os.environ.get("G16_BENCH_FAIL_INLIB"):   # exercises the fallback on a 1-GPU box
            raise RuntimeError("G16_BENCH_FAIL_INLIB is set")
        devices = [dev0] * n_gpus if one_gpu else list(range(n_gpus))
        prover = cc.Prover(pk, mats, devices=devices, **kw)
    else:
        dist_wm = os.environ.get("G16_BENCH_DIST_WM", "1") != "0"
        prover = cc.Prover(pk, mats, device=local_rank, rank=rank, world=world, dist_wm=dist_wm, **kw)
    w_dev = None
    if mode == "inlib":
        # resident on EVERY device before the timed region, as at N = 1 (g16_witness_upload): the ctx
        # would otherwise peer-broadcast 32 N bytes from the first device inside every proof
        w_ptr = prover.upload_witness(w)
        if world > 1:
            # first proof of a path no 1-GPU box can exercise: it has to verify (GPU verifier; the
            # checker legs below still run on the timed proof) or the per-rank path takes over
            trial = prover.prove_dev(rs[0], rs[1], w_ptr)
            if not cc.verify_batch(pk.vk, [trial], [[w_ints[1]]], device=dev0)[0]:
                prover.close()
                raise RuntimeError("the first in-library proof does not verify")
    else:
        w_dev = torch.from_numpy(w.view(np.int64)).to(f"cuda:{torch.cuda.current_device()}")
        w_ptr = w_dev.data_ptr()
    return True, pk, mats, mats.num_constraints, prover, w_dev, w_ptr, zkey_path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2", type=int, default=22, help="log2 of the domain (m = 2^k - 2 constraints)")
    ap.add_argument("--workload", choices=["chain", "dense-skewed", "poseidon", "poseidon-shaped", "complex-circuit"], default="chain")
    ap.add_argument("--mode", choices=["prove", "parts"], default="prove")
    ap.add_argument("--cpu-log2", type=int, default=17, help="probe size of the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-own", action="store_true",
                    help="CPU baseline = ONE proof of the bench's own inputs whatever it costs (no size climbing): "
                         "how the sizes beyond --cpu-budget are byte-compared (scripts/r4_chain26.sh)")
    ap.add_argument("--cpu-budget", type=float, default=60.0,
                    help="seconds of CPU work the baseline sample may take")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--planes", type=int, default=0)
    ap.add_argument("--tables", choices=["auto", "on", "off"], default="auto",
                    help="g16_options.fixed_tables: small keys through fixed-base tables (auto = the library's rule)")
    ap.add_argument("--shard", choices=["auto", "points", "buckets"], default="auto",
                    help="N > 1: MSMs cut by point range or by bucket range (auto = points)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc passes that measure roofline.traffic (N = 1, default workload)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block of the default N = 1 line (BASELINE configs[1] and configs[4], one child run each)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = max(args.gpus, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    visible = torch.cuda.device_count()
    # G16_BENCH_BACKEND=gloo G16_BENCH_DEVICE=0: run the N > 1 code path with every rank on ONE GPU
    # -- a functional check of this script on a 1-GPU box, never a measurement
    backend = os.environ.get("G16_BENCH_BACKEND", "nccl")
    one_gpu = backend != "nccl"
    mode = os.environ.get("G16_BENCH_MODE", "")
    if n_gpus == 1:
        mode = "single"
    elif not mode:
        mode = "inlib" if (visible >= n_gpus or one_gpu or world == 1) else "ranks"
    if mode == "ranks" and world != n_gpus:
        raise SystemExit("mode 'ranks' needs torch.distributed.run with one process per GPU")
    if one_gpu:
        local_rank = int(os.environ.get("G16_BENCH_DEVICE", 0))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL's internal stream with high priority: a normal-priority stream would share a hardware
        # queue with the ctx's MSM streams and the all-to-all would run behind them
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if mode == "ranks" and backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:  # in-library mode: the idle ranks only keep the barriers -- on the CPU, off the GPUs
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import circom_compat_amd as cc
    from circom_compat_amd import _binding
    lib_path = _binding.load().path
    if os.environ.get("G16_AMD_LIB"):
        if "emu" in os.path.basename(lib_path):
            raise SystemExit("G16_AMD_LIB points at the emulator build: not a measurement")
        print(f"bench.py: product library overridden by G16_AMD_LIB={lib_path}", file=sys.stderr)
    if args.mode == "parts":
        os.environ["G16_NO_OVERLAP"] = "1"    # one stream: every stage timer is an uncontended time

    k = args.log2
    t_setup = time.time()
    zkey_path = None
    if args.workload == "chain":
        mats, (A, B, Cm), w_ints, n_vars = chain_circuit(cc, k)
        desc = f"synthetic squaring-chain R1CS, 2^{k}-2 constraints"
    elif args.workload == "dense-skewed":
        mats, (A, B, Cm), w_ints, n_vars = dense_skewed_circuit(cc, k)
        desc = (f"SYNTHETIC substitute for BASELINE configs[4] (not circom-generated): dense-rows R1CS (3-term A / "
                f"2-term B rows), 2^{k}-2 constraints, skewed witness, "
                "key through the snarkjs .zkey writer + read_zkey")
    elif args.workload == "poseidon":
        mats, (A, B, Cm), w_ints, n_vars = poseidon_chain_circuit(cc, k)
        desc = (f"Poseidon(2) hash chain (circomlib parameters: Grain-LFSR constants + Cauchy MDS, t = 3, R_F = 8, "
                f"R_P = 57; KAT-pinned: h_1 = circomlibjs poseidon([1, 2])), R1CS NOT circom-compiled (no circom / snarkjs / "
                f"ptau offline): {mats.num_constraints // 240} hashes x 240 S-box rows = {mats.num_constraints} "
                f"constraints in the 2^{k} domain, {n_vars} wires, linear layers folded into rows of up to 61 "
                f"full-width terms ({len(A.col) / mats.num_constraints:.1f} / {len(B.col) / mats.num_constraints:.1f} nnz per "
                "A / B row), key through the snarkjs .zkey writer + read_zkey (Coefs path)")
    elif args.workload == "poseidon-shaped":
        mats, (A, B, Cm), w_ints, n_vars = poseidon_shaped_circuit(cc, k)
        desc = (f"SYNTHETIC Poseidon-shaped SUBSTITUTE for BASELINE configs[4] (NOT circom-generated: seeded constants, "
                f"no circom / snarkjs / ptau offline): hash-chain R1CS (width 3, 8 + 57 rounds, x^5 as 3 rows, 4-term linear "
                f"combinations with full-width MDS / round constants, uniform witness), 2^{k}-2 constraints, "
                "key through the snarkjs .zkey writer + read_zkey")
    else:
        mats, (A, B, Cm), w_ints, n_vars = complex_circuit(cc)
        k = (mats.num_constraints + 2 - 1).bit_length()
        desc = "complex-circuit-10000-10000.r1cs (the reference bench's circuit, benches/groth16.rs:87-108), a = 3"
    m = mats.num_constraints
    frac01 = sum(1 for x in w_ints if x in (0, 1)) / len(w_ints)
    rng = random.Random(k)
    tox = [rng.randrange(1, R_MOD) for _ in range(5)]
    rs_rng = random.Random(1000 + k)
    r, s = rs_rng.randrange(R_MOD), rs_rng.randrange(R_MOD)
    rs = cc.fr_from_ints([r, s])
    w = cc.fr_from_ints(w_ints)

    # Under torch.distributed.run BOTH N > 1 paths are timed by default (round 5): first the in-library
    # ctx (rank 0 drives every device: peer copies over xGMI, no collective library), then one ctx per
    # process with RCCL all_to_all / all_gather in between -- north_star's wording -- on the same key,
    # witness and (r, s); `value` is the faster, `value_inlib` / `value_rccl` / `rccl_ranks` say which
    # ran how.  If the in-library ctx cannot be built or its first proof does not verify, only the
    # per-process path is timed and the line says so (config.parallelism, fallback_reason).
    pg_nccl = mode == "ranks" and backend == "nccl"   # the default process group lives on the GPUs
    both = world > 1 and mode == "inlib" and not os.environ.get("G16_BENCH_MODE")
    st = {"grp": None, "fallback_reason": None, "rccl_ranks": 0}

    def barrier():
        if dist:
            dist.barrier()

    def measure(mode):
        """ctx + resident witness for one launch mode, W untimed proofs, EXACTLY K timed proofs between
        barrier + synchronize on both sides, max over ranks."""
        trial = mode == "inlib" and world > 1
        try:
            (active, pk_, mats_, m_, prover, w_dev, w_ptr, zk) = setup_prover(
                cc, torch, args, mode, rank, local_rank, n_gpus, world, one_gpu, A, B, Cm, n_vars, tox,
                mats, w, rs, w_ints)
            ok, why = True, ""
        except Exception as e:  # noqa: BLE001 -- anything the trial throws selects the per-process path
            if not trial:
                raise
            ok, why = False, f"{type(e).__name__}: {e}"
        if trial:
            flag = [ok, why]
            dist.broadcast_object_list(flag, src=0)
            ok, why = flag
            if not ok:
                if rank == 0:
                    print(f"bench.py: in-library multi-device ctx failed ({why}); only the per-process path "
                          "(RCCL collectives) is timed", file=sys.stderr)
                st["fallback_reason"] = why
                import gc
                gc.collect()                      # a half-built in-library ctx frees its HBM here
                return None
        dev = f"cuda:{torch.cuda.current_device()}" if active else "cpu"
        if active:
            torch.cuda.synchronize()

        # ---- per-rank path: RCCL collectives on a torch stream the library orders itself against
        if mode == "ranks":
            grp = st["grp"]
            xs = torch.cuda.Stream(priority=-1)   # high priority: shares a hardware queue with the aux stream, not with the MSM streams
            prover.set_exchange_stream(xs.cuda_stream)
            if not prover.dist_wm:
                raise SystemExit("mode 'ranks' runs the fully sharded prover (G16_BENCH_DIST_WM=1)")
            nbytes = prover.exchange_bytes()
            send = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            recv = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            part_t = cc.device_tensor(prover.partial_buffer(), 1024, dev)
            gath_t = cc.device_tensor(prover.gather_buffer(), world * 1024, dev)
            if backend == "nccl":
                st["rccl_ranks"] = dist.get_world_size(grp)

            def exchange():
                with torch.cuda.stream(xs):
                    if backend == "nccl":
                        dist.all_to_all_single(recv, send, group=grp)
                    else:
                        xs.synchronize()
                        hs, hr = send.cpu(), torch.empty(nbytes, dtype=torch.uint8)
                        dist.all_to_all_single(hr, hs)
                        recv.copy_(hr)

            def gather():
                with torch.cuda.stream(xs):
                    if backend == "nccl":
                        dist.all_gather_into_tensor(gath_t, part_t, group=grp)
                    else:
                        xs.synchronize()
                        parts = [torch.empty(1024, dtype=torch.uint8) for _ in range(world)]
                        dist.all_gather(parts, part_t.cpu())
                        gath_t.copy_(torch.cat(parts))

        def step():
            if mode in ("single", "inlib"):
                return prover.prove_dev(rs[0], rs[1], w_ptr)
            prover.dist_phase1(rs[0], rs[1], w_ptr, send.data_ptr())
            exchange()
            prover.dist_phase2(recv.data_ptr(), send.data_ptr())
            exchange()
            prover.dist_phase3_dev(recv.data_ptr())
            gather()
            return prover.prove_finish_dev(rs[0], rs[1])

        proof = None
        for _ in range(args.warmup):
            if active:
                proof = step()
        if active:
            prover.set_profiling(True)
        sampler = ClockSampler(local_rank if active else None) if rank == 0 else None
        barrier()
        if active:
            torch.cuda.synchronize()
        if sampler:
            sampler.mark("timed")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if active:
                proof = step()
        if active:
            torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        if sampler:
            sampler.mark("after")
        if dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if pg_nccl else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return dict(mode=mode, elapsed=elapsed, proof=proof, prover=prover, active=active, pk=pk_, mats=mats_,
                    m=m_, w_dev=w_dev, w_ptr=w_ptr, zkey=zk, dev=dev, step=step, sampler=sampler)

    def release(res):
        if res and res["active"]:
            if res["sampler"]:
                res["sampler"].stop()
            res["prover"].set_profiling(False)
            res["prover"].close()
            res["prover"] = None
            if res["zkey"] and os.path.exists(res["zkey"]):
                os.remove(res["zkey"])
            import gc
            gc.collect()

    def rccl_group():
        if backend == "nccl" and not pg_nccl:
            torch.cuda.set_device(local_rank)
            st["grp"] = dist.new_group(backend="nccl")

    res_inlib = res_ranks = None
    if both:
        res_inlib = measure("inlib")
        rccl_group()
        if res_inlib is not None and rank == 0:
            keep = dict(elapsed=res_inlib["elapsed"], proof=res_inlib["proof"].raw)
            release(res_inlib)
            res_inlib = keep
        res_ranks = measure("ranks")
        res = res_ranks
        mode = "ranks"
    else:
        res = measure(mode)
        if res is None:                       # forced in-library mode that failed under torch.distributed.run
            rccl_group()
            mode = "ranks"
            res = res_ranks = measure("ranks")
    fallback_reason = st["fallback_reason"]
    active, prover, proof, elapsed = res["active"], res["prover"], res["proof"], res["elapsed"]
    w_dev, w_ptr, zkey_path, dev, sampler = res["w_dev"], res["w_ptr"], res["zkey"], res["dev"], res["sampler"]
    if res["pk"] is not None:
        pk, mats, m = res["pk"], res["mats"], res["m"]
    t_setup = time.time() - t_setup
    if not active or rank != 0:
        if dist:
            dist.barrier()          # rank 0's checker legs run while the others wait here
            dist.destroy_process_group()
        return
    if args.pmc_child:                # the profiled child of measure_pmc_traffic(): the proofs above are all it is for
        if zkey_path and os.path.exists(zkey_path):
            os.remove(zkey_path)
        return
    both_vals = None
    if both:
        both_vals = {"value_rccl": m * args.steps / res_ranks["elapsed"], "ms_per_step_rccl": res_ranks["elapsed"] / args.steps * 1e3,
                     "value_inlib": None, "ms_per_step_inlib": None, "rccl_ranks": st["rccl_ranks"]}
        if isinstance(res_inlib, dict):
            both_vals["value_inlib"] = m * args.steps / res_inlib["elapsed"]
            both_vals["ms_per_step_inlib"] = res_inlib["elapsed"] / args.steps * 1e3
            both_vals["inlib_and_rccl_proofs_identical"] = bool(res_inlib["proof"] == proof.raw)
            if res_inlib["elapsed"] < elapsed:
                elapsed = res_inlib["elapsed"]       # `value` = the faster path; the checker legs below run on the RCCL path's proof
                both_vals["value_is"] = "in-library"
            else:
                both_vals["value_is"] = "rccl"
    stages = prover.stage_times()
    prover.set_profiling(False)
    info = prover.info()

    # ---- the SURVEY 8(d) step: host witness (page-locked staging buffer of the ctx) -> H2D -> proof
    host_ms = None
    if mode in ("single", "inlib"):
        host_w = prover.witness_host_buffer()
        host_w[:] = w
        prover.prove(rs[0], rs[1], host_w)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            p_host = prover.prove(rs[0], rs[1], host_w)
        host_ms = (time.perf_counter() - t1) / args.steps * 1e3
        assert p_host.raw == proof.raw

    # ---- throughput mode (never `value`): two proofs in flight on two ctxs that share the point planes
    # (g16_ctx_create_sibling), one host thread each -- the front of one proof (digit sort, witness map)
    # runs under the bucket reductions / finalisation of the other
    pipelined = None
    kw = dict(window_bits=args.window_bits, planes=args.planes, tables={"auto": 0, "on": 1, "off": -1}[args.tables])
    if mode == "single" and args.mode == "prove" and not os.environ.get("G16_BENCH_NO_PIPELINE"):
        import threading
        try:
            sib = cc.Prover(pk, mats, device=local_rank, sibling_of=prover, **kw)
            per = max(args.steps // 2, 2)
            outs = [None, None]

            def worker(i, p):
                torch.cuda.set_device(local_rank)
                for _ in range(per):
                    outs[i] = p.prove_dev(rs[0], rs[1], w_ptr)

            sib.prove_dev(rs[0], rs[1], w_ptr)                     # warm-up of the sibling
            torch.cuda.synchronize()
            ths = [threading.Thread(target=worker, args=(i, p)) for i, p in enumerate((prover, sib))]
            t1 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            assert outs[0].raw == proof.raw and outs[1].raw == proof.raw
            pipelined = {"value": m * 2 * per / dt, "unit": "constraints/s", "ms_per_proof": dt / (2 * per) * 1e3,
                         "proofs": 2 * per, "in_flight": 2,
                         "what": "two ctxs sharing the point planes (g16_ctx_create_sibling), one host thread each; "
                                 "both produce the timed proof's bytes"}
            # the same from HOST witnesses (g16_prove, what a create_proof caller hands over): each
            # ctx's H2D runs under the other ctx's proof
            hosts = [prover.witness_host_buffer(), sib.witness_host_buffer()]
            for hb in hosts:
                hb[:] = w

            def worker_host(i, p):
                torch.cuda.set_device(local_rank)
                for _ in range(per):
                    outs[i] = p.prove(rs[0], rs[1], hosts[i])

            sib.prove(rs[0], rs[1], hosts[1])
            torch.cuda.synchronize()
            ths = [threading.Thread(target=worker_host, args=(i, p)) for i, p in enumerate((prover, sib))]
            t1 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            assert outs[0].raw == proof.raw and outs[1].raw == proof.raw
            pipelined["ms_per_proof_from_host_witness"] = dt / (2 * per) * 1e3
            sib.close()
        except Exception as e:  # noqa: BLE001 -- an extra, never the headline
            pipelined = {"error": f"{type(e).__name__}: {e}"}

    # ---- config 2: the witness map and every MSM on their own (device-resident operands)
    parts = None
    if args.mode == "parts" and mode == "single":
        h_dev = torch.empty((info["domain_size"], 4), dtype=torch.int64, device=dev)

        def timed(fn, reps=3):
            fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                fn()
            return (time.perf_counter() - t1) / reps * 1e3

        wptr = w_ptr
        parts = {"witness_map_ms": timed(lambda: prover.witness_map_dev(wptr, h_dev.data_ptr())),
                 "msm_A_ms": timed(lambda: prover.msm_g1_dev(0, wptr + 32, n_vars - 1)),
                 "msm_B1_ms": timed(lambda: prover.msm_g1_dev(1, wptr + 32, n_vars - 1)),
                 "msm_L_ms": timed(lambda: prover.msm_g1_dev(2, wptr + 64, n_vars - 2)),
                 "msm_H_ms": timed(lambda: prover.msm_g1_dev(3, h_dev.data_ptr(), info["domain_size"])),
                 "msm_B2_ms": timed(lambda: prover.msm_g2_dev(wptr + 32, n_vars - 1)),
                 "note": "each MSM = its own digit sort + bucket accumulation + reduction + affine result; "
                         "the full prove shares one sort among A, B1, L, B2 and overlaps the witness map"}

    clock = None
    if sampler:
        sampler.stop()
        clock = sampler.summary()
        clock["note"] = ("`timed` = samples inside the timed region; `after` = the PCIe-inclusive and two-in-flight legs that "
                         "follow it (continuous proving); DESIGN.md section 5: ~2.07-2.12 GHz under 2^22 proofs on a normal box")

    # ---------------- parity (outside the timed region; oracle = checker only) ----------------
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bn254_ref as o
    import helpers as H
    vk = dict(alpha_g1=o.g1_from_bytes(bytes(pk.vk.alpha_g1)), beta_g2=o.g2_from_bytes(bytes(pk.vk.beta_g2)),
              gamma_g2=o.g2_from_bytes(bytes(pk.vk.gamma_g2)), delta_g2=o.g2_from_bytes(bytes(pk.vk.delta_g2)),
              ic=[o.g1_from_bytes(bytes(x)) for x in pk.vk.gamma_abc_g1])
    verified = bool(o.verify_proof(vk, [w_ints[1]], H.proof_from_bytes(proof.raw)))
    rejected_wrong = not o.verify_proof(vk, [(w_ints[1] + 1) % R_MOD], H.proof_from_bytes(proof.raw))
    parity = {"proof_verifies": verified, "wrong_public_input_rejected": bool(rejected_wrong)}

    # ---------------- CPU baseline: the SAME (pk, r, s, w) when it fits the time bound -----------
    # A 2^cpu_log2 probe proof sizes the sample: the bench's own inputs are proved on the CPU (and
    # byte-compared with the GPU proof) when one such proof is estimated to fit --cpu-budget seconds;
    # otherwise the largest smaller chain circuit that does is timed and compared instead.
    cpu = None
    if args.cpu_log2 > 0:
        import cpu_ref
        # the CPU restatement's thread pool follows what the host GRANTS (a cgroup quota below the visible CPU
        # count throttles surplus threads): 2 x the granted cores, `cores` = the threads used, the grant beside it
        # (cpu_ref.lib() does the sizing at load: G16_CPU_THREADS overrides)
        granted, grant_how = cpu_ref.host_cpu_grant()
        omp_default = cpu_ref.omp_default_threads()

        def cpu_prove(pk_c, mats_c, wc, reps):
            t_cpu, out = 0.0, None
            for _ in range(reps):
                t1 = time.perf_counter()
                out = cpu_ref.prove(pk_c, mats_c, rs[0:1].copy(), rs[1:2].copy(), wc)
                t_cpu += time.perf_counter() - t1
            return out, t_cpu

        def small_case(kc):
            mats_c, (Ac, Bc, Cc), wc_ints, nvc = chain_circuit(cc, kc)
            pk_c = cc.trapdoor_setup(Ac, Bc, Cc, nvc, 1, tox, device=torch.cuda.current_device())
            wc = cc.fr_from_ints(wc_ints)
            pr_c = cc.Prover(pk_c, mats_c, device=torch.cuda.current_device())
            gpu_small = pr_c.prove(rs[0], rs[1], wc)
            pr_c.close()
            return pk_c, mats_c, wc, gpu_small.raw

        def ark_windows(nscal):          # ark-ec msm_bigint: c = ln-ish(n) + 2, one rayon task per window
            if nscal < 32:
                c_ = 3
            else:
                c_ = (nscal - 1).bit_length() * 69 // 100 + 2
            return -(-254 // c_)

        samples = []
        if args.cpu_own and n_gpus == 1:
            # the bench's own inputs, once, whatever it costs (sizes beyond --cpu-budget)
            pk_c, mats_c, wc, _ = small_case(min(14, k))
            cpu_ref.prove(pk_c, mats_c, rs[0:1].copy(), rs[1:2].copy(), wc)  # thread pool warm-up
            out, t_prev = cpu_prove(pk, mats, w, 1)
            parity["bit_identical_to_cpu_at_2^%d" % k] = bool(out == proof.raw)
            k_prev, m_prev, own, spent = k, m, True, t_prev
            samples = [t_prev]
        else:
            # N > 1: only a small byte comparison (torchrun pins OMP_NUM_THREADS=1: the timed baseline
            # belongs to the N = 1 line)
            kp = min(args.cpu_log2, k) if n_gpus == 1 else min(args.cpu_log2, k, 14)
            pk_c, mats_c, wc, gpu_small = small_case(kp)
            if n_gpus == 1:
                cpu_ref.prove(pk_c, mats_c, rs[0:1].copy(), rs[1:2].copy(), wc)  # thread pool warm-up
            out, t_probe = cpu_prove(pk_c, mats_c, wc, 1)
            parity["bit_identical_to_cpu_at_2^%d" % kp] = bool(out == gpu_small)
            # climb two sizes at a time while the next size is estimated to fit what is left of the
            # budget (small probes over-estimate: fixed per-proof costs, windows shrink with n), ending
            # on the bench's OWN inputs when they fit
            spent, t_prev, same_prev = t_probe, t_probe, bool(out == gpu_small)
            k_prev = min(kp, k - 1)      # a probe of the bench's own size still leads to its own inputs
            m_prev, own = mats_c.num_constraints, False
            last = (pk_c, mats_c, wc)
            while n_gpus == 1 and k_prev < k:
                nxt = min(k, k_prev + 2)
                est = t_prev * (2 ** (nxt - k_prev)) * 1.1
                if spent + est > args.cpu_budget:
                    break
                if nxt == k:
                    out, t_cpu = cpu_prove(pk, mats, w, 1)
                    same_prev, m_prev, own = bool(out == proof.raw), m, True
                    last = (pk, mats, w)
                else:
                    pk_c, mats_c, wc, gpu_small = small_case(nxt)
                    out, t_cpu = cpu_prove(pk_c, mats_c, wc, 1)
                    same_prev, m_prev = bool(out == gpu_small), mats_c.num_constraints
                    last = (pk_c, mats_c, wc)
                parity["bit_identical_to_cpu_at_2^%d" % nxt] = same_prev
                spent, t_prev, k_prev = spent + t_cpu, t_cpu, nxt
            # a distribution, not one shot (the reference's criterion bench reports one,
            # benches/groth16.rs:69-84): up to two more proofs of the final sample while they fit the budget
            samples = [t_prev]
            while n_gpus == 1 and len(samples) < 3 and spent + 1.05 * min(samples) <= args.cpu_budget:
                _, t_more = cpu_prove(*last, 1)
                samples.append(t_more)
                spent += t_more
            t_prev = sorted(samples)[len(samples) // 2]     # median (of 1, 2 -> the larger, or 3)
        if n_gpus > 1:
            cpu = None      # the timed baseline belongs to the N = 1 line (torchrun pins OMP_NUM_THREADS=1)
        else:
            what = (f"the bench's own inputs: {desc}" if own else
                    f"the 2^{k_prev}-constraint squaring-chain circuit (the next size towards 2^{k} was "
                    f"estimated beyond --cpu-budget = {args.cpu_budget:.0f} s)")
            nwin = ark_windows(max(m_prev, 2))
            sample = (f"{len(samples)} proof(s) of {what}: {', '.join('%.2f' % t for t in samples)} s, "
                      f"value = constraints / median")
            cpu = {"value": m_prev / t_prev, "unit": "constraints/s", "samples_s": [round(t, 3) for t in samples],
                   "min_s": round(min(samples), 3), "median_s": round(t_prev, 3)}
        if cpu and n_gpus == 1:
            # the conservative column: the same proof with every MSM window cut into chunks of
            # bases so that windows x chunks tasks keep ALL host threads busy (g16cpu_set_msm_chunks) -- ark-ec's
            # msm_bigint runs one rayon task per window (`value`), which leaves most of a many-core host idle
            try:
                T = cpu_ref.max_threads()
                nw = ark_windows(max(m_prev, 2))
                # candidates (window bits, chunks per window): ark-ec's window cut into chunks, and smaller windows
                # whose buckets stay in a core's cache when every thread fills its own (2^16 buckets x 96 B x 128
                # threads do not); the fastest on a 2^18 probe is timed on the sample's own inputs
                cands = [(0, max(2, -(-T // nw))), (0, 2), (0, 4)]
                cands += [(c_, max(1, -(-T // (-(-254 // c_))))) for c_ in (12, 13, 14, 15)]
                # the shape ark-ec 0.5's msm_bigint_wnaf_parallel is RECALLED to have (the crate is not vendored:
                # unverifiable offline): threads / 2 chunks of bases, every chunk a window-parallel MSM with the
                # window of ITS size, the chunk sums added
                ch_ark = max(1, T // 2)
                n_ch = max(32, -(-max(m_prev, 2) // ch_ark))
                cands.append(((n_ch - 1).bit_length() * 69 // 100 + 2, ch_ark))
                pk_p, mats_p, wc_p, _ = small_case(min(18, k))
                probe = []
                for c_, ch_ in cands:
                    cpu_ref.set_msm_window(c_)
                    cpu_ref.set_msm_chunks(ch_)
                    _, t_p = cpu_prove(pk_p, mats_p, wc_p, 1)
                    probe.append((t_p, c_, ch_))
                t_best, c_best, ch_best = min(probe)
                cpu_ref.set_msm_window(c_best)
                cpu_ref.set_msm_chunks(ch_best)
                ac_case = (pk, mats, w) if own else last
                out_ac, t_ac = cpu_prove(*ac_case, 1)
                cpu["value_all_cores"] = m_prev / t_ac
                cpu["all_cores"] = {"seconds": round(t_ac, 3), "window_bits": c_best or "ark-ec's", "msm_chunks_per_window": ch_best,
                                    "probe_2^%d_s" % min(18, k): {"c=%s x %d chunks" % (c_ or "ark", ch_): round(t_, 3) for t_, c_, ch_ in probe},
                                    "bit_identical_to_gpu": bool(out_ac == proof.raw) if own else None,
                                    "what": "same proof with the MSMs re-shaped for a many-core host: windows cut into chunks of "
                                            "bases (one task per window and chunk), window bits chosen on a probe -- NOT how "
                                            "ark-ec 0.5 schedules msm_bigint (one rayon task per window: `value`); the faster of "
                                            "the two CPU figures is the conservative bound quoted as gpu_over_cpu_all_cores"}
            except Exception as e:  # noqa: BLE001 -- an extra column
                cpu["all_cores"] = {"error": f"{type(e).__name__}: {e}"}
            finally:
                cpu_ref.set_msm_chunks(1)
                cpu_ref.set_msm_window(0)
        if cpu:
            cpu.update({"cores": cpu_ref.max_threads(), "kind": "port",
                        # the arkworks-shaped MSM is window-parallel (one task per window, reference
                        # upstream ark-ec msm_bigint; oracle/groth16_cpu.c G##_msm): at most `nwin` of the
                        # `cores` threads are busy during the MSMs that dominate the proof
                        "msm_parallelism": nwin,
                        "sample": sample + f"; MSMs run <= {nwin} threads wide (one per window, as ark-ec does), FFTs and "
                                  "digit decomposition on all threads; C restatement of ark-groth16 0.5 prove() built with "
                                  + ("-O3 -mbmi2 -madx" if cpu_ref.variant() == "adx" else "-O3")
                                  + " (arkworks itself is not buildable offline)",
                        "host_cpu_count": os.cpu_count(), "host_cpu_granted": granted,
                        "host_cpu_grant": grant_how + f"; OpenMP's own default was {omp_default} threads"})

    # ---------------- roofline: the three instantiations of k_bucket_accumulate ----------------
    # A proof launches it four times: <Fq2, 1, false> for B2 (the longest launch of a step: the line's
    # `roofline.kernel`), <Fq, 2, true> once over the interleaved A|B1 pair, <Fq, 1, false> for L and H.
    # Algorithmic bytes: point + 32-byte scalar per point of the query.
    g2_ms, g2_cnt = stages.get("msm_accumulate_g2", (0.0, 0))
    acc_ms, acc_cnt = stages["msm_accumulate_g1"]
    pair_ms, pair_cnt = stages.get("msm_accumulate_g1_pair", (0.0, 0))
    shard_w, shard_h = info["shard_w"], info["shard_h"]
    W_w, W_h = info["W_w"], info["W_h"]
    import hashlib
    lib_sha = hashlib.sha256(open(lib_path, "rb").read()).hexdigest()[:16]
    pmc, pmc_note, pmc_live_note = {}, "no PMC record for this size / workload", None
    if n_gpus == 1 and not args.no_pmc and args.mode == "prove" and not os.environ.get("G16_AMD_LIB"):
        prover.close()                        # every timing leg is done: the profiled child gets the HBM
        prover = None
        rec, pmc_live_note = measure_pmc_traffic(args, k)
        if rec:
            pmc, pmc_note = rec, pmc_live_note
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not pmc and n_gpus == 1 and os.path.exists(tpath) and args.workload == "chain":
        t = json.load(open(tpath))
        if t.get("log2_domain") != k:
            pmc_note = f"profiles/pmc_traffic.json was measured at 2^{t.get('log2_domain')}"
        elif t.get("library_sha16") != lib_sha:
            pmc_note = (f"profiles/pmc_traffic.json was measured on library {t.get('library_sha16')}, this run "
                        f"loaded {lib_sha}: traffic dropped (re-run scripts/pmc_passes.sh + scripts/pmc_traffic.py)")
        else:
            pmc = t
            pmc_note = (f"separate rocprofv3 --pmc passes (scripts/pmc_passes.sh) on library {lib_sha}, "
                        f"{t.get('measured', '?')}; NOT measured by this run")
        if pmc_live_note:
            pmc_note += f" [this run's own passes: {pmc_live_note}]"
    elif not pmc and pmc_live_note:
        pmc_note = pmc_live_note

    def kern(name, what, ms, cnt, bytes_per_point, npoints, madds, vmad_per_madd, tkey):
        if not cnt or ms <= 0:
            return None
        per = ms / cnt
        algb = float(bytes_per_point) * npoints
        ach = algb / (per * 1e-3) / 1e9
        return {"kernel": name, "what": what, "avg_launch_ms": per, "launches_per_step": cnt / args.steps,
                "algorithmic_bytes_per_launch": algb, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                "traffic": pmc.get(tkey) if tkey else None, "mixed_additions_per_s": madds / (per * 1e-3),
                "vmad_frac_of_issue_peak": madds * vmad_per_madd / (per * 1e-3) / 30.1e12}
    VMAD_G1, VMAD_G2 = 1557.0, 4878.0          # multiply-adds per mixed addition
    if pair_cnt:
        g1_len, g1_madds = (shard_w + shard_h) / 2.0, (shard_w * W_w + shard_h * W_h) / 2.0   # L and H
    else:                                       # G16_NO_PAIR_AB=1: A, B1, L and H
        g1_len, g1_madds = (3 * shard_w + shard_h) / 4.0, (3 * shard_w * W_w + shard_h * W_h) / 4.0
    kerns = [kern("k_bucket_accumulate<Fq2, 1, false>", "B2 query (G2)", g2_ms, g2_cnt, 160, shard_w,
                  shard_w * W_w, VMAD_G2, "g2_traffic_bytes_per_launch"),
             kern("k_bucket_accumulate<Fq, 2, true>", "A and B1 over the interleaved A_i|B1_i records", pair_ms,
                  pair_cnt, 160, shard_w, 2 * shard_w * W_w, VMAD_G1, "pair_traffic_bytes_per_launch"),
             kern("k_bucket_accumulate<Fq, 1, false>", "L and H queries" if pair_cnt else "A, B1, L, H queries",
                  acc_ms, acc_cnt, 96, g1_len, g1_madds, VMAD_G1, "traffic_bytes_per_launch")]
    # small keys (g16_options.fixed_tables, automatic): the MSMs are table lookups + tree sums (csrc/msm_table.hip);
    # a stage span there = k_tbl_msm + k_tbl_final of one launch pair
    tg1_ms, tg1_cnt = stages.get("msm_table_g1", (0.0, 0))
    tg2_ms, tg2_cnt = stages.get("msm_table_g2", (0.0, 0))
    if tg1_cnt or tg2_cnt:
        kerns += [kern("k_tbl_msm<Fq2, 128> + k_tbl_final", "B2 query through fixed-base tables (32 lookups per point)",
                       tg2_ms, tg2_cnt, 160, shard_w, shard_w * 32, VMAD_G2, None),
                  kern("k_tbl_msm<Fq, 256> + k_tbl_final", "A, B1, L, s*A, r*B1 in one launch pair; H in another",
                       tg1_ms, tg1_cnt, 96, (3 * shard_w + shard_h) / 2.0, (5 * shard_w + shard_h) * 32 / 2.0, VMAD_G1, None)]
    kerns = [x for x in kerns if x]
    head = max(kerns, key=lambda x: x["avg_launch_ms"])        # the per-step dominant launch
    tot_b = sum(x["algorithmic_bytes_per_launch"] * x["launches_per_step"] for x in kerns)
    tot_ms = sum(x["avg_launch_ms"] * x["launches_per_step"] for x in kerns)
    roofline = {"bound": "hbm", "kernel": head["kernel"], "achieved": head["achieved"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": head["frac"], "traffic": head["traffic"],
                "traffic_unit": "HBM bytes per launch", "traffic_source": pmc_note,
                "avg_launch_ms": head["avg_launch_ms"], "launches_per_step": head["launches_per_step"],
                "algorithmic_bytes_per_launch": head["algorithmic_bytes_per_launch"],
                "all_accumulate_launches": kerns,
                "time_weighted": {"achieved": tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms else 0.0,
                                  "frac": tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if tot_ms else 0.0,
                                  "accumulate_ms_per_step": tot_ms},
                "library_sha16": lib_sha,
                "note": "the kernel is integer-ALU bound (254-bit Montgomery arithmetic on v_mad_i64_i32), "
                        "not HBM bound: see `alu` and DESIGN.md section 4-5"}
    # supplementary: the same launches against the micro-benchmarked integer multiply-add issue peak
    g1k = next((x for x in kerns if x["kernel"].startswith("k_bucket_accumulate<Fq, 1")), head)
    if tg1_cnt or tg2_cnt:
        roofline["note"] = ("small key: every MSM is table lookups + a tree sum (latency bound: ~21 dependent EC additions per MSM); "
                            "the HBM fraction of a 1 ms proof says nothing -- see ms_per_step")
    alu = {"kernel": g1k["kernel"], "mixed_additions_per_s": g1k["mixed_additions_per_s"],
           "vmad_peak_per_s": 30.1e12, "frac": g1k["vmad_frac_of_issue_peak"],
           "alu_only_ceiling_mixed_additions_per_s": 16.7e9,
           "by_kernel": {x["kernel"]: x["vmad_frac_of_issue_peak"] for x in kerns}}

    ms_per_step = elapsed / args.steps * 1e3
    if mode == "single":
        par = "single-gpu"
    else:
        cut = ("msm-bucket-range-shard (A/B1/B2/L points on every GPU, 1/N of the sorted bucket list per rank; H by point range)"
               if info.get("shard_mode") == "buckets" else "msm-point-range-shard")
        par = (f"{cut} x{n_gpus} + four-step witness map (2 all-to-all), "
               + ("one g16_ctx_create_multi ctx in one process: peer copies over xGMI inside the library"
                  if mode == "inlib" else "one process per GPU: RCCL all_to_all / all_gather, event hand-offs"))
        if both_vals and both_vals.get("value_inlib"):
            par = (f"{cut} x{n_gpus} + four-step witness map (2 all-to-all); BOTH launch shapes timed: one g16_ctx_create_multi "
                   "ctx in one process (peer copies over xGMI inside the library) and one process per GPU (RCCL all_to_all / "
                   f"all_gather, event hand-offs); `value` = the faster ({both_vals['value_is']})")
        if fallback_reason:
            par += " [fallback: the in-library ctx failed on this node]"
        if one_gpu:
            par += " [functional run: every rank on ONE GPU]"
    out = {
        "metric": "Groth16 constraints/sec (BN254)", "value": m * args.steps / elapsed,
        "unit": "constraints/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "int32x9 (29-bit limbs of 254-bit Montgomery integers)",
        "data": "synthetic",
        "config": {"workload": f"{desc}, BN254, full prove (witness map + 4 G1 MSM + 1 G2 MSM + finalize), "
                               "trapdoor key minted on GPU",
                   "log2_domain": k, "num_constraints": m, "n_vars": n_vars,
                   "witness_fraction_in_{0,1}": round(frac01, 4),
                   "parallelism": par, "msm": info},
        "roofline": roofline, "alu": alu, "cpu_baseline": cpu, "parity": parity,
        "stages_ms_per_step": {n: ms / args.steps for n, (ms, _c) in stages.items()},
        "setup_s": t_setup,
        "value_pcie_inclusive": (m / (host_ms * 1e-3)) if host_ms else None,
        "ms_per_step_pcie_inclusive": host_ms,
        "pcie_inclusive_note": "SURVEY 8(d) step: witness in the ctx's page-locked host buffer -> H2D -> ... "
                               "-> 256 B D2H; `value` keeps the witness resident in HBM (bench contract)",
        "library": lib_path,
    }
    if (n_gpus == 1 and args.mode == "prove" and args.workload == "chain" and k == 22 and not args.no_secondary
            and not os.environ.get("G16_AMD_LIB")):
        out["secondary"] = secondary_lines(args)
    if both_vals:
        out.update(both_vals)
    elif n_gpus > 1:
        out["rccl_ranks"] = st["rccl_ranks"]
    if clock:
        out["clock_mhz"] = clock["timed"]["sclk_mhz_median"] or clock["after"]["sclk_mhz_median"]
        out["power_w"] = clock["timed"]["power_w_median"] or clock["after"]["power_w_median"]
        out["clock"] = clock
    if pipelined:
        out["value_pipelined"] = pipelined
    if fallback_reason:
        out["fallback_reason"] = fallback_reason
    if parts:
        out["parts_ms"] = parts
    if cpu:
        # a prove() caller hands over a HOST witness: the ratio that compares like with like is the
        # PCIe-inclusive step over the CPU restatement; the resident-witness ratio beside it
        if out["value_pcie_inclusive"]:
            out["gpu_over_cpu"] = out["value_pcie_inclusive"] / cpu["value"]
            out["gpu_over_cpu_note"] = ("PCIe-inclusive GPU step (host witness -> proof) / CPU restatement, the CPU side on "
                                        f"{cpu['cores']} threads with MSMs {cpu['msm_parallelism']} threads wide; a baseline, not a kernel-quality figure")
        out["gpu_resident_over_cpu"] = out["value"] / cpu["value"]
        if cpu.get("value_all_cores"):
            best = max(cpu["value"], cpu["value_all_cores"])
            out["gpu_over_cpu_all_cores"] = (out["value_pcie_inclusive"] or out["value"]) / best
            out["gpu_over_cpu_all_cores_note"] = ("the same ratio against the FASTER CPU figure (chunk-parallel MSMs on every "
                                                  "host thread): the conservative bound")
    print(json.dumps(out), flush=True)
    if zkey_path and os.path.exists(zkey_path):
        os.remove(zkey_path)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
