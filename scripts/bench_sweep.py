"""The reference bench's sweep (benches/groth16.rs:87-104, feature bench-complex-all: 10^i variables x
10^j constraints, 3 <= i <= j <= 5, circuits of test-vectors/complex-circuit/complex-circuit.circom.template)
on one MI355X: latency of create_proof_with_reduction_and_matrices through the C ABI next to the CPU
restatement's on the same (pk, r, s, w), proofs byte-compared.  Keys: trapdoor setup on the GPU (the
snapshot ships no .zkey for these circuits).
    python scripts/bench_sweep.py [reps=20] > profiles/rNN_bench_sweep.txt"""
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import bench
import circom_compat_amd as cc
import cpu_ref

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rows = []
for i in (3, 4, 5):
    for j in range(i, 6):
        V, Cn = 10 ** i, 10 ** j
        mats, (A, B, Cm), w_ints, n_vars = bench.complex_shape_circuit(cc, V, Cn)
        if (V, Cn) == (10 ** 4, 10 ** 4):   # the shipped artefact is this member of the family: same matrices
            ref = cc.R1CS.from_file(os.path.join(ROOT, "tests", "golden", "complex-circuit-10000-10000.r1cs"))
            assert np.array_equal(ref.a.col, A.col) and np.array_equal(ref.c.col, Cm.col) and np.array_equal(ref.a.coeff, A.coeff)
        rng = random.Random(1000 * i + j)
        tox = [rng.randrange(1, bench.R_MOD) for _ in range(5)]
        pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
        rs = cc.fr_from_ints([rng.randrange(bench.R_MOD), rng.randrange(bench.R_MOD)])
        w = cc.fr_from_ints(w_ints)
        pr = cc.Prover(pk, mats)
        w_dev = torch.from_numpy(w.view(np.int64)).cuda()
        proof = pr.prove_dev(rs[0], rs[1], w_dev.data_ptr())
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            proof = pr.prove_dev(rs[0], rs[1], w_dev.data_ptr())
        gpu_ms = (time.perf_counter() - t) / reps * 1e3
        host = pr.witness_host_buffer()
        host[:] = w
        t = time.perf_counter()
        for _ in range(reps):
            pr.prove(rs[0], rs[1], host)
        gpu_host_ms = (time.perf_counter() - t) / reps * 1e3
        cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
        t = time.perf_counter()
        want = cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w)
        cpu_ms = (time.perf_counter() - t) * 1e3
        info = pr.info()
        rows.append(dict(variables=V, constraints=Cn, domain=info["domain_size"], c_w=info["c_w"], fixed_tables=info["fixed_tables"], gpu_ms=gpu_ms,
                         gpu_ms_witness_from_host=gpu_host_ms, cpu_ms=cpu_ms, bytes_equal=bool(proof.raw == want)))
        pr.close()
print("| variables | constraints | domain | window c | fixed-base tables | GPU ms (witness in HBM) | GPU ms (witness from host) | CPU restatement ms | proofs byte-equal |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['variables']} | {r['constraints']} | {r['domain']} | {r['c_w']} | {'yes' if r['fixed_tables'] else 'no'} | {r['gpu_ms']:.2f} | {r['gpu_ms_witness_from_host']:.2f} | "
          f"{r['cpu_ms']:.0f} | {r['bytes_equal']} |")
print()
print(json.dumps({"reps": reps, "cpu_threads": cpu_ref.max_threads(), "cpu_variant": cpu_ref.variant(), "rows": rows}))
