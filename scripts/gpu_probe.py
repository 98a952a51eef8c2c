"""First-contact GPU probe: stage timings of witness map and MSMs on synthetic data (cycled
points: cost-equivalent to a real key, not pairing-valid).  Writes gpurun_out/probe.json."""
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bn254_ref as o
import circom_compat_amd as cc
import helpers as H


def chain_matrices(k):
    m = (1 << k) - 2
    n_vars = m + 2
    wire = np.arange(m, dtype=np.uint32) + 2
    rp = np.arange(m + 1, dtype=np.uint32)
    minus1 = H.fr_mont_arr([o.R_MOD - 1])[0]
    one = H.fr_mont_arr([1])[0]
    A = cc.Csr(rp, wire, np.tile(minus1, (m, 1)))
    B = cc.Csr(rp, wire, np.tile(one, (m, 1)))
    return cc.ConstraintMatrices(2, n_vars - 1, m, A, B), n_vars


def cycled_key(N, dom, K=256, seed=1):
    rng = random.Random(seed)
    g1 = [o.G1.mul(o.G1_GEN, rng.randrange(1, o.R_MOD)) for _ in range(K)]
    g2 = [o.G2.mul(o.G2_GEN, rng.randrange(1, o.R_MOD)) for _ in range(K)]
    b1, b2 = H.g1_arr(g1), H.g2_arr(g2)
    idx = np.arange(N) % K
    vk = cc.VerifyingKey(o.g1_to_bytes(g1[0]), o.g2_to_bytes(g2[0]), o.g2_to_bytes(g2[1]),
                         o.g2_to_bytes(g2[2]), b1[:2].copy())
    return cc.ProvingKey(N, 1, dom, vk, o.g1_to_bytes(g1[1]), o.g1_to_bytes(g1[2]), b1[idx].copy(),
                         b1[(idx + 1) % K].copy(), b2[idx].copy(), b1[(np.arange(N - 2) + 5) % K].copy(),
                         b1[(np.arange(dom) * 3) % K].copy())


def main():
    out = []
    ks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,20").split(",")]
    cfgs = [tuple(int(y) for y in x.split(":")) for x in
            (sys.argv[2] if len(sys.argv) > 2 else "0:0").split(",")]
    for k in ks:
        mats, n_vars = chain_matrices(k)
        pk = cycled_key(n_vars, 1 << k)
        rng = np.random.default_rng(k)
        w = rng.integers(0, 1 << 62, size=(n_vars, 4), dtype=np.uint64)
        w[:, 3] &= np.uint64((1 << 60) - 1)
        for (c, planes) in cfgs:
            t0 = time.time()
            pr = cc.Prover(pk, mats, window_bits=c, planes=planes)
            t_create = time.time() - t0
            info = pr.info()
            pr.prove(5, 7, w)                      # warm-up
            pr.set_profiling(True)
            reps = 3
            t0 = time.time()
            for _ in range(reps):
                pr.prove(5, 7, w)
            wall = (time.time() - t0) / reps
            st = pr.stage_times()
            rec = dict(k=k, c=c, planes=planes, info=info, create_s=t_create, prove_wall_ms=wall * 1e3,
                       stages={n: (ms / reps, cnt // reps) for n, (ms, cnt) in st.items()},
                       constraints_per_s=((1 << k) - 2) / wall)
            print(json.dumps(rec), flush=True)
            out.append(rec)
            pr.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
