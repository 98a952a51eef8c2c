"""Per-rank timing of the fully sharded prover with all ranks emulated on ONE GPU (the exchanges are
device-to-device copies): what one rank spends in each phase, i.e. T_G without the xGMI time.
    python scripts/dist_phase_timing.py [log2=22] [world=8]  -> gpurun_out/dist_phase_timing.json"""
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import circom_compat_amd as cc

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
rng = random.Random(k)
tox = [rng.randrange(1, bench.R_MOD) for _ in range(5)]
pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
r, s = rng.randrange(bench.R_MOD), rng.randrange(bench.R_MOD)
w = cc.fr_from_ints(w_ints)
w_dev = torch.from_numpy(w.view(np.int64)).cuda()
single_p = cc.Prover(pk, mats)
single = single_p.prove_dev(r, s, w_dev.data_ptr())
t0 = time.perf_counter()
for _ in range(3):
    single_p.prove_dev(r, s, w_dev.data_ptr())
t_single = (time.perf_counter() - t0) / 3
single_p.close()
provers = [cc.Prover(pk, mats, rank=g, world=world, dist_wm=True) for g in range(world)]
nbytes = provers[0].exchange_bytes()
send = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(world)]
recv = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(world)]
chunk = nbytes // world


def all_to_all():
    for dst in range(world):
        for src in range(world):
            recv[dst][src * chunk:(src + 1) * chunk] = send[src][dst * chunk:(dst + 1) * chunk]
    torch.cuda.synchronize()


def timed(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    return out, (time.perf_counter() - t) * 1e3


res = []
for rep in range(3):
    ph = {"phase1": [], "phase2": [], "phase3": []}
    for g, p in enumerate(provers):
        _, t = timed(lambda: (p.dist_phase1(r, s, w_dev.data_ptr(), send[g].data_ptr()), torch.cuda.synchronize()))
        ph["phase1"].append(t)   # includes this rank's A/B1/L/B2 MSMs (device-wide sync)
    all_to_all()
    for g, p in enumerate(provers):
        _, t = timed(lambda: p.dist_phase2(recv[g].data_ptr(), send[g].data_ptr()))
        ph["phase2"].append(t)
    all_to_all()
    parts = []
    for g, p in enumerate(provers):
        part, t = timed(lambda: p.dist_phase3(recv[g].data_ptr()))
        ph["phase3"].append(t)
        parts.append(part)
    proof, t_fin = timed(lambda: provers[0].prove_finish(r, s, b"".join(parts)))
    assert proof.raw == single.raw, "sharded proof differs from the single-GPU proof"
    res.append({k2: float(np.mean(v)) for k2, v in ph.items()} | {"finish": t_fin})
last = res[-1]
per_rank = last["phase1"] + last["phase2"] + last["phase3"] + last["finish"]
out = {"log2": k, "world": world, "single_gpu_ms": t_single * 1e3, "per_rank_ms": last, "per_rank_total_ms": per_rank,
       "exchange_bytes_per_rank": nbytes, "projected_efficiency_without_comm": t_single * 1e3 / (world * per_rank),
       "note": "all ranks time-share one GPU here; phase1 includes the rank's witness-scalar MSMs"}
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "dist_phase_timing.json"), "w"), indent=1)
