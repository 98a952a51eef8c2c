"""Projected strong-scaling efficiency T1 / (G x T_rank) on ONE MI355X (the only hardware available
to the builder): the single-GPU proof and ONE rank of the fully sharded prover (rank 0 of G) are timed
on the same box with the same library.  The rank runs exactly what it runs on an 8-GPU node --
phases 1-3 with the device-side hand-offs (g16_dist_set_exchange_stream: no host syncs), its MSM
shards, the partial record and the finish -- except that the two all-to-all exchanges and the
all-gather are replaced by local copies of the same byte counts on the exchange stream (xGMI time is
therefore NOT in T_rank: 3n/G x 36 B per rank and exchange).
    python scripts/dist_projection.py [log2=22] [worlds=2,4,8] [reps=5]"""
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
worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,4,8").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
t0 = time.time()
mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
rng = random.Random(k)
tox = [rng.randrange(1, bench.R_MOD) for _ in range(5)]
pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
rs = cc.fr_from_ints([rng.randrange(bench.R_MOD), rng.randrange(bench.R_MOD)])
w_dev = torch.from_numpy(cc.fr_from_ints(w_ints).view(np.int64)).cuda()
torch.cuda.synchronize()
print(f"setup {time.time() - t0:.1f} s", file=sys.stderr)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[len(ts) // 2]


if os.environ.get("G16_PROJ_T1"):          # reuse a single-GPU time measured on this box (knob sweeps)
    t1, info1 = float(os.environ["G16_PROJ_T1"]), {"c_w": 0, "W_w": 0}
else:
    single = cc.Prover(pk, mats)
    t1 = timed(lambda: single.prove_dev(rs[0], rs[1], w_dev.data_ptr()))
    info1 = single.info()
    single.close()
    del single
out = {"log2": k, "single_gpu_ms": t1, "single_msm": {x: info1[x] for x in ("c_w", "W_w")}, "ranks": {}}
# high priority: shares a hardware queue with the aux stream, not with the MSM streams (G16_PROJ_XS_PRIO=0: A/B)
xs = torch.cuda.Stream(priority=-1 if os.environ.get('G16_PROJ_XS_PRIO', '1') != '0' else 0)
for G in worlds:
    p = cc.Prover(pk, mats, rank=0, world=G, dist_wm=True)
    p.set_exchange_stream(xs.cuda_stream)
    nbytes = p.exchange_bytes()
    send = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    recv = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    part = cc.device_tensor(p.partial_buffer(), 1024)
    gath = cc.device_tensor(p.gather_buffer(), G * 1024)

    def rank_step():
        p.dist_phase1(rs[0], rs[1], w_dev.data_ptr(), send.data_ptr())
        with torch.cuda.stream(xs):
            recv.copy_(send, non_blocking=True)
        p.dist_phase2(recv.data_ptr(), send.data_ptr())
        with torch.cuda.stream(xs):
            recv.copy_(send, non_blocking=True)
        p.dist_phase3_dev(recv.data_ptr())
        with torch.cuda.stream(xs):
            for g in range(G):
                gath[g * 1024:(g + 1) * 1024].copy_(part, non_blocking=True)
        p.prove_finish_dev(rs[0], rs[1])

    tr = timed(rank_step)
    info = p.info()
    out["ranks"][str(G)] = {"per_rank_ms": tr, "efficiency_before_xgmi": t1 / (G * tr),
                            "exchange_MB_per_rank": nbytes / 1e6, "c_w": info["c_w"], "W_w": info["W_w"],
                            "shard_w": info["shard_w"]}
    p.close()
    del p, send, recv
    torch.cuda.empty_cache()
print(json.dumps(out))
