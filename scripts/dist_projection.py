"""Projected strong-scaling efficiency T1 / (G x T_rank) on ONE MI355X (the only hardware available
to the builder): the single-GPU proof and ONE rank of the fully sharded prover are timed on the same
box with the same library.  The rank runs exactly what it runs on an 8-GPU node -- the witness-map
phases with the device-side hand-offs (g16_dist_set_exchange_stream: no host syncs), its share of the
MSMs, the partial record and the finish -- except that the exchanges are replaced by local copies
of the same byte counts on the exchange stream.  xGMI time is therefore NOT in T_rank; the bytes a
rank sends per proof and the link time they take at a stated per-link rate are printed beside it.

    python scripts/dist_projection.py [log2=22] [worlds=2,4,8] [reps=5] [modes=points,buckets] [knobs]

knobs: "name:K=V,K=V;name2:K=V" -- every (mode, world) is timed once per knob set (environment knobs the
       library reads per ctx / per proof), on the same resident key: same-box A/B of schedule variants

modes: points  = MSMs cut by point range (rank 0 is timed: all ranks alike)
       buckets = witness-scalar MSMs cut by bucket range (every rank holds all A/B1/B2/L points; the
                 slowest of rank 0 -- the dense low partitions, fewest buckets -- and a middle rank)"""
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
modes = (sys.argv[4] if len(sys.argv) > 4 else "points,buckets").split(",")
knob_sets = [("default", {})]
if len(sys.argv) > 5 and sys.argv[5]:
    for item in sys.argv[5].split(";"):
        name, _, kv = item.partition(":")
        knob_sets.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
LINK_GBS = float(os.environ.get("G16_PROJ_LINK_GBS", "48"))   # one xGMI link, one direction, achieved
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
out = {"log2": k, "single_gpu_ms": t1, "single_msm": {x: info1[x] for x in ("c_w", "W_w")},
       "link_GBs_assumed": LINK_GBS, "ranks": {}}
# high priority: shares a hardware queue with the aux stream, not with the MSM streams (G16_PROJ_XS_PRIO=0: A/B)
xs = torch.cuda.Stream(priority=-1 if os.environ.get('G16_PROJ_XS_PRIO', '1') != '0' else 0)


def one_rank(G, mode, rank):
    p = cc.Prover(pk, mats, rank=rank, world=G, dist_wm=True, shard=mode,
                  window_bits=int(os.environ.get("G16_PROJ_WINDOW_BITS", "0")))
    p.set_exchange_stream(xs.cuda_stream)
    nbytes = p.exchange_bytes()
    send = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    recv = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    part = cc.device_tensor(p.partial_buffer(), 1024)
    gath = cc.device_tensor(p.gather_buffer(), G * 1024)
    gath_src = torch.zeros(max(G - 1, 1) * 1024, dtype=torch.uint8, device="cuda")   # the peers' records (infinity)

    def rank_step():
        p.dist_phase1(rs[0], rs[1], w_dev.data_ptr(), send.data_ptr())
        with torch.cuda.stream(xs):
            recv.copy_(send, non_blocking=True)
        p.dist_phase2(recv.data_ptr(), send.data_ptr())
        with torch.cuda.stream(xs):
            recv.copy_(send, non_blocking=True)
        p.dist_phase3_dev(recv.data_ptr())
        with torch.cuda.stream(xs):     # the all-gather of the 1 KiB records: ONE collective = one 8 KiB copy
            gath[:1024].copy_(part, non_blocking=True)
            if G > 1:
                gath[1024:].copy_(gath_src, non_blocking=True)
        p.prove_finish_dev(rs[0], rs[1])

    tr = timed(rank_step)
    info = p.info()
    p.set_profiling(True)
    rank_step()
    torch.cuda.synchronize()
    stages = {n: round(ms, 3) for n, (ms, _c) in p.stage_times().items()}
    p.close()
    # bytes this rank SENDS per proof: (G-1)/G of each all-to-all buffer (+ 1 KiB records)
    sent = 2 * nbytes * (G - 1) / G
    links = min(G - 1, 7)
    return tr, info, stages, sent, sent / links / (LINK_GBS * 1e9) * 1e3


for mode in modes:
  for G in worlds:
    for kname, kenv in knob_sets:
        for kk, vv in kenv.items():
            os.environ[kk] = vv
        cand = [0] if mode == "points" else sorted({0, G // 2})
        res = [one_rank(G, mode, r) for r in cand]
        for kk in kenv:
            del os.environ[kk]
        tr, info, stages, sent, link_ms = max(res, key=lambda x: x[0])
        out["ranks"][f"{mode}:{G}" + ("" if kname == "default" else ":" + kname)] = {
            "knobs": kenv,
            "mode": mode, "G": G, "per_rank_ms": tr, "ranks_timed": {str(r): x[0] for r, x in zip(cand, res)},
            "efficiency_before_xgmi": t1 / (G * tr),
            "sent_MB_per_rank_per_proof": sent / 1e6, "link_ms_if_exposed": link_ms,
            "efficiency_if_all_link_time_exposed": t1 / (G * (tr + link_ms)),
            "c_w": info["c_w"], "W_w": info["W_w"], "shard_w": info["shard_w"],
            "stages_ms_alone": stages}
        torch.cuda.empty_cache()
print(json.dumps(out))
