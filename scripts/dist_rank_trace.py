"""One rank of the fully sharded prover (rank 0 of `world`), phases 1-3 + finish, repeated: run under
`rocprofv3 --kernel-trace` and feed the database to rocpd_timeline.py to see where a rank's time goes.
The exchanges are skipped (recv = stale bytes): values are garbage, kernel timing is not.
    python scripts/dist_rank_trace.py [log2=22] [world=8] [window_bits=0 (cost model)]"""
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
wbits = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
rng = random.Random(k)
tox = [rng.randrange(1, bench.R_MOD) for _ in range(5)]
pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
r, s = rng.randrange(bench.R_MOD), rng.randrange(bench.R_MOD)
w_dev = torch.from_numpy(cc.fr_from_ints(w_ints).view(np.int64)).cuda()
p = cc.Prover(pk, mats, rank=0, world=world, dist_wm=True, window_bits=wbits)
nbytes = p.exchange_bytes()
send = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
recv = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
for rep in range(4):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    p.dist_phase1(r, s, w_dev.data_ptr(), send.data_ptr()); t.append(time.perf_counter())
    recv.copy_(send); torch.cuda.current_stream().synchronize(); t.append(time.perf_counter())
    p.dist_phase2(recv.data_ptr(), send.data_ptr()); t.append(time.perf_counter())
    recv.copy_(send); torch.cuda.current_stream().synchronize(); t.append(time.perf_counter())
    part = p.dist_phase3(recv.data_ptr()); t.append(time.perf_counter())
    p.prove_finish(r, s, part * world); t.append(time.perf_counter())
    print("rep", rep, " ".join(f"{(b - a) * 1e3:.3f}" for a, b in zip(t, t[1:])), "total", f"{(t[-1] - t[0]) * 1e3:.3f}")
print(p.info())
