"""One rank of the fully sharded prover, all phases + finish with the device-side hand-offs, repeated:
run under `rocprofv3 --kernel-trace` and feed the database to rocpd_timeline.py to see where a rank's
time goes.  The exchanges are local copies of the same size on the exchange stream (values are
garbage / random, kernel timing is not).
    python scripts/dist_rank_trace.py [log2=22] [world=8] [mode=buckets] [rank=world/2] [reps=4]"""
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
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mode = sys.argv[3] if len(sys.argv) > 3 else "buckets"
rank = int(sys.argv[4]) if len(sys.argv) > 4 else G // 2
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
rng = random.Random(k)
tox = [rng.randrange(1, bench.R_MOD) for _ in range(5)]
pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
rs = cc.fr_from_ints([rng.randrange(bench.R_MOD), rng.randrange(bench.R_MOD)])
w_dev = torch.from_numpy(cc.fr_from_ints(w_ints).view(np.int64)).cuda()
xs = torch.cuda.Stream(priority=-1)
p = cc.Prover(pk, mats, rank=rank, world=G, dist_wm=True, shard=mode)
p.set_exchange_stream(xs.cuda_stream)
nbytes = p.exchange_bytes()
send = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
recv = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
part = cc.device_tensor(p.partial_buffer(), 1024)
gath = cc.device_tensor(p.gather_buffer(), G * 1024)
torch.cuda.synchronize()
for rep in range(reps):
    t0 = time.perf_counter()
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
    torch.cuda.synchronize()
    print("rep", rep, f"{(time.perf_counter() - t0) * 1e3:.3f} ms")
print(p.info())
