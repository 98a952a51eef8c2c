"""One process per GPU, the collectives issued by the LIBRARY (g16_dist_attach_rccl / g16_prove_dist): what a host
without a collective framework runs.  torch.distributed (gloo) is only the launcher's messenger here: it carries the
ncclUniqueId from rank 0 to the others and the barriers around the timed region; the communicator is created with a
bare ctypes binding of RCCL (the stand-in for the Rust shim's own binding) and handed to the library.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/rccl_inlib_ranks.py [log2=22] [steps=10] [shard=points]

Prints one JSON line on rank 0: ms per proof (max over ranks), every rank's proof bytes equal, == the single-GPU
proof of rank 0's device, pairing-verified.  Never run on two distinct devices by the builder (one-GPU boxes): the
world-1 case is tests/test_gpu_large.py::test_rccl_collectives_issued_by_the_library_world1 and
`torchrun --nproc-per-node 1` of this script (scripts/hardware_day.sh runs it at N = 2, 4, 8)."""
import ctypes as C
import json
import os
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import bench  # noqa: E402
import circom_compat_amd as cc  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
shard = sys.argv[3] if len(sys.argv) > 3 else "points"
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29811")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(local)
torch.zeros(1, device="cuda")
dist.init_process_group("gloo", rank=rank, world_size=world)


class Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
uid = Uid()
if rank == 0:
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
box = [bytes(uid)]
dist.broadcast_object_list(box, src=0)
uid = Uid.from_buffer_copy(box[0])
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
st = rccl.ncclCommInitRank(C.byref(comm), world, uid, rank)
assert st == 0, f"ncclCommInitRank: {st}"

mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
rng = random.Random(k)
tox = [rng.randrange(1, bench.R_MOD) for _ in range(5)]
pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox, device=local)
w = cc.fr_from_ints(w_ints)
w_dev = torch.from_numpy(w.view(np.int64)).cuda()
rs_rng = random.Random(1000 + k)
rs = cc.fr_from_ints([rs_rng.randrange(bench.R_MOD), rs_rng.randrange(bench.R_MOD)])
p = cc.Prover(pk, mats, device=local, rank=rank, world=world, dist_wm=True, shard=shard)
p.attach_rccl(comm.value)
assert p.rccl_ranks() == world
proof = p.prove_dist(rs[0], rs[1], w_dev.data_ptr())            # warm-up (RCCL sets its channels up here)
proof = p.prove_dist(rs[0], rs[1], w_dev.data_ptr())
torch.cuda.synchronize()
dist.barrier()
t0 = time.perf_counter()
for _ in range(steps):
    proof = p.prove_dist(rs[0], rs[1], w_dev.data_ptr())
torch.cuda.synchronize()
dist.barrier()
dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
dist.all_reduce(dt, op=dist.ReduceOp.MAX)
proofs = [None] * world
dist.all_gather_object(proofs, proof.raw)
p.close()
if rank == 0:
    single = cc.Prover(pk, mats, device=local).prove(rs[0], rs[1], w)
    import bn254_ref as o
    import helpers as H
    vk = dict(alpha_g1=o.g1_from_bytes(bytes(pk.vk.alpha_g1)), beta_g2=o.g2_from_bytes(bytes(pk.vk.beta_g2)),
              gamma_g2=o.g2_from_bytes(bytes(pk.vk.gamma_g2)), delta_g2=o.g2_from_bytes(bytes(pk.vk.delta_g2)),
              ic=[o.g1_from_bytes(bytes(x)) for x in pk.vk.gamma_abc_g1])
    print(json.dumps({"what": "g16_prove_dist: one process per GPU, ncclAllToAll x 2 + ncclAllGather issued by the library",
                      "log2_domain": k, "n_gpus": world, "shard": shard, "steps": steps,
                      "ms_per_step": float(dt.item()) / steps * 1e3, "value": mats.num_constraints * steps / float(dt.item()),
                      "all_ranks_same_proof": all(x == proofs[0] for x in proofs),
                      "equals_single_gpu_proof": proofs[0] == single.raw,
                      "proof_verifies": bool(o.verify_proof(vk, [w_ints[1]], H.proof_from_bytes(proofs[0])))}), flush=True)
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
dist.barrier()
dist.destroy_process_group()
