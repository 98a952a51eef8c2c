"""world_size = 2 and 4 over gloo on the CPU: the sharded paths (per-rank MSM partials over point
ranges and over bucket ranges -> all_gather -> local EC add -> finish; distributed witness map with
its two all-to-all exchanges) give the oracle's proof bytes.
Runs the kernel sources on the SIMT emulator (tests only); on the GPU the same harness code in
bench.py uses backend nccl (= RCCL)."""
import os
import random
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, random
    import numpy as np
    import torch, torch.distributed as dist
    ROOT = sys.argv[1]
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import bn254_ref as o, helpers as H
    import circom_compat_amd as cc
    from circom_compat_amd import _binding
    lib = _binding.Library(os.path.join(ROOT, "tests", "emu", "libg16_emu.so"))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if sys.argv[3] == "dense":   # uneven rows, skewed witness, several public inputs
        cons, w, n_vars, _ = H.dense_skewed_circuit((1 << int(sys.argv[2])) - 5, seed=3, long_rows=(7,))
        n_pub = 3
    else:
        cons, w, n_vars, n_pub = H.squaring_chain(int(sys.argv[2]))
    ni = n_pub + 1
    rng = random.Random(42)
    tox = [rng.randrange(1, o.R_MOD) for _ in range(5)]
    opk = o.trapdoor_setup(cons, n_vars, n_pub, *tox)
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    mats = H.matrices_from_rows(a_rows, b_rows, ni, n_vars, lib)
    pk = H.pk_from_oracle(opk)
    r, s = rng.randrange(o.R_MOD), rng.randrange(o.R_MOD)
    pr = cc.Prover(pk, mats, lib=lib, rank=rank, world=world, shard="points")
    part = pr.prove_partial(r, s, w)
    mine = torch.frombuffer(bytearray(part), dtype=torch.uint8)
    gathered = torch.empty(world * 1024, dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered, mine)
    proof = pr.prove_finish(r, s, gathered.numpy().tobytes())
    want = o.create_proof_with_reduction_and_matrices(opk, r, s, dict(a=a_rows, b=b_rows), ni, len(cons), w)
    assert proof.raw == o.proof_to_bytes(want), "sharded proof differs from the oracle"
    # fully sharded: the witness map is distributed too (four-step NTTs, two all-to-all exchanges)
    pd = cc.Prover(pk, mats, lib=lib, rank=rank, world=world, dist_wm=True, shard="points")
    nbytes = pd.exchange_bytes()
    send = torch.empty(nbytes, dtype=torch.uint8)
    recv = torch.empty(nbytes, dtype=torch.uint8)
    w_arr = H.fr_mont_arr(w)                      # the emulator's "device" memory is host memory
    pd.dist_phase1(r, s, w_arr.ctypes.data, send.data_ptr())
    dist.all_to_all_single(recv, send)
    pd.dist_phase2(recv.data_ptr(), send.data_ptr())
    dist.all_to_all_single(recv, send)
    part2 = pd.dist_phase3(recv.data_ptr())
    g2 = torch.empty(world * 1024, dtype=torch.uint8)
    dist.all_gather_into_tensor(g2, torch.frombuffer(bytearray(part2), dtype=torch.uint8))
    proof2 = pd.prove_finish(r, s, g2.numpy().tobytes())
    assert proof2.raw == o.proof_to_bytes(want), "fully sharded proof differs from the oracle"
    assert o.verify_proof(opk, w[1:ni], H.proof_from_bytes(proof.raw))
    # MSMs sharded by BUCKET range (every rank holds the whole key, keeps 1/world of the sorted list):
    # replicated witness map first ...
    pb = cc.Prover(pk, mats, lib=lib, rank=rank, world=world, shard="buckets")
    assert pb.info()["shard_mode"] == "buckets" and pb.info()["shard_w"] == n_vars - 1
    g3 = torch.empty(world * 1024, dtype=torch.uint8)
    dist.all_gather_into_tensor(g3, torch.frombuffer(bytearray(pb.prove_partial(r, s, w)), dtype=torch.uint8))
    assert pb.prove_finish(r, s, g3.numpy().tobytes()).raw == o.proof_to_bytes(want), "bucket-sharded proof differs"
    # ... then fully sharded: the same three phases and two exchanges as the point-range ranks
    pq = cc.Prover(pk, mats, lib=lib, rank=rank, world=world, dist_wm=True, shard="buckets")
    assert pq.info()["shard_mode"] == "buckets" and pq.info()["shard_h"] == pk.domain_size // world
    nbytes = pq.exchange_bytes()
    send = torch.empty(nbytes, dtype=torch.uint8)
    recv = torch.empty(nbytes, dtype=torch.uint8)
    pq.dist_phase1(r, s, w_arr.ctypes.data, send.data_ptr())
    dist.all_to_all_single(recv, send)
    pq.dist_phase2(recv.data_ptr(), send.data_ptr())
    dist.all_to_all_single(recv, send)
    part4 = pq.dist_phase3(recv.data_ptr())
    g4 = torch.empty(world * 1024, dtype=torch.uint8)
    dist.all_gather_into_tensor(g4, torch.frombuffer(bytearray(part4), dtype=torch.uint8))
    assert pq.prove_finish(r, s, g4.numpy().tobytes()).raw == o.proof_to_bytes(want), "fully bucket-sharded proof differs"
    # every rank must have produced the identical proof
    t = torch.frombuffer(bytearray(proof.raw), dtype=torch.uint8).clone()
    ref = t.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(t, ref)
    dist.destroy_process_group()
    sys.stdout.write(f"rank {rank} ok" + chr(10))    # one write(): ranks share the pipe
    sys.stdout.flush()
''')


@pytest.mark.parametrize("world,logm,kind", [(2, 4, "chain"), (4, 6, "chain"), (2, 5, "dense")])
def test_sharded_prove_gloo(emu, tmp_path, world, logm, kind):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    import socket
    with socket.socket() as sk:  # a port that is free right now (a fixed range collided with TIME_WAIT leftovers)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script), ROOT, str(logm), kind]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for k in range(world):
        assert f"rank {k} ok" in r.stdout
