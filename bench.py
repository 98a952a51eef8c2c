#!/usr/bin/env python
"""bench.py -- Groth16 constraints/sec (BN254) on MI355X: BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full proof (CircomReduction witness map + 4 G1 MSMs + 1 G2 MSM + finalisation) of the
synthetic squaring-chain circuit of SURVEY.md section 8(d) with m = 2^k - 2 constraints (default
k = 22, BASELINE.json configs[2]); the proving key is a real trapdoor key minted on the GPU, the
witness is resident in HBM when the timed region starts, every rank holds its point-range shard.
N > 1: the MSMs are sharded by point range, the five partial sums are all-gathered over RCCL and
combined locally ("all-reduce" of EC points), i.e. strong scaling of ONE proof.

Rank 0 prints ONE JSON line: metric/value/unit/... plus `roofline` (dominant kernel, live HIP-event
timing on the kernel's own stream), `cpu_baseline` (the oracle's multithreaded C restatement of the
reference's CPU prover timed on this box's host cores on a bounded sample) and `parity`.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2", type=int, default=22, help="log2 of the domain (m = 2^k - 2 constraints)")
    ap.add_argument("--cpu-log2", type=int, default=17, help="size of the CPU baseline sample (0 = skip)")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--planes", type=int, default=0)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 through torch.distributed.run (one process per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # G16_BENCH_BACKEND=gloo G16_BENCH_DEVICE=0: run the N > 1 code path with every rank on ONE GPU
    # and the exchanges staged through the host -- a functional check of this script on a 1-GPU box,
    # never a measurement
    backend = os.environ.get("G16_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = int(os.environ.get("G16_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import circom_compat_amd as cc
    k = args.log2
    t_setup = time.time()
    mats, (A, B, Cm), w_ints, n_vars = chain_circuit(cc, k)
    m = mats.num_constraints
    rng = random.Random(k)
    tox = [rng.randrange(1, R_MOD) for _ in range(5)]
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox, device=local_rank)
    dist_wm = world > 1 and os.environ.get("G16_BENCH_DIST_WM", "1") != "0"
    prover = cc.Prover(pk, mats, device=local_rank, rank=rank, world=world,
                       window_bits=args.window_bits, planes=args.planes, dist_wm=dist_wm)
    w = cc.fr_from_ints(w_ints)
    rs_rng = random.Random(1000 + k)
    r, s = rs_rng.randrange(R_MOD), rs_rng.randrange(R_MOD)
    rs = cc.fr_from_ints([r, s])
    # witness resident in HBM before the timed region (torch owns the buffer: plumbing only)
    w_dev = torch.from_numpy(w.view(np.int64)).to(f"cuda:{local_rank}")
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    dev = f"cuda:{local_rank}"
    gathered = torch.empty(world * 1024, dtype=torch.uint8, device=dev) if world > 1 else None
    if dist_wm:
        nbytes = prover.exchange_bytes()
        send = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        recv = torch.empty(nbytes, dtype=torch.uint8, device=dev)

    def exchange():
        # RCCL all-to-all over xGMI on torch's stream; only that stream is waited for, so the ctx's
        # own MSM stream keeps running underneath
        if backend == "nccl":
            dist.all_to_all_single(recv, send)
            torch.cuda.current_stream().synchronize()
        else:
            hs, hr = send.cpu(), torch.empty(nbytes, dtype=torch.uint8)
            dist.all_to_all_single(hr, hs)
            recv.copy_(hr)
            torch.cuda.current_stream().synchronize()

    def step():
        if world == 1:
            return prover.prove_dev(rs[0], rs[1], w_dev.data_ptr())
        if dist_wm:
            prover.dist_phase1(rs[0], rs[1], w_dev.data_ptr(), send.data_ptr())
            exchange()
            prover.dist_phase2(recv.data_ptr(), send.data_ptr())
            exchange()
            part = prover.dist_phase3(recv.data_ptr())
        else:
            part = prover.prove_partial(rs[0], rs[1], w_dev_ptr=w_dev.data_ptr())
        if backend == "nccl":
            mine = torch.frombuffer(bytearray(part), dtype=torch.uint8).to(dev)
            dist.all_gather_into_tensor(gathered, mine)
            return prover.prove_finish(rs[0], rs[1], gathered.cpu().numpy().tobytes())
        parts = [torch.empty(1024, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, torch.frombuffer(bytearray(part), dtype=torch.uint8))
        return prover.prove_finish(rs[0], rs[1], b"".join(p.numpy().tobytes() for p in parts))

    for _ in range(args.warmup):
        proof = step()
    prover.set_profiling(True)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stages = prover.stage_times()
    prover.set_profiling(False)
    info = prover.info()
    # PCIe-inclusive variant (host witness -> H2D inside the call): reported beside, never as `value`
    host_ms = None
    if world == 1:
        prover.prove(rs[0], rs[1], w)
        t1 = time.perf_counter()
        for _ in range(2):
            prover.prove(rs[0], rs[1], w)
        host_ms = (time.perf_counter() - t1) / 2 * 1e3

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

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

    # ---------------- CPU baseline: bounded sample on this box's host cores ----------------
    # The sample size adapts to the box: a 2^cpu_log2 probe proof is timed first, then the largest
    # circuit (<= the GPU's) whose two proofs fit in ~20 s of CPU work is timed and byte-compared.
    cpu = None
    if args.cpu_log2 > 0:
        import cpu_ref

        def cpu_case(kc, max_reps, budget_s):
            mats_c, (Ac, Bc, Cc), wc_ints, nvc = chain_circuit(cc, kc)
            pk_c = cc.trapdoor_setup(Ac, Bc, Cc, nvc, 1, tox, device=local_rank)
            wc = cc.fr_from_ints(wc_ints)
            pr_c = cc.Prover(pk_c, mats_c, device=local_rank)
            gpu_small = pr_c.prove(rs[0], rs[1], wc)
            pr_c.close()
            reps, t_cpu, same = 0, 0.0, True
            while reps < 1 or (t_cpu < budget_s and reps < max_reps):
                t1 = time.perf_counter()
                cpu_proof = cpu_ref.prove(pk_c, mats_c, rs[0:1].copy(), rs[1:2].copy(), wc)
                t_cpu += time.perf_counter() - t1
                reps += 1
                same = same and (cpu_proof == gpu_small.raw)
            return mats_c.num_constraints, reps, t_cpu, same

        # N > 1: only the byte-comparison of a small single-GPU proof (the timed baseline belongs to
        # the N = 1 line; torchrun also pins OMP_NUM_THREADS=1)
        kc = min(args.cpu_log2, k) if world == 1 else min(args.cpu_log2, k, 14)
        m_c, reps, t_cpu, same = cpu_case(kc, 1, 0.0)
        parity["bit_identical_to_cpu_at_2^%d" % kc] = bool(same)
        per_proof = t_cpu / reps
        grow = 0
        while world == 1 and kc + grow < min(k, 22) and per_proof * (2 ** (grow + 1)) * 2 <= 20.0:
            grow += 1
        if grow > 0:
            kc += grow
            m_c, reps, t_cpu, same = cpu_case(kc, 4, 12.0)
            parity["bit_identical_to_cpu_at_2^%d" % kc] = bool(same)
        cpu = None if world > 1 else {"value": m_c * reps / t_cpu, "unit": "constraints/s",
               "cores": cpu_ref.max_threads(), "kind": "port",
               "sample": f"{reps} proof(s) of the 2^{kc}-constraint squaring-chain circuit "
                         f"({t_cpu:.1f} s of CPU work); C restatement of ark-groth16 0.5 prove() "
                         "(arkworks itself is not buildable offline)",
               "host_cpu_count": os.cpu_count()}

    # ---------------- roofline of the dominant kernel (k_bucket_accumulate<Fq>) ----------------
    acc_ms, acc_cnt = stages["msm_accumulate_g1"]
    per_launch_ms = acc_ms / max(acc_cnt, 1)
    # one G1 MSM of length L = 96 L algorithmic bytes (64 B point + 32 B scalar)
    shard_w, shard_h = info["shard_w"], info["shard_h"]
    avg_len = (3 * shard_w + shard_h) / 4.0          # launches per step: A, B1, L (witness) and H
    alg_bytes = 96.0 * avg_len
    achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if world == 1 and os.path.exists(tpath):
        t = json.load(open(tpath))
        if t.get("log2_domain") == k:
            traffic = t["traffic_bytes_per_launch"]
    roofline = {"bound": "hbm", "kernel": "k_bucket_accumulate<Fq>", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_unit": "bytes per launch (PMC: profiles/pmc_traffic.json)",
                "avg_launch_ms": per_launch_ms, "launches_per_step": acc_cnt / args.steps,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "the kernel is integer-ALU bound (254-bit Montgomery arithmetic on v_mad_i64_i32), "
                        "not HBM bound: see `alu` and DESIGN.md section 4-5"}
    # supplementary: the same launches against the micro-benchmarked integer multiply-add issue peak
    W_w, W_h = info["W_w"], info["W_h"]
    madds = (3 * shard_w * W_w + shard_h * W_h) / 4.0          # mixed additions per launch (upper bound)
    VMAD_PER_MADD = 1557.0                                     # 8 products + 2 squarings on 9x29-bit limbs
    VMAD_PEAK = 30.1e12                                        # profiles/r01_instr_rates.txt
    alu = {"kernel": "k_bucket_accumulate<Fq>", "mixed_additions_per_s": madds / (per_launch_ms * 1e-3),
           "vmad_per_s": madds * VMAD_PER_MADD / (per_launch_ms * 1e-3), "vmad_peak_per_s": VMAD_PEAK,
           "frac": madds * VMAD_PER_MADD / (per_launch_ms * 1e-3) / VMAD_PEAK,
           "alu_only_ceiling_mixed_additions_per_s": 16.7e9}

    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": "Groth16 constraints/sec (BN254)", "value": m * args.steps / elapsed,
        "unit": "constraints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "int32x9 (29-bit limbs of 254-bit Montgomery integers)",
        "data": "synthetic",
        "config": {"workload": f"synthetic squaring-chain R1CS, 2^{k}-2 constraints, BN254, full prove "
                               "(witness map + 4 G1 MSM + 1 G2 MSM + finalize), trapdoor key minted on GPU",
                   "log2_domain": k, "num_constraints": m, "n_vars": n_vars,
                   "parallelism": (f"msm-point-range-shard x{world}" + (" + four-step witness map (2 all-to-all)" if dist_wm else " (witness map replicated)")) if world > 1 else "single-gpu",
                   "msm": info},
        "roofline": roofline, "alu": alu, "cpu_baseline": cpu, "parity": parity,
        "stages_ms_per_step": {n: ms / args.steps for n, (ms, _c) in stages.items()},
        "setup_s": t_setup, "ms_per_step_with_host_witness_upload": host_ms,
    }
    if cpu:
        out["gpu_over_cpu"] = out["value"] / cpu["value"]
    print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
