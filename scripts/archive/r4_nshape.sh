#!/bin/bash
# round 4: the driver's N > 1 launch shapes, functionally, on ONE GPU (ranks time-share device 0)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r4_nshape; mkdir -p $O; export TMPDIR=/tmp
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', 'n_gpus', d['n_gpus'], round(d['ms_per_step'],2), 'ms', d['parity'], d['config'].get('parallelism'), d.get('fallback_reason'))" 2>&1 | cut -c1-300; }
G16_BENCH_BACKEND=gloo G16_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 8 --log2 22 --steps 3 --warmup 1 > $O/r04_bench_inlib8_one_gpu_functional.json 2> $O/inlib8.err; show $O/r04_bench_inlib8_one_gpu_functional.json "single-process in-library N=8 @2^22"
G16_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --log2 18 --steps 3 --warmup 1 > $O/torchrun_inlib2.json 2> $O/torchrun_inlib2.err; show $O/torchrun_inlib2.json "torchrun in-library N=2 @2^18"
G16_BENCH_BACKEND=gloo G16_BENCH_FAIL_INLIB=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --log2 18 --steps 3 --warmup 1 > $O/torchrun_fallback2.json 2> $O/torchrun_fallback2.err; show $O/torchrun_fallback2.json "torchrun fallback to ranks N=2 @2^18"
G16_BENCH_BACKEND=gloo G16_BENCH_MODE=ranks timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 4 --log2 18 --steps 3 --warmup 1 > $O/torchrun_ranks4.json 2> $O/torchrun_ranks4.err; show $O/torchrun_ranks4.json "torchrun ranks N=4 @2^18"
tail -3 $O/*.err | cut -c1-200 | tail -20
