# the driver's N = 8 command shapes at the headline size, every rank on ONE GPU (functional only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), d['n_gpus'], d['config']['parallelism'][:90], d.get('fallback_reason'), d['parity'])"; }
G16_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 5 --warmup 1 2>gpurun_out/r30_a.err | tail -1 | tee gpurun_out/r02_bench_inlib8_one_gpu_functional.json | show world1-inlib8
G16_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 8 --steps 5 --warmup 1 2>gpurun_out/r30_b.err | tail -1 | show torchrun8-inlib
tail -2 gpurun_out/r30_b.err
