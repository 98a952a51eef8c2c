# bench.py launch shapes after the fallback restructuring (functional, every rank on ONE GPU)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['parallelism'][:110], d.get('fallback_reason'), d['parity'])"; }
python bench.py --log2 18 --steps 3 --warmup 1 --cpu-log2 14 2>/dev/null | tail -1 | show single
G16_BENCH_BACKEND=gloo python bench.py --gpus 2 --log2 18 --steps 3 --warmup 1 --cpu-log2 14 2>/dev/null | tail -1 | show inlib-world1
G16_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --log2 18 --steps 3 --warmup 1 2>gpurun_out/r15_a.err | tail -1 | show torchrun-inlib
G16_BENCH_FAIL_INLIB=1 G16_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --log2 18 --steps 3 --warmup 1 2>gpurun_out/r15_b.err | tail -1 | show torchrun-fallback
grep "bench.py:" gpurun_out/r15_b.err | head -3
G16_BENCH_MODE=ranks G16_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 4 --log2 18 --steps 3 --warmup 1 2>gpurun_out/r15_c.err | tail -1 | show torchrun-ranks4
# the real thing on this 1-GPU box: nccl backend with world 1 is plain single
tail -3 gpurun_out/r15_a.err
