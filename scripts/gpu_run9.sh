# functional runs of bench.py's N > 1 launch shapes on a 1-GPU box (every rank on GPU 0; never a measurement)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
export G16_BENCH_BACKEND=gloo
for N in 2 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N)) bench.py --gpus $N --log2 18 --steps 3 --warmup 1 --cpu-log2 14 > gpurun_out/r02_torchrun_inlib_$N.json 2> gpurun_out/r02_torchrun_inlib_$N.err; echo "inlib N=$N rc=$?"
  python -c "
import json; d=json.loads([l for l in open('gpurun_out/r02_torchrun_inlib_$N.json') if l.startswith('{')][-1]); print(d['n_gpus'], d['ms_per_step'], d['parity'], d['config']['parallelism'][:80])"
  G16_BENCH_MODE=ranks timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) bench.py --gpus $N --log2 18 --steps 3 --warmup 1 --cpu-log2 14 > gpurun_out/r02_torchrun_ranks_$N.json 2> gpurun_out/r02_torchrun_ranks_$N.err; echo "ranks N=$N rc=$?"
  python -c "
import json; d=json.loads([l for l in open('gpurun_out/r02_torchrun_ranks_$N.json') if l.startswith('{')][-1]); print(d['n_gpus'], d['ms_per_step'], d['parity'], d['config']['parallelism'][:80])"
done
tail -5 gpurun_out/r02_torchrun_ranks_4.err
