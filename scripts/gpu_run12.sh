# functional runs of the in-library multi-device path at size / with the skewed workload (all ranks on GPU 0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
export G16_BENCH_BACKEND=gloo
timeout 600 python bench.py --workload dense-skewed --log2 18 --gpus 4 --steps 3 --cpu-log2 14 > gpurun_out/r02_inlib4_dense18.json 2> gpurun_out/r02_inlib4_dense18.err; echo "dense inlib4 rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r02_inlib4_dense18.json')); print(d['ms_per_step'], d['parity'], d['config']['witness_fraction_in_{0,1}'])"
timeout 900 python bench.py --log2 22 --gpus 8 --steps 3 --cpu-log2 14 > gpurun_out/r02_inlib8_k22_onegpu.json 2> gpurun_out/r02_inlib8_k22_onegpu.err; echo "k22 inlib8 rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r02_inlib8_k22_onegpu.json')); print(d['ms_per_step'], d['parity'], d['config']['msm'])"
tail -3 gpurun_out/r02_inlib8_k22_onegpu.err
