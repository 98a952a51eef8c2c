# same-box A/B: two-dimensional bucket reduction (row / column sums + two small weighted sums) vs running sums
# NOTE: the kernels / knob this script A/B-ed were measured and dropped; the env variables are no longer read.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
one() { env "${@:2}" python bench.py --steps 10 --warmup 3 --cpu-log2 0 --log2 ${K:-22} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['stages_ms_per_step']
print('$1 k=${K:-22}', round(d['ms_per_step'],3), 'reduce', round(st.get('msm_reduce',0),3), 'acc', round(st['msm_accumulate_g1']+st['msm_accumulate_g1_pair']+st['msm_accumulate_g2'],2), d['parity'])"
}
for i in 1 2; do
  one old G16_RED2D_MIN_LOG2=0
  one 2d G16_RED2D_MIN_LOG2=17
done
K=21 one old G16_RED2D_MIN_LOG2=0
K=21 one 2d G16_RED2D_MIN_LOG2=17
K=20 one old G16_RED2D_MIN_LOG2=0
K=20 one 2d16 G16_RED2D_MIN_LOG2=16
timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "headline_sizes and not 24" 2>&1 | tail -2
