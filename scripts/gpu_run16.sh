# same-box A/B: A|B1 interleaved (one 128-byte line per gather, one paired launch) vs separate arrays
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
one() { # $1 = label, rest = env
  env "${@:2}" python bench.py --steps 10 --warmup 3 --cpu-log2 0 --log2 ${K:-22} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['stages_ms_per_step']
print('$1 k=${K:-22}', round(d['ms_per_step'],3), 'acc_g1', round(st.get('msm_acc_g1',0),3), 'acc_g2', round(st.get('msm_acc_g2',0),3), 'reduce', round(st.get('msm_reduce',0),3), d['parity'])"
}
for i in 1 2; do
  one separate G16_NO_PAIR_AB=1
  one paired G16_NO_PAIR_AB=0
done
K=20 one separate G16_NO_PAIR_AB=1
K=20 one paired G16_NO_PAIR_AB=0
K=18 one separate G16_NO_PAIR_AB=1
K=18 one paired G16_NO_PAIR_AB=0
timeout 900 python -m pytest tests/test_kernels.py tests/test_gpu_large.py -m gpu -x -q 2>&1 | tail -3
