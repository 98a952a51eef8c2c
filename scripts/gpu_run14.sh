# G1 reductions underneath the G2 accumulation (G16_HIDE_REDUCE: 0 = serial order, 1 = lean kernel on red, 2 = plain kernel on red)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for h in 0 1 2; do
  G16_HIDE_REDUCE=$h python bench.py --steps 10 --warmup 2 --cpu-log2 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d['stages_ms_per_step']
print('hide=$h rep$rep ms=%.3f acc_g1=%.3f acc_g2=%.3f red=%.2f verifies=%s' % (d['ms_per_step'], s['msm_accumulate_g1'], s['msm_accumulate_g2'], s['msm_reduce'], d['parity']['proof_verifies']))"
done; done
for k in 21 24; do for h in 0 1; do
  G16_HIDE_REDUCE=$h python bench.py --log2 $k --steps 5 --warmup 1 --cpu-log2 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('hide=$h k=$k ms=%.3f verifies=%s' % (d['ms_per_step'], d['parity']['proof_verifies']))"
done; done
