cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do for v in before now; do
  if [ $v = before ]; then export G16_AMD_LIB=$PWD/circom_compat_amd/libg16_var_before.so; else unset G16_AMD_LIB; fi
  python bench.py --steps 10 --warmup 2 --cpu-log2 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d['stages_ms_per_step']
print('$v rep$rep ms=%.3f acc_g1=%.3f acc_g2=%.3f red=%.2f verifies=%s pcie=%.2f' % (d['ms_per_step'], s['msm_accumulate_g1'], s['msm_accumulate_g2'], s['msm_reduce'], d['parity']['proof_verifies'], d['ms_per_step_pcie_inclusive']))"
done; done
unset G16_AMD_LIB
for k in 16 20; do for v in before now; do
  if [ $v = before ]; then export G16_AMD_LIB=$PWD/circom_compat_amd/libg16_var_before.so; else unset G16_AMD_LIB; fi
  python bench.py --log2 $k --steps 20 --warmup 3 --cpu-log2 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$v k=$k ms=%.3f' % d['ms_per_step'])"
done; done
unset G16_AMD_LIB
python bench.py --workload dense-skewed --log2 20 --steps 10 --cpu-log2 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('dense20 ms=%.3f frac01=%s verifies=%s' % (d['ms_per_step'], d['config']['witness_fraction_in_{0,1}'], d['parity']['proof_verifies']))"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
