cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
export G16_AMD_LIB=$PWD/circom_compat_amd/libg16_var_gather.so
for m in none 0x3ff 0xfffff none 0x3ff; do
  if [ $m = none ]; then unset G16_DEBUG_GATHER_MASK; else export G16_DEBUG_GATHER_MASK=$m; fi
  python bench.py --steps 10 --warmup 2 --cpu-log2 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d['stages_ms_per_step']
print('mask=$m ms=%.3f acc_g1=%.3f acc_g2=%.3f red=%.2f verifies=%s' % (d['ms_per_step'], s['msm_accumulate_g1'], s['msm_accumulate_g2'], s['msm_reduce'], d['parity']['proof_verifies']))"
done
