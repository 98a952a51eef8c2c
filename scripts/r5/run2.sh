#!/bin/bash
# round 5, GPU call 2: single-level NTT tables A/B, bench.py's new legs (PMC by the run itself, clock sampler, both N > 1 shapes)
O=gpurun_out/r5_2; mkdir -p $O
export TMPDIR=/tmp
ls /sys/class/drm/ > $O/sysfs.txt 2>&1
for c in /sys/class/drm/card*/device; do echo "== $c" >> $O/sysfs.txt; cat $c/vendor >> $O/sysfs.txt 2>&1; ls $c/hwmon/*/ >> $O/sysfs.txt 2>&1; done
for f in /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input /sys/class/drm/card*/device/hwmon/hwmon*/power1_*; do echo "$f: $(cat $f 2>&1)" >> $O/sysfs.txt; done
timeout 900 python -m pytest tests/test_kernels.py tests/test_gpu_large.py -m gpu -x -q -k "ntt or witness_map or headline_sizes or libsnark" --durations=5 > $O/pytest_ntt.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ntt.log
tail -4 $O/pytest_ntt.log
# witness map alone, new vs two-level tables (same box)
for tl in 0 1; do
  G16_NTT_TWO_LEVEL=$tl timeout 300 python bench.py --mode parts --log2 22 --steps 5 --warmup 1 --cpu-log2 0 --no-pmc > $O/parts22_tl$tl.json 2> $O/parts22_tl$tl.err
  python -c "
import json; d=json.loads(open('$O/parts22_tl$tl.json').read().strip().splitlines()[-1]); print('two_level=$tl', d['parts_ms']['witness_map_ms'], d['ms_per_step'])"
done
for tl in 1 0 1 0; do
  G16_NTT_TWO_LEVEL=$tl timeout 300 python bench.py --steps 20 --warmup 3 --cpu-log2 0 --no-pmc > $O/b22_tl$tl.json 2> $O/b22_tl$tl.err
  python -c "
import json; d=json.loads(open('$O/b22_tl$tl.json').read().strip().splitlines()[-1]); print('two_level=$tl', d['ms_per_step'], d.get('clock_mhz'), d.get('power_w'), d['stages_ms_per_step']['witness_map'])"
done
# the default line with every new leg
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], d.get('clock_mhz'), d.get('power_w'), r['traffic'], r['traffic_source'][:200]); print(d.get('clock'))"
timeout 900 python -m pytest tests/test_bench_shapes.py -m gpu -x -q > $O/pytest_shapes.log 2>&1; echo "shapes rc=$?"; tail -5 $O/pytest_shapes.log
