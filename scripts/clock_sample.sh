#!/bin/bash
# clock_sample.sh LABEL CMD... -- runs CMD and samples `rocm-smi --showclocks --showpower` every ~0.3 s
# while it runs (lines prefixed with LABEL on stdout): the DVFS evidence of DESIGN.md section 5.
label=$1; shift
"$@" &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -e "s/^/$label /"
  sleep 0.3
done
wait $pid
