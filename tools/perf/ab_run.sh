#!/bin/bash
# On the GPU box: bench the previous revision (scratch/ab_prev) and the working tree alternately, ABAB.
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
: > gpurun_out/ab.txt
ARGS=${AB_ARGS:---steps 40 --no-cpu-baseline --no-g-forward --no-kernel-timer --no-f32-mode}
for i in 1 2; do
  for side in prev new; do
    if [ $side = prev ]; then cd $ROOT/scratch/ab_prev; else cd $ROOT; fi
    python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$side', d['value'], d['ms_per_step'])" >> $ROOT/gpurun_out/ab.txt
  done
done
cat $ROOT/gpurun_out/ab.txt
