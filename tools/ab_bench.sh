#!/bin/bash
# Whole-network A/B of library builds in ONE gpurun call (boxes differ by a few %): alternates the variants.
# usage: bash tools/ab_bench.sh gpurun_ablate/lib_old.so gpurun_ablate/lib_new.so
for round in 1 2; do
  for lib in "$@"; do
    echo -n "$(basename $lib) round $round: "
    CF_LIB_PATH=$PWD/$lib python bench.py --no-cpu-baseline --no-roofline --steps 15 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'faces/s', d['ms_per_step'], 'ms')"
  done
done
