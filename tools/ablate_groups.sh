#!/bin/bash
# What a run of launches costs END TO END in the overlapped step (timing only, results invalid): same box, interleaved rounds, a diagnostic build of the
# library (depthany.cpp compiled with -DVISP_TIMING_ABLATIONS, see Makefile ABLATE=1) that skips the launch indices named by VISP_ABLATE_LAUNCHES.
#   tools/ablate_groups.sh variants/ablate.so ROUNDS "" 39-48 28-38 ...
lib=$1; rounds=$2; shift 2
for rnd in $(seq 1 $rounds); do
  for r in "$@"; do
    echo "round $rnd skip [$r]: $(VISP_ABLATE_LAUNCHES=$r VISP_LIBRARY=$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --min-seconds 0 --no-pipeline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
  done
done
