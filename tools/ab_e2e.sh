#!/bin/bash
# Same-box end-to-end A/B of whole libraries (guide rule 24: interleaved rounds, one device):
#   tools/ab_e2e.sh ROUNDS [ENV=VAL,...,]lib1.so [ENV=VAL,...,]lib2.so ...
# prints ms_per_step / images/s of `bench.py` (hipGraph step, 40 steps) and the serial group times per entry and round.
rounds=$1; shift
for rnd in $(seq 1 $rounds); do
  for ent in "$@"; do
    lib=${ent##*,}; envs=""
    if [ "$lib" != "$ent" ]; then envs=$(echo "${ent%,*}" | tr ',' ' '); fi
    echo "round $rnd $ent: $(env $envs VISP_LIBRARY=$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --min-seconds 0 --no-pipeline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"], {k: d["kernel_groups_ms"][k] for k in ("block","attention")}, "dpt", round(sum(v for k, v in d["kernel_groups_ms"].items() if k.split("_")[0] in ("fusion","bilinear","neck","head")), 3), "mae" , d.get("cpu_baseline",{}).get("mae_gpu_vs_cpu"))')"
  done
done
