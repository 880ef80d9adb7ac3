#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel trace of bench.py, then the PMC passes (one counter set per pass, kernel trace
# only -- no sys/hip trace next to --pmc) on tools/bench_block.py --pmc. Outputs under gpurun_out/prof_<tag>/.
set -e
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --min-seconds 0 --no-pipeline > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
db=$(ls $out/trace/*/*.db $out/trace/*.db 2>/dev/null | head -1)
echo "db: $db"
python3 tools/rocpd_stats.py "$db" $out/kernel_stats.csv
python3 tools/rocpd_stats.py "$db" $out/kernel_stats_by_grid.csv --by-grid
rm -rf $out/trace
for pass in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" GRBM_GUI_ACTIVE; do
  name=$(echo $pass | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/pmc_$name -o p -- python3 tools/bench_block.py --pmc --quick > $out/pmc_$name.log 2>&1 && echo "pass $name ok"
  f=$(ls $out/pmc_$name/*/*counter_collection.csv $out/pmc_$name/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|dino_block|attention_kernel" "$f" > $out/${name}_pass.csv
  rm -rf $out/pmc_$name
done
ls -la $out
