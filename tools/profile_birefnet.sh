#!/bin/bash
# rocprofv3 kernel trace of the BiRefNet and SWIN bench workloads + their bench lines (outputs under gpurun_out/prof_birefnet/)
set -e
out=gpurun_out/prof_birefnet
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 500 python3 bench.py --workload birefnet --profile-groups > $out/bench_birefnet.json 2> $out/bench_birefnet_groups.txt
timeout -k 10 300 python3 bench.py --workload swin --profile-groups > $out/bench_swin.json 2> $out/bench_swin_groups.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bf -- python3 bench.py --workload birefnet --steps 5 --warmup 2 --no-cpu-baseline > $out/under_rocprof.json 2> $out/under_rocprof.err
db=$(ls $out/trace/*/*.db $out/trace/*.db 2>/dev/null | head -1)
python3 tools/rocpd_stats.py "$db" $out/kernel_stats.csv
python3 tools/rocpd_stats.py "$db" $out/kernel_stats_by_grid.csv --by-grid
rm -rf $out/trace
ls -la $out
