#!/bin/bash
# L2 / fabric counter passes on the block kernel at 118 / 247 / 343 workgroups (tools/bench_block.py --l2): is the per-workgroup time at a
# full chip an L2 effect (hit rate, requests) or the clock? One counter set per pass, kernel trace only. Output: gpurun_out/pmc_l2/<pass>.csv
out=gpurun_out/pmc_l2
mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
for pass in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/raw_$name -o p -- python3 tools/bench_block.py --l2 --quick > $out/$name.log 2>&1 && echo "pass $name ok" || echo "pass $name FAILED"
  f=$(ls $out/raw_$name/*/*counter_collection.csv $out/raw_$name/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" > $out/$name.txt <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "dino_block16" not in r["Kernel_Name"]: continue
    wgs = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
    acc[(wgs, r["Counter_Name"])].append(float(r["Counter_Value"]))
    acc[(wgs, "_us")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc): print(k[0], k[1], sum(acc[k]) / len(acc[k]), len(acc[k]))
PY
  rm -rf $out/raw_$name
done
cat $out/*.txt
