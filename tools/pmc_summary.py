#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (two separate passes, tools/bench_kernels.py
--quick) into per-kernel HBM traffic per launch. Units and corrections per MI355X_MICROARCH.md section HBM:
both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced streaming reads
(TCC_EA0_RDREQ tallied at 64 B for 128-B requests), so it is doubled; WRITE_SIZE is exact for 16-B/lane stores."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

GROUPS = {"attention_kernel": "attention", "ELi1ELi5ELb0": "gemm_qkv", "1, 5, false": "gemm_qkv", "1, 1, false": "gemm_fc1",
          "1, 3, false": "gemm_resid(fc2+out avg)"}


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for key, grp in GROUPS.items():
            if key in r["Kernel_Name"]:
                acc[grp].append(float(r["Counter_Value"]) * 1024.0)
                break
    return {k: sum(v) / len(v) for k, v in acc.items()}


if __name__ == "__main__":
    fetch_csv, write_csv, out = sys.argv[1:4]
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fb, wb = 2.0 * f.get(k, 0.0), w.get(k, 0.0)
        res[k] = {"fetch_bytes_corrected_x2": round(fb), "write_bytes": round(wb), "hbm_bytes_per_launch": round(fb + wb)}
    Path(out).write_text(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/bench_kernels.py --quick "
                                               "(B=32, T=1370, D=384); FETCH_SIZE doubled per MI355X_MICROARCH.md", "kernels": res}, indent=1))
    print(json.dumps(res, indent=1))
