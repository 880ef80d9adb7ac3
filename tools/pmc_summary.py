#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of tools/profile_round.sh (one counter set per pass, kernel trace only, target
tools/bench_block.py --pmc: the north-star launch shapes B = 32, T = 1370, D = 384) into per-kernel figures per launch:

  * HBM traffic = FETCH_SIZE + WRITE_SIZE. Both counters are in KiB (MI355X_MICROARCH.md, HBM section). On gfx950 FETCH_SIZE tallies a 128-byte
    read request as 64 bytes, so it is DOUBLED. Round 4 settled this for the block kernel too (round 3 left it un-doubled because the raw
    counter happened to sit near the row bytes): TCC_EA0_RDREQ_32B = 0 and TCC_EA0_RDREQ x 128 B = the residual / attention rows + eight
    first-touch copies of the 3.4 MB weight stream (one per XCD L2) at 118, 247 and 343 workgroups (profiles/r04_pmc_block_l2/): every fabric
    read of these kernels is a 128-byte request -- the LDS-DMA weight stream and the 16-byte-per-lane row loads alike. WRITE_SIZE is exact.
  * MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs). SQ_VALU_MFMA_BUSY_CYCLES counts 32
    per v_mfma_f32_32x32x16_f16 summed over the chip (block kernel: 151.7 M = 32 x 4.74 M MFMAs = 155.1 GFLOP / 32768);
    GRBM_GUI_ACTIVE is summed over the 8 XCDs.

    python tools/pmc_summary.py gpurun_out/prof_r02 profiles/r02_pmc/traffic.json"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

GROUPS = {"dino_block16_kernel<true, true": "block", "dino_block16_kernel<false, true": "block_qkv0",
          "attention_kernel": "attention"}
FETCH_X2 = {"attention": True, "block": True, "block_qkv0": True}


def collect(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        for key, grp in GROUPS.items():
            if key in r["Kernel_Name"]:
                wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
                grp = f"{grp}@{wgs}wg"   # one entry per launch shape
                acc[grp][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[grp]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                acc[grp]["_wgs"].append(wgs)
                break
    return {g: {k: sum(v) / len(v) for k, v in d.items()} for g, d in acc.items()}


if __name__ == "__main__":
    src, out = Path(sys.argv[1]), Path(sys.argv[2])
    fetch, write = collect(src / "fetch_size_pass.csv"), collect(src / "write_size_pass.csv")
    sq, grbm = collect(src / "sq_valu_mfma_busy_cycles_pass.csv"), collect(src / "grbm_gui_active_pass.csv")
    res = {}
    for g in sorted(fetch):
        raw = fetch[g]["FETCH_SIZE"] * 1024.0
        x2 = FETCH_X2[g.split("@")[0]]
        fb = raw * (2.0 if x2 else 1.0)
        wb = write[g]["WRITE_SIZE"] * 1024.0
        cyc = grbm[g]["GRBM_GUI_ACTIVE"] / 8.0
        res[g] = {"workgroups": round(fetch[g]["_wgs"]), "fetch_bytes_raw": round(raw), "fetch_doubled": x2, "fetch_bytes": round(fb), "write_bytes": round(wb),
                  "hbm_bytes_per_launch": round(fb + wb), "launch_us_under_pmc": round(grbm[g]["_us"], 1),
                  "shader_clock_ghz": round(cyc / grbm[g]["_us"] / 1e3, 2),
                  "mfma_busy_pct": round(100.0 * sq[g]["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 1),  # of all 1024 SIMDs of the chip
                  "sq": {k: round(v) for k, v in sq[g].items() if not k.startswith("_")}}
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps({"source": "rocprofv3 --pmc passes of tools/profile_round.sh on tools/bench_block.py --pmc (sub-batch 11 = the product launch shape, and batch 32; T=1370, D=384); "
                                         "see tools/pmc_summary.py for units and corrections", "kernels": res}, indent=1))
    print(json.dumps(res, indent=1))
