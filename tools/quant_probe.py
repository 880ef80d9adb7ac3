#!/usr/bin/env python3
"""Grid-quantisation probe: attention launch time against the number of (image, head, q-block) blocks.
2112 blocks (batch 32) on 1024 resident slots (4 blocks per CU) is 2.06 rounds; if the time steps up between
2048 and 2112 blocks the tail round is what the step pays for."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from bench_kernels import attn_case  # noqa: E402

for rnd in range(2):
    for B in (15, 16, 17, 23, 24, 30, 31, 32, 33, 36, 40, 46, 47, 48):
        attn_case(B, 6, 1370)
