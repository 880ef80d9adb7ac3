#!/bin/bash
# diagnostic sweep of the block kernel's DBG variants (library built with -DVISP_BLOCK_DIAG): phase stamps per variant
for d in ${DBGS:-0 1 3 16 17}; do
  echo "== VISP_BLOCK_DBG=$d"
  VISP_BLOCK_DBG=$d timeout -k 10 120 python - <<'PY'
import sys
sys.path.insert(0, 'tools')
from bench_block import block_case, block_stamps
block_case(23)
block_stamps(23)
PY
done
