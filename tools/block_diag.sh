#!/bin/bash
# diagnostic sweep of the block kernel's DBG variants (library built with -DVISP_BLOCK_DIAG)
for d in 0 1 3 4 7 11 15; do
  echo "== VISP_BLOCK_DBG=$d"
  VISP_BLOCK_DBG=$d timeout -k 10 120 python - <<'PY'
import sys
sys.path.insert(0, 'tools')
from bench_block import block_case
block_case(23)
block_case(23)
PY
done
