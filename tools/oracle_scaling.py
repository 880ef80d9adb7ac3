#!/usr/bin/env python3
"""Times the CPU oracle (Depth-Anything-V2-Small, one 518x518 image) at several OpenMP thread counts."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from oracle import oracle  # noqa: E402
from visioncpp_amd import synth  # noqa: E402

cfg = synth.SMALL
sd = synth.state_dict(cfg, 0)
tensors, conv2d = synth.gguf_tensors(sd)
om = oracle.Model(tensors, conv2d, "whcn")
params = oracle.make_params(cfg.patch_size, cfg.embed_dim, cfg.n_layers, cfg.n_heads, cfg.image_size, 14, cfg.feature_layers)
img = synth.images(1, 518, 518, seed=1234)[0]
om.compute(params, img)
for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    oracle.set_num_threads(n)
    t0 = time.perf_counter()
    om.compute(params, img)
    print(f"threads={n:4d}: {time.perf_counter() - t0:.3f} s/image", flush=True)
