#!/usr/bin/env python3
"""Where does the overlapped host pipeline (visp_depthany_pipeline_*) lose time against the device-resident step? Times, at the
north-star shape, the resident hipGraph step, the pipeline fed from pageable memory (host memcpy into the pinned slot) and the
pipeline with the slot filled in place (no host copy), for 2..4 slots."""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

if "--torch" in sys.argv:
    import torch  # before the library touches HIP, as bench.py does

    torch.cuda.init()

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from visioncpp_amd import synth, vision  # noqa: E402

B, W, H = 32, 518, 518
path = Path(tempfile.gettempdir()) / "probe_da.gguf"
synth.write_gguf(path, synth.SMALL, seed=0)
dev = vision.Device.init(vision.Backend.gpu)
model = vision.Model.load(path, dev, vision.Arch.depth_anything)
imgs = synth.images(4, W, H, seed=1)
imgs = np.concatenate([imgs] * 8)[:B]
print("HIP runtime mapped in this process:", sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln or "libhsa-runtime" in ln}), flush=True)
model.use_graph(True)
if "--torch" in sys.argv:  # as bench.py: torch owns the buffers and the stream the graph is captured on
    src = torch.from_numpy(imgs).cuda()
    dst = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream().cuda_stream
    for _ in range(3):
        model.compute_batch_device(src.data_ptr(), B, W, H, dst.data_ptr(), None, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        model.compute_batch_device(src.data_ptr(), B, W, H, dst.data_ptr(), None, stream)
    torch.cuda.synchronize()
    print(f"resident (torch stream, async): {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
else:
    rgb = vision.DeviceBuffer.from_numpy(imgs)
    out = vision.DeviceBuffer(B * W * H * 4)
    for _ in range(3):
        model.compute_batch_device(rgb.ptr, B, W, H, out.ptr)
    t0 = time.perf_counter()
    for _ in range(30):
        model.compute_batch_device(rgb.ptr, B, W, H, out.ptr)
    print(f"resident (blocking call per step): {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
for slots in (3,):
    for in_place in (False, True):
        pipe = vision.DepthPipeline(model, B, W, H, n_slots=slots)
        tickets = []
        for _ in range(slots):
            pipe.input_view()[...] = imgs
            tickets.append(pipe.submit(None))
        while tickets:
            pipe.wait(tickets.pop(0), copy=False)
        if not in_place:  # the host copy alone: pageable numpy batch -> pinned slot
            v = pipe.input_view()
            t0 = time.perf_counter()
            for _ in range(10):
                v[...] = imgs
            dt = (time.perf_counter() - t0) / 10
            print(f"host copy into the pinned slot: {dt * 1e3:.3f} ms per batch = {imgs.nbytes / dt / 1e9:.1f} GB/s", flush=True)
        n = 40
        t0 = time.perf_counter()
        for i in range(n):
            if in_place:
                pipe.input_view()  # the producer would decode into it; nothing copied here
                tickets.append(pipe.submit(None))
            else:
                tickets.append(pipe.submit(imgs))
            if len(tickets) == slots:
                pipe.wait(tickets.pop(0), copy=False)
        while tickets:
            pipe.wait(tickets.pop(0), copy=False)
        dt = (time.perf_counter() - t0) / n
        print(f"pipeline slots={slots} {'in-place input' if in_place else 'pageable input '}: {dt * 1e3:.3f} ms/step", flush=True)
        pipe.close()
