#!/usr/bin/env python3
"""Design study (one GPU): time of every tile of the dragon frame rendered on its own, for a few tile sizes, so that
tile orders and sizes for N GPUs can be compared offline (tools/sim_tiles.py).  Writes gpurun_out/tile_costs.json."""
import importlib
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

srt = importlib.import_module("simple-ray-tracer_amd")
tiling = importlib.import_module("simple-ray-tracer_amd.distributed")
srt.set_device(0)
d, cfg = bench.workload("dragon", srt)
W, H = cfg["width"], cfg["height"]
bvh = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
cam = srt.make_camera(W, H)
out = {}
for tw, th in ((480, 270), (240, 270), (240, 135), (480, 135)):
    rects = tiling.tile_rects(W, H, tw, th)
    buf = torch.zeros((th, tw, 3), dtype=torch.float32, device="cuda")
    costs = []
    for rep in range(2):
        costs = []
        for r in rects:
            p = bvh.params(cam, cfg["features"], cfg["nsx"], cfg["nsy"], cfg["max_depth"], 0, r, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bvh.render_device(p, buf.data_ptr(), tw, th, r[0], r[1], torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            costs.append((time.perf_counter() - t0) * 1e3)
    out[f"{tw}x{th}"] = costs
    print(tw, th, len(costs), round(sum(costs), 1), round(max(costs), 2), flush=True)
json.dump(out, open("gpurun_out/tile_costs.json", "w"))
