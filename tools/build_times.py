#!/usr/bin/env python3
"""Where the time of BvhInterface's constructor goes on the dragon stand-in: host build, flatten, upload."""
import importlib
import os
import sys
import time

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
srt = importlib.import_module("simple-ray-tracer_amd")
standin = importlib.import_module("simple-ray-tracer_amd.standin")
d = standin.dragon_standin()
d["lights"] = standin.dragon_lights()
scene = srt.Scene.from_arrays(d)
for rep in range(3):
    t0 = time.perf_counter()
    hb = srt.HostBvh(scene, srt.F_ACCEL)
    t1 = time.perf_counter()
    flat = hb.flatten(0)
    t2 = time.perf_counter()
    print(f"host build {1e3 * (t1 - t0):7.1f} ms   flatten (incl. copy to numpy) {1e3 * (t2 - t1):7.1f} ms", flush=True)
if torch.cuda.is_available():
    srt.set_device(0)
    for rep in range(3):
        t0 = time.perf_counter()
        b = srt.Bvh(scene, srt.F_ACCEL)
        t1 = time.perf_counter()
        print(f"BvhInterface constructor (build + flatten + upload) {1e3 * (t1 - t0):7.1f} ms", flush=True)
