#!/usr/bin/env python3
"""Gather-rate probe over the dragon stand-in's records: windows from L1-sized to everything, per-lane and quad fetches."""
import importlib
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
srt = importlib.import_module("simple-ray-tracer_amd")
standin = importlib.import_module("simple-ray-tracer_amd.standin")
srt.set_device(0)
d = standin.dragon_standin()
d["lights"] = standin.dragon_lights()
b = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
for window in (8 << 10, 16 << 10, 64 << 10, 512 << 10, 2 << 20, 8 << 20, 0):
    for quad in (False, True):
        gbs, used = b.probe_gather(window, 512, quad)
        print(f"window {used >> 10:6d} KiB  {'quad    ' if quad else 'per-lane'}  {gbs:9.1f} GB/s  = {gbs / 64 / 256 / 2.4:6.3f} records/cycle/CU at 2.4 GHz", flush=True)
