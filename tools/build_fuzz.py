"""Random triangle soups through the device-side BVH build against the host builder, bit for bit (test infrastructure).
usage (GPU box): python3 tools/build_fuzz.py [cases]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import numpy as np
srt = importlib.import_module("simple-ray-tracer_amd")
import orc


def soup(rng):
    n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 1100), rng.integers(1000, 2200), rng.integers(2000, 60000)]))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        c = rng.integers(0, 5, (n, 3, 3)).astype(np.float32)                       # few distinct keys: long runs of ties
    elif kind == 1:
        c = rng.standard_normal((n, 3, 3)).astype(np.float32)
    elif kind == 2:
        c = np.sort(rng.standard_normal((n, 1, 3)).astype(np.float32), axis=0) + np.float32(0)   # already sorted on every axis
        c = np.repeat(c, 3, axis=1)
    elif kind == 3:
        c = np.repeat(np.sort(rng.standard_normal((n, 1, 3)).astype(np.float32), axis=0)[::-1], 3, axis=1)  # reverse sorted
    elif kind == 4:
        c = np.zeros((n, 3, 3), np.float32) + np.float32(rng.standard_normal())     # all keys equal
    else:
        c = (rng.integers(0, 3, (n, 1, 3)) + rng.random((n, 3, 3)) * 1e-6).astype(np.float32)  # clusters
    d = orc.empty_scene()
    v = np.zeros((3 * n, 8), np.float32); v[:, :3] = c.reshape(-1, 3); v[:, 5] = 1.0
    d["verts"], d["tris"] = v, np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    d["mesh_vert_off"], d["mesh_tri_off"] = np.array([0, 3 * n], np.uint32), np.array([0, n], np.uint32)
    d["materials"] = np.array([[1, 1, 1, 0.2, 0.2, 0.2, 10.0, 1.0]], np.float32); d["mesh_tex"] = np.array([-1], np.int32)
    return d, n, kind


def run(first, count):
    srt.set_device(0)
    bad = 0
    for seed in range(first, first + count):
        d, n, kind = soup(np.random.default_rng(seed))
        scene = srt.Scene.from_arrays(d)
        got = srt.Bvh(scene, srt.F_ACCEL).download()
        want = srt.HostBvh(scene, srt.F_ACCEL).flatten(0)
        if not all(got[k].tobytes() == want[k].tobytes() for k in ("nodes", "tris", "attrs", "scene_order")):
            bad += 1
            print("seed", seed, "n", n, "kind", kind, "MISMATCH", flush=True)
    return bad


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    print("done", N, "soups,", run(0, N), "mismatches")
