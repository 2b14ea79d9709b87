#!/usr/bin/env python3
"""Design study (CPU only): shadow queries of the dragon workload walked as 64-ray packets (one node per wave step,
every ray applying its own box tests) against the same queries walked one per lane.  Prints wave steps per ray."""
import importlib
import sys

import numpy as np

sys.path.insert(0, ".")
srt = importlib.import_module("simple-ray-tracer_amd")
standin = importlib.import_module("simple-ray-tracer_amd.standin")
from oracle import orc  # noqa: E402

LEAF = 0x80000000
rng = np.random.default_rng(1)
d = standin.dragon_standin()
d["lights"] = standin.dragon_lights()
scene = srt.Scene.from_arrays(d)
flat = srt.HostBvh(scene, srt.F_ACCEL).flatten()
nodes, tris = flat["nodes"], flat["tris"]
oscene = orc.Scene(d)
obvh = orc.Bvh(oscene, orc.F_ACCEL)
W, H = 3840, 2160
cam = orc.make_camera(W, H)
NPIX = int(sys.argv[1]) if len(sys.argv) > 1 else 200
# candidate pixels: random, keep those whose centre ray hits
px = rng.integers(0, W, 20000)
py = rng.integers(0, H, 20000)
hits = []
for x, y in zip(px, py):
    sub = [((x + (i + 0.5) / 4) / W * 2 - 1, (y + (j + 0.5) / 2) / H * 2 - 1) for j in range(2) for i in range(4)]
    rays = orc.camera_rays(cam, np.array(sub, np.float32))
    r, _ = obvh.intersect(rays, orc.F_ACCEL | orc.F_MISS_CULL)
    if r["hit"].any():
        for k in range(8):
            if r["hit"][k]:
                hits.append(rays[k, :3] + rays[k, 3:6] * r["t"][k])
        if len(hits) >= NPIX * 8:
            break
hits = np.array(hits, np.float32)
print("hit points", len(hits))
# shadow items in issue order: per hit the point light, then 10 x 10 jittered samples of the area light
L = d["lights"]
items_o, items_d = [], []
v0, e1, e2 = np.array([-0.2, 1.6, -0.2], np.float32), np.array([0.4, 0, 0], np.float32), np.array([0, 0, 0.4], np.float32)
for ip in hits:
    tg = [np.array([-1, 1, -1], np.float32)]
    for a in range(10):
        for b in range(10):
            tg.append(v0 + e1 * np.float32((a + rng.random()) / 10) + e2 * np.float32((b + rng.random()) / 10))
    for t in tg:
        dv = (t - ip).astype(np.float32)
        items_o.append(ip + dv / np.linalg.norm(dv) * np.float32(0.001))
        items_d.append(dv)
O = np.array(items_o, np.float32)
D = np.array(items_d, np.float32)
n_items = len(O)
print("shadow items", n_items)


def slab(lo, hi, o, dd):
    with np.errstate(all="ignore"):
        t0 = (lo[None, :] - o) / dd
        t1 = (hi[None, :] - o) / dd
    tin = np.minimum(t0, t1).max(axis=1)
    tout = np.maximum(t0, t1).min(axis=1)
    return ~((tin > tout) | (tout < 0) | (tin > 1.0))


def tri_hit(first, cnt, o, dd):
    occl = np.zeros(len(o), bool)
    for k in range(first, first + cnt):
        t = tris[k]
        dn = dd @ t["n"]
        on = o @ t["n"]
        with np.errstate(all="ignore"):
            nt = (t["D"] - on) / dn
        ok = (dn != 0) & (nt >= 0) & (nt <= 1)
        p = o + dd * nt[:, None]
        for (a, b, c) in ((t["v1"], t["v2"], t["c0"]), (t["v0"], t["v2"], t["c1"]), (t["v0"], t["v1"], t["c2"])):
            ok &= (np.cross(b - a, p - a) @ c) >= 0
        occl |= ok & (nt < 0.99)
    return occl


def walk(o, dd):
    """Packet walk; returns (wave inner steps, wave leaf steps, lane inner visits, lane leaf visits)."""
    n = len(o)
    alive = slab(flat["root_box"][:3], flat["root_box"][3:], o, dd)
    stack = [(flat["root_ref"], alive.copy())]
    wi = wl = li = ll = 0
    while stack:
        ref, m = stack.pop()
        m = m & alive
        c = int(m.sum())
        if c == 0:
            continue
        if ref & LEAF:
            wl += 1
            ll += c
            idx = np.nonzero(m)[0]
            oc = tri_hit(ref & 0xFFFFFF, (ref >> 24) & 0x7F, o[idx], dd[idx])
            alive[idx[oc]] = False
            continue
        wi += 1
        li += c
        nd = nodes[ref]
        idx = np.nonzero(m)[0]
        pl = np.zeros(n, bool)
        pr = np.zeros(n, bool)
        pl[idx] = slab(nd["lbox"][:3], nd["lbox"][3:], o[idx], dd[idx])
        pr[idx] = slab(nd["rbox"][:3], nd["rbox"][3:], o[idx], dd[idx])
        if pl.any():
            stack.append((int(nd["left"]), pl))
        if pr.any():
            stack.append((int(nd["right"]), pr))
    return wi, wl, li, ll


tot = np.zeros(4, np.int64)
for s in range(0, n_items - 63, 64):
    tot += walk(O[s:s + 64], D[s:s + 64])
nr = (n_items // 64) * 64
print(f"per ray: lane inner visits {tot[2] / nr:.2f}, lane leaf visits {tot[3] / nr:.2f}")
print(f"packet walk: wave inner steps/ray {tot[0] / nr:.3f} ({tot[2] / max(tot[0], 1):.1f} lanes/step), "
      f"wave leaf steps/ray {tot[1] / nr:.3f} ({tot[3] / max(tot[1], 1):.1f} lanes/step)")
print("per-lane kernel as measured: wave inner steps/ray 0.741 (47.5 lanes/step), wave leaf steps/ray 0.164 (22.6 lanes/step)")
