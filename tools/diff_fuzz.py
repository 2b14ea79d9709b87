"""Differential fuzz (test infrastructure: it runs the CPU oracle): random scenes - degenerate triangles, integer grids,
flat and tiny clusters, magnitudes outside the fast-division ranges, 1 to 3000 triangles - with random rays (axis
parallel, zero directions, origins on vertices), HIP path against oracle: hits, t bits, triangle ids, attributes and
visit counters bit for bit under both visiting rules and without the tree; every fourth scene also as a frame under a
rotating feature set (reflections, transparency, multisampling, motion blur, depth of field, environment map).
usage (GPU box): python3 tools/diff_fuzz.py [scenes]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import numpy as np
srt = importlib.import_module("simple-ray-tracer_amd")
import orc

def scene(rng):
    n = int(rng.integers(1, 120)) if rng.random() < 0.7 else int(rng.integers(120, 3000))
    kind = rng.integers(0, 5)
    if kind == 0:
        c = rng.integers(-3, 4, (n, 3, 3)).astype(np.float32)                       # integer grid: many ties, degenerate triangles
    elif kind == 1:
        c = rng.standard_normal((n, 3, 3)).astype(np.float32)
    elif kind == 2:
        c = (rng.standard_normal((n, 3, 3)) * 10.0 ** rng.integers(-30, 25)).astype(np.float32)  # outside the fast-division ranges
    elif kind == 3:
        c = rng.standard_normal((n, 3, 3)).astype(np.float32); c[:, :, int(rng.integers(0, 3))] = np.float32(rng.integers(-1, 2))  # flat: axis-aligned planes
    else:
        base = rng.standard_normal((n, 1, 3)).astype(np.float32); c = base + (rng.standard_normal((n, 3, 3)) * 1e-3).astype(np.float32)
    d = orc.empty_scene()
    v = np.zeros((3 * n, 8), np.float32); v[:, :3] = c.reshape(-1, 3); v[:, 3:6] = rng.standard_normal((3 * n, 3)).astype(np.float32); v[:, 6:8] = rng.random((3 * n, 2)).astype(np.float32)
    d["verts"], d["tris"] = v, np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    d["mesh_vert_off"], d["mesh_tri_off"] = np.array([0, 3 * n], np.uint32), np.array([0, n], np.uint32)
    d["materials"] = np.array([[0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 12.0, float(rng.choice([1.0, 1.0, 0.4, 1.5]))]], np.float32); d["mesh_tex"] = np.array([-1], np.int32)
    d["lights"] = np.stack([orc.point_light(tuple(rng.standard_normal(3) * 2)), orc.segment_light(tuple(rng.standard_normal(3)), tuple(rng.standard_normal(3)), (1, 0, 0), (0, 1, 0))]).astype(np.float32)
    return d, c

def rays(rng, c, m):
    lo, hi = c.reshape(-1, 3).min(0), c.reshape(-1, 3).max(0)
    span = np.maximum(hi - lo, np.float32(1e-30))
    org = (lo + span * (rng.random((m, 3)) * 3 - 1)).astype(np.float32)
    tgt = (lo + span * rng.random((m, 3))).astype(np.float32)
    dirs = (tgt - org).astype(np.float32)
    z = rng.random((m, 3)) < 0.1
    dirs[z] = 0.0                                      # axis-parallel rays, some with all components zero
    org[rng.random((m, 3)) < 0.05] = 0.0
    t = np.where(rng.random((m, 1)) < 0.5, np.float32(1.0), np.finfo(np.float32).max).astype(np.float32)
    r = np.concatenate([org, dirs, t], 1)
    # rays that start exactly on vertices / pass exactly through vertices
    k = m // 4
    vv = c.reshape(-1, 3)
    r[:k, :3] = vv[rng.integers(0, len(vv), k)]
    r[k:2 * k, 3:6] = vv[rng.integers(0, len(vv), k)] - r[k:2 * k, :3]
    return np.ascontiguousarray(r.astype(np.float32))

def run(first, count):
    """Scenes first .. first + count - 1; returns the number of mismatches (each is printed)."""
    srt.set_device(0)
    bad = 0
    for seed in range(first, first + count):
        if (seed - first) % 500 == 499:
            print("...", seed - first + 1, "scenes,", bad, "mismatches", flush=True)  # a long run must keep writing
        rng = np.random.default_rng(seed)
        d, c = scene(rng)
        try:
            ob, gb = orc.Bvh(d, orc.F_ACCEL), srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
        except Exception as e:
            print("seed", seed, "build:", type(e).__name__, str(e)[:100]); continue
        r = rays(rng, c, 600)
        for feat in (orc.F_ACCEL, orc.F_ACCEL | orc.F_MISS_CULL, orc.F_ACCEL | orc.F_MISS_CULL | orc.F_NORMAL_INTERP | orc.F_TEXTURE, 0):
            want, wc = ob.intersect(r.copy(), feat)
            got, _, gc = gb.intersect(r.copy(), feat)
            same = (np.array_equal(got["hit"], want["hit"]) and np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32))
                    and np.array_equal(got["tri"], want["tri"]) and gc["inner_visits"] == wc["inner_visits"] and gc["tri_tests"] == wc["tri_tests"])
            h = want["hit"] == 1
            if same and h.any():
                for f in ("normal", "bary", "uv"):
                    a, b = got[f][h].view(np.uint32), want[f][h].view(np.uint32)
                    nan = np.isnan(got[f][h]) & np.isnan(want[f][h])
                    same = same and bool(((a == b) | nan).all())
            if not same:
                bad += 1
                print("seed", seed, "feat", feat, "MISMATCH hits", int((got["hit"] != want["hit"]).sum()), "t", int((got["t"].view(np.uint32) != want["t"].view(np.uint32)).sum()),
                      "visits", gc["inner_visits"], wc["inner_visits"], "tri", gc["tri_tests"], wc["tri_tests"], flush=True)
        if seed % 4 == 0:
            feat = orc.F_ACCEL | orc.F_SHADING | orc.F_HARD_SHADOW | orc.F_SOFT_SHADOW | orc.F_RECURSIVE | orc.F_MISS_CULL
            k = (seed // 4) % 9
            nsx = nsy = 1
            if k == 1:
                feat |= orc.F_TRANSPARENCY
            if k == 2:
                feat |= orc.F_MULTISAMPLE | orc.F_NORMAL_INTERP
                nsx, nsy = 2, 2
            if k in (0, 2, 7):
                feat |= orc.F_COARSE_SHADOW       # shadow queries on the coarse four-wide nodes
            if k == 3:
                feat &= ~orc.F_MISS_CULL          # the reference's own visiting rule
            if k == 4:
                feat &= ~orc.F_ACCEL              # every triangle, no tree
            if k == 5:
                feat &= ~(orc.F_SOFT_SHADOW | orc.F_RECURSIVE)
            depth = int(rng.integers(0, 6))
            lens, env, gscene = None, None, gb
            if k == 6:
                feat |= orc.F_MOTION_BLUR
                lens = dict(samples=int(rng.integers(1, 5)), time0=0.1, time1=0.8, direction=tuple(float(x) for x in rng.standard_normal(3)), focal_length=3, aperture=0.05)
            if k == 7:
                feat |= orc.F_DEPTH_OF_FIELD
                lens = dict(samples=int(rng.integers(1, 5)), time0=0.0, time1=1.0, direction=(1.0, 0.0, 0.0), focal_length=int(rng.integers(1, 5)), aperture=float(rng.random() * 0.2))
            if k == 8:
                feat |= orc.F_ENVIRONMENT | (orc.F_TRANSPARENCY if seed % 8 == 0 else 0)
                env = np.ascontiguousarray(rng.random((int(rng.integers(3, 40)), int(rng.integers(4, 50)), 3)).astype(np.float32))
                sc = srt.Scene.from_arrays(d)
                sc.set_environment(env)
                gscene = srt.Bvh(sc, srt.F_ACCEL)
            extra = {}
            if lens is not None:
                extra["lens"] = lens
            want = ob.render(orc.make_camera(24, 18), feat, nsx=nsx, nsy=nsy, max_depth=depth, seed=seed, **extra, **({"environment": env} if env is not None else {}))
            got = gscene.render(srt.make_camera(24, 18), feat, nsx=nsx, nsy=nsy, max_depth=depth, seed=seed, **extra)
            ok = np.array_equal(got["tri"], want["tri"]) and np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32))
            fin = np.isfinite(want["rgb"]) & np.isfinite(got["rgb"])
            err = float(np.abs(got["rgb"][fin] - want["rgb"][fin]).max()) if fin.any() else 0.0
            same_nonfinite = np.array_equal(np.isfinite(want["rgb"]), np.isfinite(got["rgb"]))
            cnt = (got["counters"]["shadow_rays"], got["counters"]["path_rays"]) == (want["counters"]["shadow_rays"], want["counters"]["path_rays"])
            if (feat & orc.F_TRANSPARENCY) and d["materials"][0][7] > 1.0:
                cnt = True  # a hit that returns black (transparency above 1) sends on no reflection ray here; the reference follows it and discards the result
            if not (ok and err <= 1 / 255 and cnt and same_nonfinite):
                bad += 1
                print("seed", seed, "FRAME mismatch (variant", k, "depth", depth, "): ids/t", ok, "err", err, "counters", cnt, "nonfinite same", same_nonfinite, flush=True)
    return bad


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    print("done", N, "scenes,", run(0, N), "mismatches")
