#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by RUNNING THE REFERENCE'S OWN OBJECT CODE
(oracle/_ref/libref.so, built from /root/reference by oracle/Makefile).  Runs only where the reference
checkout exists; the fixtures it writes are plain data (inputs and the reference's outputs).

  tests/golden/scenes/<name>.npz   the reference loader's output for each prebuilt scene
                                   (loadScenePrebuilt -> loadMesh -> tinyobj/stb; src/scene.cpp:5-94)
  tests/golden/leaf_math.npz       inputs and reference outputs of the leaf functions
  tests/golden/camera.npz          glm camera arithmetic as used by framework/src/trackball.cpp
  tests/golden/obj_cases/*.npz     the reference loader's output for the OBJ files authored in that directory
  tests/golden/screen_bmp.npz      float pixels and the bytes of the .bmp file the reference's writer makes of them

usage: make_golden.py [scenes|leaf|camera|obj|bmp ...]   (default: all)
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import orc  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DATA = "/root/reference/data"

# SceneType enum values, src/scene.h:15-26
SCENES = {"single_triangle": 0, "cube": 1, "cube_textured": 2, "cornell": 3, "cornell_parallelogram": 4,
          "monkey": 5, "teapot": 6, "custom": 9}

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


def p(a, t=f32p):
    return a.ctypes.data_as(t)


def gen_scenes():
    R = orc.ref()
    os.makedirs(os.path.join(GOLD, "scenes"), exist_ok=True)
    out = {}
    for name, typ in SCENES.items():
        h = R.ref_scene_load_prebuilt(typ, DATA.encode())
        assert h, name
        d = orc.ref_scene_to_dict(h)
        R.ref_scene_free(h)
        np.savez_compressed(os.path.join(GOLD, "scenes", name + ".npz"), **{k: d[k] for k in orc.SCENE_KEYS})
        print(f"{name}: meshes={len(d['materials'])} verts={len(d['verts'])} tris={len(d['tris'])} "
              f"textures={len(d['tex_w'])} lights={len(d['lights'])}")
        out[name] = d
    return out


def special_floats(rng, n):
    """Random values with exact zeros, signed zeros, repeated values and large/small magnitudes mixed in."""
    x = rng.standard_normal(n).astype(np.float32)
    k = rng.integers(0, 12, n)
    x[k == 0] = 0.0
    x[k == 1] = -0.0
    x[k == 2] = np.float32(1.0)
    x[k == 3] = np.float32(-1.0)
    x[k == 4] *= np.float32(1e-20)
    x[k == 5] *= np.float32(1e6)
    return x


def gen_leaf(scenes):
    R = orc.ref()
    rng = np.random.default_rng(20261111)
    g = {}

    # --- ray / triangle: triangles of the monkey and Cornell meshes hit by rays aimed at (or near) them
    tri_list = []
    for name in ("monkey", "cornell", "cube"):
        d = scenes[name]
        for m in range(len(d["materials"])):
            v = d["verts"][d["mesh_vert_off"][m]:d["mesh_vert_off"][m + 1], :3]
            t = d["tris"][d["mesh_tri_off"][m]:d["mesh_tri_off"][m + 1]]
            tri_list.append(v[t].reshape(-1, 9))
    tris = np.concatenate(tri_list)
    n = 6000
    pick = rng.integers(0, len(tris), n)
    T = tris[pick].copy()
    w = rng.random((n, 3)).astype(np.float32)
    w /= w.sum(1, keepdims=True)
    # a third of the targets exactly on a vertex / edge, a third outside
    kind = rng.integers(0, 6, n)
    w[kind == 0] = [1, 0, 0]
    w[kind == 1] = [0.5, 0.5, 0]
    w[kind == 2] = w[kind == 2] * np.float32(3) - np.float32(1)
    target = (T.reshape(n, 3, 3) * w[:, :, None]).sum(1).astype(np.float32)
    origin = (rng.standard_normal((n, 3)) * 2).astype(np.float32)
    direction = target - origin
    unit = rng.integers(0, 2, n).astype(bool)
    direction[unit] /= np.linalg.norm(direction[unit], axis=1, keepdims=True)
    tmax = np.where(rng.integers(0, 3, n) == 0, np.float32(1.0), np.finfo(np.float32).max).astype(np.float32)
    rays = np.concatenate([origin, direction, tmax[:, None]], 1).astype(np.float32)
    # degenerate triangles and axis-parallel rays
    T[:50, 3:6] = T[:50, 0:3]
    rays[50:150, 3 + rng.integers(0, 3)] = 0.0
    # fully random block with special values
    m2 = 2000
    T2 = special_floats(rng, m2 * 9).reshape(m2, 9)
    r2 = special_floats(rng, m2 * 7).reshape(m2, 7)
    r2[:, 6] = np.abs(r2[:, 6]) * 10
    T = np.ascontiguousarray(np.concatenate([T, T2]).astype(np.float32))
    rays = np.ascontiguousarray(np.concatenate([rays, r2]).astype(np.float32))
    out_rays = rays.copy()
    hit = np.zeros(len(rays), np.int32)
    R.ref_ray_triangle(p(T), p(out_rays), p(hit, i32p), len(rays))
    g["tri_tri9"], g["tri_rays"], g["tri_hit"], g["tri_t"] = T, rays, hit, out_rays[:, 6].copy()
    print("ray/triangle: %d cases, %d hits" % (len(rays), hit.sum()))

    # --- ray / box
    n = 6000
    lo = (rng.standard_normal((n, 3))).astype(np.float32)
    hi = lo + np.abs(rng.standard_normal((n, 3))).astype(np.float32)
    hi[:100] = lo[:100]  # flat boxes
    o = (rng.standard_normal((n, 3)) * 2).astype(np.float32)
    inside = rng.integers(0, 3, n) == 0
    o[inside] = ((lo + hi) * np.float32(0.5))[inside]
    onface = rng.integers(0, 8, n) == 0
    o[onface, 0] = lo[onface, 0]  # origin exactly on a face: 0/0 when d.x == 0
    tgt = lo + (hi - lo) * rng.random((n, 3)).astype(np.float32) * np.float32(1.5)
    d = (tgt - o).astype(np.float32)
    zero = rng.integers(0, 5, n) == 0
    d[zero, rng.integers(0, 3)] = 0.0
    tmax = np.where(rng.integers(0, 3, n) == 0, np.float32(1.0), np.finfo(np.float32).max).astype(np.float32)
    tmax[::7] = np.abs(rng.standard_normal(len(tmax[::7]))).astype(np.float32)
    box = np.ascontiguousarray(np.concatenate([lo, hi], 1).astype(np.float32))
    rays = np.ascontiguousarray(np.concatenate([o, d, tmax[:, None]], 1).astype(np.float32))
    out_rays = rays.copy()
    hit = np.zeros(n, np.int32)
    R.ref_ray_aabb(p(box), p(out_rays), p(hit, i32p), n)
    g["box_box6"], g["box_rays"], g["box_hit"], g["box_t"] = box, rays, hit, out_rays[:, 6].copy()
    print("ray/box: %d cases, %d pass" % (n, hit.sum()))

    # --- barycentrics + interpolation
    n = 4000
    pick = rng.integers(0, len(tris), n)
    T = np.ascontiguousarray(tris[pick].astype(np.float32))
    w = rng.random((n, 3)).astype(np.float32)
    w /= w.sum(1, keepdims=True)
    P = np.ascontiguousarray((T.reshape(n, 3, 3) * w[:, :, None]).sum(1).astype(np.float32))
    P[:20] = T[:20, 0:3]
    T[20:40, 3:6] = T[20:40, 0:3]
    T[20:40, 6:9] = T[20:40, 0:3]
    P[20:40] = T[20:40, 0:3]  # zero-area: sum == 0 branch
    bary = np.zeros((n, 3), np.float32)
    R.ref_barycentric(p(T), p(P), p(bary), n)
    N9 = np.ascontiguousarray(rng.standard_normal((n, 9)).astype(np.float32))
    nout = np.zeros((n, 3), np.float32)
    R.ref_interp_normal(p(N9), p(bary), p(nout), n)
    T6 = np.ascontiguousarray(rng.random((n, 6)).astype(np.float32))
    tout = np.zeros((n, 2), np.float32)
    R.ref_interp_texcoord(p(T6), p(bary), p(tout), n)
    g["bary_tri9"], g["bary_p"], g["bary_out"] = T, P, bary
    g["interp_n9"], g["interp_n_out"], g["interp_t6"], g["interp_t_out"] = N9, nout, T6, tout

    # --- texel fetch on default.png through the cube-textured scene (indices kept inside the image)
    h = R.ref_scene_load_prebuilt(SCENES["cube_textured"], DATA.encode())
    n = 3000
    uv = rng.random((n, 2)).astype(np.float32)
    uv[:, 1] = np.maximum(uv[:, 1], np.float32(1e-3))  # v == 0 reads past the image in the reference
    uv[:, 0] = np.minimum(uv[:, 0], np.float32(0.999))
    uv = np.ascontiguousarray(uv)
    tx = np.zeros((n, 3), np.float32)
    R.ref_acquire_texel(h, 0, p(uv), p(tx), n)
    R.ref_scene_free(h)
    g["texel_uv"], g["texel_out"] = uv, tx

    # --- shading and reflection
    n = 4000
    lp = np.ascontiguousarray((rng.standard_normal((n, 3)) * 2).astype(np.float32))
    lc = np.ascontiguousarray(rng.random((n, 3)).astype(np.float32))
    o = (rng.standard_normal((n, 3)) * 2).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    t = (rng.random(n) * 3).astype(np.float32)
    rays = np.ascontiguousarray(np.concatenate([o, d, t[:, None]], 1).astype(np.float32))
    nrm = np.ascontiguousarray(rng.standard_normal((n, 3)).astype(np.float32))
    kd = np.ascontiguousarray(rng.random((n, 3)).astype(np.float32))
    ks = np.ascontiguousarray(rng.random((n, 3)).astype(np.float32))
    sh = np.ascontiguousarray(rng.choice(np.array([1, 4, 10, 250, 0.5], np.float32), n))
    col = np.zeros((n, 3), np.float32)
    R.ref_compute_shading(p(lp), p(lc), p(rays), p(nrm), p(kd), p(ks), p(sh), p(col), n)
    refl = np.zeros((n, 7), np.float32)
    R.ref_reflection_ray(p(rays), p(nrm), p(refl), n)
    g.update(shade_lp=lp, shade_lc=lc, shade_rays=rays, shade_n=nrm, shade_kd=kd, shade_ks=ks, shade_sh=sh,
             shade_out=col, refl_out=refl)
    np.savez_compressed(os.path.join(GOLD, "leaf_math.npz"), **g)


def gen_camera():
    R = orc.ref()
    rng = np.random.default_rng(7)
    cfgs = [(50.0, 3.0, (0, 0, 0), (20, 20, 0), 64, 64), (50.0, 3.0, (0, 0, 0), (20, 20, 0), 1920, 1080),
            (35.0, 4.5, (0.1, -0.2, 0.3), (-33, 125, 10), 800, 600), (90.0, 1.0, (0, 0, 0), (0, 0, 0), 3840, 2160)]
    g = {"cfg": np.array([[c[0], c[1], *c[2], *c[3], c[4], c[5]] for c in cfgs], np.float64)}
    for i, (fov, dist, la, rot, w, h) in enumerate(cfgs):
        la_a, rot_a = np.array(la, np.float32), np.array(rot, np.float32)
        o, q, hw = np.zeros(3, np.float32), np.zeros(4, np.float32), np.zeros(2, np.float32)
        R.ref_glm_camera(fov, dist, p(la_a), p(rot_a), w, h, p(o), p(q), p(hw))
        ndc = np.ascontiguousarray((rng.random((512, 2)) * 2 - 1).astype(np.float32))
        ndc[:4] = [[-1, -1], [0, 0], [1, 1], [-1, 1]]
        rays = np.zeros((512, 7), np.float32)
        R.ref_glm_camera_rays(fov, dist, p(la_a), p(rot_a), w, h, p(ndc), p(rays), 512)
        g[f"origin{i}"], g[f"quat{i}"], g[f"half{i}"], g[f"ndc{i}"], g[f"rays{i}"] = o, q, hw, ndc, rays
    np.savez_compressed(os.path.join(GOLD, "camera.npz"), **g)


def gen_obj_cases():
    """The reference loader (loadMesh) on the OBJ files authored under tests/golden/obj_cases."""
    R = orc.ref()
    case_dir = os.path.join(GOLD, "obj_cases")
    for fn in sorted(os.listdir(case_dir)):
        if not fn.endswith(".obj"):
            continue
        for normalize in (0, 1):
            h = R.ref_scene_load_file(os.path.join(case_dir, fn).encode(), normalize)
            assert h, fn
            d = orc.ref_scene_to_dict(h)
            R.ref_scene_free(h)
            out = os.path.join(case_dir, f"{fn[:-4]}.norm{normalize}.npz")
            np.savez_compressed(out, **{k: d[k] for k in orc.SCENE_KEYS})
            print(f"{fn} normalize={normalize}: meshes={len(d['materials'])} verts={len(d['verts'])} tris={len(d['tris'])}")


def gen_bitmap():
    """stb's BMP writer fed as Screen::writeBitmapToFile feeds it (src/screen.cpp:93-104)."""
    import tempfile
    R = orc.ref()
    rng = np.random.default_rng(11)
    w, h = 7, 5  # odd sizes: no row padding exists in the 32-bit layout, and none must appear
    rgb = (rng.random((h, w, 3)) * 1.6 - 0.3).astype(np.float32)  # some values below 0 and above 1
    rgb[0, 0] = (0.0, 1.0, 0.5)
    rgb[0, 1] = (254.999 / 255.0, 1.0 / 255.0, 0.999 / 255.0)  # truncation, not rounding
    rgb[h - 1, w - 1] = (2.0, -1.0, 1.0)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "out.bmp")
        assert R.ref_write_bmp(path.encode(), w, h, p(np.ascontiguousarray(rgb)))
        data = np.frombuffer(open(path, "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "screen_bmp.npz"), rgb=rgb, bmp=data)
    print("bmp fixture:", len(data), "bytes")


if __name__ == "__main__":
    if not orc.ref_available():
        orc.build()
    what = sys.argv[1:] or ["scenes", "leaf", "camera", "obj", "bmp"]
    if "scenes" in what or "leaf" in what:
        scenes = gen_scenes()
        if "leaf" in what:
            gen_leaf(scenes)
    if "camera" in what:
        gen_camera()
    if "obj" in what:
        gen_obj_cases()
    if "bmp" in what:
        gen_bitmap()
    print("golden fixtures written to", GOLD)
