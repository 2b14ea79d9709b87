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
  tests/golden/image_cases/        small JPEG / interlaced PNG files written here with Pillow, and expected.npz: what the
                                   reference's Image loader (stb_image) decodes each of them to; cube_map.npz: SHA-256 and
                                   a 1/16 sub-grid of the reference's decode of its own data/cube_map.jpg

usage: make_golden.py [scenes|leaf|camera|obj|bmp|images ...]   (default: all)
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


def gen_images():
    """Test images made here (a photo-like synthetic picture saved in the formats the loader has to read) and the
    reference loader's decode of each."""
    import hashlib
    from PIL import Image as PI
    out_dir = os.path.join(GOLD, "image_cases")
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(11)

    def picture(w, h):
        y, x = np.mgrid[0:h, 0:w].astype(np.float64)
        r = 127 + 120 * np.sin(x / 5.0) * np.cos(y / 7.0)
        g = 255 * (x + y) / max(1, w + h - 2)
        b = 255 * (((x // 6 + y // 4) % 2) * 0.7 + 0.15) + rng.normal(0, 6, (h, w))
        return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)

    cases = []
    def jpeg(name, w, h, mode="RGB", **kw):
        img = PI.fromarray(picture(w, h)).convert(mode)
        img.save(os.path.join(out_dir, name + ".jpg"), "JPEG", **kw)
        cases.append(name + ".jpg")

    def png(name, w, h, mode="RGB", bits=None, **kw):
        img = PI.fromarray(picture(w, h)).convert(mode)
        if mode == "P" and bits:
            img = PI.fromarray(picture(w, h)).quantize(1 << bits)
        # Pillow cannot write interlaced files: interlace them here (Adam7), from Pillow's non-interlaced output
        path = os.path.join(out_dir, name + ".png")
        img.save(path, "PNG", bits=bits, **kw) if bits and mode == "P" else img.save(path, "PNG", **kw)
        _interlace_png(path)
        cases.append(name + ".png")

    jpeg("base444", 40, 24, quality=90, subsampling=0)
    jpeg("base420_odd", 37, 29, quality=85, subsampling=2)
    jpeg("base422", 35, 18, quality=80, subsampling=1)
    jpeg("base_gray", 33, 17, mode="L", quality=88)
    jpeg("base420_restart", 50, 41, quality=75, subsampling=2, restart_marker_blocks=3)
    jpeg("base_optimized", 48, 32, quality=60, subsampling=2, optimize=True)
    jpeg("prog444", 40, 27, quality=90, subsampling=0, progressive=True)
    jpeg("prog420_odd", 45, 35, quality=80, subsampling=2, progressive=True)
    jpeg("prog_gray", 31, 22, mode="L", quality=85, progressive=True)
    jpeg("prog422_restart", 52, 30, quality=70, subsampling=1, progressive=True, restart_marker_rows=1)
    jpeg("tiny_1x1", 1, 1, quality=90)
    jpeg("low_quality", 64, 48, quality=8, subsampling=2)
    png("ilace_rgb", 13, 11)
    png("ilace_rgba", 9, 10, mode="RGBA")
    png("ilace_gray", 17, 5, mode="L")
    png("ilace_pal4", 21, 19, mode="P", bits=4)
    png("ilace_1x1", 1, 1)
    png("ilace_2x3", 2, 3)
    expected = {}
    for f in cases:
        rgb = orc.ref_image(os.path.join(out_dir, f))
        assert rgb is not None, f
        u8 = np.rint(rgb * 255.0).astype(np.uint8)
        assert np.array_equal(u8.astype(np.float32) / np.float32(255.0), rgb)
        expected[f] = u8
        print(f"{f}: {u8.shape[1]}x{u8.shape[0]}")
    np.savez_compressed(os.path.join(out_dir, "expected.npz"), **expected)
    big = orc.ref_image(os.path.join(DATA, "cube_map.jpg"))
    u8 = np.rint(big * 255.0).astype(np.uint8)
    assert np.array_equal(u8.astype(np.float32) / np.float32(255.0), big)
    np.savez_compressed(os.path.join(out_dir, "cube_map.npz"), sha256=np.frombuffer(hashlib.sha256(u8.tobytes()).digest(), np.uint8),
                        shape=np.array(u8.shape, np.int64), grid16=u8[::16, ::16].copy())
    print("cube_map.jpg:", u8.shape, hashlib.sha256(u8.tobytes()).hexdigest())


def _interlace_png(path):
    """Rewrite a non-interlaced 8/16-bit or palette PNG as an Adam7-interlaced one (filter type 0 on every line)."""
    import struct
    import zlib
    data = open(path, "rb").read()
    pos, chunks = 8, []
    while pos < len(data):
        n = struct.unpack(">I", data[pos:pos + 4])[0]
        chunks.append((data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]))
        pos += 12 + n
    ihdr = dict(chunks)[b"IHDR"]
    w, h, depth, ctype = struct.unpack(">IIBB", ihdr[:10])
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bits = ch * depth
    stride = (w * bits + 7) // 8
    raw = zlib.decompress(b"".join(d for t, d in chunks if t == b"IDAT"))
    bpp = max(1, bits // 8)
    rows, prev = [], bytearray(stride)
    for y in range(h):
        f = raw[y * (stride + 1)]
        line = bytearray(raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)])
        for i in range(stride):
            a = line[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 1:
                line[i] = (line[i] + a) & 255
            elif f == 2:
                line[i] = (line[i] + b) & 255
            elif f == 3:
                line[i] = (line[i] + ((a + b) >> 1)) & 255
            elif f == 4:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                line[i] = (line[i] + (a if pa <= pb and pa <= pc else (b if pb <= pc else c))) & 255
        rows.append(line)
        prev = line

    def pixel_bits(y, x):
        v = int.from_bytes(rows[y], "big")
        return (v >> (stride * 8 - (x + 1) * bits)) & ((1 << bits) - 1)

    out = bytearray()
    for x0, y0, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
        pw, ph = (w - x0 + dx - 1) // dx if w > x0 else 0, (h - y0 + dy - 1) // dy if h > y0 else 0
        if pw <= 0 or ph <= 0:
            continue
        ps = (pw * bits + 7) // 8
        for j in range(ph):
            acc = 0
            for i in range(pw):
                acc = (acc << bits) | pixel_bits(y0 + j * dy, x0 + i * dx)
            acc <<= ps * 8 - pw * bits
            out += b"\0" + acc.to_bytes(ps, "big")
    new = bytearray(data[:8])
    done = False
    for t, d in chunks:
        if t == b"IHDR":
            d = d[:12] + b"\1"
        if t == b"IDAT":
            if done:
                continue
            d, done = zlib.compress(bytes(out)), True
        new += struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    open(path, "wb").write(bytes(new))


if __name__ == "__main__":
    if not orc.ref_available():
        orc.build()
    what = sys.argv[1:] or ["scenes", "leaf", "camera", "obj", "bmp", "images"]
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
    if "images" in what:
        gen_images()
    print("golden fixtures written to", GOLD)
