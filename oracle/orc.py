"""ctypes front end of the CPU oracle (oracle/liborc.so) and, where built, of the reference harness
(oracle/_ref/libref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by the smoke check and by bench.py's cpu_baseline leg.
The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

F_SHADING = 1 << 0
F_RECURSIVE = 1 << 1
F_HARD_SHADOW = 1 << 2
F_SOFT_SHADOW = 1 << 3
F_NORMAL_INTERP = 1 << 4
F_TEXTURE = 1 << 5
F_ACCEL = 1 << 6
F_MULTISAMPLE = 1 << 7
F_SAH = 1 << 8
F_TRANSPARENCY = 1 << 10
F_MOTION_BLUR = 1 << 11
F_DEPTH_OF_FIELD = 1 << 12
F_ENVIRONMENT = 1 << 13
F_MISS_CULL = 1 << 16
F_COARSE_SHADOW = 1 << 17

LIGHT_STRIDE = 22

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)


class SceneStruct(C.Structure):
    _fields_ = [
        ("n_meshes", C.c_uint32),
        ("mesh_vert_off", _u32p),
        ("mesh_tri_off", _u32p),
        ("verts", _f32p),
        ("tris", _u32p),
        ("materials", _f32p),
        ("mesh_tex", _i32p),
        ("n_textures", C.c_uint32),
        ("tex_w", _u32p),
        ("tex_h", _u32p),
        ("tex_off", _u64p),
        ("texels", _f32p),
        ("n_lights", C.c_uint32),
        ("lights", _f32p),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("origin", C.c_float * 3),
        ("quat", C.c_float * 4),
        ("half_w", C.c_float),
        ("half_h", C.c_float),
        ("width", C.c_int32),
        ("height", C.c_int32),
    ]


class Extras(C.Structure):
    _fields_ = [("samples", C.c_int32), ("time0", C.c_float), ("time1", C.c_float), ("direction", C.c_float * 3),
                ("focal_length", C.c_float), ("aperture", C.c_float), ("env_texels", _f32p), ("env_w", C.c_int32),
                ("env_h", C.c_int32)]


def make_extras(lens=None, environment=None):
    """lens: dict(samples, time0, time1, direction, focal_length, aperture); environment: float RGB (H, W, 3).
    Returns (struct, objects to keep alive)."""
    e = Extras()
    keep = []
    if lens:
        e.samples = int(lens.get("samples", 0))
        e.time0, e.time1 = float(lens.get("time0", 0.0)), float(lens.get("time1", 0.0))
        e.direction[:] = [float(x) for x in lens.get("direction", (0, 0, 0))]
        e.focal_length, e.aperture = float(lens.get("focal_length", 0)), float(lens.get("aperture", 0.0))
    if environment is not None:
        env = np.ascontiguousarray(environment, np.float32)
        e.env_texels = _ptr(env, _f32p)
        e.env_h, e.env_w = env.shape[0], env.shape[1]
        keep.append(env)
    return e, keep


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("path_rays", "shadow_rays", "node_visits", "inner_visits", "box_tests", "tri_tests", "shaded_hits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


HIT_DTYPE = np.dtype([("hit", "<i4"), ("t", "<f4"), ("mesh", "<i4"), ("tri", "<i4"), ("normal", "<f4", 3),
                      ("bary", "<f4", 3), ("uv", "<f4", 2), ("kd", "<f4", 3)])


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


_lib = None


def build():
    """Compile liborc.so (and _ref/libref.so when the reference checkout is present)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_bvh_build.restype = C.c_void_p
        L.orc_bvh_build.argtypes = [C.POINTER(SceneStruct), C.c_uint32]
        L.orc_bvh_free.argtypes = [C.c_void_p]
        for n in ("orc_bvh_num_levels", "orc_bvh_num_leaves", "orc_bvh_num_nodes"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = C.c_int32
        L.orc_bvh_dump.argtypes = [C.c_void_p, _i32p, _f32p]
        L.orc_bvh_tri_order.argtypes = [C.c_void_p, _i32p]
        L.orc_bvh_tri_order.restype = C.c_int32
        L.orc_intersect_batch.argtypes = [C.c_void_p, _f32p, C.c_int64, C.c_uint32, C.c_void_p, C.POINTER(Counters)]
        L.orc_final_color_batch.argtypes = [C.c_void_p, _f32p, C.c_int64, C.c_uint32, C.c_int32, C.c_uint32, _f32p,
                                            C.POINTER(Counters)]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_uint32, _i32p, _f32p, _i32p, _i32p, _f32p, C.c_int32, C.POINTER(Counters)]
        L.orc_render_ex.argtypes = L.orc_render.argtypes + [C.POINTER(Extras)]
        L.orc_final_color_batch_ex.argtypes = L.orc_final_color_batch.argtypes + [C.POINTER(Extras)]
        L.orc_camera_make.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, _f32p, _f32p, C.c_int32, C.c_int32]
        L.orc_camera_ray.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, _f32p]
        L.orc_ray_triangle.argtypes = [_f32p, _f32p, _f32p, _f32p]
        L.orc_ray_triangle.restype = C.c_int32
        L.orc_ray_aabb.argtypes = [_f32p, _f32p]
        L.orc_ray_aabb.restype = C.c_int32
        L.orc_boxval.argtypes = [_f32p, _f32p]
        L.orc_boxval.restype = C.c_float
        L.orc_barycentric.argtypes = [_f32p] * 5
        L.orc_compute_shading.argtypes = [_f32p] * 6 + [C.c_float, _f32p]
        L.orc_reflection_ray.argtypes = [_f32p] * 3
        L.orc_texel.argtypes = [_f32p, C.c_int32, C.c_int32, C.c_float, C.c_float, _f32p]
        L.orc_rand01.argtypes = [C.c_uint32] * 6
        L.orc_rand01.restype = C.c_float
        L.orc_bloom.argtypes = [_f32p, C.c_int32, C.c_int32, C.c_float, C.c_int32]
        _lib = L
    return _lib


# ------------------------------------------------------------------------------------------ scenes

SCENE_KEYS = ("mesh_vert_off", "mesh_tri_off", "verts", "tris", "materials", "mesh_tex", "tex_w", "tex_h", "tex_off",
              "texels_u8", "lights")


def load_scene_npz(path):
    """Scene fixture -> dict of numpy arrays (texels expanded to float RGB exactly as the reference's
    image loader does: u8 / 255.0f, framework/src/image.cpp:30-32)."""
    with np.load(path, allow_pickle=False) as z:
        d = {k: np.ascontiguousarray(z[k]) for k in SCENE_KEYS}
    d["texels"] = np.ascontiguousarray(d["texels_u8"].astype(np.float32) / np.float32(255.0))
    return d


def empty_scene():
    return {
        "mesh_vert_off": np.zeros(1, np.uint32), "mesh_tri_off": np.zeros(1, np.uint32),
        "verts": np.zeros((0, 8), np.float32), "tris": np.zeros((0, 3), np.uint32),
        "materials": np.zeros((0, 8), np.float32), "mesh_tex": np.zeros(0, np.int32),
        "tex_w": np.zeros(0, np.uint32), "tex_h": np.zeros(0, np.uint32), "tex_off": np.zeros(0, np.uint64),
        "texels": np.zeros((0, 3), np.float32), "texels_u8": np.zeros((0, 3), np.uint8),
        "lights": np.zeros((0, LIGHT_STRIDE), np.float32),
    }


def scene_from_meshes(meshes, lights=None):
    """meshes: list of (verts[n,8], tris[m,3], material8) without textures."""
    d = empty_scene()
    vo, to = [0], [0]
    for v, t, _ in meshes:
        vo.append(vo[-1] + len(v))
        to.append(to[-1] + len(t))
    d["mesh_vert_off"] = np.array(vo, np.uint32)
    d["mesh_tri_off"] = np.array(to, np.uint32)
    d["verts"] = np.ascontiguousarray(np.concatenate([np.asarray(v, np.float32).reshape(-1, 8) for v, _, _ in meshes]))
    d["tris"] = np.ascontiguousarray(np.concatenate([np.asarray(t, np.uint32).reshape(-1, 3) for _, t, _ in meshes]))
    d["materials"] = np.ascontiguousarray(np.array([m for _, _, m in meshes], np.float32).reshape(-1, 8))
    d["mesh_tex"] = np.full(len(meshes), -1, np.int32)
    if lights is not None:
        d["lights"] = np.ascontiguousarray(np.asarray(lights, np.float32).reshape(-1, LIGHT_STRIDE))
    return d


def point_light(pos, col=(1, 1, 1)):
    l = np.zeros(LIGHT_STRIDE, np.float32)
    l[0] = 0
    l[1:4] = pos
    l[4:7] = col
    return l


def segment_light(e0, e1, c0, c1):
    l = np.zeros(LIGHT_STRIDE, np.float32)
    l[0] = 1
    l[1:4], l[4:7], l[7:10], l[10:13] = e0, e1, c0, c1
    return l


def parallelogram_light(v0, e01, e02, c0, c1, c2, c3):
    l = np.zeros(LIGHT_STRIDE, np.float32)
    l[0] = 2
    l[1:4], l[4:7], l[7:10], l[10:13], l[13:16], l[16:19], l[19:22] = v0, e01, e02, c0, c1, c2, c3
    return l


class Scene:
    """Keeps the numpy arrays alive next to the C struct that points into them."""

    def __init__(self, d):
        self.d = d
        s = SceneStruct()
        s.n_meshes = len(d["materials"])
        s.mesh_vert_off = _ptr(d["mesh_vert_off"], _u32p)
        s.mesh_tri_off = _ptr(d["mesh_tri_off"], _u32p)
        s.verts = _ptr(d["verts"], _f32p)
        s.tris = _ptr(d["tris"], _u32p)
        s.materials = _ptr(d["materials"], _f32p)
        s.mesh_tex = _ptr(d["mesh_tex"], _i32p)
        s.n_textures = len(d["tex_w"])
        s.tex_w = _ptr(d["tex_w"], _u32p)
        s.tex_h = _ptr(d["tex_h"], _u32p)
        s.tex_off = _ptr(d["tex_off"], _u64p)
        s.texels = _ptr(d["texels"], _f32p)
        s.n_lights = len(d["lights"])
        s.lights = _ptr(d["lights"], _f32p)
        self.struct = s

    @property
    def n_tris(self):
        return int(self.d["mesh_tri_off"][-1])


class Bvh:
    def __init__(self, scene, features=0):
        self.scene = scene if isinstance(scene, Scene) else Scene(scene)
        self.h = lib().orc_bvh_build(C.byref(self.scene.struct), features)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:  # at interpreter shutdown the module's globals may be gone
            lib().orc_bvh_free(self.h)
            self.h = None

    @property
    def num_levels(self):
        return lib().orc_bvh_num_levels(self.h)

    @property
    def num_leaves(self):
        return lib().orc_bvh_num_leaves(self.h)

    @property
    def num_nodes(self):
        return lib().orc_bvh_num_nodes(self.h)

    def dump(self):
        n = self.num_nodes
        meta = np.zeros((n, 4), np.int32)
        box = np.zeros((n, 6), np.float32)
        lib().orc_bvh_dump(self.h, _ptr(meta, _i32p), _ptr(box, _f32p))
        return meta, box

    def tri_order(self):
        n = lib().orc_bvh_tri_order(self.h, None)
        out = np.zeros((n, 2), np.int32)
        lib().orc_bvh_tri_order(self.h, _ptr(out, _i32p))
        return out

    def intersect(self, rays7, features):
        rays = np.ascontiguousarray(rays7, np.float32).copy()
        out = np.zeros(len(rays), HIT_DTYPE)
        cnt = Counters()
        lib().orc_intersect_batch(self.h, _ptr(rays, _f32p), len(rays), features, out.ctypes.data, C.byref(cnt))
        return out, cnt.as_dict()

    def final_color(self, rays7, features, max_depth=5, seed=0, environment=None):
        rays = np.ascontiguousarray(rays7, np.float32)
        rgb = np.zeros((len(rays), 3), np.float32)
        cnt = Counters()
        ex, keep = make_extras(None, environment)
        lib().orc_final_color_batch_ex(self.h, _ptr(rays, _f32p), len(rays), features, max_depth, seed,
                                       _ptr(rgb, _f32p), C.byref(cnt), C.byref(ex))
        return rgb, cnt.as_dict()

    def render(self, cam, features, nsx=1, nsy=1, max_depth=5, seed=0, rect=None, threads=0, primary=True, lens=None,
               environment=None):
        W, H = cam.width, cam.height
        rgb = np.zeros((H, W, 3), np.float32)
        pm = np.full((H, W), -1, np.int32)
        pt = np.full((H, W), -1, np.int32)
        tt = np.zeros((H, W), np.float32)
        cnt = Counters()
        r = None if rect is None else _ptr(np.array(rect, np.int32), _i32p)
        ex, keep = make_extras(lens, environment)
        lib().orc_render_ex(self.h, C.byref(cam), features, nsx, nsy, max_depth, seed, r, _ptr(rgb, _f32p),
                            _ptr(pm, _i32p) if primary else None, _ptr(pt, _i32p) if primary else None,
                            _ptr(tt, _f32p) if primary else None, threads, C.byref(cnt), C.byref(ex))
        return {"rgb": rgb, "mesh": pm, "tri": pt, "t": tt, "counters": cnt.as_dict()}


def bloom(rgb, threshold, box_size):
    """Screen::applyBloomFilter on a frame stored top row first; returns the filtered copy."""
    out = np.ascontiguousarray(rgb, np.float32).copy()
    lib().orc_bloom(_ptr(out, _f32p), out.shape[1], out.shape[0], threshold, box_size)
    return out


def make_camera(width, height, fovy_deg=50.0, dist=3.0, look_at=(0, 0, 0), rot_deg=(20, 20, 0)):
    """Default arguments are the reference's CameraConfig defaults (src/config.h:16-21)."""
    cam = Camera()
    la = np.array(look_at, np.float32)
    rd = np.array(rot_deg, np.float32)
    lib().orc_camera_make(C.byref(cam), fovy_deg, dist, _ptr(la, _f32p), _ptr(rd, _f32p), width, height)
    return cam


def camera_rays(cam, ndc):
    ndc = np.asarray(ndc, np.float32).reshape(-1, 2)
    out = np.zeros((len(ndc), 7), np.float32)
    tmp = (C.c_float * 7)()
    for i, (x, y) in enumerate(ndc):
        lib().orc_camera_ray(C.byref(cam), float(x), float(y), tmp)
        out[i] = tmp[:]
    return out


def pixel_ndc(width, height):
    """NDC of every pixel centre-less corner as the reference computes it (src/render.cpp:190-193);
    returned in row-major (y, x) order."""
    xs = np.arange(width, dtype=np.float32) / np.float32(width) * np.float32(2) - np.float32(1)
    ys = np.arange(height, dtype=np.float32) / np.float32(height) * np.float32(2) - np.float32(1)
    g = np.zeros((height, width, 2), np.float32)
    g[..., 0] = xs[None, :]
    g[..., 1] = ys[:, None]
    return g


# ------------------------------------------------------------------------------------------ reference harness

_ref = None


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libref.so"))


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(os.path.join(HERE, "_ref", "libref.so"))
        R.ref_scene_load_prebuilt.restype = C.c_void_p
        R.ref_scene_load_prebuilt.argtypes = [C.c_int, C.c_char_p]
        R.ref_scene_load_file.restype = C.c_void_p
        R.ref_scene_load_file.argtypes = [C.c_char_p, C.c_int]
        R.ref_scene_free.argtypes = [C.c_void_p]
        for n in ("ref_scene_num_meshes", "ref_scene_num_textures", "ref_scene_num_lights"):
            getattr(R, n).argtypes = [C.c_void_p]
        R.ref_scene_mesh_sizes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        R.ref_scene_mesh_get.argtypes = [C.c_void_p, C.c_int, _f32p, _u32p, _f32p]
        R.ref_scene_texture_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        R.ref_scene_texture_get.argtypes = [C.c_void_p, C.c_int, _f32p]
        R.ref_scene_light_get.argtypes = [C.c_void_p, C.c_int, _f32p]
        R.ref_ray_triangle.argtypes = [_f32p, _f32p, _i32p, C.c_long]
        R.ref_ray_aabb.argtypes = [_f32p, _f32p, _i32p, C.c_long]
        R.ref_barycentric.argtypes = [_f32p, _f32p, _f32p, C.c_long]
        R.ref_interp_normal.argtypes = [_f32p, _f32p, _f32p, C.c_long]
        R.ref_interp_texcoord.argtypes = [_f32p, _f32p, _f32p, C.c_long]
        R.ref_acquire_texel.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, C.c_long]
        R.ref_compute_shading.argtypes = [_f32p] * 8 + [C.c_long]
        R.ref_reflection_ray.argtypes = [_f32p, _f32p, _f32p, C.c_long]
        R.ref_glm_camera.argtypes = [C.c_float, C.c_float, _f32p, _f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p]
        R.ref_glm_camera_rays.argtypes = [C.c_float, C.c_float, _f32p, _f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_long]
        R.ref_write_bmp.argtypes = [C.c_char_p, C.c_int, C.c_int, _f32p]
        R.ref_image_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _f32p]
        _ref = R
    return _ref


def ref_scene_to_dict(h):
    """Copy a scene loaded by the reference's own loader into the fixture layout."""
    R = ref()
    nm = R.ref_scene_num_meshes(h)
    vo, to, verts, tris, mats, mtex = [0], [0], [], [], [], []
    for m in range(nm):
        nv, nt = C.c_int(), C.c_int()
        R.ref_scene_mesh_sizes(h, m, C.byref(nv), C.byref(nt))
        v = np.zeros((nv.value, 8), np.float32)
        t = np.zeros((nt.value, 3), np.uint32)
        k = np.zeros(8, np.float32)
        tex = R.ref_scene_mesh_get(h, m, _ptr(v, _f32p), _ptr(t, _u32p), _ptr(k, _f32p))
        vo.append(vo[-1] + nv.value)
        to.append(to[-1] + nt.value)
        verts.append(v)
        tris.append(t)
        mats.append(k)
        mtex.append(tex)
    d = empty_scene()
    d["mesh_vert_off"] = np.array(vo, np.uint32)
    d["mesh_tri_off"] = np.array(to, np.uint32)
    if nm:
        d["verts"] = np.ascontiguousarray(np.concatenate(verts))
        d["tris"] = np.ascontiguousarray(np.concatenate(tris))
        d["materials"] = np.ascontiguousarray(np.stack(mats))
    d["mesh_tex"] = np.array(mtex, np.int32)
    ntex = R.ref_scene_num_textures(h)
    tw, th, toff, tex = [], [], [], []
    off = 0
    for t in range(ntex):
        w, hh = C.c_int(), C.c_int()
        R.ref_scene_texture_size(h, t, C.byref(w), C.byref(hh))
        px = np.zeros((w.value * hh.value, 3), np.float32)
        R.ref_scene_texture_get(h, t, _ptr(px, _f32p))
        tw.append(w.value)
        th.append(hh.value)
        toff.append(off)
        off += w.value * hh.value
        tex.append(px)
    d["tex_w"] = np.array(tw, np.uint32)
    d["tex_h"] = np.array(th, np.uint32)
    d["tex_off"] = np.array(toff, np.uint64)
    if ntex:
        d["texels"] = np.ascontiguousarray(np.concatenate(tex))
        u8 = np.rint(d["texels"] * 255.0).astype(np.uint8)
        assert np.array_equal(u8.astype(np.float32) / np.float32(255.0), d["texels"]), "texture is not u8/255"
        d["texels_u8"] = u8
    nl = R.ref_scene_num_lights(h)
    L = np.zeros((nl, LIGHT_STRIDE), np.float32)
    for i in range(nl):
        R.ref_scene_light_get(h, i, _ptr(L[i], _f32p))
    d["lights"] = L
    return d


def ref_image(path):
    """The reference's Image(path): float RGB (H, W, 3), or None where the reference throws."""
    R = ref()
    w, h = C.c_int(), C.c_int()
    if R.ref_image_load(os.fsencode(path), C.byref(w), C.byref(h), None) != 0:
        return None
    rgb = np.zeros((h.value, w.value, 3), np.float32)
    R.ref_image_load(os.fsencode(path), C.byref(w), C.byref(h), _ptr(rgb, _f32p))
    return rgb
