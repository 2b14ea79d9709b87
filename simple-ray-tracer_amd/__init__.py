"""simple-ray-tracer_amd: MI355X-native back end for the Simple-Ray-Tracer BVH / intersection / shading path.

This module is a thin ctypes binding of host/libsrt_host.so (the C++ mirror of the reference's
Scene / BvhInterface / Screen / renderRayTracing interface), which in turn drives csrc/libsrt_hip.so (the
hand-written gfx950 kernels behind the C ABI of include/srt_hip.h).  There is no CPU rendering path: if the
libraries are missing, or no GPU is visible, calls raise.

Import with importlib (the directory name carries a hyphen):
    srt = importlib.import_module("simple-ray-tracer_amd")
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# SRT_LIB_ROOT points at another build of the same two libraries (kernel A/B experiments); default: in-tree
LIB_ROOT = os.environ.get("SRT_LIB_ROOT", HERE)
HIP_LIB = os.path.join(LIB_ROOT, "csrc", "libsrt_hip.so")
HOST_LIB = os.path.join(LIB_ROOT, "host", "libsrt_host.so")

# feature bits (include/srt_hip.h)
F_SHADING = 1 << 0
F_RECURSIVE = 1 << 1
F_HARD_SHADOW = 1 << 2
F_SOFT_SHADOW = 1 << 3
F_NORMAL_INTERP = 1 << 4
F_TEXTURE = 1 << 5
F_ACCEL = 1 << 6
F_MULTISAMPLE = 1 << 7
F_SAH = 1 << 8           # host-side build option (extra.enableBvhSahBinning)
F_TRANSPARENCY = 1 << 10  # extra.enableTransparency
F_MOTION_BLUR = 1 << 11   # extra.enableMotionBlur (Scene.set_lens)
F_DEPTH_OF_FIELD = 1 << 12  # extra.enableDepthOfField
F_ENVIRONMENT = 1 << 13   # extra.enableEnvironmentMapping (Scene.set_environment)
F_BLOOM = 1 << 9         # host-side pass of the C++ mirror (extra.enableBloomEffect); not a kernel flag
F_MISS_CULL = 1 << 16    # traversal policy; clear it for the reference's exact visiting rule
F_COARSE_SHADOW = 1 << 17  # with F_MISS_CULL: shadow queries on four-wide nodes of outward-rounded 16-bit boxes

RENDER_STATS = 1 << 0
RENDER_TIMING = 1 << 1

LIGHT_STRIDE = 22

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)


class SrtError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("quat", C.c_float * 4), ("half_w", C.c_float), ("half_h", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


class Lens(C.Structure):
    _fields_ = [("samples", C.c_int32), ("time0", C.c_float), ("time1", C.c_float), ("direction", C.c_float * 3),
                ("focal_length", C.c_float), ("aperture", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("camera", Camera), ("features", C.c_uint32), ("samples_x", C.c_int32), ("samples_y", C.c_int32),
                ("max_depth", C.c_int32), ("seed", C.c_uint32), ("rect", C.c_int32 * 4), ("flags", C.c_uint32),
                ("lens", Lens)]


class Output(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("prim_mesh", C.c_void_p), ("prim_tri", C.c_void_p), ("prim_t", C.c_void_p),
                ("out_width", C.c_int32), ("out_height", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("on_device", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("path_rays", "shadow_rays", "shaded_hits", "inner_visits", "tri_tests", "box_tests", "wave_inner_steps",
                                           "wave_leaf_steps")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


HIT_DTYPE = np.dtype([("hit", "<i4"), ("t", "<f4"), ("mesh", "<i4"), ("tri", "<i4"), ("normal", "<f4", 3),
                      ("bary", "<f4", 3), ("uv", "<f4", 2), ("kd", "<f4", 3)])
NODE_DTYPE = np.dtype([("lbox", "<f4", 6), ("rbox", "<f4", 6), ("left", "<u4"), ("right", "<u4"), ("left_level", "<u4"),
                       ("pad", "<u4")])
TRI_DTYPE = np.dtype([("n", "<f4", 3), ("D", "<f4"), ("v0", "<f4", 3), ("mesh", "<u4"), ("v1", "<f4", 3), ("tri", "<u4"),
                      ("v2", "<f4", 3), ("c0", "<f4", 3), ("c1", "<f4", 3), ("c2", "<f4", 3)])
ATTR_DTYPE = np.dtype([("n0", "<f4", 3), ("n1", "<f4", 3), ("n2", "<f4", 3), ("uv0", "<f4", 2), ("uv1", "<f4", 2),
                       ("uv2", "<f4", 2), ("mesh", "<u4")])
assert NODE_DTYPE.itemsize == 64 and TRI_DTYPE.itemsize == 96 and ATTR_DTYPE.itemsize == 64 and HIT_DTYPE.itemsize == 60

# every symbol include/srt_hip.h declares
HIP_ABI_SYMBOLS = ("srt_last_error", "srt_device_count", "srt_device_create", "srt_device_destroy",
                   "srt_device_synchronize", "srt_device_read_counters", "srt_device_read_kernel_counters",
                   "srt_device_kernel_times",
                   "srt_scene_upload", "srt_scene_destroy", "srt_scene_build", "srt_scene_info", "srt_scene_download", "srt_render", "srt_render_noalloc", "srt_intersect_batch",
                   "srt_final_color_batch", "srt_generate_rays", "srt_selftest_division", "srt_probe_gather")


def build():
    """Compile both libraries in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


_host = None


def host():
    """libsrt_host.so; raises when the product libraries have not been built."""
    global _host
    if _host is not None:
        return _host
    for p in (HIP_LIB, HOST_LIB):
        if not os.path.exists(p):
            raise SrtError(f"{p} is missing: run `make -C {HERE}` (there is no CPU fallback)")
    # PyTorch's ROCm wheels carry their own HIP runtime.  Loaded first, it is the one libsrt_hip.so binds to (same
    # soname) and the process has a single runtime; loaded after the system runtime that libsrt_hip.so would otherwise
    # pull in, it would be a second one whose device initialisation fails.  So: PyTorch first, wherever it is installed
    # (bench.py and distributed.py pass torch buffers and streams to this library anyway).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(HOST_LIB)
    L.srth_last_error.restype = C.c_char_p
    L.srth_scene_create.restype = C.c_void_p
    L.srth_scene_destroy.argtypes = [C.c_void_p]
    L.srth_scene_add_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, _f32p]
    L.srth_scene_add_mesh.argtypes = [C.c_void_p, _f32p, C.c_int, _u32p, C.c_int, _f32p, C.c_int]
    L.srth_scene_add_light.argtypes = [C.c_void_p, _f32p]
    L.srth_scene_set_environment.argtypes = [C.c_void_p, C.c_int, C.c_int, _f32p]
    L.srth_scene_load_environment.argtypes = [C.c_void_p, C.c_char_p]
    L.srth_scene_set_lens.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, _f32p, C.c_int, C.c_float, C.c_int]
    L.srth_image_load.argtypes = [C.c_char_p, _i32p, _i32p, _f32p]
    L.srth_scene_load_obj.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.srth_scene_load_prebuilt.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    L.srth_scene_num_meshes.argtypes = [C.c_void_p]
    L.srth_scene_num_lights.argtypes = [C.c_void_p]
    L.srth_scene_mesh_sizes.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p]
    L.srth_scene_mesh_get.argtypes = [C.c_void_p, C.c_int, _f32p, _u32p, _f32p, _i32p]
    L.srth_scene_mesh_texture.argtypes = [C.c_void_p, C.c_int, _f32p]
    L.srth_scene_light_get.argtypes = [C.c_void_p, C.c_int, _f32p]
    L.srth_hostbvh_create.restype = C.c_void_p
    L.srth_hostbvh_create.argtypes = [C.c_void_p, C.c_uint32]
    L.srth_hostbvh_destroy.argtypes = [C.c_void_p]
    for n in ("srth_hostbvh_num_levels", "srth_hostbvh_num_leaves", "srth_hostbvh_num_nodes"):
        getattr(L, n).argtypes = [C.c_void_p]
    L.srth_hostbvh_dump.argtypes = [C.c_void_p, _i32p, _f32p]
    L.srth_hostbvh_tri_order.argtypes = [C.c_void_p, _i32p]
    L.srth_hostbvh_flatten.argtypes = [C.c_void_p, C.c_int, _u32p, C.c_void_p, C.c_void_p, C.c_void_p, _u32p, _f32p, _u32p]
    L.srth_bvh_create.restype = C.c_void_p
    L.srth_bvh_create.argtypes = [C.c_void_p, C.c_uint32]
    L.srth_bvh_destroy.argtypes = [C.c_void_p]
    L.srth_bvh_num_levels.argtypes = [C.c_void_p]
    L.srth_bvh_num_leaves.argtypes = [C.c_void_p]
    L.srth_bvh_intersect.argtypes = [C.c_void_p, _f32p, C.c_int64, C.c_uint32, C.c_void_p, C.POINTER(Counters)]
    L.srth_final_color.argtypes = [C.c_void_p, _f32p, C.c_int64, C.c_uint32, C.c_int, C.c_uint32, _f32p]
    L.srth_camera_make.argtypes = [C.c_float, C.c_float, _f32p, _f32p, C.c_int, C.c_int, C.POINTER(Camera)]
    L.srth_generate_rays.argtypes = [C.POINTER(Camera), _f32p, C.c_int64, _f32p]
    L.srth_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.POINTER(Output), C.c_void_p]
    L.srth_render_noalloc.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.POINTER(Output), C.c_void_p]
    L.srth_read_counters.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Counters), C.c_int]
    L.srth_bvh_download.argtypes = [C.c_void_p, _u32p, C.c_void_p, C.c_void_p, C.c_void_p, _u32p, _f32p, _u32p]
    L.srth_probe_gather.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.srth_read_kernel_counters.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Counters), C.c_int]
    L.srth_kernel_times.argtypes = [C.c_void_p, _f32p, _i32p, C.c_int]
    L.srth_synchronize.argtypes = [C.c_void_p, C.c_void_p]
    L.srth_write_bmp.argtypes = [C.c_char_p, _f32p, C.c_int, C.c_int]
    L.srth_set_device.argtypes = [C.c_int]
    L.srth_set_device_context.argtypes = [C.c_int, C.c_int]
    L.srth_selftest_division.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
    L.srth_mirror_render.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_float, C.c_float, _f32p, _f32p, C.c_int, _f32p]
    L.srth_bvh_add_replica.argtypes = [C.c_void_p, C.c_int]
    L.srth_bvh_num_render_devices.argtypes = [C.c_void_p]
    L.srth_mirror_render_frames.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, _f32p]
    L.srth_tiles_of.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _i32p, C.c_int]
    L.srth_run_workers_selftest.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p]
    L.srth_mirror_intersect_one.argtypes = [C.c_void_p, _f32p, C.c_uint32, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _i32p]
    L.srth_mirror_final_color.argtypes = [C.c_void_p, _f32p, C.c_uint32, _f32p]
    L.srth_mirror_generate_ray.argtypes = [C.c_float, C.c_float, C.c_float, _f32p, _f32p, _f32p, _f32p]
    L.srth_bloom.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, C.c_int]
    L.srth_config_describe.restype = C.c_long
    L.srth_config_describe.argtypes = [C.c_char_p, C.c_char_p, C.c_long]
    L.srth_tilecounter_open.restype = C.c_void_p
    L.srth_tilecounter_open.argtypes = [C.c_char_p, C.c_int]
    L.srth_tilecounter_next.restype = C.c_int64
    L.srth_tilecounter_next.argtypes = [C.c_void_p, C.c_int]
    L.srth_tilecounter_reset.argtypes = [C.c_void_p, C.c_int]
    L.srth_tilecounter_close.argtypes = [C.c_void_p]
    L.srth_frameboard_open.restype = C.c_void_p
    L.srth_frameboard_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.srth_frameboard_post.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, _i32p, _i32p]
    L.srth_frameboard_wait.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    L.srth_frameboard_read.argtypes = [C.c_void_p, C.c_int64, C.c_int, _i32p, _i32p]
    L.srth_frameboard_close.argtypes = [C.c_void_p]
    _host = L
    return L


def _check(rc):
    if rc != 0:
        raise SrtError(host().srth_last_error().decode())


def _p(a, t=_f32p):
    return a.ctypes.data_as(t)


def set_device(gpu, context=0):
    """Select (and initialise) the GPU used by objects created afterwards on this thread.  Objects of different
    contexts have independent back-end handles on the GPU, so their renders may overlap on different streams."""
    _check(host().srth_set_device_context(int(gpu), int(context)))


class Scene:
    """Mirror of the reference's Scene: a list of meshes (each with one material) and lights."""

    def __init__(self):
        self.h = host().srth_scene_create()
        self._keep = []

    def __del__(self):
        if getattr(self, "h", None) and host is not None:  # module globals are gone at interpreter exit
            host().srth_scene_destroy(self.h)
            self.h = None

    @classmethod
    def from_arrays(cls, d):
        """d: the fixture layout (mesh_vert_off, mesh_tri_off, verts, tris, materials, mesh_tex, tex_w, tex_h,
        tex_off, texels, lights)."""
        s = cls()
        L = host()
        tex_ids = []
        for t in range(len(d["tex_w"])):
            w, h, off = int(d["tex_w"][t]), int(d["tex_h"][t]), int(d["tex_off"][t])
            px = np.ascontiguousarray(d["texels"][off:off + w * h], np.float32)
            tex_ids.append(L.srth_scene_add_texture(s.h, w, h, _p(px)))
        for m in range(len(d["materials"])):
            v = np.ascontiguousarray(d["verts"][d["mesh_vert_off"][m]:d["mesh_vert_off"][m + 1]], np.float32)
            t = np.ascontiguousarray(d["tris"][d["mesh_tri_off"][m]:d["mesh_tri_off"][m + 1]], np.uint32)
            k = np.ascontiguousarray(d["materials"][m], np.float32)
            tex = int(d["mesh_tex"][m])
            L.srth_scene_add_mesh(s.h, _p(v), len(v), _p(t, _u32p), len(t), _p(k), tex_ids[tex] if tex >= 0 else -1)
        for l in np.asarray(d["lights"], np.float32).reshape(-1, LIGHT_STRIDE):
            _check(L.srth_scene_add_light(s.h, _p(np.ascontiguousarray(l))))
        return s

    @classmethod
    def from_obj(cls, path, normalize=False, lights=()):
        """loadSceneFromFile: meshes from an OBJ file plus caller-supplied lights."""
        s = cls()
        _check(host().srth_scene_load_obj(s.h, os.fsencode(path), int(normalize)))
        for l in lights:
            _check(host().srth_scene_add_light(s.h, _p(np.ascontiguousarray(l, np.float32))))
        return s

    @classmethod
    def prebuilt(cls, scene_type, data_dir):
        """loadScenePrebuilt(type, dataDir)."""
        s = cls()
        _check(host().srth_scene_load_prebuilt(s.h, int(scene_type), os.fsencode(data_dir)))
        return s

    def set_environment(self, rgb):
        """Scene::environment: float RGB image (H, W, 3), top row first; set it before building the Bvh."""
        rgb = np.ascontiguousarray(rgb, np.float32)
        _check(host().srth_scene_set_environment(self.h, rgb.shape[1], rgb.shape[0], _p(rgb)))

    def load_environment(self, data_dir):
        """Scene::environment = <data_dir>/cube_map.jpg, the file the reference's environment map is read from."""
        _check(host().srth_scene_load_environment(self.h, os.fsencode(data_dir)))

    def set_lens(self, mb_samples=0, time0=0.0, time1=0.0, direction=(0, 0, 0), focal_length=0, aperture=0.0, dof_samples=0):
        """The Scene's motion-blur and depth-of-field variables (used by mirror_render)."""
        d = np.array(direction, np.float32)
        _check(host().srth_scene_set_lens(self.h, int(mb_samples), float(time0), float(time1), _p(d), int(focal_length), float(aperture), int(dof_samples)))

    def add_light(self, light22):
        _check(host().srth_scene_add_light(self.h, _p(np.ascontiguousarray(light22, np.float32))))

    def to_arrays(self):
        """Back to the fixture layout (used to compare the OBJ loader with the reference loader's output)."""
        L = host()
        nm = L.srth_scene_num_meshes(self.h)
        vo, to, verts, tris, mats, mtex = [0], [0], [], [], [], []
        tw, th, toff, tex = [], [], [], []
        off = 0
        for m in range(nm):
            nv, nt = C.c_int32(), C.c_int32()
            L.srth_scene_mesh_sizes(self.h, m, C.byref(nv), C.byref(nt))
            v = np.zeros((nv.value, 8), np.float32)
            t = np.zeros((nt.value, 3), np.uint32)
            k = np.zeros(8, np.float32)
            wh = np.zeros(2, np.int32)
            has = L.srth_scene_mesh_get(self.h, m, _p(v), _p(t, _u32p), _p(k), _p(wh, _i32p))
            vo.append(vo[-1] + nv.value)
            to.append(to[-1] + nt.value)
            verts.append(v), tris.append(t), mats.append(k)
            if has:
                px = np.zeros((int(wh[0]) * int(wh[1]), 3), np.float32)
                L.srth_scene_mesh_texture(self.h, m, _p(px))
                mtex.append(len(tw))
                tw.append(int(wh[0])), th.append(int(wh[1])), toff.append(off), tex.append(px)
                off += len(px)
            else:
                mtex.append(-1)
        nl = L.srth_scene_num_lights(self.h)
        lights = np.zeros((nl, LIGHT_STRIDE), np.float32)
        for i in range(nl):
            L.srth_scene_light_get(self.h, i, _p(lights[i]))
        return {
            "mesh_vert_off": np.array(vo, np.uint32), "mesh_tri_off": np.array(to, np.uint32),
            "verts": np.concatenate(verts) if verts else np.zeros((0, 8), np.float32),
            "tris": np.concatenate(tris) if tris else np.zeros((0, 3), np.uint32),
            "materials": np.stack(mats) if mats else np.zeros((0, 8), np.float32),
            "mesh_tex": np.array(mtex, np.int32), "tex_w": np.array(tw, np.uint32), "tex_h": np.array(th, np.uint32),
            "tex_off": np.array(toff, np.uint64),
            "texels": np.concatenate(tex) if tex else np.zeros((0, 3), np.float32), "lights": lights,
        }


class HostBvh:
    """BoundingVolumeHierarchy built and flattened on the host only (no GPU needed)."""

    def __init__(self, scene, features=F_ACCEL):
        self.scene = scene
        self.h = host().srth_hostbvh_create(scene.h, features)
        if not self.h:
            raise SrtError(host().srth_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None) and host is not None:  # module globals are gone at interpreter exit
            host().srth_hostbvh_destroy(self.h)
            self.h = None

    num_levels = property(lambda self: host().srth_hostbvh_num_levels(self.h))
    num_leaves = property(lambda self: host().srth_hostbvh_num_leaves(self.h))
    num_nodes = property(lambda self: host().srth_hostbvh_num_nodes(self.h))

    def dump(self):
        n = self.num_nodes
        meta, box = np.zeros((n, 4), np.int32), np.zeros((n, 6), np.float32)
        host().srth_hostbvh_dump(self.h, _p(meta, _i32p), _p(box))
        return meta, box

    def tri_order(self):
        n = host().srth_hostbvh_tri_order(self.h, None)
        out = np.zeros((n, 2), np.int32)
        host().srth_hostbvh_tri_order(self.h, _p(out, _i32p))
        return out

    def flatten(self, bfs_levels=0):
        sizes = np.zeros(2, np.uint32)
        _check(host().srth_hostbvh_flatten(self.h, bfs_levels, _p(sizes, _u32p), None, None, None, None, None, None))
        nodes = np.zeros(int(sizes[0]), NODE_DTYPE)
        tris = np.zeros(int(sizes[1]), TRI_DTYPE)
        attrs = np.zeros(int(sizes[1]), ATTR_DTYPE)
        order = np.zeros(int(sizes[1]), np.uint32)
        root_box = np.zeros(6, np.float32)
        root_ref = np.zeros(1, np.uint32)
        _check(host().srth_hostbvh_flatten(self.h, bfs_levels, _p(sizes, _u32p), nodes.ctypes.data, tris.ctypes.data,
                                           attrs.ctypes.data, _p(order, _u32p), _p(root_box), _p(root_ref, _u32p)))
        return {"nodes": nodes, "tris": tris, "attrs": attrs, "scene_order": order, "root_box": root_box,
                "root_ref": int(root_ref[0])}


class Bvh:
    """BvhInterface: builds the BVH on the host, uploads the flattened scene to the current GPU."""

    def __init__(self, scene, features=F_ACCEL):
        self.scene = scene
        self.h = host().srth_bvh_create(scene.h, features)
        if not self.h:
            raise SrtError(host().srth_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None) and host is not None:  # module globals are gone at interpreter exit
            host().srth_bvh_destroy(self.h)
            self.h = None

    num_levels = property(lambda self: host().srth_bvh_num_levels(self.h))
    num_leaves = property(lambda self: host().srth_bvh_num_leaves(self.h))
    num_render_devices = property(lambda self: host().srth_bvh_num_render_devices(self.h))

    def add_replica(self, gpu):
        """BvhInterface::addReplica: one more copy of the scene on GPU `gpu` (the same GPU may be named again); frames
        rendered through mirror_render / mirror_render_frames are then shared tile by tile between the copies."""
        _check(host().srth_bvh_add_replica(self.h, int(gpu)))

    def download(self):
        """The scene's arrays as they sit on the device, in the layout of HostBvh.flatten()."""
        sizes = np.zeros(4, np.uint32)
        root_box, root_ref = np.zeros(6, np.float32), np.zeros(1, np.uint32)
        _check(host().srth_bvh_download(self.h, _p(sizes, _u32p), None, None, None, None, _p(root_box), _p(root_ref, _u32p)))
        nodes, tris = np.zeros(int(sizes[0]), NODE_DTYPE), np.zeros(int(sizes[1]), TRI_DTYPE)
        attrs, order = np.zeros(int(sizes[1]), ATTR_DTYPE), np.zeros(int(sizes[1]), np.uint32)
        _check(host().srth_bvh_download(self.h, _p(sizes, _u32p), nodes.ctypes.data, tris.ctypes.data, attrs.ctypes.data, _p(order, _u32p),
                                        _p(root_box), _p(root_ref, _u32p)))
        return {"nodes": nodes, "tris": tris, "attrs": attrs, "scene_order": order, "root_box": root_box, "root_ref": int(root_ref[0]),
                "num_levels": int(sizes[2]), "num_leaves": int(sizes[3])}

    def intersect(self, rays7, features):
        """BvhInterface::intersect for many rays; returns (hits, updated rays, counters)."""
        rays = np.ascontiguousarray(rays7, np.float32).copy()
        hits = np.zeros(len(rays), HIT_DTYPE)
        cnt = Counters()
        _check(host().srth_bvh_intersect(self.h, _p(rays), len(rays), features, hits.ctypes.data, C.byref(cnt)))
        return hits, rays, cnt.as_dict()

    def final_color(self, rays7, features, max_depth=5, seed=0):
        rays = np.ascontiguousarray(rays7, np.float32)
        rgb = np.zeros((len(rays), 3), np.float32)
        _check(host().srth_final_color(self.h, _p(rays), len(rays), features, max_depth, seed, _p(rgb)))
        return rgb

    def params(self, cam, features, nsx=1, nsy=1, max_depth=5, seed=0, rect=None, flags=0, lens=None):
        """lens: dict(samples, time0, time1, direction, focal_length, aperture) for F_MOTION_BLUR / F_DEPTH_OF_FIELD."""
        p = RenderParams()
        if lens:
            p.lens.samples = int(lens.get("samples", 0))
            p.lens.time0, p.lens.time1 = float(lens.get("time0", 0.0)), float(lens.get("time1", 0.0))
            p.lens.direction[:] = [float(x) for x in lens.get("direction", (0, 0, 0))]
            p.lens.focal_length, p.lens.aperture = float(lens.get("focal_length", 0)), float(lens.get("aperture", 0.0))
        p.camera = cam
        p.features = features
        p.samples_x, p.samples_y = nsx, nsy
        p.max_depth = max_depth
        p.seed = seed
        r = rect if rect is not None else (0, 0, cam.width, cam.height)
        p.rect[:] = list(r)
        p.flags = flags
        return p

    def render(self, cam, features, nsx=1, nsy=1, max_depth=5, seed=0, rect=None, flags=0, primary=True, lens=None):
        """renderRayTracing into host arrays in Screen layout; returns rgb (+ primary hit of sample 0)."""
        W, H = cam.width, cam.height
        rgb = np.zeros((H, W, 3), np.float32)
        pm = np.full((H, W), -1, np.int32)
        pt = np.full((H, W), -1, np.int32)
        tt = np.zeros((H, W), np.float32)
        p = self.params(cam, features, nsx, nsy, max_depth, seed, rect, flags, lens)
        o = Output()
        o.rgb = rgb.ctypes.data
        if primary:
            o.prim_mesh, o.prim_tri, o.prim_t = pm.ctypes.data, pt.ctypes.data, tt.ctypes.data
        o.out_width, o.out_height, o.x0, o.y0, o.on_device = W, H, 0, 0, 0
        self.read_counters(reset=True)
        _check(host().srth_render(self.h, C.byref(p), C.byref(o), None))
        return {"rgb": rgb, "mesh": pm, "tri": pt, "t": tt, "counters": self.read_counters(reset=True)}

    def render_device(self, params, rgb_ptr, out_w, out_h, x0, y0, stream=None):
        """Enqueue one rectangle into a device buffer (tile-local or whole frame); does not synchronize."""
        o = Output()
        o.rgb = rgb_ptr
        o.out_width, o.out_height, o.x0, o.y0, o.on_device = out_w, out_h, x0, y0, 1
        _check(host().srth_render(self.h, C.byref(params), C.byref(o), stream))

    def render_device_noalloc(self, params, rgb_ptr, out_w, out_h, x0, y0, stream=None):
        """render_device through the entry point that allocates nothing (fixed scratch area, bounded rectangle)."""
        o = Output()
        o.rgb = rgb_ptr
        o.out_width, o.out_height, o.x0, o.y0, o.on_device = out_w, out_h, x0, y0, 1
        _check(host().srth_render_noalloc(self.h, C.byref(params), C.byref(o), stream))

    def read_counters(self, stream=None, reset=False):
        c = Counters()
        _check(host().srth_read_counters(self.h, stream, C.byref(c), int(reset)))
        return c.as_dict()

    FAMILIES = ("trace_primary", "trace_secondary", "shade", "resolve", "trace_shadow")

    def read_kernel_counters(self, stream=None, reset=False):
        """Counters split by kernel family."""
        c = (Counters * 5)()
        _check(host().srth_read_kernel_counters(self.h, stream, c, int(reset)))
        return {name: c[i].as_dict() for i, name in enumerate(self.FAMILIES)}

    def kernel_times(self, reset=True):
        """Device ms and launch count per kernel family over the render calls made with RENDER_TIMING."""
        ms, n = np.zeros(5, np.float32), np.zeros(5, np.int32)
        _check(host().srth_kernel_times(self.h, _p(ms), _p(n, _i32p), int(reset)))
        return {name: {"ms": float(ms[i]), "launches": int(n[i])} for i, name in enumerate(self.FAMILIES)}

    def synchronize(self, stream=None):
        _check(host().srth_synchronize(self.h, stream))

    def probe_gather(self, window_bytes=0, loads_per_lane=256, quad=False):
        """GB/s of 64-byte record gathers over the first window_bytes of the node + triangle records (0: all); quad:
        four lanes fetch one record together instead of every lane its own."""
        gbs, used = C.c_double(), C.c_uint64()
        _check(host().srth_probe_gather(self.h, int(window_bytes), int(loads_per_lane), int(bool(quad)), C.byref(gbs), C.byref(used)))
        return float(gbs.value), int(used.value)


def make_camera(width, height, fovy_deg=50.0, dist=3.0, look_at=(0, 0, 0), rot_deg=(20, 20, 0)):
    """Trackball set up as the reference's CLI does; defaults are its CameraConfig defaults (src/config.h:16-21)."""
    cam = Camera()
    la, rd = np.array(look_at, np.float32), np.array(rot_deg, np.float32)
    _check(host().srth_camera_make(fovy_deg, dist, _p(la), _p(rd), width, height, C.byref(cam)))
    return cam


def generate_rays(cam, ndc):
    ndc = np.ascontiguousarray(ndc, np.float32).reshape(-1, 2)
    rays = np.zeros((len(ndc), 7), np.float32)
    _check(host().srth_generate_rays(C.byref(cam), _p(ndc), len(ndc), _p(rays)))
    return rays


def selftest_division(n=1 << 24, seed=1):
    """Mismatches between the fast slab test's division and plain division on n random operand pairs (expect 0)."""
    bad = C.c_uint64()
    _check(host().srth_selftest_division(seed, n, C.byref(bad)))
    return int(bad.value)


def mirror_render(bvh, features, width, height, num_rays=1, fovy_deg=50.0, dist=3.0, look_at=(0, 0, 0), rot_deg=(20, 20, 0)):
    """The C++ mirror as a reference call site uses it: Screen, Trackball, renderRayTracing (src/main.cpp:472-476)."""
    rgb = np.zeros((height, width, 3), np.float32)
    la, rd = np.array(look_at, np.float32), np.array(rot_deg, np.float32)
    _check(host().srth_mirror_render(bvh.h, features, width, height, fovy_deg, dist, _p(la), _p(rd), num_rays, _p(rgb)))
    return rgb


def mirror_render_frames(bvh, features, width, height, cameras, num_rays=1, tile=(0, 0)):
    """renderFrames: several cameras of one scene in one call (what the command-line front end does with its [[cameras]]).
    cameras: rows of (fovy_deg, dist, look_at xyz, rot_deg xyz); returns (n, H, W, 3)."""
    cams = np.ascontiguousarray(cameras, np.float32).reshape(-1, 8)
    rgb = np.zeros((len(cams), height, width, 3), np.float32)
    _check(host().srth_mirror_render_frames(bvh.h, features, width, height, len(cams), _p(cams), num_rays, int(tile[0]), int(tile[1]), _p(rgb)))
    return rgb


def tiles_of(width, height, tile_w, tile_h):
    """tilesOf of the host mirror: a frame's tiles (x0, y0, x1, y1; y up) in the order they are handed out."""
    n = host().srth_tiles_of(width, height, tile_w, tile_h, None, 0)
    out = np.zeros((n, 4), np.int32)
    host().srth_tiles_of(width, height, tile_w, tile_h, _p(out, _i32p), n)
    return out


def run_workers_selftest(workers, items, fail_at=-1):
    """runWorkers of the host mirror on a work function that only records who ran what; returns (runs per item, items per
    worker); raises SrtError when the item fail_at threw."""
    done, by = np.zeros(items, np.int32), np.zeros(workers, np.int32)
    rc = host().srth_run_workers_selftest(workers, items, fail_at, _p(done, _i32p), _p(by, _i32p))
    if rc != 0:
        raise SrtError(host().srth_last_error().decode())
    return done, by


def mirror_intersect_one(bvh, ray7, features):
    """BvhInterface::intersect(Ray&, HitInfo&, features) on one ray; returns (hit, t, HitInfo fields)."""
    r = np.ascontiguousarray(ray7, np.float32).copy()
    n, b, uv, kd, ks = (np.zeros(k, np.float32) for k in (3, 3, 2, 3, 3))
    sh, tex = np.zeros(1, np.float32), np.zeros(1, np.int32)
    rc = host().srth_mirror_intersect_one(bvh.h, _p(r), features, _p(n), _p(b), _p(uv), _p(kd), _p(ks), _p(sh), _p(tex, _i32p))
    if rc < 0:
        _check(rc)
    return bool(rc), float(r[6]), {"normal": n, "bary": b, "uv": uv, "kd": kd, "ks": ks, "shininess": float(sh[0]), "has_texture": bool(tex[0])}


def mirror_final_color(bvh, ray7, features):
    r = np.ascontiguousarray(ray7, np.float32)
    rgb = np.zeros(3, np.float32)
    _check(host().srth_mirror_final_color(bvh.h, _p(r), features, _p(rgb)))
    return rgb


def mirror_generate_ray(ndc, aspect, fovy_deg=50.0, dist=3.0, look_at=(0, 0, 0), rot_deg=(20, 20, 0)):
    la, rd, nd = np.array(look_at, np.float32), np.array(rot_deg, np.float32), np.array(ndc, np.float32)
    ray = np.zeros(7, np.float32)
    _check(host().srth_mirror_generate_ray(aspect, fovy_deg, dist, _p(la), _p(rd), _p(nd), _p(ray)))
    return ray


def bloom(rgb, threshold, box_size):
    """Screen::applyBloomFilter on a frame (top row first); returns the filtered copy."""
    out = np.ascontiguousarray(rgb, np.float32).copy()
    _check(host().srth_bloom(_p(out), out.shape[1], out.shape[0], threshold, box_size))
    return out


def config_describe(toml_text):
    """The command-line front end's reading of a configuration text, printed as the reference prints its Config."""
    n = host().srth_config_describe(toml_text.encode(), None, 0)
    if n < 0:
        raise SrtError(host().srth_last_error().decode())
    buf = C.create_string_buffer(n)
    host().srth_config_describe(toml_text.encode(), buf, n)
    return buf.value.decode()


CLI = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "srt_render")


def load_image(path):
    """Image(path) of the host mirror (PNG, JPEG): float RGB (H, W, 3), top row first, u8 / 255."""
    w, h = C.c_int32(), C.c_int32()
    _check(host().srth_image_load(os.fsencode(path), C.byref(w), C.byref(h), None))
    rgb = np.zeros((h.value, w.value, 3), np.float32)
    _check(host().srth_image_load(os.fsencode(path), C.byref(w), C.byref(h), _p(rgb)))
    return rgb


def write_bmp(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    _check(host().srth_write_bmp(os.fsencode(path), _p(rgb), rgb.shape[1], rgb.shape[0]))


class TileCounter:
    """Frame-indexed tile counters shared by the per-GPU processes of a node (POSIX shared memory, one atomic
    fetch-add per tile).  The creating process must open it before the others."""
    SLOTS = 64

    def __init__(self, name, create):
        self.h = host().srth_tilecounter_open(name.encode(), int(create))
        if not self.h:
            raise SrtError(host().srth_last_error().decode())

    def next(self, frame):
        return int(host().srth_tilecounter_next(self.h, frame % self.SLOTS))

    def reset(self, frame):
        host().srth_tilecounter_reset(self.h, frame % self.SLOTS)

    def close(self):
        if self.h:
            host().srth_tilecounter_close(self.h)
            self.h = None



class FrameBoard:
    """Per-frame record of who rendered which tile (and what it cost), shared by the per-GPU processes of a node through
    POSIX shared memory: every process posts its list when its tiles are done and then reads everybody's.  No GPU work
    and no collective is involved.  The creating process must open it before the others."""

    def __init__(self, name, create, world, n_tiles):
        self.world, self.n_tiles = int(world), int(n_tiles)
        self.h = host().srth_frameboard_open(name.encode(), int(create), self.world, self.n_tiles)
        if not self.h:
            raise SrtError(host().srth_last_error().decode())

    def post(self, frame, rank, ids, costs):
        ids = np.ascontiguousarray(ids, np.int32)
        costs = np.ascontiguousarray(costs, np.int32)
        _check(host().srth_frameboard_post(self.h, int(frame), int(rank), len(ids), _p(ids, _i32p), _p(costs, _i32p)))

    def collect(self, frame, timeout_s=600.0):
        """Waits for every process's post of `frame`; returns [(ids, costs)] by rank."""
        _check(host().srth_frameboard_wait(self.h, int(frame), int(timeout_s * 1000)))
        out = []
        ids, costs = np.zeros(self.n_tiles, np.int32), np.zeros(self.n_tiles, np.int32)
        for r in range(self.world):
            k = host().srth_frameboard_read(self.h, int(frame), r, _p(ids, _i32p), _p(costs, _i32p))
            out.append((ids[:k].copy(), costs[:k].copy()))
        return out

    def close(self):
        if self.h:
            host().srth_frameboard_close(self.h)
            self.h = None
