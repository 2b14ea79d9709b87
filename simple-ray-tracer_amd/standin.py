"""Deterministic procedural stand-in for the reference's `dragon.obj`, which is not part of its checkout
(the file is listed as a missing large blob).  A scaly tube wound along a (2,3) torus knot: a closed,
self-occluding surface with strong depth complexity, tessellated on a regular grid.  With the default grid it
has 262 144 triangles, i.e. more than 2^17, so the reference's BVH builder hits its depth cap of 17 levels and
produces multi-triangle leaves, the regime a dragon-sized mesh is in.

The mesh is centred and scaled to the unit sphere the way the reference's loader does for its Dragon scene
(loadMesh(..., normalize = true), reference framework/src/mesh.cpp:151-176) and carries the default material the
reference's other large models use (kd 1, ks 0.2, shininess 10; no texture, since the surface has no authored uv).
"""
import numpy as np

LIGHT_STRIDE = 22

# SHA-256 of dragon_standin()'s default `verts` and `tris` arrays (C order, little endian): the geometry every
# published dragon number was measured on.  dragon_standin(check=True) refuses to hand out anything else.
DRAGON_STANDIN_SHA256 = {
    "verts": "2afc533ca39bf5a8d22ee96dcf149c03c9a40c33829330cb529a6e18be642449",
    "tris": "00a1ca6a1023d821bdd573a272c35de5477dfc9fd9d7f94bf825f959d4b18511",
}


def dragon_standin(nu=1024, nv=128, check=False):
    """Returns the scene in the fixture layout (one mesh, no lights).  check: raise unless the default grid's
    arrays carry the pinned hashes (numpy's sin / cos could round differently on another build)."""
    u = (np.arange(nu, dtype=np.float64) / nu) * 2.0 * np.pi
    v = (np.arange(nv, dtype=np.float64) / nv) * 2.0 * np.pi

    def centre(t):
        r = 2.0 + np.cos(3.0 * t)
        return np.stack([r * np.cos(2.0 * t), r * np.sin(2.0 * t), np.sin(3.0 * t)], -1)

    c = centre(u)
    h = 1e-4
    tangent = centre(u + h) - centre(u - h)
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    side = np.cross(tangent, up)
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    binorm = np.cross(tangent, side)
    U, V = np.meshgrid(u, v, indexing="ij")
    # body thickness along the knot, a dorsal ridge and small scales
    radius = 0.42 * (1.0 + 0.30 * np.sin(5.0 * U)) * (1.0 + 0.18 * np.cos(V) ** 8) + 0.035 * np.sin(48.0 * U) * np.sin(12.0 * V)
    offset = side[:, None, :] * np.cos(V)[..., None] + binorm[:, None, :] * np.sin(V)[..., None]
    pos = c[:, None, :] + radius[..., None] * offset
    pos = pos.reshape(-1, 3).astype(np.float32)
    nrm = offset.reshape(-1, 3).astype(np.float32)

    # centre and scale to the unit sphere in float32, in the reference loader's order of operations
    centre32 = np.zeros(3, np.float32)
    for k in range(3):
        centre32[k] = np.float32(np.add.reduce(pos[:, k], dtype=np.float32))
    centre32 = centre32 / np.float32(len(pos))
    d = pos - centre32
    max_d = np.float32(np.sqrt(((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).max()))
    pos = (d / max_d).astype(np.float32)

    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    a = (i * nv + j).ravel()
    b = (((i + 1) % nu) * nv + j).ravel()
    cc = (((i + 1) % nu) * nv + (j + 1) % nv).ravel()
    dd = (i * nv + (j + 1) % nv).ravel()
    tris = np.empty((2 * nu * nv, 3), np.uint32)
    tris[0::2] = np.stack([a, b, cc], 1)
    tris[1::2] = np.stack([a, cc, dd], 1)

    verts = np.zeros((len(pos), 8), np.float32)
    verts[:, 0:3] = pos
    verts[:, 3:6] = nrm
    if check:
        import hashlib
        for key, arr in (("verts", verts), ("tris", tris)):
            if (nu, nv) != (1024, 128) or hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest() != DRAGON_STANDIN_SHA256[key]:
                raise RuntimeError(f"dragon stand-in: `{key}` differs from the pinned geometry")
    return {
        "mesh_vert_off": np.array([0, len(verts)], np.uint32),
        "mesh_tri_off": np.array([0, len(tris)], np.uint32),
        "verts": verts,
        "tris": tris,
        "materials": np.array([[1, 1, 1, 0.2, 0.2, 0.2, 10.0, 1.0]], np.float32),
        "mesh_tex": np.array([-1], np.int32),
        "tex_w": np.zeros(0, np.uint32), "tex_h": np.zeros(0, np.uint32), "tex_off": np.zeros(0, np.uint64),
        "texels": np.zeros((0, 3), np.float32), "texels_u8": np.zeros((0, 3), np.uint8),
        "lights": np.zeros((0, LIGHT_STRIDE), np.float32),
    }


SCENE_KEYS = ("mesh_vert_off", "mesh_tri_off", "verts", "tris", "materials", "mesh_tex", "tex_w", "tex_h", "tex_off",
              "texels_u8", "lights")


def load_scene_npz(path):
    """Scene stored as plain arrays (the layout Scene.from_arrays takes) -> dict; texels are expanded to float RGB
    the way the reference's image loader does it: u8 / 255.0f (reference framework/src/image.cpp:30-32)."""
    with np.load(path, allow_pickle=False) as z:
        d = {k: np.ascontiguousarray(z[k]) for k in SCENE_KEYS}
    d["texels"] = np.ascontiguousarray(d["texels_u8"].astype(np.float32) / np.float32(255.0))
    return d


def point_light(pos, col=(1, 1, 1)):
    l = np.zeros(LIGHT_STRIDE, np.float32)
    l[1:4], l[4:7] = pos, col
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


def dragon_lights():
    """The Dragon scene's point light (reference src/scene.cpp:70-75) plus one area light for soft shadows: the
    reference's Cornell parallelogram light (src/scene.cpp:47-55) lifted above the unit-sphere model."""
    return np.stack([
        point_light((-1, 1, -1)),
        parallelogram_light((-0.2, 1.6, -0.2), (0.4, 0, 0), (0, 0, 0.4), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1)),
    ])
