"""Host-side mirror of the reference's renderer seams over the C-ABI of librt_amd.so.

    TracerScene(scene)                  <- AABBTree::AABBTree          (reference src/AABBTree.cpp:3-16)
    TracerScene.trace(org, dir, t0, t1) <- get_hit                     (src/render_thread.cpp:5-23)
    TracerScene.render_rect(opt, rect)  <- RenderThread::operator()    (src/render_thread.cpp:25-488)
    resolve_u8(rgb)                     <- save_jpg's 8-bit conversion (src/main.cpp:157-167)

There is no CPU fallback: importing works without a GPU (so that symbols can be checked), every compute call needs
the HIP library and a gfx950 device and raises otherwise.
"""
import ctypes as C
import os

import numpy as np

from ._abi import CameraDesc, LightDesc, MaterialDesc, MeshDesc, Options, Stats, pack_scene

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_amd.so")
_LIB = None

EXPORTS = ["rt_init", "rt_last_error", "rt_scene_create", "rt_scene_create_ex", "rt_scene_check_tree", "rt_scene_destroy", "rt_scene_info", "rt_scene_set_level_limit", "rt_tree_check", "rt_trace",
           "rt_trace_counted", "rt_render_rect", "rt_render_rect_noalloc", "rt_render_frame_multi", "rt_resolve_u8", "rt_save_jpeg", "rt_encode_jpeg",
           "rt_host_scene_load", "rt_host_scene_desc", "rt_host_scene_free", "rt_host_last_error", "rt_options_default", "rt_options_parse_line"]


class RtError(RuntimeError):
    pass


def lib():
    """Loads librt_amd.so (built by raytracer_amd/csrc/Makefile); raises if it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RtError("%s is missing: build it with `make -C raytracer_amd/csrc` (hipcc, gfx950)" % LIB_PATH)
        # PyTorch wheels bundle their own copy of the HIP runtime.  If this library pulled in the system's copy first and
        # torch were imported afterwards, the process would hold two runtimes and the second one finds no GPU; importing
        # torch first makes the dynamic loader hand its copy (same soname) to this library too.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, i64, f32 = C.c_void_p, C.c_int64, C.c_float
        L.rt_init.argtypes = [C.c_int]
        L.rt_last_error.restype = C.c_char_p
        L.rt_scene_create.argtypes = [C.POINTER(MeshDesc), C.c_int, C.POINTER(MaterialDesc), C.c_int,
                                      C.POINTER(LightDesc), C.c_int, C.POINTER(CameraDesc), C.POINTER(vp)]
        L.rt_scene_create_ex.argtypes = [C.POINTER(MeshDesc), C.c_int, C.POINTER(MaterialDesc), C.c_int,
                                         C.POINTER(LightDesc), C.c_int, C.POINTER(CameraDesc), C.c_uint32, C.POINTER(vp)]
        L.rt_scene_check_tree.argtypes = [vp, C.POINTER(MeshDesc), C.c_int, C.POINTER(i64)]
        L.rt_scene_destroy.argtypes = [vp]
        L.rt_scene_info.argtypes = [vp, C.POINTER(i64)]
        L.rt_scene_set_level_limit.argtypes = [vp, i64]
        L.rt_tree_check.argtypes = [C.POINTER(MeshDesc), C.c_int, C.c_int, C.POINTER(i64)]
        L.rt_trace.argtypes = [vp, vp, vp, f32, f32, i64, C.c_int, C.c_int, vp, vp, vp, vp]
        L.rt_trace_counted.argtypes = [vp, vp, vp, f32, f32, i64, C.c_int, C.c_int, vp, vp, vp, vp, vp]
        L.rt_render_rect.argtypes = [vp, C.POINTER(Options), C.c_int, C.c_int, C.c_int, C.c_int, vp,
                                     C.POINTER(Stats), vp]
        L.rt_render_rect_noalloc.argtypes = L.rt_render_rect.argtypes
        L.rt_render_frame_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Options), C.c_int, vp, C.POINTER(Stats)]
        L.rt_resolve_u8.argtypes = [vp, C.c_int, C.c_int, vp, vp]
        L.rt_save_jpeg.argtypes = [C.c_char_p, vp, C.c_int, C.c_int, C.c_int, C.c_char_p]
        L.rt_encode_jpeg.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_char_p, vp, i64, C.POINTER(i64)]
        L.rt_host_scene_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.rt_host_scene_desc.argtypes = [vp, C.POINTER(C.POINTER(MeshDesc)), C.POINTER(C.c_int), C.POINTER(C.POINTER(MaterialDesc)), C.POINTER(C.c_int),
                                         C.POINTER(C.POINTER(LightDesc)), C.POINTER(C.c_int), C.POINTER(C.POINTER(CameraDesc))]
        L.rt_host_scene_free.argtypes = [vp]
        L.rt_host_last_error.restype = C.c_char_p
        L.rt_options_default.argtypes = [C.POINTER(Options)]
        L.rt_options_parse_line.argtypes = [C.POINTER(Options), C.c_char_p, C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def _check(rc, what):
    if rc != 0:
        raise RtError("%s failed (%d): %s" % (what, rc, lib().rt_last_error().decode()))


def tree_check(scene, which):
    """Builds one of the two trees of a scene on the host (0: the reference's trees flattened, 1: the tree the pruning
    traversal walks), verifies its invariants and returns its shape.  Needs no GPU."""
    meshes, _, _, _, keep = pack_scene(scene)
    c = (C.c_int64 * 8)()
    _check(lib().rt_tree_check(meshes, len(scene.meshes), int(which), c), "rt_tree_check")
    del keep
    return dict(tris=c[0], nodes=c[1], leaves=c[2], depth=c[3], max_leaf=c[4], reached_tris=c[5], reached_nodes=c[6])


def init(device=-1):
    _check(lib().rt_init(device), "rt_init")


def _ptr(a):
    """Address of a numpy array or of a torch tensor (host or device)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()


class TracerScene:
    """A scene resident on the current device: flattened tree, triangles, materials, lights, camera."""

    def __init__(self, scene, device_build=None):
        """device_build: True builds the pruning traversal's tree on the GPU (rt_scene_create_ex with
        RT_BUILD_TREE_ON_DEVICE), False on host threads, None leaves it to the RT_AMD_BUILDER environment variable."""
        meshes, mats, lights, cam, keep = pack_scene(scene)
        h = C.c_void_p()
        if device_build is None:
            _check(lib().rt_scene_create(meshes, len(scene.meshes), mats, len(scene.materials), lights,
                                         len(scene.lights), C.byref(cam), C.byref(h)), "rt_scene_create")
        else:
            _check(lib().rt_scene_create_ex(meshes, len(scene.meshes), mats, len(scene.materials), lights,
                                            len(scene.lights), C.byref(cam), 1 if device_build else 0, C.byref(h)), "rt_scene_create_ex")
        self._h = h
        self._destroy = lib().rt_scene_destroy        # bound now: at interpreter shutdown module globals and builtins may be gone
        self.scene = scene

    def check_tree(self):
        """Downloads the pruning traversal's tree and verifies its invariants on the host (rt_scene_check_tree)."""
        meshes, _, _, _, keep = pack_scene(self.scene)
        c = (C.c_int64 * 8)()
        _check(lib().rt_scene_check_tree(self._h, meshes, len(self.scene.meshes), c), "rt_scene_check_tree")
        del keep
        return dict(tris=c[0], nodes=c[1], leaves=c[2], depth=c[3], max_leaf=c[4], reached_tris=c[5], reached_nodes=c[6],
                    built_on_device=bool(c[7]))

    def close(self):
        """Frees the scene's device memory.  Safe to call twice, from __del__, and while the interpreter shuts down (it
        touches nothing but this object's own attributes)."""
        d = self.__dict__
        h, destroy = d.get("_h"), d.get("_destroy")
        d["_h"] = None
        if h and destroy is not None:
            destroy(h)

    def __del__(self):
        try:
            self.close()
        except:     # noqa: E722  (nothing may escape a finaliser, and at shutdown even exception classes may be gone)
            pass

    def info(self):
        c = (C.c_int64 * 4)()
        _check(lib().rt_scene_info(self._h, c), "rt_scene_info")
        return dict(tris=c[0], inner=c[1], leaves=c[2], depth=c[3])

    def set_level_limit(self, max_items):
        """Largest number of light-ray items one level of a band's ray tree may hold before the band is split (0: default)."""
        _check(lib().rt_scene_set_level_limit(self._h, int(max_items)), "rt_scene_set_level_limit")

    def trace(self, org, dir, t0, t1, any_hit=False, exact_boxes=False, counted=False, stream=None):
        """Nearest (or any) hit of n rays.  numpy in -> numpy out (t, tri, normal[, nodes, tris])."""
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        dir = np.ascontiguousarray(dir, dtype=np.float32).reshape(-1, 3)
        n = org.shape[0]
        t = np.empty(n, np.float32)
        tri = np.empty(n, np.int32)
        if counted:
            cn = np.empty(n, np.uint32)
            ct = np.empty(n, np.uint32)
            _check(lib().rt_trace_counted(self._h, _ptr(org), _ptr(dir), t0, t1, n, int(any_hit), int(exact_boxes),
                                          _ptr(t), _ptr(tri), _ptr(cn), _ptr(ct), stream), "rt_trace_counted")
            return t, tri, cn, ct
        nrm = np.empty((n, 3), np.float32)
        _check(lib().rt_trace(self._h, _ptr(org), _ptr(dir), t0, t1, n, int(any_hit), int(exact_boxes),
                              _ptr(t), _ptr(tri), _ptr(nrm), stream), "rt_trace")
        return t, tri, nrm

    def render_rect(self, opt, x0, y0, x1, y1, out=None, stats=True, stream=None, noalloc=False):
        """Renders the pixel rectangle; `out` may be a numpy array or a CUDA torch tensor [y1-y0, x1-x0, 3] float32
        (device tensors are written in place, asynchronously on `stream`).  noalloc: through rt_render_rect_noalloc
        (fixed workspace, device output only)."""
        if out is None:
            out = np.empty((y1 - y0, x1 - x0, 3), np.float32)
        st = Stats() if stats else None
        fn = lib().rt_render_rect_noalloc if noalloc else lib().rt_render_rect
        _check(fn(self._h, C.byref(opt), x0, y0, x1, y1, _ptr(out),
                  C.byref(st) if st is not None else None, stream), fn.__name__)
        return out, st


def render_frame_multi(tracers, opt, band_rows=8, out=None, stats=True):
    """One frame rendered by several scene handles (one per GPU) from this one process; see rt_render_frame_multi."""
    if out is None:
        out = np.empty((opt.height, opt.width, 3), np.float32)
    hs = (C.c_void_p * len(tracers))(*[t._h for t in tracers])
    st = Stats() if stats else None
    _check(lib().rt_render_frame_multi(hs, len(tracers), C.byref(opt), band_rows, _ptr(out),
                                       C.byref(st) if st is not None else None), "rt_render_frame_multi")
    return out, st


def resolve_u8(rgb, out=None, stream=None):
    """[H, W, 3] float32 (row 0 = bottom) -> [H, W, 3] uint8 (row 0 = top)."""
    h, w = rgb.shape[0], rgb.shape[1]
    if out is None:
        out = np.empty((h, w, 3), np.uint8)
    _check(lib().rt_resolve_u8(_ptr(rgb), w, h, _ptr(out), stream), "rt_resolve_u8")
    return out


def load_scene_file(path):
    """A scene file through the library's own loader (rt_host_scene_load: binary FBX as the reference's load_scene reads it,
    src/main.cpp:256-395, or .rtscene) as a raytracer_amd.scene.Scene.  Host-only, needs no GPU."""
    from .scene import Camera, Light, Material, Mesh, Scene
    L = lib()
    h = C.c_void_p()
    if L.rt_host_scene_load(str(path).encode(), C.byref(h)) != 0:
        raise RtError("rt_host_scene_load failed: %s" % L.rt_host_last_error().decode())
    try:
        pm, pmat, pl, pc = C.POINTER(MeshDesc)(), C.POINTER(MaterialDesc)(), C.POINTER(LightDesc)(), C.POINTER(CameraDesc)()
        nm, nmat, nl = C.c_int(), C.c_int(), C.c_int()
        L.rt_host_scene_desc(h, C.byref(pm), C.byref(nm), C.byref(pmat), C.byref(nmat), C.byref(pl), C.byref(nl), C.byref(pc))
        f3 = lambda a: np.array(list(a), np.float32)
        mats = [Material(pmat[i].kind, f3(pmat[i].base), f3(pmat[i].ambient), f3(pmat[i].spec), f3(pmat[i].transp), pmat[i].shininess,
                         pmat[i].reflectivity, pmat[i].refractivity) for i in range(nmat.value)]
        lights = [Light(pl[i].kind, f3(pl[i].vec), f3(pl[i].color)) for i in range(nl.value)]
        cam = Camera(f3(pc[0].pos), f3(pc[0].up), f3(pc[0].w), f3(pc[0].side), pc[0].near_clip, pc[0].far_clip)
        meshes = []
        for i in range(nm.value):
            n = pm[i].n_tris
            pos = np.ctypeslib.as_array(C.cast(pm[i].pos, C.POINTER(C.c_float)), (n, 3, 3)).copy() if n else np.zeros((0, 3, 3), np.float32)
            nrm = np.ctypeslib.as_array(C.cast(pm[i].nrm, C.POINTER(C.c_float)), (n, 3, 3)).copy() if n else np.zeros((0, 3, 3), np.float32)
            meshes.append(Mesh(pm[i].name.decode(), pm[i].material, pos, nrm))
        return Scene(meshes, mats, lights, cam)
    finally:
        L.rt_host_scene_free(h)


def save_jpeg(path, img_u8, quality=90, comment="Render Result"):
    """Writes an [H, W, 3] uint8 image (row 0 = top) the way the reference's save_jpg does (src/main.cpp:148-180):
    baseline JPEG, quality 90, no chroma subsampling, comment "Render Result".  Host-only, needs no GPU."""
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w = img.shape[0], img.shape[1]
    _check(lib().rt_save_jpeg(str(path).encode(), img.ctypes.data, w, h, int(quality), comment.encode() if comment else None),
           "rt_save_jpeg")
