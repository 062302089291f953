"""ctypes mirror of include/rt_amd.h (struct layouts and argument packing)."""
import ctypes as C

import numpy as np

RT_OK = 0
RT_JITTER_OFF, RT_JITTER_HASH = 0, 1
RT_MAX_LIGHTS = 8


class MeshDesc(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("material", C.c_int32), ("n_tris", C.c_int32),
                ("pos", C.c_void_p), ("nrm", C.c_void_p)]


class MaterialDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("base", C.c_float * 3), ("ambient", C.c_float * 3),
                ("spec", C.c_float * 3), ("transp", C.c_float * 3), ("shininess", C.c_float),
                ("reflectivity", C.c_float), ("refractivity", C.c_float)]


class LightDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("vec", C.c_float * 3), ("color", C.c_float * 3), ("pad_", C.c_float)]


class CameraDesc(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("up", C.c_float * 3), ("w", C.c_float * 3), ("side", C.c_float * 3),
                ("near_clip", C.c_float), ("far_clip", C.c_float), ("pad_", C.c_float * 2)]


class Options(C.Structure):
    """rt_options; defaults are the reference's (src/main.cpp:74-75,103-111)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("hfov", C.c_float), ("spp", C.c_int32),
                ("spr", C.c_int32), ("max_bounce", C.c_int32), ("max_refrac_bounce", C.c_int32),
                ("eps", C.c_float), ("ray_eps", C.c_float), ("jitter_mode", C.c_int32), ("seed", C.c_uint32),
                ("exact_boxes", C.c_int32), ("count_traversal", C.c_int32), ("row_stride", C.c_int32), ("row_band", C.c_int32), ("row_phase", C.c_int32),
                ("skip_unlit", C.c_int32)]

    def __init__(self, width=960, height=540, hfov=54.43, spp=2, spr=4, max_bounce=3, max_refrac_bounce=3,
                 eps=1e-4, ray_eps=0.1, jitter_mode=RT_JITTER_OFF, seed=1, exact_boxes=0, count_traversal=0, row_stride=0, row_band=0, row_phase=0,
                 skip_unlit=0):
        super().__init__(width, height, hfov, spp, spr, max_bounce, max_refrac_bounce, eps, ray_eps,
                         jitter_mode, seed, exact_boxes, count_traversal, row_stride, row_band, row_phase, skip_unlit)


class Stats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_any", C.c_uint64), ("ref_get_hit_calls", C.c_uint64),
                ("nodes_extend", C.c_uint64), ("tris_extend", C.c_uint64), ("nodes_shadow", C.c_uint64),
                ("tris_shadow", C.c_uint64), ("ms_total", C.c_float),
                ("ms_extend", C.c_float), ("ms_shadow", C.c_float), ("ms_shade", C.c_float),
                ("levels", C.c_int32), ("reserved_", C.c_int32 * 3)]


def _f3(a):
    return (C.c_float * 3)(*[float(x) for x in a])


def pack_scene(scene):
    """Scene -> (mesh[], material[], light[], camera, keepalive) ctypes arguments."""
    scene.validate()
    keep = []
    meshes = (MeshDesc * len(scene.meshes))()
    for d, m in zip(meshes, scene.meshes):
        pos = np.ascontiguousarray(m.pos, dtype=np.float32)
        nrm = np.ascontiguousarray(m.nrm, dtype=np.float32)
        keep += [pos, nrm]
        d.name = m.name.encode()
        d.material = m.material
        d.n_tris = m.n_tris
        d.pos = pos.ctypes.data
        d.nrm = nrm.ctypes.data
    mats = (MaterialDesc * len(scene.materials))()
    for d, m in zip(mats, scene.materials):
        d.kind = m.kind
        d.base, d.ambient, d.spec, d.transp = _f3(m.base), _f3(m.ambient), _f3(m.spec), _f3(m.transp)
        d.shininess, d.reflectivity, d.refractivity = m.shininess, m.reflectivity, m.refractivity
    lights = (LightDesc * max(1, len(scene.lights)))()
    for d, l in zip(lights, scene.lights):
        d.kind = l.kind
        d.vec, d.color = _f3(l.vec), _f3(l.color)
    cam = CameraDesc()
    c = scene.camera
    cam.pos, cam.up, cam.w, cam.side = _f3(c.pos), _f3(c.up), _f3(c.w), _f3(c.side)
    cam.near_clip, cam.far_clip = c.near, c.far
    return meshes, mats, lights, cam, keep
