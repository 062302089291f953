"""ctypes wrapper of liboracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd._abi import CameraDesc, LightDesc, MaterialDesc, MeshDesc, Options, Stats, pack_scene

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_scene_create.argtypes = [C.POINTER(MeshDesc), C.c_int, C.POINTER(MaterialDesc), C.c_int,
                                       C.POINTER(LightDesc), C.c_int, C.POINTER(CameraDesc), C.POINTER(C.c_void_p)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_destroy.restype = None
        L.orc_scene_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.orc_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int64, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render_rect.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.POINTER(Stats)]
        L.orc_render_frame_mt.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.c_int, C.c_void_p, C.POINTER(Stats)]
        L.orc_render_frame_mt.restype = C.c_double
        L.orc_resolve_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_resolve_u8.restype = None
        _LIB = L
    return _LIB


class OracleScene:
    def __init__(self, scene):
        meshes, mats, lights, cam, keep = pack_scene(scene)
        h = C.c_void_p()
        rc = lib().orc_scene_create(meshes, len(scene.meshes), mats, len(scene.materials), lights,
                                    len(scene.lights), C.byref(cam), C.byref(h))
        if rc != 0:
            raise RuntimeError("orc_scene_create failed: %d" % rc)
        self._h = h
        self.scene = scene

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def info(self):
        c = (C.c_int64 * 4)()
        lib().orc_scene_info(self._h, c)
        return dict(tris=c[0], inner=c[1], leaves=c[2], depth=c[3])

    def trace(self, org, dir, t0, t1, any_hit=False, count=False):
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        dir = np.ascontiguousarray(dir, dtype=np.float32).reshape(-1, 3)
        n = org.shape[0]
        t = np.empty(n, np.float32)
        tri = np.empty(n, np.int32)
        nrm = np.empty((n, 3), np.float32)
        cnt = np.zeros(2, np.uint64)
        lib().orc_trace(self._h, org.ctypes.data, dir.ctypes.data, t0, t1, n, int(any_hit), t.ctypes.data,
                        tri.ctypes.data, nrm.ctypes.data, cnt.ctypes.data if count else None)
        if count:
            return t, tri, nrm, cnt
        return t, tri, nrm

    def render_rect(self, opt, x0, y0, x1, y1):
        out = np.empty((y1 - y0, x1 - x0, 3), np.float32)
        st = Stats()
        rc = lib().orc_render_rect(self._h, C.byref(opt), x0, y0, x1, y1, out.ctypes.data, C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_render_rect failed: %d" % rc)
        return out, st

    def render_frame_mt(self, opt, hw_threads, prog_disp_span=100):
        out = np.empty((opt.height, opt.width, 3), np.float32)
        st = Stats()
        secs = lib().orc_render_frame_mt(self._h, C.byref(opt), hw_threads, prog_disp_span, out.ctypes.data, C.byref(st))
        return out, st, secs


def resolve_u8(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    out = np.empty((h, w, 3), np.uint8)
    lib().orc_resolve_u8(rgb.ctypes.data, w, h, out.ctypes.data)
    return out
