"""Host-side scene description: the inputs the tile renderer consumes.

Mirrors what the reference's scene loader hands to its renderer
(reference src/main.cpp:256-395): per-mesh world-space triangles with vertex
normals in face order (src/geometry.cpp:360-376), a material per mesh chosen
by name (src/main.cpp:331-349, src/materials.cpp:10-118), lights
(src/lights.cpp:7-25) and one camera basis (src/cameras.cpp:3-18).

The on-disk form (".rtscene") is a flat little-endian binary so that a C++
host can read it without any parser dependency.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List

import numpy as np

MAT_DEFAULT = 0   # no recognised material: solid yellow, no shadowing (src/materials.cpp:4-8)
MAT_LAMBERT = 1   # src/materials.cpp:28-43
MAT_PHONG = 2     # src/materials.cpp:68-88
MAT_REFRAC = 3    # src/materials.cpp:120-140

LIGHT_DIRECTIONAL = 0  # src/lights.cpp:13-20
LIGHT_POINT = 1        # src/lights.cpp:27-54

_MAGIC = b"RTSCN01\x00"


def _f3(v):
    a = np.asarray(v, dtype=np.float32).reshape(3)
    return a


@dataclass
class Material:
    kind: int = MAT_LAMBERT
    base: np.ndarray = field(default_factory=lambda: _f3([0.5, 0.5, 0.5]))
    ambient: np.ndarray = field(default_factory=lambda: _f3([0.05, 0.05, 0.05]))
    spec: np.ndarray = field(default_factory=lambda: _f3([0.5, 0.5, 0.5]))
    transp: np.ndarray = field(default_factory=lambda: _f3([1.0, 1.0, 1.0]))
    shininess: float = 32.0
    reflectivity: float = 0.5
    refractivity: float = 1.333

    def packed(self) -> np.ndarray:
        """16 x 4-byte record: kind + 15 floats (the C-ABI's rt_material_desc)."""
        rec = np.zeros(16, dtype=np.float32)
        rec[0:1].view(np.int32)[0] = self.kind
        rec[1:4] = _f3(self.base)
        rec[4:7] = _f3(self.ambient)
        rec[7:10] = _f3(self.spec)
        rec[10:13] = _f3(self.transp)
        rec[13] = self.shininess
        rec[14] = self.reflectivity
        rec[15] = self.refractivity
        return rec

    @staticmethod
    def unpack(rec: np.ndarray) -> "Material":
        rec = np.asarray(rec, dtype=np.float32)
        return Material(int(rec[0:1].view(np.int32)[0]), rec[1:4].copy(), rec[4:7].copy(),
                        rec[7:10].copy(), rec[10:13].copy(), float(rec[13]), float(rec[14]),
                        float(rec[15]))


@dataclass
class Light:
    kind: int
    vec: np.ndarray    # directional: unit vector TOWARD the light; point: position
    color: np.ndarray

    def packed(self) -> np.ndarray:
        rec = np.zeros(8, dtype=np.float32)
        rec[0:1].view(np.int32)[0] = self.kind
        rec[1:4] = _f3(self.vec)
        rec[4:7] = _f3(self.color)
        return rec


@dataclass
class Camera:
    pos: np.ndarray
    up: np.ndarray
    w: np.ndarray      # unit vector pointing BACKWARD (the reference's Camera::lookat)
    side: np.ndarray
    near: float = 0.1
    far: float = 10000.0

    def packed(self) -> np.ndarray:
        rec = np.zeros(16, dtype=np.float32)
        rec[0:3] = _f3(self.pos)
        rec[3:6] = _f3(self.up)
        rec[6:9] = _f3(self.w)
        rec[9:12] = _f3(self.side)
        rec[12] = self.near
        rec[13] = self.far
        return rec

    @staticmethod
    def look(pos, up, lookat_dir, near=0.1, far=10000.0) -> "Camera":
        """Builds the basis the way src/cameras.cpp:3-18 does, in float32."""
        f = np.float32

        def norm(v):
            v = np.asarray(v, dtype=f)
            p = f(0)
            for i in range(3):
                p = f(p + f(v[i] * v[i]))
            k = f(f(1) / f(np.sqrt(p)))
            return np.array([f(v[i] * k) for i in range(3)], dtype=f)

        up_n = norm(up)
        w = norm(np.asarray(lookat_dir, dtype=f) * f(-1.0))
        side = np.array([f(f(up_n[1] * w[2]) - f(up_n[2] * w[1])),
                         f(f(up_n[2] * w[0]) - f(up_n[0] * w[2])),
                         f(f(up_n[0] * w[1]) - f(up_n[1] * w[0]))], dtype=f)
        side = norm(side)
        return Camera(_f3(pos), up_n, w, side, float(near), float(far))


@dataclass
class Mesh:
    name: str
    material: int
    pos: np.ndarray   # [n_tris, 3, 3] float32 world-space corner positions
    nrm: np.ndarray   # [n_tris, 3, 3] float32 world-space corner normals

    def __post_init__(self):
        self.pos = np.ascontiguousarray(self.pos, dtype=np.float32).reshape(-1, 3, 3)
        self.nrm = np.ascontiguousarray(self.nrm, dtype=np.float32).reshape(-1, 3, 3)
        if self.pos.shape != self.nrm.shape:
            raise ValueError("mesh %r: pos/nrm shape mismatch" % self.name)
        if len(self.name.encode()) > 31:
            raise ValueError("mesh name longer than 31 bytes: %r" % self.name)

    @property
    def n_tris(self) -> int:
        return self.pos.shape[0]


@dataclass
class Scene:
    meshes: List[Mesh]
    materials: List[Material]
    lights: List[Light]
    camera: Camera

    @property
    def n_tris(self) -> int:
        return sum(m.n_tris for m in self.meshes)

    def validate(self):
        names = [m.name for m in self.meshes]
        if len(set(names)) != len(names):
            # the reference shades once per mesh whose name matches the hit
            # (src/render_thread.cpp:178-180); duplicate names are not supported here
            raise ValueError("mesh names must be unique")
        for m in self.meshes:
            if not (0 <= m.material < len(self.materials)):
                raise ValueError("mesh %r: bad material index" % m.name)

    # ---------------------------------------------------------------- file IO
    def save(self, path):
        self.validate()
        with open(path, "wb") as f:
            f.write(_MAGIC)
            f.write(struct.pack("<III", len(self.meshes), len(self.materials), len(self.lights)))
            f.write(self.camera.packed().tobytes())
            for m in self.materials:
                f.write(m.packed().tobytes())
            for l in self.lights:
                f.write(l.packed().tobytes())
            for m in self.meshes:
                f.write(m.name.encode().ljust(32, b"\x00"))
                f.write(struct.pack("<ii", m.material, m.n_tris))
                f.write(m.pos.tobytes())
                f.write(m.nrm.tobytes())

    @staticmethod
    def load(path) -> "Scene":
        b = open(path, "rb").read()
        if b[:8] != _MAGIC:
            raise ValueError("%s: not an .rtscene file" % path)
        nm, nmat, nl = struct.unpack_from("<III", b, 8)
        p = 20
        c = np.frombuffer(b, np.float32, 16, p)
        p += 64
        cam = Camera(c[0:3].copy(), c[3:6].copy(), c[6:9].copy(), c[9:12].copy(), float(c[12]), float(c[13]))
        mats = []
        for _ in range(nmat):
            mats.append(Material.unpack(np.frombuffer(b, np.float32, 16, p)))
            p += 64
        lights = []
        for _ in range(nl):
            r = np.frombuffer(b, np.float32, 8, p)
            p += 32
            lights.append(Light(int(r[0:1].view(np.int32)[0]), r[1:4].copy(), r[4:7].copy()))
        meshes = []
        for _ in range(nm):
            name = b[p:p + 32].split(b"\x00")[0].decode()
            mat, nt = struct.unpack_from("<ii", b, p + 32)
            p += 40
            pos = np.frombuffer(b, np.float32, nt * 9, p).reshape(nt, 3, 3).copy()
            p += nt * 36
            nrm = np.frombuffer(b, np.float32, nt * 9, p).reshape(nt, 3, 3).copy()
            p += nt * 36
            meshes.append(Mesh(name, mat, pos, nrm))
        return Scene(meshes, mats, lights, cam)
