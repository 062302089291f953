"""MI355X-native ray-tracing hot path (BVH traversal, ray/triangle tests, shadow/reflect/refract bounce loop)."""
from ._abi import Options, Stats, RT_JITTER_OFF, RT_JITTER_HASH  # noqa: F401
from .scene import Scene, Mesh, Material, Light, Camera  # noqa: F401
from .api import TracerScene, resolve_u8, init, lib, RtError  # noqa: F401
