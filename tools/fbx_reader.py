"""Minimal binary-FBX reader (data only, executes nothing from the file).

Binary FBX layout (public Blender/Autodesk format notes): 27-byte header,
then nested node records {end_offset, num_props, prop_list_len, name_len,
name, properties, children..., 13/25-byte NUL sentinel}.  Offsets are
u32 before version 7500 and u64 from 7500 on.  Property type codes:
Y i16, C bool, I i32, F f32, D f64, L i64, f/d/l/i/b arrays (optionally
zlib-deflated), S string, R raw bytes.
"""
import struct
import zlib
import numpy as np


class Node:
    __slots__ = ("name", "props", "children")

    def __init__(self, name, props, children):
        self.name, self.props, self.children = name, props, children

    def find(self, name):
        for c in self.children:
            if c.name == name:
                return c
        return None

    def find_all(self, name):
        return [c for c in self.children if c.name == name]


_ARR = {b"f": "<f4", b"d": "<f8", b"l": "<i8", b"i": "<i4", b"b": "u1"}


def _read_prop(b, p):
    t = b[p:p + 1]
    p += 1
    if t == b"Y":
        return struct.unpack_from("<h", b, p)[0], p + 2
    if t == b"C":
        return bool(b[p]), p + 1
    if t == b"I":
        return struct.unpack_from("<i", b, p)[0], p + 4
    if t == b"F":
        return struct.unpack_from("<f", b, p)[0], p + 4
    if t == b"D":
        return struct.unpack_from("<d", b, p)[0], p + 8
    if t == b"L":
        return struct.unpack_from("<q", b, p)[0], p + 8
    if t in _ARR:
        n, enc, clen = struct.unpack_from("<III", b, p)
        p += 12
        raw = b[p:p + clen]
        if enc == 1:
            raw = zlib.decompress(raw)
        return np.frombuffer(raw, dtype=_ARR[t], count=n).copy(), p + clen
    if t in (b"S", b"R"):
        n = struct.unpack_from("<I", b, p)[0]
        p += 4
        raw = b[p:p + n]
        return (raw if t == b"R" else raw.decode("latin-1")), p + n
    raise ValueError("unknown FBX property type %r at %d" % (t, p))


def _read_node(b, p, wide):
    if wide:
        end, nprops, plen = struct.unpack_from("<QQQ", b, p)
        p += 24
    else:
        end, nprops, plen = struct.unpack_from("<III", b, p)
        p += 12
    nlen = b[p]
    p += 1
    if end == 0:
        return None, p  # sentinel
    name = b[p:p + nlen].decode("latin-1")
    p += nlen
    props = []
    pend = p + plen
    for _ in range(nprops):
        v, p = _read_prop(b, p)
        props.append(v)
    p = pend
    children = []
    while p < end:
        c, p = _read_node(b, p, wide)
        if c is None:
            break
        children.append(c)
    return Node(name, props, children), end


def read_fbx(path):
    b = open(path, "rb").read()
    if b[:21] != b"Kaydara FBX Binary  \x00":
        raise ValueError("not a binary FBX file")
    version = struct.unpack_from("<I", b, 23)[0]
    wide = version >= 7500
    p = 27
    top = []
    while p < len(b):
        n, p = _read_node(b, p, wide)
        if n is None:
            break
        top.append(n)
    return Node("", [version], top)


def dump(node, depth=0, maxdepth=3, out=None):
    import sys
    out = out or sys.stdout
    for c in node.children:
        ps = []
        for v in c.props:
            if isinstance(v, np.ndarray):
                ps.append("<%s[%d]>" % (v.dtype, v.size))
            elif isinstance(v, (bytes, str)) and len(v) > 60:
                ps.append(repr(v[:60]) + "...")
            else:
                ps.append(repr(v))
        out.write("  " * depth + c.name + ": " + ", ".join(ps) + "\n")
        if depth + 1 < maxdepth:
            dump(c, depth + 1, maxdepth, out)


if __name__ == "__main__":
    import sys
    dump(read_fbx(sys.argv[1]), maxdepth=int(sys.argv[2]) if len(sys.argv) > 2 else 3)
