"""Input readers for the data either side of the hot path.

parse_lidar_scan mirrors common/parse_scan.cpp:18-50 (CSV "x,y" per line; rows where either
coordinate is +-inf are dropped).  read_ply_vertices is the minimal stand-in for what
tests/test_ply.cpp:32-46 gets from PCL: the x,y,z vertex properties, float32 cast to f64.
Clouds are returned as (n, dim) C-contiguous float64 arrays, whose memory is exactly the
reference's column-major Dim x n PointCloud (geo.h:27).
"""
from __future__ import annotations

import math

import numpy as np


def parse_lidar_scan(path: str) -> np.ndarray:
    try:
        f = open(path, "r")
    except OSError as e:  # parse_scan.cpp:20-22 throws std::runtime_error
        raise RuntimeError("failed to read lidar scan: failed to open file") from e
    pts = []
    with f:
        for line in f:
            line = line.rstrip("\n")
            index = line.find(",")
            xs = line[:index] if index >= 0 else line
            ys = line[index + 1:]
            x = _strtod(xs)
            y = _strtod(ys)
            if math.isinf(x) or math.isinf(y):
                continue
            pts.append((x, y))
    return np.array(pts, dtype=np.float64).reshape(-1, 2)


def _strtod(s: str) -> float:
    """std::strtod semantics for the inputs the scans hold: longest valid prefix, 0.0 if none."""
    s = s.strip()
    try:
        return float(s)
    except ValueError:
        # longest parsable prefix
        for k in range(len(s) - 1, 0, -1):
            try:
                return float(s[:k])
            except ValueError:
                continue
        return 0.0


_PLY_TYPES = {
    "char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4",
    "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2",
    "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8",
}


def read_ply_vertices(path: str) -> np.ndarray:
    """Vertex x,y,z of a binary_little_endian or ascii PLY as an (n, 3) float64 array."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.find(b"end_header")
    if end < 0 or not data.startswith(b"ply"):
        raise RuntimeError("Failed to read PLY file: " + path)
    body = end + len(b"end_header")
    body = data.index(b"\n", body) + 1
    header = data[:end].decode("ascii", "replace").splitlines()
    fmt = None
    n_vertex = None
    props = []
    in_vertex = False
    seen_vertex = False
    for ln in header:
        tok = ln.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            if tok[1] == "vertex":
                if seen_vertex:
                    raise RuntimeError("PLY: duplicate vertex element")
                n_vertex = int(tok[2])
                in_vertex = True
                seen_vertex = True
            else:
                if not seen_vertex:
                    raise RuntimeError("PLY: vertex element must come first")
                in_vertex = False
        elif tok[0] == "property" and in_vertex:
            if tok[1] == "list":
                raise RuntimeError("PLY: list property in vertex element")
            props.append((tok[2], _PLY_TYPES[tok[1]]))
    if n_vertex is None or not all(k in dict(props) for k in ("x", "y", "z")):
        raise RuntimeError("Failed to read PLY file: " + path)
    if fmt == "binary_little_endian":
        dt = np.dtype([(name, "<" + t) for name, t in props])
        v = np.frombuffer(data, dtype=dt, count=n_vertex, offset=body)
        out = np.stack([v["x"], v["y"], v["z"]], axis=1)
    elif fmt == "ascii":
        rows = data[body:].decode("ascii").split("\n")[:n_vertex]
        arr = np.array([[float(t) for t in r.split()] for r in rows])
        names = [p[0] for p in props]
        out = arr[:, [names.index("x"), names.index("y"), names.index("z")]]
        out = out.astype(np.float32)  # PCL's PointXYZ holds float32 (test_ply.cpp:42)
    else:
        raise RuntimeError("PLY: unsupported format " + str(fmt))
    return np.ascontiguousarray(out, dtype=np.float64)
