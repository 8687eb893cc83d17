"""Test tool (tests/test_gpu_fuzz.py): random stress of the 3D grid converge — f32 filter records + lean sorted-domain kernel (default)
against the f64 records (ICPX_GRID_FILTER=0) and against unseeded queries (ICPX_GRID_SEED=0): every result must be
bit-identical.  Shapes: uniform, clustered, planar, duplicated, anisotropic, far from the origin, tiny extents,
sources sticking out of the targets' box.  usage: grid_filter_stress.py [cases [seed]]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import icp_amd
torch.cuda.is_available()
lib = icp_amd._lib.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)

def rot(rng, amp):
    a = rng.normal(size=3); a /= np.linalg.norm(a); th = rng.uniform(-amp, amp)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K

def cloud(rng, kind, m):
    if kind == "uniform": t = rng.uniform(-10, 10, (m, 3))
    elif kind == "clustered":
        c = rng.uniform(-10, 10, (40, 3)); t = c[rng.integers(0, 40, m)] + rng.normal(0, 0.15, (m, 3))
    elif kind == "planar":
        t = rng.uniform(-10, 10, (m, 3)); t[:, 2] = 0.3 * np.sin(t[:, 0]) + rng.normal(0, 0.01, m)
    elif kind == "dups":
        b = rng.uniform(-5, 5, (m // 3 + 1, 3)); t = np.repeat(b, 3, axis=0)[:m]
    elif kind == "aniso": t = rng.uniform(-1, 1, (m, 3)) * np.array([100.0, 1.0, 0.01])
    elif kind == "lattice": t = rng.integers(0, 40, (m, 3)).astype(float)
    else: raise ValueError(kind)
    return t

def run(method, s, t, it, md):
    cfg = icp_amd.Config(); cfg.set("nn", "grid"); cfg.set("max_distance", md)
    d = icp_amd.ICPDriver(icp_amd.ICP3.from_method(method, cfg, device=0))
    d.set_max_iterations(it); d.set_transform_tolerance(1e-9, 1e-9)
    r = d.converge(s, t)
    nb = d.icp_.debug_neighbors()
    return r.iteration_count, r.transform.copy(), np.array(nb[0]), np.array(nb[1])

bad = 0
for c in range(cases):
    kind = ["uniform", "clustered", "planar", "dups", "aniso", "lattice"][c % 6]
    m = int(rng.integers(3000, 150000)); n = int(rng.integers(3000, 150000))
    t = cloud(rng, kind, m)
    s = t[rng.integers(0, m, n)] + rng.normal(0, 0.02 * (0.01 if kind == "aniso" else 1.0), (n, 3))
    s = s @ rot(rng, 0.1).T + rng.normal(0, 0.2, 3) * (0.01 if kind == "aniso" else 1.0)
    if c % 5 == 1: s = s * 1.3                       # sticks out of the targets' box
    off = [0.0, 7.0e5, -3.0e3][c % 3]; sc = [1.0, 1.0, 1e-5][c % 3]
    s, t = s * sc + off, t * sc + off
    method = ["vanilla", "trimmed"][c % 2]
    md = float(np.sqrt(0.35 * sc)) if method == "trimmed" else 1.0
    outs = []
    for filt, seed in (("1", "1"), ("0", "1"), ("1", "0")):
        lib.icpx_debug_knob(b"ICPX_GRID_FILTER", filt.encode()); lib.icpx_debug_knob(b"ICPX_GRID_SEED", seed.encode())
        outs.append(run(method, s, t, 6, md))
    lib.icpx_debug_knob(b"ICPX_GRID_FILTER", None); lib.icpx_debug_knob(b"ICPX_GRID_SEED", None)
    ok = all(o[0] == outs[0][0] and np.array_equal(o[1], outs[0][1], equal_nan=True) and np.array_equal(o[2], outs[0][2]) and
             np.array_equal(o[3], outs[0][3], equal_nan=True) for o in outs[1:])
    bad += 0 if ok else 1
    print(f"case {c:2d} {kind:9s} {method:7s} n={n:6d} m={m:6d} off={off:g} scale={sc:g}: {'identical' if ok else 'DIFFERENT'}", flush=True)
print("all identical" if bad == 0 else f"{bad} cases differ")
sys.exit(1 if bad else 0)
