import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gpu_voxel_raytracer_amd import host as H
from oracle import oracle as O
rng = np.random.default_rng(1)
for fn, lo, hi in [("sin", -50, 50), ("cos", -50, 50), ("tan", -1.5, 1.5), ("exp", -110, 90), ("log", 0, 1e6), ("sqrt", 0, 1e6), ("chain", -3, 3)]:
    x = rng.uniform(lo, hi, 1 << 20).astype(np.float32)
    y = rng.uniform(-2, 2, 1 << 20).astype(np.float32)
    dev = H.detmath_probe(fn, x, y)
    cpu = O.detmath(fn, x, y) if fn != "chain" else None
    if cpu is None:
        print(fn, "dev sample", dev[:4]); continue
    bad = np.flatnonzero(dev.view(np.uint32) != cpu.view(np.uint32))
    print(fn, "mismatches", len(bad))
    for i in bad[:8]:
        print("   x=%r (%08x) dev=%r (%08x) cpu=%r (%08x)" % (x[i], x[i:i+1].view(np.uint32)[0], dev[i], dev[i:i+1].view(np.uint32)[0], cpu[i], cpu[i:i+1].view(np.uint32)[0]))
