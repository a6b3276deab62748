"""ORACLE-BASED DIAGNOSTIC (not collected by pytest): device vs host bit equality (signs of zero included) on special values."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from gpu_voxel_raytracer_amd import host as H
from oracle import oracle as O

sp = np.array([0.0, -0.0, 0.25, 0.5, 0.75, 0.125, 1 - 2.0 ** -24, 1.0, -1.0, 2.0, -2.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40], np.float32)
x, y = [a.ravel() for a in np.meshgrid(sp, sp)]
for fn in ("hemi_y", "hemi_z", "mul", "sub", "flip", "sin", "cos", "sqrt", "div"):
    d, o = H.detmath_probe(fn, x, y), O.detmath(fn, x, y)
    bad = (d.view(np.uint32) != o.view(np.uint32)) & ~(np.isnan(d) & np.isnan(o))
    print(fn, "differing", int(bad.sum()), "of", x.size)
    for i in np.argwhere(bad).ravel()[:8]:
        print(f"   x {x[i]!r} y {y[i]!r}: device {d[i]!r} ({d[i:i+1].view(np.uint32)[0]:#x}) host {o[i]!r} ({o[i:i+1].view(np.uint32)[0]:#x})")
