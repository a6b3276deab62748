#!/bin/bash
# usage: scripts/ab_build.sh <tag> [extra hipcc flags...]  -> gpu_voxel_raytracer_amd/libvxrt_<tag>.so  (select with VXRT_LIB=...)
# A/B builds of the library for same-box comparisons (boxes differ by a few per cent, so A and B must run in one gpurun call).
set -e
tag=$1; shift
cd "$(dirname "$0")/.."
python - "$tag" "$@" <<'PY'
import sys
from gpu_voxel_raytracer_amd import _build
tag, flags = sys.argv[1], sys.argv[2:]
variants = "-DVXRT_VARIANTS=1" in flags          # the variants' sources come with the flag
flags = [f for f in flags if f != "-DVXRT_VARIANTS=1"]
_build.build(extra_flags=flags, variants=variants, out=_build.LIB.replace("libvxrt.so", f"libvxrt_{tag}.so"))
print("built", tag, flags)
PY
