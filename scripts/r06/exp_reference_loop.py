import json, sys
sys.path.insert(0, '.')
import bench
from gpu_voxel_raytracer_amd import ALL, TIMED, Camera, Context, scenes
print(json.dumps(bench.measure_reference_loop(Context, Camera, (ALL, TIMED), scenes, 0), indent=1))
