"""RCCL on this box, once: a world of one rank over the nccl backend — init, two all-reduces, a barrier, and the halo messages of
distributed.HaloExchange sent GPU to GPU through RCCL with both neighbours mapped onto the rank itself
(tests/gpu_rccl_self_worker.py).  Prints one JSON line; keep it under profiles/rNN/nccl_sanity.json.

    python scripts/nccl_sanity.py > gpurun_out/nccl_sanity.json
"""
import os
import runpy
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.argv = [os.path.join(ROOT, "tests", "gpu_rccl_self_worker.py")]
runpy.run_path(sys.argv[0], run_name="__main__")
