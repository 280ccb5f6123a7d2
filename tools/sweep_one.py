"""One config of tools/bench_sweep.py (for rocprofv3):  python3 tools/sweep_one.py wire2d|siren|gauss|relu|posenc"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sweep as b

kind = sys.argv[1]
if kind == "wire2d":
    b.run("wire2d", 1024, 256, first_omega_0=10.0, hidden_omega_0=10.0, scale=10.0, steps=4)
elif kind == "posenc":
    b.run("relu", 512, 256, pos_encode=True, sidelength=512, steps=4)
else:
    b.run(kind, 512, 256, first_omega_0=30.0, hidden_omega_0=30.0, steps=4)
