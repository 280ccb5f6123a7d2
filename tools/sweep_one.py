import sys; sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import bench_sweep as b
kind = sys.argv[1]
if kind == "wire2d": b.run("wire2d", 1024, 256, first_omega_0=10.0, hidden_omega_0=10.0, scale=10.0, steps=4)
else: b.run(kind, 512, 256, first_omega_0=30.0, hidden_omega_0=30.0, steps=4)
