#!/usr/bin/env python3
"""bench.py -- coord-samples/sec, forward+backward(+Adam), 4x256 complex WIRE MLP.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): image fit with a
4-hidden-layer WIRE of 256 COMPLEX features per layer (``hidden_features=363``
through the reference's API, modules/wire.py:119), D=2, O=3, omega0=20,
sigma0=30; one step = one full pass of the hot path over a batch of 262 144
coordinates per GPU (a 512 x 512 image per GPU; the N-GPU job fits a
512 x 512N image, coordinate batch sharded contiguously, one RCCL all-reduce of
the 2.1 MB flat gradient per step -> weak scaling): device-side randperm ->
on-device coordinates -> forward -> MSE -> backward -> all-reduce -> Adam.
Synthetic data (U[0,1) target), reference init under torch.manual_seed(0).

The JSON line also carries
  roofline     : dominant kernel class (the layer GEMMs) timed live with HIP
                 events on the launch stream; achieved = algorithmic flops (8
                 flop per complex MAC) / average launch duration.
                 Default path (split-bf16, wire_gemmx3.hip): every fp32 product
                 is 6 bf16 MFMA products, so the ceiling of the algorithm is
                 the dense bf16 MFMA peak / 6 = 416.7 fp32-equivalent TFLOP/s;
                 with WIRE_SPLIT_BF16=0 (fp32-MFMA kernels) it is 157.3.
  cpu_baseline : the oracle's eager-PyTorch restatement of the reference's CPU
                 path (kind "port"), timed on this host's cores on a bounded
                 sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 256 flop/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (16x the fp32 MFMA rate)
HIDDEN_FEATURES = 363           # -> K = int(363/sqrt(2)) = 256 complex features
L, D, O = 4, 2, 3
OMEGA0, SIGMA0 = 20.0, 30.0
SIDE = 512                      # 512 x 512 = 262 144 coordinates per GPU


def host_cores() -> int:
    """CPU cores this process may actually use: min(affinity, cgroup quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(n_sample: int, iters: int):
    """Reference CPU path (eager PyTorch restatement from oracle/) on a bounded
    sample: fwd + bwd + Adam over n_sample coordinates of the same workload."""
    from oracle import torch_ref, wire_oracle as wo
    cores = host_cores()
    torch.set_num_threads(cores)
    p = torch_ref.init_wire_params(D, HIDDEN_FEATURES, L, O, seed=0)
    coords = torch.tensor(wo.image_coords(SIDE, SIDE))[:n_sample][None]
    g = torch.Generator().manual_seed(0)
    target = torch.rand(1, n_sample, O, generator=g)
    torch_ref.train_steps(p, coords, target, L, OMEGA0, OMEGA0, SIGMA0, 5e-3, 1)      # warm-up
    t0 = time.perf_counter()
    torch_ref.train_steps(p, coords, target, L, OMEGA0, OMEGA0, SIGMA0, 5e-3, iters)
    dt = time.perf_counter() - t0
    return {"value": n_sample * iters / dt, "unit": "coord-samples/s", "cores": cores, "kind": "port",
            "sample": f"{iters} fwd+bwd+Adam steps over the first {n_sample} coords of the 512x512 grid, "
                      f"4x256 complex WIRE, torch {torch.__version__} CPU eager (oracle/torch_ref.py)"}


def extras(dev, lib):
    """Secondary numbers (not the headline): the same net through the reference's API width
    (hidden_features=256 -> K=181, padded to 192 on the matrix cores) and forward-only inference."""
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    res = {}
    g = torch.Generator().manual_seed(0)
    target = torch.rand(SIDE * SIDE, O, generator=g)
    for tag, hf in (("k181_api_hidden_features_256", 256), ("k256_literal", HIDDEN_FEATURES)):
        torch.manual_seed(0)
        model = models.get_INR(nonlin="wire", in_features=D, out_features=O, hidden_features=hf,
                               hidden_layers=L, first_omega_0=OMEGA0, hidden_omega_0=OMEGA0, scale=SIGMA0).to(dev)
        K = model._arch["width"]
        tr = FusedTrainer(model, (SIDE, SIDE), target, lr=5e-3, niters=2000)
        if hf == 256:
            for _ in range(2):
                tr.step(torch.randperm(SIDE * SIDE, device=dev))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                tr.step(torch.randperm(SIDE * SIDE, device=dev))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            F = 24 * K * K * L + 4 * D * K + 12 * K * O
            res[tag] = {"train_samples_per_s": SIDE * SIDE / dt, "K": K,
                        "frac_of_fp32_mfma_peak": SIDE * SIDE / dt * F / 1e12 / PEAK_FP32_MFMA_TFLOPS}
        else:
            tr.render()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                tr.render()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            Ff = 8 * K * K * L + 2 * D * K + 4 * K * O
            res["forward_only_" + tag] = {"samples_per_s": SIDE * SIDE / dt, "K": K,
                                          "frac_of_fp32_mfma_peak": SIDE * SIDE / dt * Ff / 1e12 / PEAK_FP32_MFMA_TFLOPS}
        del tr, model
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--micro-shards", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--hidden-features", type=int, default=HIDDEN_FEATURES)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU.  (Rehearsal on a 1-GPU box: WIRE_BENCH_BACKEND=gloo lets several ranks
    # share cuda:0 -- RCCL itself refuses two ranks on one device.)
    ndev = max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    backend = os.environ.get("WIRE_BENCH_BACKEND", "nccl")
    if world > 1 or os.environ.get("WIRE_DP_FORCE", "0") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    from wire_amd import _lib
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    lib = _lib.lib()

    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=D, out_features=O,
                           hidden_features=args.hidden_features, hidden_layers=L,
                           first_omega_0=OMEGA0, hidden_omega_0=OMEGA0, scale=SIGMA0).to(dev)
    K = model._arch["width"]
    H, W = SIDE, SIDE * world
    npts = H * W
    g = torch.Generator().manual_seed(0)
    target = torch.rand(npts, O, generator=g)
    tr = FusedTrainer(model, (H, W), target, lr=5e-3, niters=2000, micro_shards=args.micro_shards)

    def one_step():
        idx = tr.permutation()                          # torch.randperm per epoch (wire_image_denoise.py:142),
        loss = tr.step(idx)                             # generated on a side stream under the previous step
        tr.scheduler_step()
        return loss

    for _ in range(args.warmup):
        one_step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    lib.wire_prof_enable(1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    fence()
    dt = time.perf_counter() - t0
    lib.wire_prof_enable(0)
    ms = (C.c_double * 4)()
    cnt = (C.c_int64 * 4)()
    fl = (C.c_double * 4)()
    lib.wire_prof_read(ms, cnt, fl)
    if world > 1:
        tmax = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    final_loss = float(loss.item())

    if rank == 0:
        n_gpu_batch = npts // world
        F = 24 * K * K * L + 4 * D * K + 12 * K * O      # SURVEY 8(d) algorithmic flop / sample
        value = npts * args.steps / dt
        split = lib.wire_tune_get(b"split_bf16") == 1
        if split:
            names = ["gemmx3_nt<gabor_fwd> (layer forward)", "gemmx3_nt<gabor_bwd> (data gradient)",
                     "gemmx3_tn (weight gradient)", "other"]
            peak = PEAK_BF16_MFMA_TFLOPS / 6.0          # 6 bf16 partial products per fp32 product
        else:
            names = ["gemm3m_nt<gabor_fwd> (layer forward)", "gemm3m_nt<gabor_bwd> (data gradient)",
                     "gemm3m_tn (weight gradient)", "other"]
            peak = PEAK_FP32_MFMA_TFLOPS
        alg_per_launch = 8.0 * K * K * (n_gpu_batch / max(1, args.micro_shards))   # per hidden-layer GEMM
        klass = max(range(3), key=lambda i: ms[i])
        avg_ms = ms[klass] / max(1, cnt[klass])
        achieved = alg_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic = None
        hbm = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                fam = tj.get("split_bf16", {}) if split else tj.get("fp32_mfma", tj)
                traffic = fam.get(str(klass))
                # HBM rate of the GEMMs from the PMC byte counts per launch (committed profile) and the launch
                # times measured in THIS run: evidence of fusion quality, not the bound
                if all(str(i) in fam and cnt[i] > 0 for i in range(3)) and args.micro_shards == 1 and world == 1:
                    gb = sum(fam[str(i)] * cnt[i] for i in range(3))
                    gms = sum(ms[i] for i in range(3))
                    hbm = {"gemm_bytes_per_step": gb / args.steps, "gemm_tb_per_s": gb / (gms * 1e-3) / 1e12,
                           "frac_of_8_tb_per_s": gb / (gms * 1e-3) / 8e12,
                           "source": "profiles/pmc_traffic.json (rocprofv3 FETCH_SIZE / WRITE_SIZE) x launches / GEMM time of this run"}
            except Exception:
                traffic = None
        out = {
            "metric": "coord-samples/sec fwd+bwd, 4x256 complex WIRE MLP",
            "value": value, "unit": "coord-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (3xbf16 split operands, bf16 MFMA, fp32 accumulate)" if split else "f32",
            "data": "synthetic",
            "config": {"workload": f"512x{512 * world} image fit, WIRE 4 hidden x {K} complex "
                                   f"(hidden_features={args.hidden_features}), D=2 O=3 omega0=20 sigma0=30, "
                                   f"batch=262144 coords/GPU, fwd+MSE+bwd+Adam per step",
                       "global_batch": npts, "parallelism": f"dp{world}", "micro_shards": args.micro_shards},
            "roofline": {"bound": "mfma", "kernel": names[klass], "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "avg_launch_ms": avg_ms, "launches": int(cnt[klass]),
                         "alg_flops_per_launch": alg_per_launch,
                         "peak_note": ("dense bf16 MFMA peak 2500 TFLOP/s / 6 partial products per fp32 product; "
                                       "executed bf16 rate = 6 x achieved") if split else "fp32 MFMA peak",
                         "executed_mfma_tflops": achieved * (6.0 if split else 0.75),
                         "frac_of_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS},
            "whole_step_tflops": value / world * F / 1e12,
            "whole_step_frac_of_fp32_mfma_peak": value / world * F / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "kernel_ms_per_step": {names[i]: ms[i] / args.steps for i in range(4)},
            "final_loss": final_loss,
        }
        if hbm is not None:
            out["hbm"] = hbm
        if world == 1 and not args.no_extras:
            out["extras"] = extras(dev, lib)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(32768, 2)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
