#!/usr/bin/env python3
"""bench.py -- coord-samples/sec, forward+backward(+Adam), 4x256 complex WIRE MLP.

    python bench.py --gpus N --steps K --warmup W

N > 1 and no WORLD_SIZE in the environment: this process starts N fresh rank processes itself
(``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...``) BEFORE it
touches the GPU, relays rank 0's JSON line and exits with the job's code.  Under a launcher (WORLD_SIZE set)
it is one rank; it exits non-zero if the process group does not have exactly N ranks.

Workload (BASELINE.json configs[1]): image fit with a
4-hidden-layer WIRE of 256 COMPLEX features per layer (``hidden_features=363``
through the reference's API, modules/wire.py:119), D=2, O=3, omega0=20,
sigma0=30; one step = one full pass of the hot path over a batch of 262 144
coordinates per GPU (a 512 x 512 image per GPU; the N-GPU job fits a
512 x 512N image, coordinate batch sharded contiguously, one RCCL all-reduce of
the 2.1 MB flat gradient per step -> weak scaling): per-rank shard of the epoch's
shuffle (position-keyed bijection, wire_perm_indices -- cost independent of N) ->
on-device coordinates -> forward -> MSE -> backward -> all-reduce -> Adam.
Synthetic data (U[0,1) target), reference init under torch.manual_seed(0), rank 0's
parameters broadcast to every replica.

The JSON line also carries
  roofline     : dominant kernel class (the layer GEMMs) timed live with HIP
                 events on the launch stream (inside the timed region, on every
                 5th step: see prof_step); achieved = algorithmic flops (8
                 flop per complex MAC) / average launch duration.
                 Default path (2 x fp16 split, wire_gemmx2h.hip): every fp32 product
                 is 3 f16 MFMA products, so the ceiling of the algorithm is
                 the dense f16 MFMA peak / 3 = 833.3 fp32-equivalent TFLOP/s
                 (WIRE_SPLIT_F16=0: the 3 x bf16 split of round 2, 6 products, 416.7;
                 WIRE_SPLIT_BF16=0: fp32-MFMA kernels, 157.3).
                 traffic / mfma_busy = (2 FETCH_SIZE + WRITE_SIZE) KiB and SQ_VALU_MFMA_BUSY_CYCLES /
                 (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) of that kernel from profiles/pmc_traffic.json, which
                 tools/pmc_traffic.py writes from rocprofv3 --pmc passes (their own runs) together with the hash
                 of the library sources; null when this tree's sources differ.
  cpu_baseline : the oracle's eager-PyTorch restatement of the reference's CPU
                 path (kind "port"), timed on this host's cores on a bounded
                 sample of the same workload.
  extras       : the exact-fp32 family on the same workload, the reference-API width, forward-only
                 inference, and BASELINE.json configs[3] / [4] (wire2d 1024^2; siren / gauss / relu /
                 relu+posenc at 4x256) with bounded step counts.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 256 flop/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / f16 MFMA (16x the fp32 MFMA rate)
PROF_EVERY = 5                  # per-launch HIP events on every 5th step of a timed region (see prof_step)
HIDDEN_FEATURES = 363           # -> K = int(363/sqrt(2)) = 256 complex features
L, D, O = 4, 2, 3
OMEGA0, SIGMA0 = 20.0, 30.0
SIDE = 512                      # 512 x 512 = 262 144 coordinates per GPU


# ---------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves (no GPU call before this point)
# ---------------------------------------------------------------------------
def spawn_ranks(n: int) -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def csrc_sha() -> str:
    """Hash of the library's sources (wire_amd/csrc, include/): ties profiles/pmc_traffic.json to the code it profiled."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "include", "wire_hip.h")]
    d = os.path.join(ROOT, "wire_amd", "csrc")
    files += sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".h")) or f == "Makefile")
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def family(lib):
    """(split products per fp32 product, dtype string, roofline peak in fp32-equivalent TFLOP/s) of the GEMM family the
    flags select for large batches: 3 = 2 x fp16 split (wire_gemmx2h.hip), 6 = 3 x bf16 split (wire_gemmx3*.hip),
    0 = fp32 MFMA."""
    if lib.wire_tune_get(b"split_bf16") != 1:
        return 0, "f32", PEAK_FP32_MFMA_TFLOPS
    if lib.wire_tune_get(b"split_f16") == 1 and lib.wire_tune_get(b"x3_h16") & 3 == 3:
        return 3, "f32 (2xfp16 split operands with power-of-two scales, f16 MFMA, fp32 accumulate)", PEAK_BF16_MFMA_TFLOPS / 3.0
    return 6, "f32 (3xbf16 split operands, bf16 MFMA, fp32 accumulate)", PEAK_BF16_MFMA_TFLOPS / 6.0


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


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(n_sample: int, iters: int, hidden_features: int = HIDDEN_FEATURES):
    """Reference CPU path (eager PyTorch restatement from oracle/) on a bounded sample:
    1 warm-up + `iters` timed fwd + bwd + Adam steps over the first n_sample coordinates of the same workload,
    and the same number of forward-only passes."""
    import torch
    from oracle import torch_ref, wire_oracle as wo
    cores = host_cores()
    torch.set_num_threads(cores)
    p = torch_ref.init_wire_params(D, hidden_features, L, O, seed=0)
    coords = torch.tensor(wo.image_coords(SIDE, SIDE))[:n_sample][None]
    g = torch.Generator().manual_seed(0)
    target = torch.rand(1, n_sample, O, generator=g)
    torch_ref.train_steps(p, coords, target, L, OMEGA0, OMEGA0, SIGMA0, 5e-3, 1)      # warm-up
    t0 = time.perf_counter()
    torch_ref.train_steps(p, coords, target, L, OMEGA0, OMEGA0, SIGMA0, 5e-3, iters)
    dt = time.perf_counter() - t0
    with torch.no_grad():
        torch_ref.wire_forward(p, coords, L, OMEGA0, OMEGA0, SIGMA0)
        t1 = time.perf_counter()
        for _ in range(iters):
            torch_ref.wire_forward(p, coords, L, OMEGA0, OMEGA0, SIGMA0)
        dtf = time.perf_counter() - t1
    return {"value": n_sample * iters / dt, "unit": "coord-samples/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "forward_only_value": n_sample * iters / dtf,
            "sample": f"{iters} timed fwd+bwd+Adam steps (after 1 warm-up) and {iters} forward-only passes over the "
                      f"first {n_sample} coords of the 512x512 grid, 4x{wo.complex_width(hidden_features)} complex WIRE "
                      f"(hidden_features={hidden_features}), torch {torch.__version__} CPU eager, {cores} threads "
                      f"(oracle/torch_ref.py)"}


def alg_flops(kind, K, Ln, Din, On):
    """SURVEY 8(d) algorithmic flop per sample, fwd + bwd."""
    if kind == "wire":
        return 24 * K * K * Ln + 4 * Din * K + 12 * K * On
    if kind == "wire2d":
        return 48 * K * K * Ln + 8 * Din * K + 12 * K * On
    return 6 * K * K * Ln + 4 * Din * K + 6 * K * On


def alg_hbm_bytes(kind, K, Ln):
    """Algorithmic HBM bytes per sample of a training step of a REAL net (siren / gauss / relu; P = padded width) on this
    dataflow -- every activation written once and read by exactly the kernels that need it:
    first layer 2 P (lin_0, out_0; relu: out_0 only); hidden layer forward P read + lin_l + out_l written (the last layer's out
    is not stored; relu stores no lin); fused final stage P read + P written; weight gradient 2 P; data gradient 2 P read + P
    written (layer 1 writes nothing: its epilogue forms the first layer's sums).  These nets are HBM-bound: this is their
    roofline, the MFMA fraction beside it is for comparison with the wire numbers."""
    if kind == "wire":
        # complex rows of P = roundup(2 K, 64) floats: first layer P (out_0); forward P read + lin + out written (the last
        # layer's out is not stored); final stage 2 P; weight gradient 2 P; data gradient 2 P read + P written, layer 1 reads
        # g_lin only (u, out_0 re-evaluated from the coordinates) and writes nothing (first-layer sums in its epilogue)
        P = (2 * K + 63) // 64 * 64
        floats = P + (2 * Ln * P + (Ln - 1) * P) + 2 * P + Ln * 2 * P + ((Ln - 1) * 3 * P + P)
        return 4 * floats
    if kind not in ("siren", "gauss", "relu"):
        return None
    P = (K + 63) // 64 * 64
    relu = kind == "relu"
    if P == 256 and os.environ.get("WIRE_FUSED_TRAIN", "1") != "0":
        # round 4, the whole-net kernels: the storing forward writes L + 1 rows (r_0 .. r_{L-1}, lin_L;
        # relu: out_0 .. out_L), the chain reads L + 1 (r_l / out_l, g_lin_L) and writes L - 1 (L with positional encoding:
        # not counted), the one weight-gradient launch reads 2 L, the final stage reads 1 and writes 1
        return 4 * (5 * Ln + 3) * P
    floats = (1 if relu else 2) * P                                   # first layer
    floats += Ln * P + (0 if relu else Ln * P) + (Ln if relu else Ln - 1) * P   # forward: reads, lin, out
    floats += 2 * P                                                   # final stage
    floats += Ln * 2 * P                                              # weight gradients
    floats += Ln * 2 * P + (Ln - 1) * P                               # data gradients
    return 4 * floats


def read_prof(lib):
    ms = (C.c_double * 4)()
    cnt = (C.c_int64 * 4)()
    fl = (C.c_double * 4)()
    lib.wire_prof_read(ms, cnt, fl)
    return list(ms), list(cnt), list(fl)


def prof_step(lib, i):
    """Per-launch HIP events (wire_prof_enable) bracket every hot kernel on its launch stream; each event is a barrier
    packet, and ~75 of them per step cost 0.16 ms of an 8.8 ms step when every step is instrumented (measured:
    8.80 ms with, 8.64 ms without).  So the timed region instruments every PROF_EVERY-th step: the per-launch
    averages of `roofline` / `kernel_ms_per_step` come from those steps, `value` from all of them.
    WIRE_BENCH_NO_PROF=1 switches the events off altogether (A/B of their cost)."""
    on = (i % PROF_EVERY == 0) and os.environ.get("WIRE_BENCH_NO_PROF") != "1"
    lib.wire_prof_enable(1 if on else 0)
    return on


def timed_config(dev, lib, kind, side, hf, steps, warmup=2, **kw):
    """fwd + MSE + bwd + Adam throughput of one net kind over a side x side grid (whole grid per step, hashed
    shuffle), with the per-launch time and algorithmic rate of its dominant GEMM class."""
    import torch
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    torch.manual_seed(0)
    model = models.get_INR(nonlin=kind, in_features=D, out_features=O, hidden_features=hf, hidden_layers=L, **kw).to(dev)
    K = model._arch["width"]
    n = side * side
    g = torch.Generator().manual_seed(0)
    tr = FusedTrainer(model, (side, side), torch.rand(n, O, generator=g), lr=5e-3, niters=2000)
    for e in range(warmup):
        tr.step_hashed(e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in range(steps):
        prof_step(lib, e)
        tr.step_hashed(warmup + e)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    lib.wire_prof_enable(0)
    ms, cnt, fl = read_prof(lib)
    Din = D if not kw.get("pos_encode") else model.positional_encoding.out_dim
    F = alg_flops(kind, K, L, Din, O)
    klass = max(range(3), key=lambda i: ms[i])
    avg_ms = ms[klass] / max(1, cnt[klass])
    gemm_tf = fl[klass] / max(1, cnt[klass]) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    _, _, peak = family(lib)
    res = {"samples_per_s": n / dt, "ms_per_step": dt * 1e3, "K": K, "coords_per_step": n, "steps": steps,
           "alg_flop_per_sample": F, "whole_step_tflops": n / dt * F / 1e12,
           "whole_step_frac": n / dt * F / 1e12 / peak,
           "whole_step_frac_of_fp32_mfma_peak": n / dt * F / 1e12 / PEAK_FP32_MFMA_TFLOPS,
           "dominant_gemm": {"class": ["forward", "data gradient", "weight gradient"][klass],
                             "avg_launch_ms": avg_ms, "alg_tflops": gemm_tf, "frac": gemm_tf / peak, "peak": peak}}
    hb = alg_hbm_bytes(kind, K, L)
    if hb is not None:
        res["hbm_bound"] = {"alg_bytes_per_sample": hb, "achieved_TBps": n / dt * hb / 1e12,
                            "frac_of_8TBps_peak": n / dt * hb / 1e12 / 8.0,
                            "note": "algorithmic bytes of this dataflow / step time; a plain 1 : 2 read : write stream reaches "
                                    "5.0 - 5.2 TB/s on this chip (profiles/r03_hbm_stream_probe.txt)"}
    del tr, model
    torch.cuda.empty_cache()
    return res


def reference_loop(dev, wire_kw, epochs=6, device_resident=False):
    import torch
    from torch.optim.lr_scheduler import LambdaLR
    from wire_amd.modules import models
    H = W = SIDE
    maxpoints = H * W
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=D, out_features=O, hidden_features=HIDDEN_FEATURES,
                           hidden_layers=L, **wire_kw)
    model.cuda()
    x = torch.linspace(-1, 1, W)
    y = torch.linspace(-1, 1, H)
    X, Y = torch.meshgrid(x, y, indexing="xy")
    coords = torch.hstack((X.reshape(-1, 1), Y.reshape(-1, 1)))[None, ...]
    if device_resident:
        coords = coords.cuda()
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(H * W, O, generator=g).cuda()[None, ...]
    gt_noisy = gt + 0.01
    optim = torch.optim.Adam(lr=5e-3 * min(1, maxpoints / (H * W)), params=model.parameters())
    scheduler = LambdaLR(optim, lambda e: 0.1 ** min(e / 2000, 1))
    mse_loss_array = torch.zeros(3 * epochs + 2, device="cuda")
    mse_array = torch.zeros(3 * epochs + 2, device="cuda")
    rec = torch.zeros_like(gt)
    t_dev = 0.0
    dt = float("inf")
    for epoch in range(3 * epochs + 2):                   # best of three blocks of `epochs` (the loop is host-paced)
        if epoch >= 2 and (epoch - 2) % epochs == 0:
            torch.cuda.synchronize()
            if epoch > 2:
                dt = min(dt, (time.perf_counter() - t0) / epochs)
            t0 = time.perf_counter()
        indices = torch.randperm(H * W, device="cuda") if device_resident else torch.randperm(H * W)
        for b_idx in range(0, H * W, maxpoints):
            b_indices = indices[b_idx:min(H * W, b_idx + maxpoints)]
            b_coords = coords[:, b_indices, ...].cuda()
            b_indices = b_indices.cuda()
            pixelvalues = model(b_coords)
            with torch.no_grad():
                rec[:, b_indices, :] = pixelvalues
            loss = ((pixelvalues - gt_noisy[:, b_indices, :]) ** 2).mean()
            optim.zero_grad()
            loss.backward()
            optim.step()
        with torch.no_grad():
            if device_resident:                           # the same bookkeeping without host round trips
                mse_loss_array[epoch] = ((gt_noisy - rec) ** 2).mean()
                mse_array[epoch] = ((gt - rec) ** 2).mean()
            else:
                mse_loss_array[epoch] = ((gt_noisy - rec) ** 2).mean().item()
                mse_array[epoch] = ((gt - rec) ** 2).mean().item()
        scheduler.step()
        if not device_resident:
            imrec = rec[0, ...].reshape(H, W, O).detach().cpu().numpy()   # noqa: F841 -- the reference's per-epoch copy
    torch.cuda.synchronize()
    dt = min(dt, (time.perf_counter() - t0) / epochs)
    del model, optim
    torch.cuda.empty_cache()
    note = ("the same calls with coords / randperm on the device and no .item() / D2H per epoch: what the module path "
            "(model(b_coords) + autograd + torch.optim.Adam) itself sustains") if device_resident else \
        ("wire_image_denoise.py:141-178 verbatim (maxpoints = H W) on the headline net through wire_amd.modules + "
         "autograd + torch.optim.Adam, host work of the reference's loop included (CPU randperm and gather, H2D, "
         "two .item(), D2H of rec per epoch)")
    return {"samples_per_s": H * W / dt, "ms_per_epoch": dt * 1e3, "epochs": epochs, "note": note}


def extras(dev, lib):
    """Secondary numbers (not the headline)."""
    import torch
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    res = {}
    wire_kw = dict(first_omega_0=OMEGA0, hidden_omega_0=OMEGA0, scale=SIGMA0)
    # the reference-API width: hidden_features=256 -> K = 181 (padded to 192 on the matrix cores)
    res["k181_api_hidden_features_256"] = timed_config(dev, lib, "wire", SIDE, 256, 16, warmup=3, **wire_kw)
    # forward-only dense inference of the headline net
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=D, out_features=O, hidden_features=HIDDEN_FEATURES,
                           hidden_layers=L, **wire_kw).to(dev)
    K = model._arch["width"]
    tr = FusedTrainer(model, (SIDE, SIDE), torch.zeros(SIDE * SIDE, O), lr=5e-3)
    for _ in range(2):
        tr.render()
    torch.cuda.synchronize()
    dt = float("inf")
    for _ in range(3):                                   # best of 3 x 5 renders (tools/forward_only.py)
        t0 = time.perf_counter()
        for _ in range(5):
            tr.render()
        torch.cuda.synchronize()
        dt = min(dt, (time.perf_counter() - t0) / 5)
    Ff = 8 * K * K * L + 2 * D * K + 4 * K * O
    res["forward_only_k256_literal"] = {"samples_per_s": SIDE * SIDE / dt, "K": K,
                                        "frac_of_fp32_mfma_peak": SIDE * SIDE / dt * Ff / 1e12 / PEAK_FP32_MFMA_TFLOPS}
    del tr, model
    torch.cuda.empty_cache()
    # forward-only of the nets that have the whole-net kernel (wire_fused.hip: activations in registers from the coordinates
    # to the output): the reference-API width and BASELINE.json configs[4]'s real nets
    _, _, peak = family(lib)
    for name, kind, hf, kw in (("forward_only_k181", "wire", 256, wire_kw),
                               ("forward_only_siren", "siren", 256, dict(first_omega_0=30.0, hidden_omega_0=30.0)),
                               ("forward_only_gauss", "gauss", 256, dict(scale=10.0)),
                               ("forward_only_relu", "relu", 256, {})):
        torch.manual_seed(0)
        model = models.get_INR(nonlin=kind, in_features=D, out_features=O, hidden_features=hf, hidden_layers=L, **kw).to(dev)
        Kk = model._arch["width"]
        tr = FusedTrainer(model, (SIDE, SIDE), torch.zeros(SIDE * SIDE, O), lr=5e-3)
        best = {}
        for knob in (1, 0):
            lib.wire_tune_set(b"fused_fwd", knob)
            tr.render()
            torch.cuda.synchronize()
            dt = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(5):
                    tr.render()
                torch.cuda.synchronize()
                dt = min(dt, (time.perf_counter() - t0) / 5)
            best[knob] = dt
        lib.wire_tune_set(b"fused_fwd", 1)
        Ff = (8 * Kk * Kk * L + 2 * D * Kk + 4 * Kk * O) if kind == "wire" else (2 * Kk * Kk * L + 2 * D * Kk + 2 * Kk * O)
        res[name] = {"samples_per_s": SIDE * SIDE / best[1], "ms": best[1] * 1e3, "K": Kk,
                     "layer_by_layer_samples_per_s": SIDE * SIDE / best[0], "layer_by_layer_ms": best[0] * 1e3,
                     "frac": SIDE * SIDE / best[1] * Ff / 1e12 / peak, "alg_flop_per_sample": Ff,
                     "kernel": "fused_fwd_kernel (wire_fused.hip) incl. packing and coordinate launches"}
        del tr, model
        torch.cuda.empty_cache()
    # the exact-fp32 family (fp32 MFMA, 3-multiplication complex GEMMs) on the headline workload: the
    # north-star's literal realisation and the fallback should the split ever be contested
    if lib.wire_tune_get(b"split_bf16") == 1:
        lib.wire_tune_set(b"split_bf16", 0)
        try:
            res["fp32_mfma_family"] = timed_config(dev, lib, "wire", SIDE, HIDDEN_FEATURES, 8, **wire_kw)
            res["fp32_mfma_family"]["dtype"] = "f32 (v_mfma_f32_32x32x2_f32, 3-multiplication complex product)"
        finally:
            lib.wire_tune_set(b"split_bf16", 1)
    # the 3 x bf16 split family (round 2's default: 6 bf16 MFMA products per fp32 product) on the headline workload
    if lib.wire_tune_get(b"split_f16") == 1:
        lib.wire_tune_set(b"split_f16", 0)
        try:
            res["bf16x3_family"] = timed_config(dev, lib, "wire", SIDE, HIDDEN_FEATURES, 8, **wire_kw)
            res["bf16x3_family"]["dtype"] = "f32 (3xbf16 split operands, bf16 MFMA, fp32 accumulate)"
        finally:
            lib.wire_tune_set(b"split_f16", 1)
    # the reference's literal epoch shuffle, torch.randperm(H*W) per step (wire_image_denoise.py:142), instead of the
    # position-keyed bijection of the headline: what the shuffle choice is worth
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=D, out_features=O, hidden_features=HIDDEN_FEATURES,
                           hidden_layers=L, **wire_kw).to(dev)
    g = torch.Generator().manual_seed(0)
    tr = FusedTrainer(model, (SIDE, SIDE), torch.rand(SIDE * SIDE, O, generator=g), lr=5e-3, niters=2000)
    for _ in range(2):
        tr.step(tr.permutation())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step(tr.permutation())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    res["randperm_shuffle"] = {"samples_per_s": SIDE * SIDE / dt, "ms_per_step": dt * 1e3,
                               "note": "headline workload with torch.randperm(H*W) per step (prefetched on a side stream) "
                                       "instead of wire_perm_indices"}
    del tr, model
    torch.cuda.empty_cache()
    # the reference's loop UNCHANGED (wire_image_denoise.py:141-178 with maxpoints = H W: one minibatch per epoch) through the
    # drop-in modules -- model(b_coords), autograd backward (wire_mlp_fwd / wire_mlp_bwd), torch.optim.Adam, and the
    # reference's own host work per epoch: CPU randperm, CPU gather of the coordinates + H2D, two .item() syncs, D2H of rec
    res["reference_loop_unchanged"] = reference_loop(dev, wire_kw)
    res["reference_loop_device_resident"] = reference_loop(dev, wire_kw, device_resident=True)
    # BASELINE.json configs[3] and [4] through the same trainer (bounded steps)
    res["cfg4_wire2d_4x256_1024x1024"] = timed_config(dev, lib, "wire2d", 1024, 256, 4, first_omega_0=10.0,
                                                      hidden_omega_0=10.0, scale=10.0)
    # (24 steps after 4 warm-up ones: at 2 ms a step the 8-step windows of round 3 read 0.1 - 0.2 ms above a longer run)
    res["cfg5_siren_4x256"] = timed_config(dev, lib, "siren", SIDE, 256, 24, warmup=4, first_omega_0=30.0, hidden_omega_0=30.0)
    res["cfg5_gauss_4x256"] = timed_config(dev, lib, "gauss", SIDE, 256, 24, warmup=4, scale=10.0)
    res["cfg5_relu_4x256"] = timed_config(dev, lib, "relu", SIDE, 256, 24, warmup=4)
    res["cfg5_relu_posenc_4x256"] = timed_config(dev, lib, "relu", SIDE, 256, 24, warmup=4, pos_encode=True, sidelength=512)
    return res


def secondary(ex):
    """Flat, driver-visible copy of the secondary claims: samples/s and the whole-step fraction of
    the family's roofline (833.3 fp32-equivalent TFLOP/s for the 2 x fp16 split; SURVEY 8(d) flop counts) for the other
    BASELINE.json configs, the reference-API width, forward-only inference and the exact-fp32 family."""
    out = {}
    for key, short in (("k181_api_hidden_features_256", "k181"), ("cfg4_wire2d_4x256_1024x1024", "cfg4_wire2d"),
                       ("cfg5_siren_4x256", "cfg5_siren"), ("cfg5_gauss_4x256", "cfg5_gauss"),
                       ("cfg5_relu_4x256", "cfg5_relu"), ("cfg5_relu_posenc_4x256", "cfg5_relu_posenc"),
                       ("fp32_mfma_family", "fp32_mfma_family"), ("bf16x3_family", "bf16x3_family")):
        if key in ex:
            out[short + "_samples_per_s"] = ex[key]["samples_per_s"]
            out[short + "_ms_per_step"] = ex[key]["ms_per_step"]
            out[short + "_whole_step_frac"] = ex[key]["whole_step_frac"]
            if "hbm_bound" in ex[key]:
                out[short + "_alg_TBps"] = ex[key]["hbm_bound"]["achieved_TBps"]
    for key in ("forward_only_k256_literal", "forward_only_k181", "forward_only_siren", "forward_only_gauss",
                "forward_only_relu"):
        if key in ex:
            out[key + "_samples_per_s"] = ex[key]["samples_per_s"]
            if "frac" in ex[key]:
                out[key + "_frac"] = ex[key]["frac"]
    for key in ("reference_loop_unchanged", "reference_loop_device_resident", "randperm_shuffle"):
        if key in ex:
            out[key + "_samples_per_s"] = ex[key]["samples_per_s"]
    return out


def rccl_probe(dev, tr, timeout_s=45.0):
    """What RCCL ITSELF reports about the job: ncclCommCount / ncclCommCuDevice of a communicator
    built from librccl's C ABI (the trainer's own when WIRE_DP_DIRECT=1, else one made here with the symmetric set-up of
    parallel.RcclDirect) and one all-reduce of ones on it, which must come back as the world size.  Runs on every rank
    AFTER the timed region, in a thread with a deadline: a communicator that cannot be built must not cost the run its
    number.  Returns a dict for config.rccl_*; "timeout" marks a probe that did not finish (the caller then leaves through
    os._exit so that a stuck rendezvous cannot hang the teardown)."""
    import threading
    import torch
    import torch.distributed as dist
    from wire_amd.parallel import RcclDirect
    res = {}

    def work():
        try:
            comm = tr.reducers[0].direct
            own = comm is None
            if own:
                comm = RcclDirect(dev)
            res["rccl_nranks"] = comm.comm_count()
            res["rccl_device"] = comm.comm_device()
            ones = torch.ones(4, device=dev, dtype=torch.float32)
            comm.all_reduce_sum_(ones)
            torch.cuda.synchronize(dev)
            res["rccl_allreduce_of_ones"] = float(ones[0].item())
            res["rccl_communicator"] = "the trainer's (WIRE_DP_DIRECT=1)" if not own else "probe only (training used torch.distributed.all_reduce)"
            if own:
                comm.close()
        except Exception as e:                              # noqa: BLE001
            res["rccl_error"] = f"{type(e).__name__}: {e}"

    if dist.get_backend() != "nccl":
        return {"rccl_nranks": None, "rccl_note": f"backend {dist.get_backend()}: no RCCL communicator in this run"}
    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        return {"rccl_nranks": None, "rccl_error": "timeout", "timeout": True}
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
    ap.add_argument("--overlap", choices=["none", "layer", "micro"], default=None,
                    help="shape of the gradient exchange at N > 1 (VERDICT r03 item 6b): none = one all-reduce of the whole "
                         "buffer after the backward (default), layer = per layer on a side stream under the backward "
                         "(WIRE_DP_OVERLAP=layer), micro = two micro-shards, the first one's all-reduce under the second "
                         "one's forward + backward (--micro-shards 2)")
    ap.add_argument("--dp-direct", type=int, choices=[0, 1], default=None,
                    help="1: ncclAllReduce of librccl's C ABI on the compute stream (parallel.RcclDirect); 0 (default): "
                         "torch.distributed.all_reduce of the process group")
    ap.add_argument("--shuffle", choices=["hashed", "randperm"], default="hashed",
                    help="hashed: per-rank slice of a position-keyed bijection (default); randperm: torch.randperm "
                         "of the whole grid on every rank, the reference's literal call")
    args = ap.parse_args()

    if args.overlap == "layer":
        os.environ["WIRE_DP_OVERLAP"] = "layer"
    elif args.overlap is not None:
        os.environ["WIRE_DP_OVERLAP"] = "none"
    if args.overlap == "micro" and args.micro_shards == 1:
        args.micro_shards = 2
    if args.dp_direct is not None:
        os.environ["WIRE_DP_DIRECT"] = str(args.dp_direct)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))          # nothing above has touched the GPU

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"error: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    backend = os.environ.get("WIRE_BENCH_BACKEND", "nccl")
    # one process per GPU.  (Rehearsal on a 1-GPU box: WIRE_BENCH_BACKEND=gloo lets several ranks
    # share cuda:0 -- RCCL itself refuses two ranks on one device.)
    ndev = max(1, torch.cuda.device_count())
    if world > ndev and backend == "nccl":
        if rank == 0:
            print(f"error: {world} ranks need {world} GPUs, {ndev} visible (WIRE_BENCH_BACKEND=gloo rehearses "
                  f"several ranks on one card)", file=sys.stderr)
        sys.exit(3)
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    if world > 1 or os.environ.get("WIRE_DP_FORCE", "0") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            print(f"error: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)

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
    epoch = [0]

    def one_step():
        if args.shuffle == "hashed":
            loss = tr.step_hashed(epoch[0])             # this rank's slice of the epoch's shuffle, on the device
        else:
            loss = tr.step(tr.permutation())            # torch.randperm(H*W) per epoch (wire_image_denoise.py:142)
        epoch[0] += 1
        tr.scheduler_step()
        return loss

    for _ in range(args.warmup):
        one_step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    n_prof = 0
    for i in range(args.steps):
        n_prof += prof_step(lib, i)
        loss = one_step()
    fence()
    dt = time.perf_counter() - t0
    lib.wire_prof_enable(0)
    ms, cnt, fl = read_prof(lib)
    if world > 1:
        tmax = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    final_loss = float(loss.item())
    rccl = None
    if tr.reducers[0].active:
        rccl = rccl_probe(dev, tr)

    if rank == 0:
        n_gpu_batch = npts // world
        F = alg_flops("wire", K, L, D, O)               # SURVEY 8(d) algorithmic flop / sample
        value = npts * args.steps / dt
        nprod, dtype_name, peak = family(lib)
        split = nprod != 0
        if nprod == 3:
            names = ["gemmx2h_nt<gabor_fwd> (layer forward, 2xfp16 split, 16x16x32 f16 MFMA)",
                     "gemmx2h_nt<gabor_bwd> (data gradient, 2xfp16 split, 16x16x32 f16 MFMA)",
                     ("gemmx2_tn16 (weight gradient, 2xfp16 split, 16x16x32 f16 MFMA)" if (2 * K + 63) // 64 * 64 in (384, 448)
                      or (2 * K) % 256 == 0 else "gemmx3_tn (weight gradient, 3xbf16 split)"), "other"]
        elif split:
            h16 = lib.wire_tune_get(b"x3_h16")
            names = [("gemmx3h_nt<gabor_fwd> (layer forward, 16x16x32 MFMA)" if h16 & 1 else "gemmx3_nt<gabor_fwd> (layer forward)"),
                     ("gemmx3h_nt<gabor_bwd> (data gradient, 16x16x32 MFMA)" if h16 & 2 else "gemmx3_nt<gabor_bwd> (data gradient)"),
                     ("gemmx3_tn16 (weight gradient, 16x16x32 MFMA)" if lib.wire_tune_get(b"x3_tn16") == 1 and (2 * K) % 256 == 0
                      else "gemmx3_tn (weight gradient)"), "other"]
        else:
            names = ["gemm3m_nt<gabor_fwd> (layer forward)", "gemm3m_nt<gabor_bwd> (data gradient)",
                     "gemm3m_tn (weight gradient)", "other"]
        alg_per_launch = 8.0 * K * K * (n_gpu_batch / max(1, args.micro_shards))   # per hidden-layer GEMM
        klass = max(range(3), key=lambda i: ms[i])
        avg_ms = ms[klass] / max(1, cnt[klass])
        achieved = alg_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic = None
        hbm = None
        mfma_busy = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        pmc_note = "profiles/pmc_traffic.json absent"
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # written by tools/pmc_traffic.py from the rocprofv3 --pmc passes of tools/profile_round.sh; only valid for the
                # library sources it profiled and for the default flags
                default_flags = nprod == 3 and args.hidden_features == HIDDEN_FEATURES
                if tj.get("csrc_sha") != csrc_sha() or not default_flags:
                    raise ValueError(f"pmc_traffic.json is for sources {tj.get('csrc_sha')}, this tree is {csrc_sha()} "
                                     f"(or non-default flags): traffic / mfma_busy not reported")
                fam = tj.get("traffic", {})
                traffic = fam.get(str(klass))
                mfma_busy = tj.get("mfma_busy", {}).get(str(klass))
                pmc_note = f"rocprofv3 --pmc passes of sources {tj.get('csrc_sha')}: {', '.join(tj.get('kernels', {}).get(str(klass), []))}"
                # HBM rate of the GEMMs from the PMC byte counts per launch (committed profile) and the launch
                # times measured in THIS run: evidence of fusion quality, not the bound
                if all(str(i) in fam and cnt[i] > 0 for i in range(3)) and args.micro_shards == 1 and world == 1:
                    gb = sum(fam[str(i)] * cnt[i] for i in range(3))
                    gms = sum(ms[i] for i in range(3))
                    hbm = {"gemm_bytes_per_step": gb / max(1, n_prof), "gemm_tb_per_s": gb / (gms * 1e-3) / 1e12,
                           "frac_of_8_tb_per_s": gb / (gms * 1e-3) / 8e12,
                           "source": "profiles/pmc_traffic.json (rocprofv3 FETCH_SIZE / WRITE_SIZE) x launches / GEMM time of this run"}
            except Exception as e:
                traffic, mfma_busy, hbm, pmc_note = None, None, None, str(e)
        out = {
            "metric": "coord-samples/sec fwd+bwd, 4x256 complex WIRE MLP",
            "value": value, "unit": "coord-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_name,
            "data": "synthetic",
            "config": {"workload": f"512x{512 * world} image fit, WIRE 4 hidden x {K} complex "
                                   f"(hidden_features={args.hidden_features}), D=2 O=3 omega0=20 sigma0=30, "
                                   f"batch=262144 coords/GPU, fwd+MSE+bwd+Adam per step",
                       "global_batch": npts, "parallelism": f"dp{world}", "micro_shards": args.micro_shards,
                       "shuffle": args.shuffle, "backend": backend if world > 1 else None,
                       "grad_allreduce": ("per layer, side stream, under the backward" if tr.overlap else
                                          ("per micro-shard, side stream, under the next micro-shard" if args.micro_shards > 1
                                           else "whole buffer after the backward")) if tr.reducers[0].active else None,
                       "overlap": args.overlap or os.environ.get("WIRE_DP_OVERLAP", "none"),
                       "dp_direct": int(tr.reducers[0].direct is not None),
                       "rccl_nranks": (rccl or {}).get("rccl_nranks"), "rccl_device": (rccl or {}).get("rccl_device"),
                       "rccl": {k: v for k, v in (rccl or {}).items() if k not in ("rccl_nranks", "rccl_device", "timeout")}},
            "roofline": {"bound": "mfma", "kernel": names[klass], "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "mfma_busy": mfma_busy, "pmc_source": pmc_note,
                         "avg_launch_ms": avg_ms, "launches": int(cnt[klass]),
                         "hbm_view": {"alg_bytes_per_step": alg_hbm_bytes("wire", K, L) * n_gpu_batch,
                                      "achieved_TBps": alg_hbm_bytes("wire", K, L) * n_gpu_batch / (dt / args.steps) / 1e12,
                                      "note": "the step's algorithmic HBM bytes (every activation written once, read by the "
                                              "kernels that need it) / step time; a plain 1 : 2 read : write stream reaches "
                                              "5.0 - 5.2 TB/s on this chip: the GEMM launches are as much HBM- as MFMA-bound"},
                         "alg_flops_per_launch": alg_per_launch,
                         "peak_note": (f"dense f16 / bf16 MFMA peak 2500 TFLOP/s / {nprod} partial products per fp32 product; "
                                       f"executed MFMA rate = {nprod} x achieved; against the 3xbf16 family's 416.7 this is "
                                       f"{achieved / (PEAK_BF16_MFMA_TFLOPS / 6.0):.3f}") if split else "fp32 MFMA peak",
                         "executed_mfma_tflops": achieved * (float(nprod) if split else 0.75),
                         "frac_of_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS},
            "whole_step_tflops": value / world * F / 1e12,
            "whole_step_frac_of_fp32_mfma_peak": value / world * F / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "kernel_ms_per_step": {names[i]: ms[i] / max(1, n_prof) for i in range(4)},
            "instrumented_steps": n_prof,
            "final_loss": final_loss,
        }
        if hbm is not None:
            out["hbm"] = hbm
        # order of the slow legs: a CPU leg, the GPU extras, the remaining CPU legs -- GPU work in the MIDDLE of the run,
        # so that an outside activity sampler meets it (the extras used to be 5 % at the very start)
        if world == 1 and not args.no_cpu_baseline:
            # N = 65 536 and 262 144, 4 x 256 and 4 x 181 complex (bounded: 3 + 1 iterations each)
            out["cpu_baseline"] = cpu_baseline(65536, 3)
            out["cpu_baseline"]["more"] = {"n65536_k181_api_hidden_features_256": cpu_baseline(65536, 3, 256)}
        if world == 1 and not args.no_extras:
            out["extras"] = extras(dev, lib)
            out["secondary"] = secondary(out["extras"])
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"]["more"]["n262144_k256"] = cpu_baseline(262144, 2)
            out["cpu_baseline"]["more"]["n262144_k181_api_hidden_features_256"] = cpu_baseline(262144, 2, 256)
        print(json.dumps(out), flush=True)
        if rccl is not None and rccl.get("rccl_nranks") not in (None, world):
            print(f"error: RCCL reports {rccl.get('rccl_nranks')} ranks, --gpus {world}", file=sys.stderr)
            sys.stdout.flush()
            os._exit(4)
    if rccl is not None and rccl.get("timeout"):
        sys.stdout.flush()
        os._exit(0)                                     # a stuck probe rendezvous must not hang the teardown
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
