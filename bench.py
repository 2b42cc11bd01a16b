#!/usr/bin/env python3
"""Headline benchmark: images/sec of the RAJNI token-pruning forward, ViT-B/16 @224, bf16,
README 4-stage schedule, batch 256 per GPU (BASELINE.json configs[1]; configs[2] when --gpus 8).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Without a launcher (`WORLD_SIZE` unset) and N > 1 this process never creates a GPU context or launches work: it counts
the visible devices (exit 3 when fewer than N - there is no silent one-rank run), starts the N ranks as a fresh child
process through `torch.distributed.run`, relays rank 0's JSON line and exits with the launcher's status.

A "step" is one forward of one 256-image synthetic batch per GPU, images already resident in HBM.
`value` is the reference's metric exactly as rajni/eval.py:44-59,74 defines it: it comes out of
`rajni_amd.evaluate_model` (per batch: sync -> model(images) -> sync, summed; SUM of images / MAX of seconds over
the ranks), with every profiling hook off.  The K back-to-back forwards behind one sync (host enqueue hidden) are
timed in a second region and reported as `pipelined_images_per_sec`; that second region is where the HIP-event
kernel timing behind `roofline` runs.
One JSON line on stdout (rank 0).  Extra objects:
  roofline     - dominant kernel (the packed-token MFMA GEMM), timed live with HIP events on the
                 launch stream inside the timed region (rajni_profile_* hooks of the C ABI);
  cpu_baseline - the oracle's torch-CPU flavour (oracle/rajni_oracle_torch.py, a port of the reference algorithm
                 on the ATen kernels the reference itself would run) timed on this box's host cores on a bounded
                 sample of the same workload (rank 0, N=1).
  other_configs - (N=1) BASELINE.json configs[3] (ViT-L/16 @384, batch 64, keep .7/.5/.3 at blocks 4/12/20) and configs[4]
                 (DeiT-3-B dims, batch 512, fp8_mfma; fp32 and bf16 residual stream), each through evaluate_model with
                 its own roofline and unpruned-base speed-up.
  pruned_torch_images_per_sec / speedup_vs_pruned_torch - the reference's own op graph (attention.py:17-60,
                 model.py:50-59: the torch-flavour oracle with ref_op_graph=True) on stock PyTorch-ROCm kernels on the
                 same GPU, same weights / schedule / batch: the reference-equivalent GPU denominator next to the
                 unpruned one.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch

README_SCHEDULE = {3: {"keep_ratio": 0.88, "update": True}, 4: {"keep_ratio": 0.88, "update": True},
                   7: {"keep_ratio": 0.80, "update": True}, 8: {"keep_ratio": 0.72, "update": True}}
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_FP8_TFLOPS = 5000.0    # dense fp8 MFMA (same table; never the 2:1-sparsity figure)
PEAK_HBM_GBS = 8000.0
L384_SCHEDULE = {4: {"keep_ratio": 0.7, "update": True}, 12: {"keep_ratio": 0.5, "update": True},
                 20: {"keep_ratio": 0.3, "update": True}}          # BASELINE.json configs[3]
GEMM_MASK = 0b0111 | (1 << 12) | (0b1111 << 13)   # kernel classes: qkv/head, fc1, fc2 (K > N), proj (K <= N), fp8 x fp8 twins (13..16)


def flops_per_image(cfg, schedule):
    from rajni_amd.wrapper.model import plan_token_counts
    from rajni_amd.ops import keep_count
    C, Hd = cfg.embed_dim, cfg.hidden_dim
    counts = plan_token_counts(cfg.num_patches + 1, cfg.depth, schedule)
    total = 2.0 * cfg.num_patches * (cfg.in_chans * cfg.patch_size ** 2) * C + 2.0 * C * cfg.num_classes
    for i, n in enumerate(counts):
        npk = keep_count(schedule[i]["keep_ratio"], n) + 1 if i in schedule else n
        total += 2.0 * n * C * 3 * C            # qkv on all N tokens
        total += 4.0 * npk * npk * C            # QK^T and PV on kept tokens
        total += 2.0 * npk * C * C              # proj
        total += 4.0 * npk * C * Hd             # fc1 + fc2
    return total, counts


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU
    box exposes 256 logical CPUs but grants one GPU's share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(cfg, schedule, seconds_budget=20.0):
    """The oracle's torch flavour (oracle/rajni_oracle_torch.py: the reference algorithm restated op for op on ATen CPU
    kernels, pinned by the reference's fixtures) on this box's host cores, fp32, same model dims and schedule."""
    from oracle import rajni_oracle_torch as ort
    from rajni_amd import timm_shaped as ts
    cores = usable_cores()
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        sd = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in ts.synth_state_dict(cfg, seed=0).items()}
        bsz = 16
        imgs = torch.randn(bsz, 3, cfg.img_size, cfg.img_size, generator=torch.Generator().manual_seed(1234))
        run = lambda: ort.vit_forward(sd, imgs, schedule, depth=cfg.depth, num_heads=cfg.num_heads, ln_eps=cfg.ln_eps)
        run()  # warm-up
        n, t0 = 0, time.time()
        while True:
            run()
            n += bsz
            dt = time.time() - t0
            if dt > seconds_budget or n >= 4096:
                break
    finally:
        torch.set_num_threads(prev)
    return {"value": round(n / dt, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} synthetic 3x{cfg.img_size}x{cfg.img_size} images in batches of {bsz}, "
                      f"same model dims and schedule, fp32 torch-CPU oracle (ATen/oneDNN kernels), {cores} threads "
                      f"(usable cores of {os.cpu_count()} logical), {dt:.1f} s"}


def reference_agreement(dev, name="base224_agree256", chunk=64, weight_format="model", residual="fp32"):
    """"top-1 delta vs the reference wrapper" with resolution.  There are no trained weights or labels here, so the
    statement is agreement with the REFERENCE'S OWN OUTPUTS on a committed fixture (tests/golden/make_golden.py
    agreement_case): 256 seeded images, ViT-B/16 dims, README schedule, seeded weights; the reference's fp32 CPU
    logits and per-stage keep_idx.  Reported: argmax agreement (eval.py:61-64 counts exactly that) and max |dlogit|,
    absolute and relative to max |logit|, (a) with the reference's selections injected (selection-conditional parity,
    SURVEY 4-3c) and (b) free-running (the device ranks its own bf16 scores); next to them the same two numbers for
    the reference's own bf16 CPU run of the same images - the yardstick for what bf16 costs on this model."""
    try:
        import rajni_amd
        from rajni_amd import timm_shaped as ts
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import load_case, case_images, pruned_blocks
        meta, data = load_case(name)
        fx = ts.create_model(ts.CONFIGS[meta["cfg_name"]], seed=meta["seed"], std=meta["std"],
                             bias_std=meta["bias_std"], round_bf16=True)
        fw = rajni_amd.RAJNIViTWrapper(fx, meta["schedule"]).to(dev).to(torch.bfloat16).eval()
        fw.set_weight_format(weight_format)
        if residual == "bf16":
            fw.set_residual_dtype(torch.bfloat16)
        images = torch.from_numpy(case_images(meta, data))
        ref = data["logits"]
        n = ref.shape[0]
        scale = float(np.abs(ref).max())

        def run(forced):
            outs = []
            for c in range(0, n, chunk):
                fw.force_keep_idx({i: torch.from_numpy(data[f"blk{i}.keep_idx"][c:c + chunk].astype(np.int32)).to(dev)
                                   for i in pruned_blocks(meta)} if forced else None)
                outs.append(fw(images[c:c + chunk].to(dev)).float().cpu().numpy())
            fw.force_keep_idx(None)
            return np.concatenate(outs)

        def cmp(got):
            d = float(np.abs(got - ref).max())
            return {"top1_agree": int((got.argmax(1) == ref.argmax(1)).sum()), "max_abs_dlogit": round(d, 5),
                    "max_abs_dlogit_over_logit_scale": round(d / scale, 5)}

        inj, free = run(True), run(False)
        out = {"fixture": f"tests/golden/{name} (reference fp32 CPU run, ViT-B/16 dims, README schedule, seeded weights)",
               "build_options": {"weight_format": weight_format, "residual_stream": residual},
               "images": int(n), "logit_scale": round(scale, 4),
               "median_top2_margin": round(float(meta.get("median_top2_margin", float("nan"))), 4),
               "injected_selections": cmp(inj), "free_running": cmp(free),
               "reference_own_bf16_run": {"top1_agree": int(meta["ref_bf16_top1_agree"]),
                                          "max_abs_dlogit": round(float(meta["ref_bf16_max_abs_dlogit"]), 5),
                                          "max_abs_dlogit_over_logit_scale": round(float(meta["ref_bf16_max_abs_dlogit"]) / scale, 5)},
               "token_counts_equal": fw.get_last_stats()["token_counts"] == data["token_counts"].tolist()}
        return out
    except Exception as e:   # the bench line must not die on the side check
        return {"error": repr(e)[:300]}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--model", default="vit_base_patch16_224")
    ap.add_argument("--weight-format", default="model", choices=["model", "fp8", "fp8_mfma"],
                    help='"fp8": block Linear weights as e4m3 + per-row scale feeding the bf16 MFMA; "fp8_mfma": e4m3 '
                         'weights AND per-row-scaled e4m3 activations on the fp8 MFMA (BASELINE configs[4]); the headline '
                         'metric is quoted on "model" (bf16 weights)')
    ap.add_argument("--schedule", default=None,
                    help="JSON text or file {block: {keep_ratio, update}} (default: the README 4-stage schedule); "
                         'BASELINE configs[3] is --model vit_large_patch16_384 --batch 64 --schedule \'{"4":{"keep_ratio":0.7},'
                         '"12":{"keep_ratio":0.5},"20":{"keep_ratio":0.3}}\'')
    ap.add_argument("--residual", default="fp32", choices=["fp32", "bf16"],
                    help="residual stream precision between blocks (RAJNIViTWrapper.set_residual_dtype): fp32 (default, the "
                         "build's accuracy choice) or bf16 (what the reference's own bf16 model keeps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-torch-baseline", action="store_true",
                    help="skip every side measurement on the GPU (agreement, opt-ins, torch baselines, other configs): profiling runs")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE configs[3] / configs[4] (other_configs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args(argv)


def visible_gpus():
    """Number of ROCm devices this process could use.  `torch.cuda.device_count()` may initialise the HIP runtime to
    count them (hipGetDeviceCount) but creates no context and launches nothing; the ranks run in a fresh child process
    (subprocess, never exec), so the parent never drives a GPU."""
    return int(torch.cuda.device_count())


def spawn_command(n, argv, port):
    """The launcher line the driver itself uses for N > 1 (one rank per GPU over RCCL)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher: start the N ranks and relay rank 0's line.  Runs BEFORE any GPU
    call in this process.  Fewer than N visible devices is an error (exit 3), never a smaller run."""
    import socket
    import subprocess
    one_device = os.environ.get("RAJNI_BENCH_ONE_DEVICE") == "1"    # rehearsal on a 1-GPU box, see worker()
    need = 1 if one_device else args.gpus
    have = visible_gpus()
    if have < need:
        print(f"bench.py: --gpus {args.gpus} needs {need} visible ROCm device(s), found {have}; refusing to run "
              "a smaller job in its place", file=sys.stderr, flush=True)
        return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(spawn_command(args.gpus, argv, port), env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:                                 # ranks other than 0 print nothing on stdout
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr, flush=True)
        rc = 4
    return rc


def latest_profile(suffix):
    """(path, record) of the newest committed `profiles/rNN_<x>_<suffix>` of the bf16 default command - or (None, None)
    when there is none, when it does not say which kernel sources it was taken on, or when those are not the sources of
    this tree (`csrc_fingerprint`, tools/srchash.py): PMC counters cannot be read from inside this process (they come
    from rocprofv3 passes over THIS command, tools/final_profile.sh), so a stale file must drop out rather than be
    reported as if measured."""
    import glob
    from srchash import csrc_fingerprint
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*" + suffix))
                   if "_fp8_" not in os.path.basename(f))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            rec = json.load(f)
    except Exception:
        return None, None
    if rec.get("csrc_fingerprint") != csrc_fingerprint():
        return None, None
    return files[-1], rec


class ResidentLoader:
    """K (images, labels) batches already in HBM - the same resident batch K times (keep counts are data independent,
    SURVEY Q1, so content does not affect timing)."""

    def __init__(self, images, labels, n):
        self.images, self.labels, self.n = images, labels, n

    def __len__(self):
        return self.n

    def __iter__(self):
        for _ in range(self.n):
            yield self.images, self.labels


def gemm_roofline(prof):
    """`roofline` object from the HIP-event records of the GEMM classes: the dominant instantiation (most time), priced
    against the MFMA roof of the pipe it runs on - except the attention projection (K <= N), whose 2 x M x N x 4 bytes of
    fp32 residual stream make it HBM bound (AI ~ 150 flop/B < the ~310 ridge)."""
    if not prof:
        return None, None
    name, rec = max(prof.items(), key=lambda kv: kv[1]["ms"])
    avg_ms = rec["ms"] / rec["launches"]
    achieved = rec["flops"] / rec["launches"] / (avg_ms * 1e-3) / 1e12
    hbm_bound = "K<=N" in name
    kpeak = PEAK_FP8_TFLOPS if name.startswith("gemm_f8") else PEAK_BF16_TFLOPS    # the pipe THIS kernel runs on
    gbps = rec["bytes"] / rec["launches"] / (avg_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm" if hbm_bound else "mfma", "kernel": name,
                "achieved": round(gbps, 1) if hbm_bound else round(achieved, 1),
                "peak": PEAK_HBM_GBS if hbm_bound else kpeak,
                "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": round(gbps / PEAK_HBM_GBS, 4) if hbm_bound else round(achieved / kpeak, 4),
                "traffic": None,
                "avg_launch_us": round(avg_ms * 1e3, 2), "launches": rec["launches"],
                "all_gemm": {k: {"launches": v["launches"], "avg_us": round(v["ms"] / v["launches"] * 1e3, 2),
                                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                                 # algorithmic bytes (operands + output + residual stream) per second
                                 "algorithmic_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 0)}
                             for k, v in prof.items()}}
    return name, roofline


def metric_run(m, images, labels, steps, dev):
    """The metric as the reference defines it (eval.py:44-59,74) through the product's own harness: per batch
    sync -> forward -> sync; under torch.distributed SUM(images) / MAX(seconds) over the ranks."""
    import contextlib
    import rajni_amd
    with contextlib.redirect_stdout(sys.stderr):      # its "Warming up" line must not land on the result stream
        _, ips = rajni_amd.evaluate_model(m, ResidentLoader(images, labels, steps), device=dev,
                                          max_batches=steps, warmup=0)
    return ips


class PrunedTorch(torch.nn.Module):
    """The reference's op graph on stock PyTorch-ROCm kernels: RAJNIViTWrapper.forward / RAJNIAttention.forward /
    compute_importance as the torch-flavour oracle restates them (ref_op_graph=True: topk + sort, explicit
    softmax-matmul attention with the [B,H,N,N] matrix materialised, three gathers, LayerNorm over all tokens) over a
    state dict resident on the device in the model dtype.  A baseline leg: never the product."""

    def __init__(self, sd, schedule, cfg):
        super().__init__()
        self.sd, self.schedule, self.cfg = sd, schedule, cfg

    def forward(self, x):
        from oracle import rajni_oracle_torch as ort
        return ort.vit_forward(self.sd, x, self.schedule, depth=self.cfg.depth, num_heads=self.cfg.num_heads,
                               ln_eps=self.cfg.ln_eps, ref_op_graph=True)[0]


def torch_baselines(cfg, schedule, images, labels, dev, steps, pruned=True):
    """img/s of (a) the unpruned timm-shaped base and (b) the reference's pruned op graph, both on stock PyTorch-ROCm
    ops in bf16 on this GPU through the same evaluate_model."""
    from rajni_amd import timm_shaped as ts
    base = ts.create_model(cfg, seed=0).to(torch.bfloat16).to(dev).eval()
    n = max(3, min(steps, 10))
    with torch.no_grad():
        for _ in range(3):
            base(images)
    out = {"unpruned": metric_run(base, images, labels, n, dev)}
    if pruned:
        sd = {k: v.detach() for k, v in base.state_dict().items()}
        pt = PrunedTorch(sd, schedule, cfg)
        for _ in range(3):
            pt(images)
        out["pruned"] = metric_run(pt, images, labels, n, dev)
    del base
    torch.cuda.empty_cache()
    return out


def side_config(dev, label, model_name, batch, schedule, weight_format, residual, steps):
    """One of the other single-GPU BASELINE configs, measured like the headline (evaluate_model, hooks off; then a
    HIP-event pass for its own roofline) plus its unpruned stock-PyTorch base."""
    import rajni_amd
    from rajni_amd import timm_shaped as ts, _native as nat
    try:
        cfg = ts.CONFIGS[model_name]
        model = ts.create_model(cfg, seed=0).to(torch.bfloat16).to(dev)
        w = rajni_amd.RAJNIViTWrapper(model, schedule).eval()
        w.set_weight_format(weight_format)
        if residual == "bf16":
            w.set_residual_dtype(torch.bfloat16)
        gen = torch.Generator(device=dev).manual_seed(4321)
        images = torch.randn(batch, 3, cfg.img_size, cfg.img_size, generator=gen, device=dev).to(torch.bfloat16)
        labels = torch.randint(0, cfg.num_classes, (batch,), generator=gen, device=dev)
        nat.profile_enable(0)
        for _ in range(3):
            w(images)
        torch.cuda.synchronize(dev)
        value = metric_run(w, images, labels, steps, dev)
        counts = w.get_last_stats()["token_counts"]
        nat.profile_reset()
        nat.profile_enable(GEMM_MASK)
        for _ in range(max(2, steps // 2)):
            w(images)
        torch.cuda.synchronize(dev)
        nat.profile_enable(0)
        _, roofline = gemm_roofline(nat.profile_collect())
        nat.profile_reset()
        fl_img, _ = flops_per_image(cfg, schedule)
        peak = PEAK_FP8_TFLOPS if weight_format == "fp8_mfma" else PEAK_BF16_TFLOPS
        del w, model
        torch.cuda.empty_cache()
        base = torch_baselines(cfg, schedule, images, labels, dev, steps, pruned=False)["unpruned"]
        return {"baseline_config": label,
                "workload": f"{model_name}, batch {batch}, schedule " + json.dumps({k: v["keep_ratio"] for k, v in schedule.items()})
                            + f", weight format {weight_format}, residual stream {residual}, synthetic randn 3x{cfg.img_size}x{cfg.img_size}, random-init weights",
                "value": round(value, 1), "unit": "images/sec", "ms_per_step": round(batch / value * 1e3, 3), "steps": steps,
                "dtype": "fp8" if weight_format == "fp8_mfma" else "bf16", "token_counts": counts,
                "model_tflops": round(value * fl_img / 1e12, 1), "model_mfma_frac": round(value * fl_img / 1e12 / peak, 4),
                "roofline": roofline,
                "unpruned_torch_images_per_sec": round(base, 1), "speedup_vs_unpruned_torch": round(value / base, 3)}
    except Exception as e:      # a side figure must not take the headline line down with it
        return {"baseline_config": label, "error": repr(e)[:300]}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)
    return worker(args)


def worker(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report one as the other")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (MI355X); the HIP path has no CPU fallback")
    # host threads of a rank, stated rather than inherited: the forward is one enqueueing thread; OMP / ATen pools of 8
    # ranks on one host must not oversubscribe it (the launcher parent sets OMP_NUM_THREADS=4 when nothing else did)
    host_threads = int(os.environ.get("OMP_NUM_THREADS", "4"))
    torch.set_num_threads(host_threads)
    # rehearsal hooks for a 1-GPU box: RAJNI_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # RAJNI_BENCH_BACKEND=gloo joins them over CPU tensors (RCCL refuses two ranks on one GPU)
    one_device = os.environ.get("RAJNI_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but only {torch.cuda.device_count()} device(s) are visible")
    backend = os.environ.get("RAJNI_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if torch.cuda.current_device() != local_rank:
        raise SystemExit(f"bench.py: rank {rank}: current device is {torch.cuda.current_device()}, expected LOCAL_RANK {local_rank}")
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    dist = None
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        one = torch.ones(1, dtype=torch.int64, device=red_dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)        # RCCL over xGMI when backend == "nccl"
        ranks_seen = int(one.item())
        if ranks_seen != world:
            raise SystemExit(f"bench.py: all_reduce saw {ranks_seen} ranks, expected {world}")

    import rajni_amd
    from rajni_amd import timm_shaped as ts, _native as nat

    cfg = ts.CONFIGS[args.model]
    schedule = README_SCHEDULE
    if args.schedule:
        text = open(args.schedule).read() if os.path.exists(args.schedule) else args.schedule
        schedule = {int(k): {"keep_ratio": float(v["keep_ratio"]), "update": bool(v.get("update", True))}
                    for k, v in json.loads(text).items()}
    B = args.batch
    model = ts.create_model(cfg, seed=0).to(torch.bfloat16).to(dev)
    wrapped = rajni_amd.RAJNIViTWrapper(model, schedule).eval()
    wrapped.set_weight_format(args.weight_format)
    if args.residual == "bf16":
        wrapped.set_residual_dtype(torch.bfloat16)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn(B, 3, cfg.img_size, cfg.img_size, generator=gen, device=dev).to(torch.bfloat16)
    labels = torch.randint(0, cfg.num_classes, (B,), generator=gen, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- headline: W untimed forwards, barrier + sync, K timed steps through evaluate_model, barrier + sync
    nat.profile_enable(0)
    for _ in range(args.warmup):
        wrapped(images)
    barrier()
    w0 = time.perf_counter()
    value = metric_run(wrapped, images, labels, args.steps, dev)   # already the whole-job figure (all-reduced inside)
    barrier()
    wall = time.perf_counter() - w0
    per_rank = list(rajni_amd.evaluate_model.last_stats["per_rank"])   # every rank's own (images, seconds), same all-reduce
    counts = wrapped.get_last_stats()["token_counts"]

    # ---- second region: K back-to-back forwards behind one sync, HIP-event timing of the GEMM classes on
    # the packed-token GEMM classes: qkv/head, fc1, fc2 (K > N), proj (K <= N), and their fp8 x fp8 twins (13-15)
    nat.profile_reset()
    nat.profile_enable(GEMM_MASK)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wrapped(images)
    barrier()
    elapsed = time.perf_counter() - t0
    nat.profile_enable(0)
    if dist is not None:
        t = torch.tensor([elapsed, wall], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, wall = (float(v) for v in t.tolist())
    prof = nat.profile_collect()
    # the HBM-bound part of the path (score -> top-k -> compact, the keep_idx gather fused into the attention
    # loads, LayerNorm): HIP-event timing of a few extra forwards
    hbm_prof = {}
    if world == 1:
        nat.profile_reset()
        nat.profile_enable((1 << 4) | (1 << 5) | (1 << 6))
        for _ in range(5):
            wrapped(images)
        torch.cuda.synchronize(dev)
        nat.profile_enable(0)
        hbm_prof = nat.profile_collect()
        nat.profile_reset()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return 0

    fl_img, _ = flops_per_image(cfg, schedule)
    fp8_mfma = args.weight_format == "fp8_mfma"
    peak_tflops = PEAK_FP8_TFLOPS if fp8_mfma else PEAK_BF16_TFLOPS
    # BASELINE.json's metric is quoted on this workload; other --model/--schedule/--batch runs are labelled as such
    headline_workload = (args.model == "vit_base_patch16_224" and not args.schedule and args.batch == 256
                         and args.weight_format == "model" and args.residual == "fp32")
    name, roofline = gemm_roofline(prof)
    if roofline is not None:
        rel = lambda pth: os.path.relpath(pth, ROOT) if pth else None
        roofline["committed_profile"] = None
        if headline_workload:      # the PMC profiles are taken over the default command only
            # traffic / mfma_busy_frac are NOT measured in this run: they are read from the newest committed PMC profile
            # of this command - and only when that profile was taken on the kernel sources of this tree
            tsrc, trec = latest_profile("hbm_traffic_pmc.json")
            msrc, mrec = latest_profile("mfma_pmc.json")
            try:
                roofline["traffic"] = trec["by_bench_class"][name]["hbm_bytes_per_launch"] if trec else None
            except Exception:
                roofline["traffic"] = None
            try:
                roofline["mfma_busy_frac"] = mrec["by_bench_class"][name]["mfma_busy_frac"] if mrec else None
            except Exception:
                roofline["mfma_busy_frac"] = None
            roofline["committed_profile"] = {
                "traffic_source": (rel(tsrc) + " (rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in separate passes; coverage "
                                   f"{trec.get('coverage')})") if roofline["traffic"] else None,
                "mfma_busy_source": (rel(msrc) + " (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; coverage "
                                     f"{mrec.get('coverage')})") if roofline.get("mfma_busy_frac") is not None else None,
                "note": "counters come from the committed profile of this command, not from this run; dropped (null) when "
                        "the profile's csrc_fingerprint is not this tree's"}

    hbm_kernels = {k: {"launches": v["launches"], "avg_us": round(v["ms"] / v["launches"] * 1e3, 2),
                       "algorithmic_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 0),
                       "frac_of_hbm_peak": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
                   for k, v in hbm_prof.items() if v["launches"]}
    wf = {"model": "", "fp8": " activations, fp8 e4m3 block weights (bf16 MFMA)",
          "fp8_mfma": " stream, fp8 e4m3 block weights and per-row-scaled e4m3 activations (fp8 MFMA)"}[args.weight_format]
    try:
        from srchash import csrc_fingerprint
        fingerprint = csrc_fingerprint()
    except Exception:
        fingerprint = None
    rates = [n / max(sec, 1e-9) for n, sec in per_rank]
    out = {"metric": "images/sec ViT-B/16@224 with README schedule" if headline_workload
                     else f"images/sec {args.model} (not the BASELINE workload: see config.workload)",
           "value": round(value, 1), "unit": "images/sec",
           "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(world * B / value * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "fp8" if fp8_mfma else "bf16", "data": "synthetic",
           "timing": "rajni_amd.evaluate_model: sum over K batches of (sync -> forward -> sync), reference eval.py:51-59,74; "
                     "MAX over ranks; profiling hooks off",
           "wall_ms_per_step": round(wall / args.steps * 1e3, 3),
           "pipelined_images_per_sec": round(world * B * args.steps / elapsed, 1),
           "pipelined_ms_per_step": round(elapsed / args.steps * 1e3, 3),
           # every rank's own rate out of evaluate_model's one all-reduce: which rank set the node time, and the spread
           "per_rank_images_per_sec": {"list": [round(r, 1) for r in rates], "min": round(min(rates), 1),
                                       "max": round(max(rates), 1)},
           "slowest_rank": int(min(range(len(rates)), key=lambda r: rates[r])),
           "host_threads_per_rank": host_threads,
           "config": {"workload": f"{args.model} bf16{wf}, "
                                  f"batch {B}/GPU, {'README 4-stage schedule {3:.88,4:.88,7:.80,8:.72}' if not args.schedule else 'schedule ' + json.dumps({k: v['keep_ratio'] for k, v in schedule.items()})}, "
                                  f"synthetic randn 3x{cfg.img_size}x{cfg.img_size}, "
                                  "random-init weights (seed 0)",
                      "global_batch": world * B, "token_counts": counts, "parallelism": f"dp{world}",
                      "residual_stream": args.residual,
                      "dims": {"C": cfg.embed_dim, "hidden": cfg.hidden_dim, "depth": cfg.depth, "heads": cfg.num_heads,
                               "classes": cfg.num_classes, "batch_per_gpu": B},
                      "csrc_fingerprint": fingerprint,
                      "collective": (f"{backend} all_reduce (SUM) of [correct,total,images] + one (images, seconds) slot per rank, "
                                     "once per run; node seconds = MAX over the slots" if world > 1 else None)},
           "model_tflops": round(value * fl_img / 1e12, 1),
           "model_mfma_frac": round(value * fl_img / 1e12 / (peak_tflops * world), 4),
           "roofline": roofline}

    side = world == 1 and not args.no_torch_baseline   # (side measurements are skipped together: profiling runs - their
    # forwards of other shapes would skew per-kernel averages)
    vitb_workload = args.model == "vit_base_patch16_224" and not args.schedule
    if vitb_workload and side:   # for the opt-in formats it prices their numerics contract
        out["reference_agreement"] = reference_agreement(dev, weight_format=args.weight_format, residual=args.residual)
    if hbm_kernels:
        # algorithmic bytes: score+select reads the K and V thirds of qkv once ((2NC + C) x 2 B per image) and writes
        # indices/scores; attention reads the kept q, k, v rows through keep_idx and writes the output (the
        # reference's separate gather copies do not exist); LayerNorm reads the fp32 stream and writes bf16
        out["hbm_kernels"] = hbm_kernels
    if side:
        # opt-in shortcut, NOT part of `value`: the last block computed for the CLS row only (the head reads
        # nothing else; same logits - tests/test_gpu_forward.py::test_cls_only_last_block_*).  `value` above is
        # the row-for-row forward, the reference's op graph.
        wrapped.set_last_block_cls_only(True)
        for _ in range(3):
            wrapped(images)
        out["cls_only_last_block_images_per_sec"] = round(metric_run(wrapped, images, labels, args.steps, dev), 1)
        wrapped.set_last_block_cls_only(False)
        if args.residual == "fp32":
            # second opt-in, NOT part of `value` either: the residual stream kept in bf16 between blocks, as the
            # reference's own bf16 model keeps it (costs ~1e-2 of the logit scale against the fp32 reference, DESIGN.md
            # section 2; `value` runs the fp32 stream)
            wrapped.set_residual_dtype(torch.bfloat16)
            for _ in range(3):
                wrapped(images)
            out["bf16_residual_stream_images_per_sec"] = round(metric_run(wrapped, images, labels, args.steps, dev), 1)
            wrapped.set_residual_dtype(torch.float32)
        # the two GPU denominators, same batch / weights / harness, stock PyTorch-ROCm ops in bf16:
        #   unpruned base (the "4x" of BASELINE.json's north_star) and the reference's own pruned op graph
        try:
            tb = torch_baselines(cfg, schedule, images, labels, dev, args.steps)
            out["unpruned_torch_images_per_sec"] = round(tb["unpruned"], 1)
            out["speedup_vs_unpruned_torch"] = round(value / tb["unpruned"], 3)
            out["pruned_torch_images_per_sec"] = round(tb["pruned"], 1)
            out["speedup_vs_pruned_torch"] = round(value / tb["pruned"], 3)
        except Exception as e:
            out["torch_baselines_error"] = repr(e)[:300]
    if side and headline_workload and not args.no_other_configs:
        del wrapped, model
        torch.cuda.empty_cache()
        ksteps = max(4, min(args.steps, 10))
        out["other_configs"] = [
            side_config(dev, "BASELINE.json configs[3]", "vit_large_patch16_384", 64, L384_SCHEDULE, "model", "fp32", ksteps),
            side_config(dev, "BASELINE.json configs[4]", "deit3_base_patch16_224", 512, README_SCHEDULE, "fp8_mfma", "fp32", ksteps),
            side_config(dev, "BASELINE.json configs[4], bf16 residual stream", "deit3_base_patch16_224", 512, README_SCHEDULE,
                        "fp8_mfma", "bf16", ksteps)]
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, schedule, args.cpu_seconds)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
