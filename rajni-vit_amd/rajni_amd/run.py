"""`python -m rajni_amd.run` - evaluation CLI with the reference's flags (`rajni/run.py:17-43`).

Same flags and printed report as the reference CLI; differences:
  * the JSON schedule's string keys are normalised, so `--schedule schedule.json` really prunes
    (the reference's never does: SURVEY B1), and the device is synchronised (B2);
  * `--data_path` is optional: without it (or with `--synthetic`) batches are seeded `randn` images
    generated on the device (no dataset or network exists in the build/bench images);
  * models come from timm when it is importable (`timm.create_model(name, pretrained=...)`),
    otherwise from the timm-shaped stand-in with seeded weights; `--weights file.safetensors|.pt`
    loads a local timm-format state dict (torch.load with weights_only=True); without `--weights` or
    `--pretrained` the weights are random and the report says so (the reference always downloads pretrained ones);
  * under `torchrun --nproc-per-node N` every rank evaluates its shard of the batches and the
    counters are all-reduced (rajni_amd.evaluate_model).
An ImageFolder loader is built only if torchvision is installed.
"""
from __future__ import annotations

import argparse
import json
import os

import torch

from . import RAJNIViTWrapper, evaluate_model
from . import timm_shaped as ts

README_SCHEDULE = {3: {"keep_ratio": 0.88, "update": True}, 4: {"keep_ratio": 0.88, "update": True},
                   7: {"keep_ratio": 0.80, "update": True}, 8: {"keep_ratio": 0.72, "update": True}}


def get_args(argv=None):
    p = argparse.ArgumentParser("RAJNI-ViT evaluation (MI355X)")
    p.add_argument("--data_path", type=str, default=None, help="ImageNet val folder (needs torchvision); omit for synthetic")
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--model", type=str, default="vit_base_patch16_224")
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--schedule", type=str, default=None, help="JSON pruning schedule (default: the README schedule)")
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--max_batches", type=int, default=None)
    p.add_argument("--compare_base", action="store_true")
    p.add_argument("--synthetic", action="store_true", help="seeded randn batches generated on the device")
    p.add_argument("--dtype", type=str, default="bfloat16", choices=["bfloat16", "float32"])
    p.add_argument("--weights", type=str, default=None, help="local timm-format state dict (.safetensors / .pt)")
    p.add_argument("--pretrained", action="store_true", help="timm pretrained weights (needs timm + network)")
    p.add_argument("--seed", type=int, default=0)
    # build-specific opt-ins (none of them exists in the reference CLI; defaults = the reference-faithful path)
    p.add_argument("--weight_format", type=str, default="model", choices=["model", "fp8", "fp8_mfma"],
                   help="RAJNIViTWrapper.set_weight_format: e4m3 block weights (fp8) / plus e4m3 activations on the fp8 matrix pipe (fp8_mfma)")
    p.add_argument("--residual", type=str, default="float32", choices=["float32", "bfloat16"],
                   help="residual stream precision between blocks (RAJNIViTWrapper.set_residual_dtype)")
    return p.parse_args(argv)


def load_schedule(path):
    if path is None:
        return dict(README_SCHEDULE)
    with open(path) as f:
        raw = json.load(f)
    return {int(k): v for k, v in raw.items()}


def create_base(args):
    try:
        import timm  # noqa: F401
        model = timm.create_model(args.model, pretrained=args.pretrained)
        source = "timm"
    except ImportError:
        if args.model not in ts.CONFIGS:
            raise SystemExit(f"timm is not installed and '{args.model}' is not a built-in config {sorted(ts.CONFIGS)}")
        model = ts.create_model(args.model, seed=args.seed)
        source = "timm-shaped stand-in, seeded random weights"
    if args.weights:
        if args.weights.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(args.weights)
        else:
            sd = torch.load(args.weights, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd and "cls_token" not in sd:
            sd = sd["state_dict"]          # a timm training checkpoint wraps the weights
        model.load_state_dict(sd, strict=True)
        source += f" + {os.path.basename(args.weights)}"
    elif not (args.pretrained and source == "timm"):
        source += " [WARNING: random weights - accuracy is meaningless; pass --weights FILE or --pretrained]"
    return model.eval(), source


class InputCast(torch.nn.Module):
    """Stock base model behind a cast of the images to its dtype: an ImageFolder batch is fp32 and
    `evaluate_model` only moves it to the device (eval.py:48), so a bf16 base would raise on its first conv.
    (The RAJNI wrapper casts its own input.)"""

    def __init__(self, model, dtype):
        super().__init__()
        self.model, self.dtype = model, dtype

    def forward(self, x):
        return self.model(x.to(self.dtype))


class ShardSampler(torch.utils.data.Sampler):
    """Rank r of `world` sees samples r, r + world, ... - every sample exactly once over the ranks, no padding
    (DistributedSampler repeats samples to even the shards out, which would count some images twice in the
    all-reduced accuracy; evaluate_model's SUM of [correct, total] is exact for ragged shards)."""

    def __init__(self, n, rank, world):
        self.n, self.rank, self.world = n, rank, world

    def __iter__(self):
        return iter(range(self.rank, self.n, self.world))

    def __len__(self):
        return len(range(self.rank, self.n, self.world))


class SyntheticLoader:
    """`n_batches` seeded (images, labels) batches resident on the device."""

    def __init__(self, n_batches, batch, img, device, dtype, seed):
        g = torch.Generator(device=device).manual_seed(seed)
        self.images = torch.randn(batch, 3, img, img, generator=g, device=device).to(dtype)
        self.labels = torch.randint(0, 1000, (batch,), generator=g, device=device)
        self.n = n_batches

    def __len__(self):
        return self.n

    def __iter__(self):
        for _ in range(self.n):
            yield self.images, self.labels


def build_loader(args, img_size, device, dtype, rank, world):
    if args.data_path and not args.synthetic:
        try:
            from torchvision import datasets, transforms
        except ImportError as e:
            raise SystemExit("--data_path needs torchvision (not installed); use --synthetic") from e
        tf = transforms.Compose([transforms.Resize(int(img_size * 256 / 224), interpolation=transforms.InterpolationMode.BICUBIC),
                                 transforms.CenterCrop(img_size), transforms.ToTensor(),
                                 transforms.Normalize((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))])
        ds = datasets.ImageFolder(args.data_path, transform=tf)
        sampler = ShardSampler(len(ds), rank, world) if world > 1 else None
        return torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=False, sampler=sampler,
                                           num_workers=args.num_workers, pin_memory=True, drop_last=False)
    n = args.max_batches if args.max_batches is not None else 20
    return SyntheticLoader(n, args.batch_size, img_size, device, dtype, 1234 + rank)


def main(argv=None):
    args = get_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = args.device
    if torch.device(device).type == "cuda":
        torch.cuda.set_device(local_rank)
        device = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl" if torch.device(device).type == "cuda" else "gloo")
    dtype = getattr(torch, args.dtype)
    say = print if rank == 0 else (lambda *a, **k: None)
    say(f"Device: {device}  world size: {world}")
    say(f"Model: {args.model}  dtype: {args.dtype}")

    base, source = create_base(args)
    img_size = getattr(getattr(base, "patch_embed", None), "img_size", (224, 224))[0]
    say(f"Weights: {source}")
    loader = build_loader(args, img_size, device, dtype, rank, world)
    say(f"Batches per rank: {len(loader)} x {args.batch_size}")

    base_acc = base_thr = None
    if args.compare_base:
        say("Evaluating base model (stock PyTorch ops)...")
        base_acc, base_thr = evaluate_model(InputCast(base.to(dtype), dtype), loader, device=device,
                                            max_batches=args.max_batches, warmup=args.warmup)
        say(f"Base accuracy: {base_acc:.2f}%  throughput: {base_thr:.1f} img/s")

    schedule = load_schedule(args.schedule)
    say(f"Pruning schedule: {schedule}")
    # the wrapper mutates its base in place (reference model.py:16-21): wrap a second model
    base2, _ = create_base(args)
    model = RAJNIViTWrapper(base2.to(dtype), schedule).to(device).eval()
    if args.weight_format != "model":
        model.set_weight_format(args.weight_format)
    if args.residual == "bfloat16":
        model.set_residual_dtype(torch.bfloat16)
    say("Evaluating RAJNI model (HIP path)...")
    acc, thr = evaluate_model(model, loader, device=device, max_batches=args.max_batches, warmup=args.warmup)
    say(f"RAJNI accuracy: {acc:.2f}%  throughput: {thr:.1f} img/s")
    say(f"Token counts: {model.get_last_stats()['token_counts']}")
    if base_thr:
        say(f"Speedup: {thr / base_thr:.2f}x  accuracy drop: {base_acc - acc:.2f}%")
    if world > 1:
        torch.distributed.destroy_process_group()
    return acc, thr


if __name__ == "__main__":
    main()
