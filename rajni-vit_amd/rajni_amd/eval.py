"""`evaluate_model` - the harness that DEFINES the benchmark metric.

Same signature and return value as the reference (`rajni/eval.py:6-75`):
    acc_percent, images_per_sec = evaluate_model(model, dataloader, device="cuda", max_batches=None, warmup=5)
images/sec = sum of batch sizes / sum of (sync -> model(images) -> sync) wall time; H2D copies and the
argmax are outside the timed region; warm-up forwards restart the loader when it runs out.

Differences, both deliberate:
  * the device is ALWAYS synchronised around the forward when it is an accelerator - the reference
    compares `device == "cuda"` and so never synchronises when handed a `torch.device`;
  * under `torch.distributed` (one process per GPU, images sharded by rank) the counters are joined
    by ONE all-reduce (SUM) of a 3 + 2 * world vector: [correct, total, images] followed by one
    (images, seconds) slot per rank that only its owner fills - so the same collective also tells every
    rank each rank's own rate; node seconds = MAX over the slots.  Every rank returns the node-level
    accuracy and node-level images/sec.  Without a process group it is the single-device function of
    the reference.
After a call, `evaluate_model.last_stats` holds what the return value cannot (the reference's signature is kept):
`{"world", "rank", "images", "seconds", "per_rank": [(images, seconds), ...]}` - the multi-GPU diagnosis surface of
`bench.py` (slowest rank, spread).
"""
from __future__ import annotations

import time

import torch

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    tqdm = None


def _is_accel(device) -> bool:
    return torch.device(device).type != "cpu"


def _sync(device) -> None:
    if _is_accel(device):
        torch.cuda.synchronize(torch.device(device) if torch.device(device).index is not None else None)


def _dist_ready() -> bool:
    return torch.distributed.is_available() and torch.distributed.is_initialized()


@torch.no_grad()
def evaluate_model(model, dataloader, device="cuda", max_batches=None, warmup=5):
    model.eval()
    model.to(device)
    rank = torch.distributed.get_rank() if _dist_ready() else 0

    # ---- warm-up (eval.py:17-26): `warmup` forwards, restarting the iterator when exhausted
    if rank == 0:
        print(f"Warming up {warmup} batches")
    it = iter(dataloader)
    for _ in range(warmup):
        try:
            x, _ = next(it)
        except StopIteration:
            it = iter(dataloader)
            x, _ = next(it)
        model(x.to(device))
    _sync(device)

    correct = 0
    total = 0
    total_images = 0
    total_time = 0.0

    try:
        n_total = max_batches if max_batches is not None else len(dataloader)
    except TypeError:
        n_total = None
    pbar = dataloader
    if tqdm is not None:
        pbar = tqdm(dataloader, desc="Evaluating", total=n_total, leave=False, disable=(rank != 0))

    for i, (images, labels) in enumerate(pbar):
        if max_batches is not None and i >= max_batches:   # eval.py:45-46
            break
        images = images.to(device)
        labels = labels.to(device)

        _sync(device)                                       # eval.py:51-52 (fixed: always)
        start = time.time()
        logits = model(images)
        _sync(device)
        total_time += time.time() - start                   # eval.py:57-59

        preds = logits.argmax(dim=1)                         # eval.py:61
        correct += (preds == labels).sum().item()
        total += labels.size(0)
        total_images += images.size(0)
        if tqdm is not None and total > 0 and rank == 0:
            pbar.set_postfix(acc=f"{100.0 * correct / total:.2f}%",
                             imgs_per_s=f"{total_images / max(total_time, 1e-6):.1f}")

    per_rank = [(total_images, total_time)]
    world = 1
    if _dist_ready() and torch.distributed.get_world_size() > 1:
        world = torch.distributed.get_world_size()
        red_dev = device if (_is_accel(device) and torch.distributed.get_backend() == "nccl") else "cpu"
        vec = [0.0] * (3 + 2 * world)
        vec[0:3] = [correct, total, total_images]
        vec[3 + 2 * rank], vec[4 + 2 * rank] = total_images, total_time      # this rank's own slot
        joined = torch.tensor(vec, dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(joined, op=torch.distributed.ReduceOp.SUM)   # the ONE collective (RCCL on device)
        vals = joined.tolist()
        correct, total, total_images = (int(round(v)) for v in vals[:3])
        per_rank = [(int(round(vals[3 + 2 * r])), float(vals[4 + 2 * r])) for r in range(world)]
        total_time = max(sec for _, sec in per_rank)                               # node time = slowest rank
    evaluate_model.last_stats = {"world": world, "rank": rank, "images": total_images, "seconds": total_time,
                                 "per_rank": per_rank}

    acc = 100.0 * correct / max(total, 1)                   # eval.py:73
    throughput = total_images / max(total_time, 1e-6)       # eval.py:74
    return acc, throughput


evaluate_model.last_stats = None
