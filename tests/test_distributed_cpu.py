"""World-size-2 `gloo` test of the N>1 path of evaluate_model: images sharded by rank,
ONE all-reduce joins [correct, total, images] (SUM) and elapsed seconds (MAX); every rank returns
the node-level accuracy, identical to a single process evaluating all shards."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Identity(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return x


def _make_batches(seed, n_batches, bsz, classes=10):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_batches):
        logits = torch.randn(bsz, classes, generator=g)
        labels = torch.randint(0, classes, (bsz,), generator=g)
        labels[::3] = logits[::3].argmax(1)      # about a third + chance are right
        out.append((logits, labels))
    return out


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "rajni-vit_amd"))
    import torch.distributed as dist
    import rajni_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ragged on purpose: rank 0 has 3 batches, rank 1 has 2 (drop_last=False semantics)
        loader = _make_batches(100 + rank, 3 - rank, 8)
        acc, thr = rajni_amd.evaluate_model(_Identity(), loader, device="cpu", max_batches=None, warmup=1)
        q.put((rank, acc, thr, rajni_amd.evaluate_model.last_stats))
    finally:
        dist.destroy_process_group()


def test_evaluate_model_world_size_2_gloo():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "rajni-vit_amd"))
    import rajni_amd
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process answer over the union of the shards
    all_batches = _make_batches(100, 3, 8) + _make_batches(101, 2, 8)
    acc1, _ = rajni_amd.evaluate_model(_Identity(), all_batches, device="cpu", warmup=0)
    assert res[0][1] == pytest.approx(acc1, abs=1e-9)
    assert res[1][1] == pytest.approx(acc1, abs=1e-9)
    assert res[0][2] == pytest.approx(res[1][2], rel=1e-9) and res[0][2] > 0   # same node-level img/s on both ranks
    # the one all-reduce also carries every rank's own (images, seconds): the multi-GPU diagnosis surface of bench.py
    for r, (_, _, thr, st) in enumerate(res):
        assert st["world"] == 2 and st["rank"] == r and st["images"] == 40
        assert [n for n, _ in st["per_rank"]] == [24, 16]                       # ragged shards: 3 and 2 batches of 8
        assert all(sec > 0 for _, sec in st["per_rank"])
        assert st["seconds"] == pytest.approx(max(sec for _, sec in st["per_rank"]))
        assert thr == pytest.approx(40 / st["seconds"], rel=1e-9)
    assert res[0][3]["per_rank"] == res[1][3]["per_rank"]
