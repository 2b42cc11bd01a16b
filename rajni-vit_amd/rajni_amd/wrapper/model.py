"""`RAJNIViTWrapper(base_model, pruning_schedule)` - drop-in for the reference wrapper
(`rajni/wrapper/model.py:6-69`) whose forward runs as ONE native plan on the MI355X.

Kept from the reference: constructor signature, in-place surgery on the base model (scheduled
blocks get a `RAJNIAttention`, every block gets `has_pruner`; model.py:12-23), parameter sharing,
`forward(images) -> logits`, `get_last_stats() -> {"token_counts": [...]}` (None before the first
forward).  Deliberate fixes: schedule keys are normalised to int (B1: a JSON-loaded
schedule prunes), and a `no_embed_class` pos-embed works (B3).

Not kept: the Python per-block loop.  `forward` builds (once per batch shape) a `rajni_vit_plan`
- packed weights, workspace, per-stage index buffers - and calls `rajni_vit_forward`, which enqueues
every kernel of the network on the current stream with no host synchronisation.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import _native as nat
from .. import ops
from .attention import RAJNIAttention


def normalise_schedule(schedule) -> Dict[int, Dict]:
    """{block index: {"keep_ratio": float, "update": bool}} with int keys.  A missing `keep_ratio`
    raises KeyError like the reference (model.py:18); `update` defaults to True (model.py:19)."""
    out: Dict[int, Dict] = {}
    for k, cfg in (schedule or {}).items():
        out[int(k)] = {"keep_ratio": float(cfg["keep_ratio"]), "update": bool(cfg.get("update", True))}
    return out


def plan_token_counts(n0: int, depth: int, schedule: Dict[int, Dict]) -> List[int]:
    """Tokens at the entry of every block (model.py:43) - a pure function of (N0, depth, schedule)."""
    counts, n = [], n0
    for i in range(depth):
        counts.append(n)
        if i in schedule:
            n = ops.keep_count(schedule[i]["keep_ratio"], n) + 1
    return counts


class RAJNIViTWrapper(nn.Module):
    def __init__(self, base_model: nn.Module, pruning_schedule: Dict[int, Dict]):
        super().__init__()
        self.m = base_model
        self.blocks = base_model.blocks
        self.pruning_schedule = normalise_schedule(pruning_schedule)

        for i, blk in enumerate(self.blocks):
            if i in self.pruning_schedule:
                cfg = self.pruning_schedule[i]
                if not isinstance(blk.attn, RAJNIAttention):
                    blk.attn = RAJNIAttention(blk.attn, keep_ratio=cfg["keep_ratio"], update=cfg["update"])
                else:  # re-wrapping an already wrapped base
                    blk.attn.keep_ratio, blk.attn.update = cfg["keep_ratio"], cfg["update"]
                blk.has_pruner = True
            else:
                blk.has_pruner = False

        self._last_stats = None
        self._weights = None       # packed device tensors (kept alive here)
        self._weights_key = None   # epoch of the last re-pack (part of the plan key)
        self._weights_sig = None   # [data_ptr..., _version...] of the base model's parameters at that re-pack
        self._weights_cfg = None
        self._plan = None          # most recent (key, VitPlan, keep-alive objects, stage buffers, counts)
        self._plans = {}           # small cache by key: a ragged last batch must not evict the main plan
        self._forced: Dict[int, torch.Tensor] = {}
        self._trace_scores = False
        # residual stream precision between blocks: fp32 (default; see DESIGN.md "numerics") or the
        # model dtype like the reference's bf16 model (`set_residual_dtype(torch.bfloat16)`)
        self._resid_bf16 = False
        # storage format of the four big Linear weights of every block: "model" = the model dtype,
        # "fp8" = e4m3 bytes + per-row fp32 scale (`set_weight_format("fp8")`, bf16 models only)
        self._weight_format = "model"
        # opt-in: compute the last block for the CLS row only (the head reads nothing else, model.py:65-66)
        self._cls_only_last = False

    # ------------------------------------------------------------------------------------------
    def get_last_stats(self):
        return self._last_stats

    def get_last_trace(self) -> Dict[int, Dict[str, torch.Tensor]]:
        """Per scheduled block: keep_idx [B,Np] int64, next_scores [B,Np] and (if enabled with
        `trace_scores(True)`) the full scores [B,N] the stage ranked.  Test/diagnostic surface."""
        if self._plan is None:
            return {}
        out = {}
        for i, bufs in self._plan[3].items():
            idx = self._forced.get(i, bufs["keep_idx"])
            d = {"keep_idx": idx.long(), "next_scores": bufs["next_scores"]}
            # a stage ranks freshly computed scores iff `update` or nothing was carried into it
            # (attention.py:25); otherwise it ranked the previous stage's next_scores
            sched = self.pruning_schedule
            recomputed = sched[i]["update"] or (i - 1) not in sched
            if bufs["scores"] is not None and recomputed:
                d["scores"] = bufs["scores"]
            elif not recomputed:
                d["scores"] = self._plan[3][i - 1]["next_scores"]
            out[i] = d
        return out

    def _drop_plans(self):
        self._plan = None
        self._plans = {}

    def set_residual_dtype(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("residual stream dtype must be torch.float32 or torch.bfloat16")
        self._resid_bf16 = dtype == torch.bfloat16
        self._drop_plans()
        return self

    def set_weight_format(self, fmt: str):
        """"model" (default), "fp8" or "fp8_mfma".
        "fp8": keep qkv/proj/fc1/fc2 weights as fp8 e4m3 with one fp32 scale per output row (BASELINE config 5,
        SURVEY 8(f)-4).  Activations, accumulation, patch-embed and head stay as they are; results equal the same
        model run with the DEQUANTISED weights (the bf16 matrix pipe does the arithmetic).
        "fp8_mfma": the same weights, AND the inputs of qkv / fc1 / fc2 as per-row-scaled e4m3 (norm1 / norm2 emit
        them, fc1's GELU epilogue re-quantises the hidden activations), so that those three products run on the
        CDNA4 fp8 matrix pipe (v_mfma_f32_16x16x128_f8f6f4, fp32 accumulation).  Results equal the model run with
        the dequantised weights and the same activation quantisation (tests/test_gpu_fp8_mfma.py holds the rule);
        the reference has no fp8 semantics, so this is an opt-in numerics contract of the build's own.  Needs
        embed dim and MLP width that are multiples of 256."""
        if fmt not in ("model", "fp8", "fp8_mfma"):
            raise ValueError('weight format must be "model", "fp8" or "fp8_mfma"')
        if fmt != self._weight_format:
            self._weight_format = fmt
            self._weights = self._weights_key = None
            self._drop_plans()
        return self

    def dequantized_state_dict(self):
        """fp32 copies of the block Linear weights as the fp8 kernels see them (q * scale), keyed like the
        base model's state_dict.  Test surface: feed these to the oracle for the fp8 parity check."""
        if self._weights is None or self._weight_format not in ("fp8", "fp8_mfma"):
            raise RuntimeError("run a forward with set_weight_format('fp8') or ('fp8_mfma') first")
        out = {}
        hid = self._weights["desc"]["hidden"]
        for i, bw in enumerate(self._weights["blocks"]):
            for ours, theirs in (("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
                w = ops.dequantize_fp8(bw[ours + "_w"], bw[ours + "_s"])
                if ours == "fc1":
                    w = w[:hid]               # rows / columns past the model's MLP width are zero padding
                elif ours == "fc2":
                    w = w[:, :hid]
                out[f"blocks.{i}.{theirs}.weight"] = w
        return out

    def set_last_block_cls_only(self, on: bool = True):
        """When the last block is not a pruning stage, run it for the CLS row only: CLS-query attention over all
        tokens, proj / MLP on B rows.  The logits are the same function of the input (the reference's head reads
        x[:, 0] only); the other rows of the last block are never formed.  Off by default: the default forward
        executes the reference's op graph row for row."""
        if bool(on) != self._cls_only_last:
            self._cls_only_last = bool(on)
            self._drop_plans()
        return self

    def trace_scores(self, on: bool = True):
        self._trace_scores = bool(on)
        self._drop_plans()
        return self

    def force_keep_idx(self, forced: Optional[Dict[int, torch.Tensor]]):
        """Test hook for selection-conditional parity: use the given keep_idx
        ([B, keep+1], CLS first, ascending) in the listed blocks instead of the device selection."""
        self._forced = {}
        for k, v in (forced or {}).items():
            self._forced[int(k)] = v.to(torch.int32).contiguous()
        self._drop_plans()
        return self

    # ------------------------------------------------------------------------------------------
    def _describe(self):
        m = self.m
        pe = m.patch_embed.proj
        if not isinstance(pe, nn.Conv2d) or pe.kernel_size != pe.stride or pe.kernel_size[0] != pe.kernel_size[1]:
            raise NotImplementedError("RAJNIViTWrapper: patch_embed.proj must be a square Conv2d with stride == kernel")
        pe_norm = getattr(m.patch_embed, "norm", None)
        if pe_norm is not None and not isinstance(pe_norm, nn.Identity):
            raise NotImplementedError("RAJNIViTWrapper: patch_embed.norm is not supported")
        if not isinstance(m.head, nn.Linear):
            raise NotImplementedError("RAJNIViTWrapper: base_model.head must be nn.Linear")
        blk0 = self.blocks[0]
        Cdim = m.cls_token.shape[-1]
        heads = blk0.attn.num_heads
        desc = dict(C=Cdim, H=heads, D=Cdim // heads, depth=len(self.blocks), hidden=blk0.mlp.fc1.out_features,
                    num_classes=m.head.out_features, patch=pe.kernel_size[0], in_chans=pe.in_channels,
                    ln_eps=float(blk0.norm1.eps), scale=float(blk0.attn.scale))
        desc["hidden_pad"] = (desc["hidden"] + 63) // 64 * 64
        for i, blk in enumerate(self.blocks):
            for ln in (blk.norm1, blk.norm2):
                if not isinstance(ln, nn.LayerNorm) or ln.weight is None or float(ln.eps) != desc["ln_eps"]:
                    raise NotImplementedError(f"block {i}: norm layers must be affine nn.LayerNorm with one eps")
            act = getattr(blk.mlp, "act", None)
            if not isinstance(act, nn.GELU) or getattr(act, "approximate", "none") != "none":
                raise NotImplementedError(f"block {i}: mlp.act must be exact-erf nn.GELU (timm default)")
            mlp_norm = getattr(blk.mlp, "norm", None)
            if mlp_norm is not None and not isinstance(mlp_norm, nn.Identity):
                raise NotImplementedError(f"block {i}: mlp.norm is not supported")
            for extra in ("q_norm", "k_norm"):
                mod = getattr(blk.attn, extra, None)
                if mod is not None and not isinstance(mod, nn.Identity):
                    raise NotImplementedError(f"block {i}: attn.{extra} is not supported (SURVEY Q5)")
            if blk.attn.num_heads != heads or float(blk.attn.scale) != desc["scale"]:
                raise NotImplementedError(f"block {i}: heads/scale differ between blocks")
        if not isinstance(m.norm, nn.LayerNorm) or float(m.norm.eps) != desc["ln_eps"]:
            raise NotImplementedError("RAJNIViTWrapper: base_model.norm must be nn.LayerNorm with the blocks' eps")
        return desc

    def _all_params(self):
        """The base model's parameters.  Walking the module tree costs ~150 us per call - exposed latency in the
        sync -> forward -> sync metric of evaluate_model - so the walk is cached, and the cache is VALIDATED on every
        call with plain dict lookups: every (parent, name, child module) and (owner, name, Parameter) slot recorded by
        the walk must still hold the same object.  A replaced module (`model.head = nn.Linear(C, 10)`) or a re-assigned
        `nn.Parameter` therefore re-walks and re-packs on the very next forward, like the reference, which reads the
        live modules on every call; in-place changes, `.to()` and `load_state_dict` are seen through data_ptr / _version."""
        slots = getattr(self, "_param_slots", None)
        if slots is not None:
            mods, pars = slots
            ok = True
            for parent, name, child in mods:
                if parent._modules.get(name) is not child:
                    ok = False
                    break
            if ok:
                for owner, name, par in pars:
                    if owner._parameters.get(name) is not par:
                        ok = False
                        break
            if ok:
                return self._param_list
        mods, pars = [], []
        for mod in self.m.modules():
            for name, child in mod._modules.items():
                mods.append((mod, name, child))
            for name, par in mod._parameters.items():
                pars.append((mod, name, par))
        self._param_slots = (mods, pars)
        self._param_list = [p for p in self.m.parameters()]
        return self._param_list

    def _pack_weights(self, device, dtype):
        params = self._all_params()
        # fingerprint of the live weights: storage pointers (`.to()`, `param.data = ...`) and version counters (in-place
        # edits, load_state_dict) - two flat list comprehensions and one C-speed list compare per forward (this sits in
        # the sync -> forward -> sync metric); `_weights_key` is just the epoch of the last re-pack
        sig = [p.data_ptr() for p in params]
        sig += [p._version for p in params]
        cfg_key = (device, dtype, self._weight_format)
        if self._weights is not None and self._weights_sig == sig and self._weights_cfg == cfg_key:
            return self._weights
        key = (getattr(self, "_weights_epoch", 0) + 1)
        self._weights_epoch = key
        desc = self._describe()
        m = self.m
        fp8 = self._weight_format in ("fp8", "fp8_mfma")
        if fp8 and dtype != torch.bfloat16:
            raise NotImplementedError("fp8 weights need a bf16 model (activations stay bf16)")
        pw = lambda w: ops.pack_weight(w, dtype, device)
        # block Linear weight -> (packed tensor, per-row scale or None)
        pq = (lambda w: ops.pack_weight_fp8(w, dtype, device)) if fp8 else (lambda w: (pw(w), None))
        pv = lambda v: ops.pack_vec(v, dtype, device)
        zeros = lambda n: torch.zeros(n, dtype=torch.float32, device=device)
        W = dict(desc=desc)
        W["patch_w"] = ops.pack_weight(m.patch_embed.proj.weight, dtype, device, k_multiple=64)
        W["patch_b"] = pv(m.patch_embed.proj.bias) if m.patch_embed.proj.bias is not None else zeros(desc["C"])
        W["cls"] = m.cls_token.detach().to(device=device, dtype=dtype).reshape(-1).contiguous()
        W["pos"] = m.pos_embed.detach().to(device=device, dtype=dtype).reshape(-1, desc["C"]).contiguous()
        W["norm_w"], W["norm_b"] = pv(m.norm.weight), pv(m.norm.bias)
        W["head_w"] = pw(m.head.weight)
        W["head_b"] = pv(m.head.bias) if m.head.bias is not None else zeros(desc["num_classes"])
        blocks = []
        for blk in self.blocks:
            a = blk.attn
            ls1 = getattr(blk, "ls1", None)
            ls2 = getattr(blk, "ls2", None)
            g1 = pv(ls1.gamma) if (ls1 is not None and hasattr(ls1, "gamma")) else None
            g2 = pv(ls2.gamma) if (ls2 is not None and hasattr(ls2, "gamma")) else None
            for ls, nm in ((ls1, "ls1"), (ls2, "ls2")):
                if ls is not None and not isinstance(ls, nn.Identity) and not hasattr(ls, "gamma"):
                    raise NotImplementedError(f"{nm} must be Identity or a LayerScale with .gamma")
            for dp in (getattr(blk, "drop_path1", None), getattr(blk, "drop_path2", None)):
                if dp is not None and not isinstance(dp, nn.Identity) and self.training:
                    raise NotImplementedError("drop_path in training mode: this is the inference path")
            (qkv_w, qkv_s), (proj_w, proj_s) = pq(a.qkv.weight), pq(a.proj.weight)
            # An MLP width that is not whole 64-wide K steps (so400m: 4304) is zero-padded: fc1 gets zero rows (it
            # has them anyway, up to the next 256) with zero bias, GELU(0) = 0 lands in the padding columns of the
            # hidden buffer, and fc2's zero-padded input columns ignore them - exact, no kernel involved.
            hid, hpad = desc["hidden"], desc["hidden_pad"]
            (fc1_w, fc1_s) = pq(blk.mlp.fc1.weight)
            if fp8:
                fc2_w, fc2_s = ops.pack_weight_fp8(blk.mlp.fc2.weight, dtype, device, k_multiple=64)
                if hpad != hid:
                    fc1_s = torch.cat([fc1_s, torch.ones(hpad - hid, dtype=fc1_s.dtype, device=fc1_s.device)])
            else:
                fc2_w, fc2_s = ops.pack_weight(blk.mlp.fc2.weight, dtype, device, k_multiple=64), None
            fc1_b = pv(blk.mlp.fc1.bias) if blk.mlp.fc1.bias is not None else zeros(hid)
            if hpad != hid:
                fc1_b = torch.cat([fc1_b, zeros(hpad - hid)])
            blocks.append(dict(
                norm1_w=pv(blk.norm1.weight), norm1_b=pv(blk.norm1.bias),
                qkv_w=qkv_w, qkv_s=qkv_s, qkv_b=pv(a.qkv.bias) if a.qkv.bias is not None else zeros(3 * desc["C"]),
                proj_w=proj_w, proj_s=proj_s, proj_b=pv(a.proj.bias) if a.proj.bias is not None else zeros(desc["C"]),
                ls1=g1, norm2_w=pv(blk.norm2.weight), norm2_b=pv(blk.norm2.bias),
                fc1_w=fc1_w, fc1_s=fc1_s, fc1_b=fc1_b,
                fc2_w=fc2_w, fc2_s=fc2_s, fc2_b=pv(blk.mlp.fc2.bias), ls2=g2))
            if self._weight_format == "fp8_mfma":
                # constants of the hidden-activation bound (rajni_layernorm_fp8): largest row norm of the DEQUANTISED
                # fc1 weight and largest |bias| as the kernels hold it
                wd = ops.dequantize_fp8(fc1_w, fc1_s)
                blocks[-1]["fc1_rownorm_max"] = float(wd.norm(dim=1).max())
                blocks[-1]["fc1_bias_absmax"] = float(fc1_b.abs().max())
                # the one scale of the e4m3 attention output (rajni_attention_fp8), from the V rows of the dequantised qkv weight
                C = desc["C"]
                vd = ops.dequantize_fp8(qkv_w, qkv_s)[2 * C:3 * C]
                blocks[-1]["attn_out_scale"] = ops.attention_out_scale(blocks[-1]["norm1_w"], blocks[-1]["norm1_b"], vd,
                                                                       blocks[-1]["qkv_b"][2 * C:3 * C])
        W["blocks"] = blocks
        self._weights, self._weights_key = W, key
        self._weights_sig, self._weights_cfg = sig, cfg_key
        self._drop_plans()
        return W

    def _build_plan(self, B: int, S: int, device, dtype):
        W = self._pack_weights(device, dtype)
        d = W["desc"]
        key = (B, S, device, dtype, self._weights_key, tuple(sorted(self._forced)), self._trace_scores, self._resid_bf16,
               self._cls_only_last)
        if self._plan is not None and self._plan[0] == key:
            return self._plan
        if key in self._plans:
            self._plan = self._plans[key]
            return self._plan
        if S % d["patch"] != 0:
            raise ValueError(f"image size {S} is not a multiple of the patch size {d['patch']}")
        n0 = (S // d["patch"]) ** 2 + 1
        pos_rows = W["pos"].shape[0]
        if pos_rows >= n0:
            pos_has_cls = 1      # reference: x + pos_embed[:, :N]  (model.py:37)
        elif pos_rows == n0 - 1:
            pos_has_cls = 0      # timm no_embed_class
        else:
            raise ValueError(f"pos_embed has {pos_rows} rows but the input yields {n0} tokens")
        counts = plan_token_counts(n0, d["depth"], self.pruning_schedule)

        blocks = (nat.Block * d["depth"])()
        bufs: Dict[int, Dict[str, Optional[torch.Tensor]]] = {}
        for i, bw in enumerate(W["blocks"]):
            cb = blocks[i]
            for name in ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1", "norm2_w", "norm2_b",
                         "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2", "qkv_s", "proj_s", "fc1_s", "fc2_s"):
                setattr(cb, name, nat.ptr(bw[name]))
            if self._weight_format == "fp8_mfma":
                cb.fc1_rownorm_max, cb.fc1_bias_absmax = bw["fc1_rownorm_max"], bw["fc1_bias_absmax"]
                cb.attn_out_scale = bw["attn_out_scale"]
            if i in self.pruning_schedule:
                cfg = self.pruning_schedule[i]
                N = counts[i]
                keep = ops.keep_count(cfg["keep_ratio"], N)
                if keep > N - 1:
                    raise ValueError(f"block {i}: keep_ratio {cfg['keep_ratio']} > 1 selects more tokens than exist")
                kb = dict(keep_idx=torch.empty((B, keep + 1), dtype=torch.int32, device=device),
                          next_scores=torch.empty((B, keep + 1), dtype=dtype, device=device),
                          scores=torch.empty((B, N), dtype=dtype, device=device) if self._trace_scores else None)
                bufs[i] = kb
                cb.keep, cb.update = keep, int(cfg["update"])
                cb.keep_idx, cb.next_scores, cb.scores = kb["keep_idx"].data_ptr(), kb["next_scores"].data_ptr(), \
                    nat.ptr(kb["scores"])
                if i in self._forced:
                    f = self._forced[i].to(device)
                    if tuple(f.shape) != (B, keep + 1):
                        raise ValueError(f"forced keep_idx for block {i} has shape {tuple(f.shape)}, want {(B, keep + 1)}")
                    self._forced[i] = f
                    cb.forced_keep_idx = f.data_ptr()
            else:
                cb.keep = 0

        plan = nat.VitPlan()
        plan.dtype = nat.dtype_code(dtype)
        plan.B, plan.in_chans, plan.img_size, plan.patch_size = B, d["in_chans"], S, d["patch"]
        plan.C, plan.H, plan.D, plan.depth, plan.hidden = d["C"], d["H"], d["D"], d["depth"], d["hidden_pad"]
        plan.num_classes, plan.ln_eps, plan.attn_scale = d["num_classes"], d["ln_eps"], d["scale"]
        plan.pos_has_cls = pos_has_cls
        plan.patch_w, plan.patch_b = W["patch_w"].data_ptr(), W["patch_b"].data_ptr()
        plan.cls_token, plan.pos_embed = W["cls"].data_ptr(), W["pos"].data_ptr()
        plan.blocks = blocks
        plan.norm_w, plan.norm_b = W["norm_w"].data_ptr(), W["norm_b"].data_ptr()
        plan.head_w, plan.head_b = W["head_w"].data_ptr(), W["head_b"].data_ptr()
        tc = (C.c_int32 * d["depth"])()
        plan.token_counts = tc
        plan.logits_ld = (d["num_classes"] + 7) // 8 * 8
        plan.resid_bf16 = int(self._resid_bf16)
        plan.cls_only_last_block = int(self._cls_only_last)
        plan.act_fp8 = int(self._weight_format == "fp8_mfma")
        nbytes = nat.lib().rajni_vit_workspace_bytes(C.byref(plan))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        plan.workspace, plan.workspace_bytes = ws.data_ptr(), nbytes
        self._plan = (key, plan, (blocks, tc, ws, W), bufs, counts)   # W: a stale optimistic launch keeps its weights alive
        if len(self._plans) >= 4:     # workspaces are large: keep only a few batch shapes alive
            self._plans.pop(next(iter(self._plans)))
        self._plans[key] = self._plan
        return self._plan

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        nat.require_device(x, "input images")
        if x.dim() != 4 or x.shape[-1] != x.shape[-2]:
            raise ValueError(f"expected images [B, C, S, S], got {tuple(x.shape)}")
        dtype = self.m.cls_token.dtype
        if self.m.cls_token.device != x.device:
            raise nat.NativeError(f"model is on {self.m.cls_token.device} but images are on {x.device}")
        if x.dtype != dtype:
            x = x.to(dtype)
        x = x.contiguous()
        B, S = x.shape[0], x.shape[-1]
        with nat.device_guard(x.device):
            def launch(entry):
                plan = entry[1]
                out = torch.empty((B, plan.logits_ld), dtype=dtype, device=x.device)
                nat.check(nat.lib().rajni_vit_forward(C.byref(plan), x.data_ptr(), out.data_ptr(),
                                                      nat.stream_ptr(x.device)), "rajni_vit_forward")
                return out
            # Optimistic launch: with a plan of this batch shape at hand the kernels are enqueued FIRST and the check
            # that the base model's weights are still the packed ones (a walk over ~150 parameters, ~0.1 ms of Python)
            # runs while the GPU works - in the sync -> forward -> sync metric of evaluate_model that check would
            # otherwise sit in front of every forward.  Packed weights are copies, so a launch on a stale plan reads
            # consistent (old) data; if the check finds a change the forward is simply enqueued again on the new plan
            # and the first result is dropped.  Option setters drop `_plan`, so they always take the slow path.
            # ONE STREAM PER WRAPPER: the stale launch's workspace and packed weights stay referenced by `cached` until
            # this function returns and are handed back to the allocator in stream order - a second stream driving the
            # same wrapper concurrently is not supported (the reference is one-forward-at-a-time too).
            cached = self._plan
            if cached is not None and cached[0][:4] == (B, S, x.device, dtype):
                logits = launch(cached)
                entry = self._build_plan(B, S, x.device, dtype)
                if entry is not cached:
                    logits = launch(entry)
            else:
                entry = self._build_plan(B, S, x.device, dtype)
                logits = launch(entry)
        plan, tc = entry[1], entry[2][1]
        self._last_stats = {"token_counts": [int(tc[i]) for i in range(plan.depth)]}   # model.py:68
        ld = plan.logits_ld
        return logits[:, : plan.num_classes] if ld != plan.num_classes else logits
