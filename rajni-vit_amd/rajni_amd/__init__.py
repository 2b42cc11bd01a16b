"""rajni_amd - MI355X-native RAJNI-ViT token-pruning forward path.

Exports what the reference package does (`rajni/__init__.py:1-2`), so
`import rajni_amd as rajni` is a drop-in:

    from rajni_amd import RAJNIViTWrapper, evaluate_model
    model = RAJNIViTWrapper(timm_vit.to(torch.bfloat16), schedule).cuda().eval()
    logits = model(images); model.get_last_stats()

All compute runs in librajni_hip.so (hand-written gfx950 HIP kernels behind the C ABI of
include/rajni_hip.h).  There is no CPU/PyTorch fallback: CPU tensors or a missing library raise.
"""
from .eval import evaluate_model
from .wrapper import RAJNIViTWrapper, RAJNIAttention, compute_importance

__version__ = "0.1.0"
__all__ = ["RAJNIViTWrapper", "RAJNIAttention", "compute_importance", "evaluate_model"]
