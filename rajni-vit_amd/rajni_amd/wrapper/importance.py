"""`compute_importance` - the RAJNI score - on the MI355X.

Same call as the reference (`rajni/wrapper/importance.py:5`): qkv [B, N, 3C] -> scores [B, N] in
qkv's dtype, with  score = mean_h softmax_n(q_cls . k / sqrt(D)) * sigmoid(zscore_n(|| mean_h v -
mean_n mean_h v ||)).  One HIP launch (`rajni_importance`); fp32 math, result rounded to qkv.dtype.
"""
import torch

from .. import ops


@torch.no_grad()
def compute_importance(qkv: torch.Tensor, num_heads: int, eps: float = 1e-6) -> torch.Tensor:
    return ops.importance(qkv, num_heads, eps)
