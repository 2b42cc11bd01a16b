"""Same three names the reference's `rajni/wrapper/__init__.py` exports."""
from .importance import compute_importance
from .attention import RAJNIAttention
from .model import RAJNIViTWrapper

__all__ = ["RAJNIViTWrapper", "RAJNIAttention", "compute_importance"]
