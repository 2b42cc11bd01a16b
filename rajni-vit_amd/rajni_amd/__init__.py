"""rajni_amd - MI355X-native RAJNI-ViT token-pruning forward path.

Drop-in for the reference package surface (`/root/reference/rajni/__init__.py:1-2`):
`from rajni_amd import RAJNIViTWrapper, evaluate_model` (or `import rajni_amd as rajni`).
"""
__version__ = "0.1.0"
