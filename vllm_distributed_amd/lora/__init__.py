"""Multi-LoRA serving."""
from .manager import LoRAManager, LoRAModel  # noqa: F401
