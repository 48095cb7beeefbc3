from .dreamer import Dreamer
from .repo import RePo

__all__ = ["Dreamer", "RePo"]
