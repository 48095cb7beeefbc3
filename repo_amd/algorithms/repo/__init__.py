from .dreamer import Dreamer
from .repo import RePo
from .tia import TIA

__all__ = ["Dreamer", "RePo", "TIA"]
