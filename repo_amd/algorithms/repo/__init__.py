from .dreamer import Dreamer
from .repo import RePo
from .repo_adapt import FinetunedRePo
from .tia import TIA

__all__ = ["Dreamer", "RePo", "TIA", "FinetunedRePo"]
