from .dreamer import Dreamer
from .dreamer_mt import MultitaskDreamer
from .repo import RePo
from .repo_adapt import FinetunedRePo
from .repo_mt import MultitaskRePo
from .tia import TIA

__all__ = ["Dreamer", "RePo", "TIA", "FinetunedRePo", "MultitaskDreamer", "MultitaskRePo"]
