from .actor_critic import ActorModel, ValueModel
from .decoder import ObservationModel, RewardModel, TIAObservationModel, VisualObservationModel
from .encoder import Encoder, VisualEncoder
from .rssm import TransitionModel
from .utils import FlatAdam, bottle

__all__ = ["ActorModel", "ValueModel", "ObservationModel", "RewardModel", "VisualObservationModel", "TIAObservationModel", "Encoder",
           "VisualEncoder", "TransitionModel", "FlatAdam", "bottle"]
