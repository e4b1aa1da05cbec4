"""unmore_amd: MI355X-native (gfx950) implementation of unMORE's stage-1 ObjectnessNet hot path."""
from .objectness_net import ObjectnessNet  # noqa: F401
from .trainer import TrainStep  # noqa: F401
from .loss import objectness_loss  # noqa: F401
from .binary_classifier import Binary_Classifier  # noqa: F401
