from .loss import Loss
from .shapematching_loss import ShapeMatchingLoss
from .latteart_loss import LatteArtLoss
from .circulation_loss import CirculationLoss
from .icecreamdynamic_loss import IceCreamDynamicLoss
from .latteartstir_loss import LatteArtStirLoss
from .icecreamstatic_loss import IceCreamStaticLoss
from .gatheringeasy_loss import GatheringEasyLoss
