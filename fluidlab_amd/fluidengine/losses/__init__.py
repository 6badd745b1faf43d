from .loss import Loss
from .shapematching_loss import ShapeMatchingLoss
from .latteart_loss import LatteArtLoss
from .circulation_loss import CirculationLoss
from .icecreamdynamic_loss import IceCreamDynamicLoss
from .latteartstir_loss import LatteArtStirLoss
from .icecreamstatic_loss import IceCreamStaticLoss
from .gatheringeasy_loss import GatheringEasyLoss
from .host_loss import HostLoss, pairwise_l1
from .pouring_loss import PouringLoss
from .transporting_loss import TransportingLoss
from .mixing_loss import MixingLoss
from .gatheringO_loss import GatheringOLoss
