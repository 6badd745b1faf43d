"""Environment registry (fluidlab/envs/__init__.py registers with gym; gym is absent here, so `make` is local)."""
from .fluid_env import FluidEnv
from .latteart_env import LatteArtEnv
from .waterblock_env import WaterBlockEnv
from .circulation_env import CirculationEnv
from .icecreamdynamic_env import IceCreamDynamicEnv
from .latteartstir_env import LatteArtStirEnv
from .icecreamstatic_env import IceCreamStaticEnv
from .gatheringeasy_env import GatheringEasyEnv
from .gatheringo_env import GatheringOEnv
from .mixing_env import MixingEnv
from .pouring_env import PouringEnv
from .transporting_env import TransportingEnv

REGISTRY = {'LatteArt-v0': LatteArtEnv, 'WaterBlock-v0': WaterBlockEnv, 'Circulation-v0': CirculationEnv,
            'IceCreamDynamic-v0': IceCreamDynamicEnv, 'LatteArtStir-v0': LatteArtStirEnv, 'IceCreamStatic-v0': IceCreamStaticEnv,
            'GatheringEasy-v0': GatheringEasyEnv, 'GatheringO-v0': GatheringOEnv, 'Mixing-v0': MixingEnv, 'Pouring-v0': PouringEnv,
            'Transporting-v0': TransportingEnv}


def make(env_name, **kwargs):
    if env_name not in REGISTRY:
        raise KeyError(f'{env_name} is not built here; available: {sorted(REGISTRY)}')
    return REGISTRY[env_name](version=int(env_name.split('-v')[-1]), **kwargs)
