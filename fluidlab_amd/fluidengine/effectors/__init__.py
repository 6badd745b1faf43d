from .effector import Effector
from .injector import Injector, BallInjector
from .rigid import Rigid
from .aircon import AirCon
