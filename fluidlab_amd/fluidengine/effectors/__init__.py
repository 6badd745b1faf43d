from .effector import Effector
from .injector import Injector, BallInjector
