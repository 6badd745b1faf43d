from .mpm_simulator import MPMSimulator
from .smoke_field import SmokeField
