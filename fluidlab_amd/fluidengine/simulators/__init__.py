from .mpm_simulator import MPMSimulator
