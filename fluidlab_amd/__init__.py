"""fluidlab_amd -- MI355X-native FluidEngine MLS-MPM core with the host-side surface of FluidLab.

The device side is fluidlab_amd/csrc (hand-written gfx950 HIP behind include/fluidengine.h);
the Python below mirrors the reference's MPMSimulator / TaichiEnv / Agent / Loss / Solver
interfaces so its optimiser and task environments drive the HIP engine unchanged."""
__version__ = '0.1.0'
