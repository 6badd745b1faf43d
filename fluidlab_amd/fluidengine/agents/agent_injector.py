"""AgentInjector -- an agent with exactly one Injector (fluidlab/fluidengine/agents/agent_injector.py)."""
from fluidlab_amd.fluidengine.effectors import Injector
from .agent import Agent


class AgentInjector(Agent):
    def build(self, sim):
        super().build(sim)
        assert self.n_effectors == 1
        assert isinstance(self.effectors[0], Injector)
        self.injector = self.effectors[0]
        self.injector.set_act_range(self.sim.particles_ng.used.to_numpy()[0])     # agent_injector.py:21
