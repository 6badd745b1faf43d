"""AgentCirculation -- one AirCon (fluidlab/fluidengine/agents/agent_circulation.py); its collide() is the identity."""
from fluidlab_amd.fluidengine.effectors import AirCon
from .agent import Agent


class AgentCirculation(Agent):
    def build(self, sim):
        super().build(sim)
        assert self.n_effectors == 1
        assert isinstance(self.effectors[0], AirCon)
        self.aircon = self.effectors[0]
