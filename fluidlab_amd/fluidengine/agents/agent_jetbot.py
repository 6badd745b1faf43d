"""AgentJetBot -- one Injector (the jet) and a collector for WATER particles that leave its boundary
(fluidlab/fluidengine/agents/agent_jetbot.py).  act() = injector then collector (:19-21); both run inside the engine's
substep.  Like AgentInjector it does not collide (agent_injector.py:35-36): the jetbot mesh is renderer data."""
from fluidlab_amd.configs.macros import WATER
from fluidlab_amd.fluidengine.boundaries import create_boundary
from .agent_injector import AgentInjector


class AgentJetBot(AgentInjector):
    def __init__(self, collector_boundary, **kwargs):
        super().__init__(**kwargs)
        self.collector_boundary = create_boundary(**collector_boundary)

    def build(self, sim):
        super().build(sim)
        sim.engine.agent_set_collector(self.collector_boundary.to_abi(sim.engine.elib), WATER)    # agent_jetbot.py:37
