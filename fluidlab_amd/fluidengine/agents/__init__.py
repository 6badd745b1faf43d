from .agent import Agent
from .agent_injector import AgentInjector
from .agent_rigid import AgentRigid
from .agent_circulation import AgentCirculation
from .agent_icecreamdynamic import AgentIceCreamDynamic
from .agent_pouring import AgentPouring
from .agent_jetbot import AgentJetBot
