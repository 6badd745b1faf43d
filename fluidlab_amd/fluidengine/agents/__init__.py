from .agent import Agent
from .agent_injector import AgentInjector
