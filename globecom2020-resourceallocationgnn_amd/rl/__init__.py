"""Callers either side of the hot path: simulator counterpart, replay memory and DQN agent glue,
so that the reference's training / test loops run on the engine where the reference's own Python is absent."""
from .environment import Environ, Vehicle                      # noqa: F401
from .sim_config import RL_Config                              # noqa: F401
from .agent import Agent, Memory                               # noqa: F401
from .batched_env import BatchedEnviron                        # noqa: F401
