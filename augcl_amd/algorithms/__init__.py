"""Agents of the hot path, same registry as src/algorithms/factory.py."""
from .factory import algorithm, make_agent  # noqa: F401
