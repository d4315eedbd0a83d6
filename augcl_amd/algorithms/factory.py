"""name -> agent class (src/algorithms/factory.py:11-25).  curl / pad / soda are auxiliary-task
variants outside the accelerated path and are not registered."""
from .augcl import AugCL
from .drq import DrQ
from .non_naive_rad import NonNaiveRAD
from .rad import RAD
from .sac import SAC
from .svea import SVEA

algorithm = {"sac": SAC, "rad": RAD, "drq": DrQ, "svea": SVEA, "augcl": AugCL, "non_naive_rad": NonNaiveRAD}


def make_agent(obs_shape, action_shape, args):
    try:
        cls = algorithm[args.algorithm]
    except KeyError:
        raise KeyError(f"algorithm {args.algorithm!r} is not on the accelerated path; have {sorted(algorithm)}") from None
    agent = cls(obs_shape, action_shape, args)
    try:  # extension flag (absent from the reference's arguments.py): replay update() as captured hipGraphs
        want_graph = bool(args.hip_graph)
    except (AttributeError, KeyError):
        want_graph = False
    if want_graph:
        agent.enable_hip_graphs()
    return agent
