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
    def opt(name, default=None):  # extension flags, absent from the reference's arguments.py
        try:
            return getattr(args, name)
        except (AttributeError, KeyError):
            return default

    agent = cls(obs_shape, action_shape, args)
    if opt("precision") is not None:  # "f32" (parity path) / "f32_x3" / "f32_mfma" / "bf16" (performance mode): THIS agent's conv stacks
        agent.set_precision(opt("precision"))
    if opt("hip_graph"):    # replay update() as captured hipGraphs
        agent.enable_hip_graphs()
    if opt("action_graph"):  # replay batch-1 select_action / sample_action as captured hipGraphs
        agent.enable_action_graphs()
    return agent
