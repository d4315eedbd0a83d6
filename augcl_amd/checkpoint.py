"""Checkpoint / curriculum hand-off compatibility.

The reference checkpoints by pickling the whole agent object (`torch.save(agent, path)`, src/train.py:224-232,
src/utils.py:333-352) -- a file that can only be opened by executing the reference's classes.  This module
exchanges the same information as plain tensors instead:

  * `agent_state(agent)` / `load_agent_state(agent, state)`: every network as a state_dict with the REFERENCE's
    parameter names (`encoder.shared_cnn.layers.2.weight`, `Q1.trunk.0.weight`, ...), `log_alpha`, and every optimiser
    in `torch.optim.Adam.state_dict()` layout (per-parameter `step` / `exp_avg` / `exp_avg_sq`, indexed by position
    in `module.parameters()` exactly as the reference's optimisers index them), so either side can load the other's.
  * `save(agent, path)` / `load(agent, path)`: the above through `torch.save` / `torch.load(weights_only=True)`.
  * `tools/export_reference_agent.py` (run inside the reference's own environment) turns one of its pickles into
    this format: `state = {"actor": agent.actor.state_dict(), ...}`.
"""
import torch

FORMAT = 1


def _adam_to_reference(opt, module):
    """FlatAdam over a FlatGroup -> torch.optim.Adam.state_dict() over list(module.parameters())."""
    g = opt.group
    by_id = {id(p): n for n, p in zip(g.names, g.params)}
    state, params = {}, list(module.parameters())
    for i, p in enumerate(params):
        n = by_id.get(id(p))
        step = opt.tensor_step(n) if n is not None else 0
        if n is None or step == 0:
            continue  # a tensor this optimiser never moves (the shared CNN inside the actor: detach=True)
        a, b = g.slices[n]
        state[i] = {"step": torch.tensor(float(step)), "exp_avg": opt.exp_avg[a:b].view(g.shapes[n]).detach().cpu().clone(),
                    "exp_avg_sq": opt.exp_avg_sq[a:b].view(g.shapes[n]).detach().cpu().clone()}
    return {"state": state, "param_groups": [{"lr": opt.lr, "betas": opt.betas, "eps": opt.eps, "weight_decay": 0,
                                              "amsgrad": False, "params": list(range(len(params)))}]}


def _adam_from_reference(opt, module, sd):
    g = opt.group
    by_id = {id(p): n for n, p in zip(g.names, g.params)}
    params = list(module.parameters())
    opt.exp_avg.zero_()
    opt.exp_avg_sq.zero_()
    steps = {}
    for i, st in sd["state"].items():
        n = by_id.get(id(params[int(i)]))
        if n is None:
            continue
        a, b = g.slices[n]
        opt.exp_avg[a:b].copy_(torch.as_tensor(st["exp_avg"]).reshape(-1))
        opt.exp_avg_sq[a:b].copy_(torch.as_tensor(st["exp_avg_sq"]).reshape(-1))
        steps[n] = int(float(st["step"]))
    counts = set(steps.values()) | ({0} if len(steps) < len(g.names) and steps else set())
    opt.step_count = max(counts) if counts else 0
    if len(counts) > 1:  # parameters that missed gradients: per-tensor counts, stepped in runs (flat.FlatAdam._apply_partial)
        opt._tensor_steps = {n: steps.get(n, 0) for n in g.names}


def agent_state(agent):
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}  # noqa: E731
    st = {"format": FORMAT, "algorithm": type(agent).__name__.lower(), "actor": cpu(agent.actor.state_dict()),
          "critic": cpu(agent.critic.state_dict()), "critic_target": cpu(agent.critic_target.state_dict()),
          "log_alpha": agent.log_alpha.detach().cpu().clone(),
          "actor_optimizer": _adam_to_reference(agent.actor_optimizer, agent.actor),
          "critic_optimizer": _adam_to_reference(agent.critic_optimizer, agent.critic),
          "log_alpha_optimizer": agent.log_alpha_optimizer.state_dict()}
    if hasattr(agent, "critic_weak"):
        st["critic_weak"] = cpu(agent.critic_weak.state_dict())
    return st


def load_agent_state(agent, state, strict=True, optimizers=True):
    """Inverse of agent_state().  Parameters are flat-buffer views: values are copied IN PLACE (the fused Adam / EMA /
    all-reduce keep seeing them).  The actor's copy of the shared CNN is the critic's (src/algorithms/sac.py:17-32):
    the critic is loaded last so its tensors win, as in the reference where both names alias one tensor."""
    with torch.no_grad():
        for name in ("actor", "critic_target", "critic"):
            getattr(agent, name).load_state_dict(state[name], strict=strict)
        if "critic_weak" in state and hasattr(agent, "critic_weak"):
            agent.critic_weak.load_state_dict(state["critic_weak"], strict=strict)
        agent.log_alpha.copy_(torch.as_tensor(state["log_alpha"]).to(agent.log_alpha))
    for grp in (agent.critic_group, agent.actor_group, agent.target_group):
        grp.rebind()
    if optimizers:
        if "actor_optimizer" in state:
            _adam_from_reference(agent.actor_optimizer, agent.actor, state["actor_optimizer"])
        if "critic_optimizer" in state:
            _adam_from_reference(agent.critic_optimizer, agent.critic, state["critic_optimizer"])
        if state.get("log_alpha_optimizer", {}).get("state"):
            agent.log_alpha_optimizer.load_state_dict(state["log_alpha_optimizer"])
    return agent


def save(agent, path):
    torch.save(agent_state(agent), path)


def load(agent, path, **kw):
    return load_agent_state(agent, torch.load(path, map_location="cpu", weights_only=True), **kw)
