#!/usr/bin/env python3
"""Run INSIDE the reference's own environment (its `src/` on sys.path): turn one of its whole-object agent pickles
(`torch.save(agent, "<step>.pt")`, src/train.py:224-232) into the plain-tensor state augcl_amd.checkpoint.load() reads.

    python export_reference_agent.py /path/to/500000.pt out_state.pt
"""
import sys

import torch

agent = torch.load(sys.argv[1])  # executes the reference's classes: only do this with checkpoints you trust
cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}  # noqa: E731
state = {"format": 1, "algorithm": type(agent).__name__.lower(), "actor": cpu(agent.actor.state_dict()),
         "critic": cpu(agent.critic.state_dict()), "critic_target": cpu(agent.critic_target.state_dict()),
         "log_alpha": agent.log_alpha.detach().cpu(), "actor_optimizer": agent.actor_optimizer.state_dict(),
         "critic_optimizer": agent.critic_optimizer.state_dict(), "log_alpha_optimizer": agent.log_alpha_optimizer.state_dict()}
if hasattr(agent, "critic_weak"):
    state["critic_weak"] = cpu(agent.critic_weak.state_dict())
torch.save(state, sys.argv[2])
print("wrote", sys.argv[2])
