"""Stability soak: N update() steps in graph and eager mode; reports finite parameters and allocator growth.
python tools/soak_update.py [N] [algo] [B]   (default svea 256; e.g. rad 256 / svea 128: the critic pass of at most 256 images runs
the gradient-carrying forward through the storing chains, augcl_enc_stack_forward_chained)"""
import sys, torch
sys.path.insert(0, '.')
from augcl_amd import bench_update, encoder_ops, utils as au
from augcl_amd.algorithms import make_agent
dev = torch.device('cuda:0')
hp = dict(discount=0.99, critic_tau=0.01, encoder_tau=0.05, actor_update_freq=2, critic_target_update_freq=2,
          num_shared_layers=11, num_head_layers=0, num_filters=32, projection_dim=100, hidden_dim=1024,
          actor_log_std_min=-10.0, actor_log_std_max=2.0, init_temperature=0.1, actor_lr=1e-3, actor_beta=0.9,
          critic_lr=1e-3, critic_beta=0.9, alpha_lr=1e-4, alpha_beta=0.5, svea_alpha=0.5, svea_beta=0.5, drq_k=1, drq_m=1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ALGO = sys.argv[2] if len(sys.argv) > 2 else "svea"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
AUG = "conv" if ALGO in ("svea", "augcl") else "shift"
for precision in ("f32", "bf16"):
    for graph in (True, False):
        encoder_ops.set_precision(precision)
        torch.manual_seed(0)
        agent = make_agent((9, 84, 84), (6,), au.dotdict(algorithm=ALGO, data_aug=AUG, aug_params=None, **hp))
        if graph:
            agent.enable_hip_graphs()
        buf = bench_update._SyntheticReplay(dev, B, 6, seed=1)
        for s in range(20):
            agent.update(buf, None, s)
        torch.cuda.synchronize()
        m0 = torch.cuda.memory_allocated()
        for s in range(20, 20 + N):
            agent.update(buf, None, s)
        torch.cuda.synchronize()
        m1 = torch.cuda.memory_allocated()
        ok = all(bool(torch.isfinite(g.param).all()) for g in (agent.critic_group, agent.actor_group, agent.target_group))
        print(ALGO, "B", B, precision, "graph" if graph else "eager", "steps", N, "finite", ok, "alpha %.4f" % float(agent.alpha),
              "mem growth MB %.1f" % ((m1 - m0) / 1e6), "critic |p| %.4f" % float(agent.critic_group.param.abs().mean()))
        del agent
encoder_ops.set_precision("f32")
