import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/golden")
import test_oracle_golden_sac as T
import test_update_gpu as U
from augcl_amd import augmentations as aug
from oracle import sac_oracle as so, aug_oracle as ao
G = np.load(ROOT + "/tests/golden/sac_golden.npz")
cuda = torch.device("cuda:0")
name, algo, data_aug, A, ci = T.CASES[1]
actor_state, critic_state, batch, noises, ov = T.make_case(name, algo, data_aug, A, ci)
agent = U.build_agent(algo, data_aug, A, actor_state, critic_state, cuda)
orc = so.OracleAgent(algo, T.OBS, A, actor_state, critic_state)
oy, ox = G[f"{name}/oy"][0], G[f"{name}/ox"][0]
obs_s = torch.from_numpy(ao.random_shift_c(batch[0].numpy(), oy, ox))
# oracle: critic loss + grads
y = orc.target_q(batch[3], batch[2], batch[4], noises[0])
q1, q2 = so.critic_forward(orc.critic, obs_s, batch[1], orc.hp)
lo = orc._mse2(q1, q2, y); lo.backward()
# product
d = lambda t: t.to(cuda)
real = torch.randn_like
torch.randn_like = lambda t, *a, **k: d(noises[0]).clone()
obs_g = aug.random_shift(d(batch[0]), oy=oy, ox=ox)
print("shift equal", torch.equal(obs_g.cpu(), obs_s))
yg = agent.calculate_target_Q(d(batch[3]), d(batch[2]), d(batch[4]))
torch.randn_like = real
print("target diff", (yg.cpu() - y).abs().max().item())
g1, g2 = agent.critic(obs_g, d(batch[1]))
lg = agent._q_loss(g1, g2, yg)
agent.critic_optimizer.zero_grad(); lg.backward()
print("loss", float(lo), float(lg))
for n, p in agent.critic.named_parameters():
    go = orc.critic[n].grad; gg = p.grad.cpu()
    print(n, "rel", float((go - gg).abs().max() / go.abs().max().clamp_min(1e-30)), float(go.abs().max()))
