import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/golden")
import test_oracle_golden_sac as T
import test_update_gpu as U
from augcl_amd import augmentations as aug
G = np.load(ROOT + "/tests/golden/sac_golden.npz")
cuda = torch.device("cuda:0")
which = sys.argv[1:] or ["sac_cartpole", "nnrad_shift_cartpole"]
for (name, algo, data_aug, A, ci) in T.CASES:
    if name not in which: continue
    actor_state, critic_state, batch, noises, ov = T.make_case(name, algo, data_aug, A, ci)
    agent = U.build_agent(algo, data_aug, A, actor_state, critic_state, cuda)
    agent.fuse_front_end = False
    dbatch = tuple(t.to(cuda) for t in batch)
    buf, L = U.FakeBuffer(dbatch), U.RecLogger()
    queue = [n.to(cuda) for n in noises]
    real = torch.randn_like
    torch.randn_like = lambda t, *a, **k: queue.pop(0).clone()
    state = {"s": 0, "c": 0}
    rs, rc = aug.random_shift, aug.random_conv
    def sh(x, pad=4, **kw):
        second = state["c"] == 1 and (algo == "drq" or (algo == "rad" and data_aug == "shift")); state["c"] += 1
        return rs(x, pad, oy=G[f"{name}/oy2" if second else f"{name}/oy"][state["s"]], ox=G[f"{name}/ox2" if second else f"{name}/ox"][state["s"]])
    aug.random_shift = sh
    if hasattr(agent, "aug_funcs"):
        if "shift" in agent.aug_funcs: agent.aug_funcs["shift"]["func"] = sh
        if "conv" in agent.aug_funcs: agent.aug_funcs["conv"]["func"] = lambda x, **kw: rc(x, weights=G[f"{name}/conv_w"][state["s"]])
    keys = [str(k) for k in G[f"{name}/log_keys"]]; want = G[f"{name}/logs"]
    for s in range(3):
        state["s"], state["c"] = s, 0
        agent.update(buf, L, 2 + s)
        for j, k in enumerate(keys):
            if not np.isnan(want[s, j]) and "variance" not in k:
                g = L.rows[(2 + s, k)]; print(name, s, k, g, want[s, j], "rel=%.2e" % (abs(g - want[s, j]) / max(abs(want[s, j]), 1e-9)))
        with torch.no_grad(): q1, q2 = agent.critic(dbatch[0], dbatch[1])
        print("  q_after maxdiff", np.abs(np.stack([q1.cpu().numpy().ravel(), q2.cpu().numpy().ravel()]) - G[f"{name}/q_after"][s]).max())
    torch.randn_like = real; aug.random_shift = rs
