"""Replay sampling: CPU -- the numpy oracle against golden vectors from the reference's own
ReplayBuffer; GPU -- the device-resident ring + gather kernel against both, bit for bit."""
import os

import numpy as np
import pytest

import inputs as gi
from oracle.replay_oracle import ReplayOracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A, CAP, B = 2, 32, 3


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(ROOT, "tests", "golden", "replay_golden.npz"))


def test_oracle_matches_reference_buffer(G):
    orc = ReplayOracle((A,), CAP)
    for t in gi.replay_stream(seed=21, n=45, A=A):
        orc.add(*t)
    assert orc.idx == int(G["idx_after"][0]) and orc.full == bool(G["full"][0])
    for k, s in enumerate((5, 6)):
        np.random.seed(s)
        idxs = orc.draw(B)
        assert np.array_equal(idxs, G[f"s{k}/idxs"])
        obs, act, rew, nobs, nd = orc.sample(idxs)
        assert np.array_equal(obs, G[f"s{k}/obs"].astype(np.float32))
        assert np.array_equal(nobs, G[f"s{k}/next_obs"].astype(np.float32))
        assert np.array_equal(act, G[f"s{k}/action"]) and np.array_equal(rew, G[f"s{k}/reward"])
        assert np.array_equal(nd, G[f"s{k}/not_done"])


@pytest.mark.gpu
def test_device_ring_matches_reference_golden(cuda, G):
    import torch

    from augcl_amd.replay import LazyFrames, ReplayBuffer
    buf = ReplayBuffer((9, 84, 84), (A,), capacity=CAP, batch_size=B)
    for of, a, r, nf, d in gi.replay_stream(seed=21, n=45, A=A):
        buf.add(LazyFrames(of), a, r, LazyFrames(nf), d)
    assert buf.idx == int(G["idx_after"][0]) and buf.full == bool(G["full"][0])
    assert buf.frame_seq == 45 + 7  # one new frame per step + one per episode start: frames are stored once
    for k, s in enumerate((5, 6)):
        np.random.seed(s)
        obs, act, rew, nobs, nd = buf.sample()
        assert obs.dtype == torch.float32 and obs.is_cuda
        assert np.array_equal(obs.cpu().numpy(), G[f"s{k}/obs"].astype(np.float32))
        assert np.array_equal(nobs.cpu().numpy(), G[f"s{k}/next_obs"].astype(np.float32))
        assert np.array_equal(act.cpu().numpy(), G[f"s{k}/action"])
        assert np.array_equal(rew.cpu().numpy(), G[f"s{k}/reward"])
        assert np.array_equal(nd.cpu().numpy(), G[f"s{k}/not_done"])
        u8 = buf.sample(idxs=G[f"s{k}/idxs"], dtype=torch.uint8)[0]
        assert u8.dtype == torch.uint8 and np.array_equal(u8.cpu().numpy(), G[f"s{k}/obs"])


@pytest.mark.gpu
def test_device_ring_vs_oracle_long_stream_and_ndarray_obs(cuda):
    import torch

    from augcl_amd.replay import ReplayBuffer
    cap = 64
    buf = ReplayBuffer((9, 84, 84), (A,), capacity=cap, batch_size=16)
    orc = ReplayOracle((A,), cap)
    for of, a, r, nf, d in gi.replay_stream(seed=3, n=200, A=A, ep_len=11):
        buf.add(np.concatenate(of, axis=0), a, r, np.concatenate(nf, axis=0), d)  # plain (9,84,84) arrays
        orc.add(of, a, r, nf, d)
    assert buf.frame_seq <= 200 + 200 // 11 + 2  # content match de-duplicates consecutive stacks
    for s in range(3):
        np.random.seed(100 + s)
        idxs = orc.draw(16)
        np.random.seed(100 + s)
        obs, act, rew, nobs, nd = buf.sample()
        o2, a2, r2, n2, d2 = orc.sample(idxs)
        assert np.array_equal(obs.cpu().numpy(), o2) and np.array_equal(nobs.cpu().numpy(), n2)
        assert np.array_equal(act.cpu().numpy(), a2) and np.array_equal(rew.cpu().numpy(), r2)
        assert np.array_equal(nd.cpu().numpy(), d2)
    sd = buf.state_dict()
    buf2 = ReplayBuffer((9, 84, 84), (A,), capacity=cap, batch_size=16)
    buf2.load_state_dict(sd)
    np.random.seed(7)
    x1 = buf.sample()[0]
    np.random.seed(7)
    assert torch.equal(x1, buf2.sample()[0])


@pytest.mark.gpu
def test_ring_overwrite_is_detected_and_agent_consumes_samples(cuda):
    import torch

    from augcl_amd import _lib
    from augcl_amd import utils as au
    from augcl_amd.algorithms import make_agent
    from augcl_amd.replay import LazyFrames, ReplayBuffer
    from oracle import sac_oracle as so
    small = ReplayBuffer((9, 84, 84), (A,), capacity=16, batch_size=4, frame_slack=-8)  # deliberately too small
    with pytest.raises(_lib.AugclError, match="frame_slack"):  # at the add() that evicts a live transition's frame
        for of, a, r, nf, d in gi.replay_stream(seed=4, n=16, A=A):
            small.add(LazyFrames(of), a, r, LazyFrames(nf), d)
    # short episodes (2 steps: a fresh frame stack every other transition) overrun the default slack of a wrapped
    # buffer: reported at once, with the slack this stream needs -- and a ring built with it holds
    cap = 64
    short = ReplayBuffer((9, 84, 84), (A,), capacity=cap, batch_size=4)
    with pytest.raises(_lib.AugclError, match="frame_slack") as ei:
        for of, a, r, nf, d in gi.replay_stream(seed=4, n=4 * cap, A=A, ep_len=2):
            short.add(LazyFrames(of), a, r, LazyFrames(nf), d)
    need = int(str(ei.value).split("frame_slack=")[1].split(")")[0])
    roomy = ReplayBuffer((9, 84, 84), (A,), capacity=cap, batch_size=4, frame_slack=need)
    for of, a, r, nf, d in gi.replay_stream(seed=4, n=4 * cap, A=A, ep_len=2):
        roomy.add(LazyFrames(of), a, r, LazyFrames(nf), d)
    roomy.sample()
    buf = ReplayBuffer((9, 84, 84), (A,), capacity=64, batch_size=8)
    for of, a, r, nf, d in gi.replay_stream(seed=5, n=40, A=A):
        buf.add(LazyFrames(of), a, r, LazyFrames(nf), d)
    agent = make_agent((9, 84, 84), (A,), au.dotdict(algorithm="drq", data_aug="identity", aug_params=None, **so.DEFAULTS))
    before = agent.critic_group.param.clone()
    np.random.seed(0)
    agent.update(buf, None, 0)
    assert not torch.equal(before, agent.critic_group.param)


@pytest.mark.gpu
def test_chunked_save_and_load_like_the_reference(cuda, tmp_path):
    """save / requeue_save / load keep the reference's "<start>_<end>.tar" incremental chunks and bookkeeping
    (src/utils.py:156-204); a buffer refilled from the chunks samples the same batches bit for bit."""
    import torch

    from augcl_amd.replay import LazyFrames, ReplayBuffer
    cap = 64
    rb = ReplayBuffer((9, 84, 84), (A,), cap, B, device=cuda)
    stream = list(gi.replay_stream(seed=31, n=40, A=A))
    d = str(tmp_path)
    for i, (fo, a, r, fn, done) in enumerate(stream):
        rb.add(LazyFrames(fo), a, r, LazyFrames(fn), done)
        if i in (14, 29):
            rb.save(d)
    rb.save(d)
    rb.save(d)  # nothing new: no empty chunk
    assert sorted(os.listdir(d), key=lambda x: int(x.split("_")[0])) == ["0_15.tar", "15_30.tar", "30_40.tar"]
    assert rb.last_save == 40
    payload = torch.load(os.path.join(d, "15_30.tar"), weights_only=True)  # plain tensors only
    assert payload["frames"].dtype == torch.uint8 and payload["frames"].shape[0] == payload["seqs"].numel()

    rb2 = ReplayBuffer((9, 84, 84), (A,), cap, B, device=cuda)
    rb2.load(d)
    assert rb2.idx == 40 and rb2.last_save == 40 and rb2.last_requeue_save == 40
    for seed in (1, 2, 3):
        np.random.seed(seed)
        want = rb.sample()
        np.random.seed(seed)
        got = rb2.sample()
        for x, y in zip(want, got):
            assert torch.equal(x, y)
    # frames shared between chunks are stored once
    assert rb2.frame_seq == rb.frame_seq
    # after the ring wraps (idx < last_save) the reference's slices are empty and it still writes "<a>_<b>.tar": same
    d2 = str(tmp_path) + "_wrapped"  # a sibling: load(d) lists d and expects nothing but chunks there
    os.makedirs(d2)
    for fo, a, r, fn, done in gi.replay_stream(seed=32, n=cap - 40 + 5, A=A):
        rb.add(LazyFrames(fo), a, r, LazyFrames(fn), done)
    assert rb.full and rb.idx == 5
    rb.save(d2)
    assert os.listdir(d2) == ["40_5.tar"] and rb.last_save == 5
    assert torch.load(os.path.join(d2, "40_5.tar"), weights_only=True)["seqs"].numel() == 0
    # end_step stops after the chunk that ends there; a gap in the chunk sequence is refused like the reference
    rb3 = ReplayBuffer((9, 84, 84), (A,), cap, B, device=cuda)
    rb3.load(d, end_step=30)
    assert rb3.idx == 30 and rb3._adds == 30
    # the counters of a resumed run (src/utils.py:177-204 restores idx; here also the transitions ever stored): load() and
    # load_state_dict() carry them over, and the ring-overwrite message of a RESUMED buffer asks for the slack its stream
    # needs -- the same figure a buffer that saw the whole stream reports, not one inflated by the lost counters
    assert rb2._adds == 40
    rb4 = ReplayBuffer((9, 84, 84), (A,), cap, B, device=cuda)
    rb4.load_state_dict(rb2.state_dict())
    assert rb4._adds == 40 and rb4.idx == 40
    from augcl_amd import _lib

    def overrun(buf):
        with pytest.raises(_lib.AugclError, match="frame_slack") as ei:
            for fo, a, r, fn, done in gi.replay_stream(seed=33, n=6 * cap, A=A, ep_len=2):
                buf.add(LazyFrames(fo), a, r, LazyFrames(fn), done)
        msg = str(ei.value)
        return int(msg.split("frame_slack=")[1].split(")")[0]), int(msg.split("(")[1].split(" transitions")[0])
    fresh = ReplayBuffer((9, 84, 84), (A,), cap, B, device=cuda)
    for fo, a, r, fn, done in stream:
        fresh.add(LazyFrames(fo), a, r, LazyFrames(fn), done)
    (need_f, adds_f), (need_r, adds_r), (need_s, adds_s) = overrun(fresh), overrun(rb2), overrun(rb4)
    assert adds_f == adds_r == adds_s and adds_f > 40
    assert need_r == need_f == need_s and need_f < 4 * cap  # 2-step episodes: ~2 new frames per transition, not dozens
    os.remove(os.path.join(d, "0_15.tar"))
    with pytest.raises(AssertionError):
        ReplayBuffer((9, 84, 84), (A,), cap, B, device=cuda).load(d)


@pytest.mark.gpu
def test_fused_sample_shift_conv_equals_sample_then_front_end(cuda):
    """ReplayBuffer.sample_shift_conv (gather -> random_shift -> random_conv in ONE pass over the uint8 ring,
    src/utils.py:206-240 + src/algorithms/svea.py:66-68,49) == sample() followed by augmentations.shift_conv, bit for
    bit: explicit parameters and the in-kernel Philox streams, the same index draw, and SVEA consumes it."""
    import torch

    from augcl_amd import augmentations as aug
    from augcl_amd import utils as au
    from augcl_amd.algorithms import make_agent
    from augcl_amd.replay import LazyFrames, ReplayBuffer
    from oracle import sac_oracle as so
    rb = ReplayBuffer((9, 84, 84), (A,), capacity=96, batch_size=B, device=cuda)
    for of, a, r, nf, d in gi.replay_stream(seed=41, n=80, A=A):
        rb.add(LazyFrames(of), a, r, LazyFrames(nf), d)
    rs = np.random.RandomState(3)
    oy, ox = rs.randint(0, 9, B).astype(np.int32), rs.randint(0, 9, B).astype(np.int32)
    w = rs.standard_normal((B, 3, 3, 3, 3)).astype(np.float32)
    for explicit in (True, False):
        kw = dict(oy=oy, ox=ox, weights=w) if explicit else {}
        np.random.seed(5)
        aug.manual_seed(77)
        obs, act, rew, nobs, nd = rb.sample()
        want = aug.shift_conv(obs, stacked=True, **kw)
        np.random.seed(5)
        aug.manual_seed(77)
        both, act2, rew2, nobs2, nd2 = rb.sample_shift_conv(**kw)
        assert both.shape == (2 * B, 9, 84, 84)
        assert torch.equal(both, want), "fused front end differs from sample() + shift_conv"
        for x, y in ((act, act2), (rew, rew2), (nobs, nobs2), (nd, nd2)):
            assert torch.equal(x, y)
    # out / out_next: the same values written into caller-owned tensors (a hipGraph's static inputs)
    np.random.seed(21)
    aug.manual_seed(5)
    ref = rb.sample_shift_conv()
    np.random.seed(21)
    aug.manual_seed(5)
    o1, o2 = torch.full_like(ref[0], float("nan")), torch.full_like(ref[3], float("nan"))
    got = rb.sample_shift_conv(out=o1, out_next=o2)
    assert got[0].data_ptr() == o1.data_ptr() and got[3].data_ptr() == o2.data_ptr()
    assert torch.equal(o1, ref[0]) and torch.equal(o2, ref[3])
    # SVEA through the fused path walks the same trajectory as through sample() + its own front end
    # (bit for bit with float32 next_obs; with next_obs kept uint8 into the target pass -- the default on the ring, ABI v5 -- the
    # first layer's products are accumulated in another order: float32 rounding)
    snaps = []
    for fused, u8 in ((True, False), (False, False), (True, True)):
        torch.manual_seed(3)
        aug.manual_seed(9)
        np.random.seed(9)
        agent = make_agent((9, 84, 84), (A,), au.dotdict(algorithm="svea", data_aug="conv", aug_params=None, **so.DEFAULTS))
        agent.u8_next_obs = u8
        buf = rb if fused else type("Plain", (), {"sample": lambda self, n=None: rb.sample(), "batch_size": B})()
        for step in range(3):
            agent.update(buf, None, step)
        torch.cuda.synchronize()
        snaps.append((agent.critic_group.param.clone(), agent.actor_group.param.clone()))
    assert torch.equal(snaps[0][0], snaps[1][0]) and torch.equal(snaps[0][1], snaps[1][1])
    for x, y in zip(snaps[0], snaps[2]):
        assert float((x.double() - y.double()).abs().max()) <= 1e-5 * max(1.0, float(x.abs().max()))


@pytest.mark.gpu
def test_svea_hip_graph_update_on_the_device_ring_and_other_frame_sizes(cuda):
    """SVEA.update() against a REAL ReplayBuffer with hipGraphs enabled: the fused sample -> shift -> conv launch writes
    straight into the captured graph's static input (SVEA._sample, out=static[0]); the trajectory equals the eager one.
    A ring whose frames are not 84 x 84 / 100 x 100 (no fused kernel) takes sample() + the un-fused front end instead of
    raising inside update()."""
    import torch

    from augcl_amd import augmentations as aug
    from augcl_amd import utils as au
    from augcl_amd.algorithms import make_agent
    from augcl_amd.replay import LazyFrames, ReplayBuffer
    from oracle import sac_oracle as so
    rb = ReplayBuffer((9, 84, 84), (A,), capacity=96, batch_size=B, device=cuda)
    for of, a, r, nf, d in gi.replay_stream(seed=43, n=80, A=A):
        rb.add(LazyFrames(of), a, r, LazyFrames(nf), d)
    snaps = []
    for graph in (False, True):
        torch.manual_seed(3)
        aug.manual_seed(9)
        np.random.seed(9)
        agent = make_agent((9, 84, 84), (A,), au.dotdict(algorithm="svea", data_aug="conv", aug_params=None, **so.DEFAULTS))
        if graph:
            agent.enable_hip_graphs(warmup=2)
        for step in range(8):
            agent.update(rb, None, step)
        torch.cuda.synchronize()
        if graph:
            assert len(agent._graph["graphs"]) >= 2 and agent._graph["static"][0].shape[0] == 2 * B
        snaps.append((agent.critic_group.param.clone(), agent.actor_group.param.clone(), agent.log_alpha.detach().clone()))
    for x, y in zip(*snaps):
        assert float((x.double() - y.double()).abs().max()) <= 1e-5 * max(1.0, float(x.abs().max()))
    # next_obs travelled as uint8 from the ring into the target pass's first conv layer (ABI v5); the same trajectory with float32
    # next_obs (u8_next_obs = False: the round 1-4 path) agrees to float32 rounding
    assert agent._graph["static"][3].dtype == torch.uint8 and agent._u8_next_obs()
    torch.manual_seed(3)
    aug.manual_seed(9)
    np.random.seed(9)
    agent32 = make_agent((9, 84, 84), (A,), au.dotdict(algorithm="svea", data_aug="conv", aug_params=None, **so.DEFAULTS))
    agent32.u8_next_obs = False
    agent32.enable_hip_graphs(warmup=2)
    for step in range(8):
        agent32.update(rb, None, step)
    torch.cuda.synchronize()
    assert agent32._graph["static"][3].dtype == torch.float32
    for x, y in zip(snaps[1], (agent32.critic_group.param, agent32.actor_group.param, agent32.log_alpha.detach())):
        assert float((x.double() - y.double()).abs().max()) <= 1e-5 * max(1.0, float(x.abs().max()))
    # another geometry: the gate looks at the ring's own frame size
    odd = ReplayBuffer((9, 64, 64), (A,), capacity=32, batch_size=4, device=cuda)
    for of, a, r, nf, d in gi.replay_stream(seed=44, n=20, A=A, H=64, W=64):
        odd.add(LazyFrames(of), a, r, LazyFrames(nf), d)
    assert not agent._fused_front_end(None, hw=(odd.h, odd.w)) and agent._fused_front_end(None, hw=(rb.h, rb.w))
    got = agent._sample(odd)
    assert len(got) == 5 and got[0].shape == (4, 9, 64, 64)  # plain sample(), not the (2B, ...) fused stack


@pytest.mark.gpu
def test_device_index_stream_is_launch_only_and_equals_explicit_indices(cuda):
    """index_stream="device": the transition indices of sample() / sample_shift_conv() are drawn by a Philox launch
    (augcl_philox_replay_indices, keyed like the augmentations) instead of np.random.randint + an H2D copy
    (src/utils.py:229-232).  RNG-internal == explicit: the batch equals sample(idxs=<the exported indices>) bit for bit;
    numpy's generator is not consumed; the indices are in range and spread over the filled part."""
    import torch

    from augcl_amd import _lib
    from augcl_amd import augmentations as aug
    from augcl_amd.replay import LazyFrames, ReplayBuffer
    rb = ReplayBuffer((9, 84, 84), (A,), capacity=96, batch_size=B, device=cuda, index_stream="device")
    for of, a, r, nf, d in gi.replay_stream(seed=45, n=70, A=A):  # not full: n_valid = 70
        rb.add(LazyFrames(of), a, r, LazyFrames(nf), d)
    aug.manual_seed(123)
    np.random.seed(7)
    state = np.random.get_state()[1].copy()
    got = rb.sample()
    used_idx = rb._idx_dev.clone()
    assert np.array_equal(np.random.get_state()[1], state), "the device index stream must not consume numpy's generator"
    idx = torch.empty(B, dtype=torch.int64, device=cuda)
    _lib.call("augcl_philox_replay_indices", 123, 0, 0, 70, idx.data_ptr(), B, _lib.current_stream(cuda))
    assert torch.equal(idx, used_idx) and int(idx.min()) >= 0 and int(idx.max()) < 70
    want = rb.sample(idxs=idx.cpu().numpy())
    for x, y in zip(got, want):
        assert torch.equal(x, y)
    # the fused front end draws its indices the same way (call 1 of the stream: the index draw; call 2: shift + conv)
    aug.manual_seed(123)
    both, act, rew, nobs, nd = rb.sample_shift_conv()
    aug.manual_seed(123)
    aug._next_call()
    want2 = rb.sample_shift_conv(idxs=idx.cpu().numpy())
    for x, y in zip((both, act, rew, nobs, nd), want2):
        assert torch.equal(x, y)
    # spread: 4096 draws over 70 transitions hit every one of them, none twice as often as uniform allows by chance
    big = torch.empty(4096, dtype=torch.int64, device=cuda)
    _lib.call("augcl_philox_replay_indices", 5, 9, 0, 70, big.data_ptr(), 4096, _lib.current_stream(cuda))
    cnt = torch.bincount(big, minlength=70)
    assert int(cnt.min()) > 20 and int(cnt.max()) < 110 and cnt.numel() == 70
