"""Seeded synthetic inputs shared by the golden generator and the tests
(np.random.RandomState is a frozen legacy stream: identical on every box).
Shapes follow SURVEY.md 8(d): obs u8 (B,9,84,84); action U(-1,1); reward U(0,4);
not_done = 1; overlay images f32 U[0,1) (B,3,84,84)."""
import numpy as np

C, H, W = 9, 84, 84


def obs_u8(seed, B, C=C, H=H, W=W):
    return np.random.RandomState(seed).randint(0, 256, size=(B, C, H, W)).astype(np.uint8)


def overlay_imgs(seed, B, H=H, W=W):
    return np.random.RandomState(seed).random_sample((B, 3, H, W)).astype(np.float32)


def shift_offsets(seed, B, pad=4):
    rs = np.random.RandomState(seed)
    return (rs.randint(0, 2 * pad + 1, size=B).astype(np.int32),
            rs.randint(0, 2 * pad + 1, size=B).astype(np.int32))


def conv_weights(seed, B):
    return np.random.RandomState(seed).standard_normal((B, 3, 3, 3, 3)).astype(np.float32)


def jitter_params(seed):
    rs = np.random.RandomState(seed)
    order = rs.permutation(4).astype(np.int32)
    return order, float(rs.uniform(0, 2)), float(rs.uniform(0, 3)), float(rs.uniform(-0.5, 0.5))


def transition_batch(seed, B, A):
    """(obs_u8, action, reward, next_obs_u8, not_done) as numpy arrays."""
    rs = np.random.RandomState(seed)
    obs = rs.randint(0, 256, size=(B, C, H, W)).astype(np.uint8)
    next_obs = rs.randint(0, 256, size=(B, C, H, W)).astype(np.uint8)
    action = rs.uniform(-1, 1, size=(B, A)).astype(np.float32)
    reward = rs.uniform(0, 4, size=(B, 1)).astype(np.float32)
    not_done = np.ones((B, 1), dtype=np.float32)
    return obs, action, reward, next_obs, not_done


def smooth_obs_u8(seed, B, C=C, H=H, W=W):
    """Image-like (spatially correlated) frames: exercises hue/saturation branches
    that uniform noise rarely hits (gray pixels, saturated primaries)."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    out = np.empty((B, C, H, W), dtype=np.uint8)
    for b in range(B):
        for c in range(C):
            fx, fy, ph = rs.uniform(0.02, 0.2), rs.uniform(0.02, 0.2), rs.uniform(0, 6.28)
            img = 127.5 + 127.5 * np.sin(fx * xx + fy * yy + ph)
            out[b, c] = np.clip(img, 0, 255).astype(np.uint8)
    # a flat gray patch and saturated primaries (eqc / maxc==r branches of rgb2hsv)
    out[:, :, :6, :6] = 90
    out[:, 0::3, 6:12, :6] = 255
    out[:, 1::3, 6:12, :6] = 0
    out[:, 2::3, 6:12, :6] = 0
    return out


def replay_stream(seed, n, A, ep_len=7, stack=3, H=H, W=W):
    """Yield n transitions (obs_frames, action, reward, next_frames, done) of an episodic stream with the frame
    sharing of src/env/wrappers.py:177-206 (FrameStack): a reset replicates the first frame `stack` times, every
    step appends ONE new (3,H,W) uint8 frame; consecutive stacks share their frame arrays (identity)."""
    rs = np.random.RandomState(seed)
    new = lambda: rs.randint(0, 256, size=(3, H, W)).astype(np.uint8)  # noqa: E731
    t = 0
    frames = None
    for _ in range(n):
        if frames is None:
            f0 = new()
            frames = [f0] * stack
            t = 0
        nxt = frames[1:] + [new()]
        t += 1
        done = t >= ep_len
        yield list(frames), rs.uniform(-1, 1, size=A).astype(np.float32), float(rs.uniform(0, 4)), list(nxt), done
        frames = None if done else nxt


def green_frame_u8(seed, H=100, W=100):
    """A rendered-frame-like (3,H,W) uint8 image: greenish background (several hues / saturations around the keying
    thresholds of src/env/wrappers.py:246-248), a non-green 'body', plus noise so that every threshold is crossed."""
    rs = np.random.RandomState(seed)
    img = np.empty((3, H, W), np.uint8)
    img[0] = rs.randint(0, 120, (H, W))
    img[1] = rs.randint(60, 256, (H, W))
    img[2] = rs.randint(0, 140, (H, W))
    img[:, H // 3:2 * H // 3, W // 4:3 * W // 4] = rs.randint(0, 256, (3, H // 3 + (H % 3 > 0) * 0, W // 2)).astype(np.uint8)[:, :2 * H // 3 - H // 3]
    img[:, :4, :4] = 128  # gray: minc == maxc branch
    img[:, 4:8, :4] = np.array([0, 255, 0], np.uint8)[:, None, None]  # pure green
    return img
