"""Host side of the live-policy TagGridWorld rollout kernel (tag_gridworld_n5_policy.hip): the packing of a
two-hidden-layer policy into the layout the kernel reads, and the float32 restatement of its forward (test
infrastructure, like oracle/cartpole_np.py::policy_probabilities for the Cartpole counterpart)."""
import numpy as np

IN, IN_STRIDE, ACTIONS = 21, 24, 5


def policy_floats(hidden):
    H = int(hidden)
    return (H * IN_STRIDE + H + H * H + H + ACTIONS * H + ACTIONS + 3) // 4 * 4


def pack(W0, b0, W1, b1, Wp, bp):
    """W0 [H, 21], b0 [H], W1 [H, H], b1 [H], Wp [5, H], bp [5] -> the flat float32 block of one policy"""
    H = W0.shape[0]
    assert W0.shape == (H, IN) and W1.shape == (H, H) and Wp.shape == (ACTIONS, H)
    w0 = np.zeros((H, IN_STRIDE), np.float32)
    w0[:, :IN] = W0
    flat = np.concatenate([w0.ravel(), b0, W1.ravel(), b1, Wp.ravel(), bp]).astype(np.float32)
    out = np.zeros(policy_floats(H), np.float32)
    out[:flat.size] = flat
    return out


def pack_model(model):
    """training.models.FullyConnected with fc_dims [H, H] and one head of 5 actions"""
    g = lambda t: t.detach().cpu().numpy().astype(np.float32)  # noqa: E731
    return pack(g(model.fc["0"][0].weight), g(model.fc["0"][0].bias), g(model.fc["1"][0].weight),
                g(model.fc["1"][0].bias), g(model.policy_head[0].weight), g(model.policy_head[0].bias))


def probabilities(packed, hidden, obs):
    """obs [R, 21] float32 -> probabilities [R, 5] float32: acc = bias, one fused multiply-add per input in index
    order (emulated in float64: the product of two float32 is exact there), ReLU, softmax with the maximum
    subtracted -- what gw5_policy_cum computes"""
    f32, H = np.float32, int(hidden)
    w = np.asarray(packed, dtype=f32)
    o = 0
    W0 = w[o:o + H * IN_STRIDE].reshape(H, IN_STRIDE)[:, :IN]; o += H * IN_STRIDE
    b0 = w[o:o + H]; o += H
    W1 = w[o:o + H * H].reshape(H, H); o += H * H
    b1 = w[o:o + H]; o += H
    Wp = w[o:o + ACTIONS * H].reshape(ACTIONS, H); o += ACTIONS * H
    bp = w[o:o + ACTIONS]

    def layer(x, W, b):
        acc = np.broadcast_to(b, (x.shape[0], W.shape[0])).astype(f32).copy()
        for j in range(W.shape[1]):
            acc = (W[None, :, j].astype(np.float64) * x[:, j:j + 1].astype(np.float64) + acc.astype(np.float64)).astype(f32)
        return acc

    x = np.asarray(obs, dtype=f32)
    h1 = np.maximum(layer(x, W0, b0), f32(0))
    h2 = np.maximum(layer(h1, W1, b1), f32(0))
    logits = layer(h2, Wp, bp)
    e = np.exp((logits - logits.max(axis=1, keepdims=True)).astype(f32)).astype(f32)
    s = np.zeros(e.shape[0], f32)
    for a in range(ACTIONS):
        s = (s + e[:, a]).astype(f32)
    return (e / s[:, None]).astype(f32)


def running_sums(p):
    """the float32 running sums the inverse-CDF sampler compares the uniform with (random.cu:51-85)"""
    c = np.zeros_like(p)
    acc = np.zeros(p.shape[0], np.float32)
    for a in range(p.shape[1]):
        acc = p[:, a] if a == 0 else (acc + p[:, a]).astype(np.float32)
        c[:, a] = acc
    return c
