"""The rollout's policy forward as ONE launch (csrc/kernels/policy_mlp.hip).

`FusedPolicyForward` wraps a `FullyConnected` policy (training/models.py; reference
models/fully_connected.py:46-120) with two hidden layers of equal width 64 / 128 / 256, observation
rows of up to 96 floats and one or two softmax heads: the kernel reads the observation rows of this
policy's agents in place ([E, N, F], the env's own array), keeps the activations in registers
(float32 MFMA, the framework's arithmetic up to summation order) and writes the probabilities
straight into the sampler's [E, N, A_h] tensors; optionally the value estimates and a copy of the rows
into the training batch.

The weights are re-packed into the order the kernel's wavefronts read them (`pack()`: call it after
every optimizer step; a few hundred KB of gathers)."""
import numpy as np
import torch

_WAVE_ROWS = 32          # observation rows (agents) per wavefront
_OUT_TILES = 2           # output rows padded to 64: all head logits + the value


def _row_of(s, h):
    """row inside a 32-row tile of accumulator register s, lane half h (32x32 MFMA C/D layout)"""
    return (s & 3) + 8 * (s >> 2) + 4 * h


def _pack_indices(n_out_tiles, n_k_tiles, first_layer):
    """(row, col) gather indices of the packed weight tensor [KT, TN, 4, 64, 4]: lane l supplies
    A[i = l & 31][k of (step, l >> 5)]; a step of the first layer contracts features
    32 kt + 16 h + s, a step of the later layers the rows (32 kt + row_of(s, h)) the previous layer's
    accumulators hold in that register."""
    kt, tn, s4, lane, e = np.meshgrid(np.arange(n_k_tiles), np.arange(n_out_tiles), np.arange(4), np.arange(64),
                                      np.arange(4), indexing="ij")
    s, h = 4 * s4 + e, lane >> 5
    rows = tn * 32 + (lane & 31)
    cols = 32 * kt + (16 * h + s if first_layer else _row_of(s, h))
    return rows, cols


def _bias_indices(n_tiles):
    tn, h, s = np.meshgrid(np.arange(n_tiles), np.arange(2), np.arange(16), indexing="ij")
    return tn * 32 + _row_of(s, h)


class FusedPolicyForward:
    HIDDEN = (64, 128, 256)
    MAX_OBS = 96
    WAVES_PER_BLOCK = 4   # wavefronts that share one LDS copy of the streamed weights

    @classmethod
    def supports(cls, model, obs_size):
        fc = getattr(model, "fc", None)
        if fc is None or len(fc) != 2:
            return False
        h1, h2 = fc["0"][0].out_features, fc["1"][0].out_features
        heads = list(model.head_sizes)
        return (h1 == h2 and h1 in cls.HIDDEN and 1 <= obs_size <= cls.MAX_OBS and 1 <= len(heads) <= 2
                and sum(heads) + 1 <= 32 * _OUT_TILES and fc["0"][0].in_features == obs_size)

    def __init__(self, function_manager, model, obs_size):
        assert self.supports(model, obs_size), "unsupported policy shape for the fused forward"
        self.model = model
        self.F = int(obs_size)
        self.H = model.fc["0"][0].out_features
        self.kt1 = (self.F + 31) // 32
        self.heads = [int(a) for a in model.head_sizes]
        name = f"HipPolicyMlp_{self.H}x{self.H}_k{self.kt1}"
        function_manager.initialize_functions([name])
        self.fn = function_manager.get_function(name)
        # two weight buffers of one k-tile; reused at the end for one [32][65] output tile (+32 row ids) per wavefront
        self.lds_bytes = max(2 * (self.H // 32) * 4096, 4 * (32 * 65 + 32) * 4)
        dev = next(model.parameters()).device
        tn = self.H // 32
        as_idx = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self._idx = [tuple(as_idx(x) for x in _pack_indices(tn, self.kt1, True)),
                     tuple(as_idx(x) for x in _pack_indices(tn, tn, False)),
                     tuple(as_idx(x) for x in _pack_indices(_OUT_TILES, tn, False))]
        self._bidx = [as_idx(_bias_indices(tn)), as_idx(_bias_indices(tn)), as_idx(_bias_indices(_OUT_TILES))]
        self._pads = [(tn * 32, self.kt1 * 32), (tn * 32, tn * 32), (_OUT_TILES * 32, tn * 32)]
        self.packed = None
        self._range_cache = {}
        self.pack()

    @torch.no_grad()
    def pack(self):
        m = self.model
        w3 = torch.cat([h.weight for h in m.policy_head] + [m.vf_head.weight], dim=0)
        b3 = torch.cat([h.bias for h in m.policy_head] + [m.vf_head.bias], dim=0)
        layers = [(m.fc["0"][0].weight, m.fc["0"][0].bias), (m.fc["1"][0].weight, m.fc["1"][0].bias), (w3, b3)]
        packed = []
        for (w, b), (rows, cols), bidx, (pr, pc) in zip(layers, self._idx, self._bidx, self._pads):
            wp = torch.zeros((pr, pc), dtype=torch.float32, device=w.device)
            wp[:w.shape[0], :w.shape[1]] = w.detach().float()
            bp = torch.zeros((pr,), dtype=torch.float32, device=w.device)
            bp[:b.shape[0]] = b.detach().float()
            packed += [wp[rows, cols].contiguous(), bp[bidx].contiguous()]
        if self.packed is None:
            self.packed = packed
        else:  # in place: a captured hipGraph of the rollout tick keeps reading the same addresses
            for dst, src in zip(self.packed, packed):
                dst.copy_(src)

    def __call__(self, obs, agent_ids, probs, values=None, obs_out=None, batch_row=None):
        """obs [E, N, F] float32 (contiguous), agent_ids int32 [n_pol], probs: one [E, N, A_h] float32
        tensor per head (rows of other agents are left alone); values [E, n_pol] or None; obs_out
        [T, E, n_pol, F] or None with batch_row an int64 device scalar selecting T."""
        E, N, F = obs.shape
        assert F == self.F and obs.is_contiguous() and obs.dtype == torch.float32
        assert agent_ids.dtype == torch.int32 and len(probs) == len(self.heads)
        n_pol = int(agent_ids.numel())
        n_rows = E * n_pol
        null = np.uint64(0)
        a1 = self.heads[1] if len(self.heads) > 1 else 0
        # a contiguous range of agents (the usual case) needs no id table in the kernel
        key = int(agent_ids.data_ptr())
        if self._range_cache.get("key") != key:
            ids_host = agent_ids.cpu().numpy()
            contiguous = bool((np.diff(ids_host) == 1).all()) if n_pol > 1 else True
            self._range_cache = {"key": key, "id0": int(ids_host[0]), "contiguous": contiguous}
        ids_arg = null if self._range_cache["contiguous"] else agent_ids
        args = [obs, np.int32(F), np.int32(N), ids_arg, np.int32(self._range_cache["id0"]), np.int32(n_pol),
                np.int32(n_rows), *self.packed,
                np.int32(self.heads[0]), np.int32(a1), probs[0], probs[1] if a1 else null,
                values if values is not None else null, obs_out if obs_out is not None else null,
                batch_row if batch_row is not None else null]
        block_rows = self.WAVES_PER_BLOCK * _WAVE_ROWS
        grid = ((n_rows + block_rows - 1) // block_rows, 1)
        self.fn(*args, block=(64 * self.WAVES_PER_BLOCK, 1, 1), grid=grid, shared=self.lds_bytes)


def rollout_policy_width(model, obs_size, widths=(32, 64)):
    """hidden width if `model` (training.models.FullyConnected) is a network the in-kernel rollout policies
    evaluate -- two hidden layers of equal width in `widths`, one action head -- else None"""
    fc = [model.fc[str(i)][0] for i in range(len(model.fc))]
    if len(fc) != 2 or len(model.policy_head) != 1 or fc[0].in_features != int(obs_size):
        return None
    w = fc[0].out_features
    if w not in widths or fc[1].in_features != w or fc[1].out_features != w or model.policy_head[0].in_features != w:
        return None
    return int(w)


@torch.no_grad()
def pack_rollout_policy(model, out=None):
    """W0 [H][obs], b0 [H], W1 [H][H], b1 [H], Wp [A][H], bp [A] as one flat float32 tensor (the layout
    csrc/kernels/cartpole.hip::cp_policy_cum reads from LDS).  `out`: refill an existing tensor in place (the
    launch plan holds its address)."""
    parts = [model.fc["0"][0].weight, model.fc["0"][0].bias, model.fc["1"][0].weight, model.fc["1"][0].bias,
             model.policy_head[0].weight, model.policy_head[0].bias]
    flat = torch.cat([p.detach().float().reshape(-1) for p in parts])
    if out is None:
        return flat.contiguous()
    out.copy_(flat)
    return out


@torch.no_grad()
def pack_gridworld_policy(model, out=None):
    """One policy of the live-policy TagGridWorld rollout (csrc/kernels/tag_gridworld_n5.hip::gw5_policy_cum): W0
    [H][24] (the 21 inputs of a row padded to 24 floats: every row starts on a 16-byte boundary), b0 [H], W1 [H][H],
    b1 [H], Wp [5][H], bp [5], float32, the block padded to a multiple of four floats.  `out`: refill in place."""
    from warp_drive_amd.envs.tag_gridworld import gridworld_policy_floats

    w0, b0 = model.fc["0"][0].weight, model.fc["0"][0].bias
    H = w0.shape[0]
    assert w0.shape[1] == 21 and model.policy_head[0].weight.shape == (5, H)
    w0p = torch.zeros((H, 24), dtype=torch.float32, device=w0.device)
    w0p[:, :21] = w0.detach().float()
    parts = [w0p, b0, model.fc["1"][0].weight, model.fc["1"][0].bias, model.policy_head[0].weight, model.policy_head[0].bias]
    flat = torch.cat([t.detach().float().reshape(-1) for t in parts])
    n = gridworld_policy_floats(H)
    if out is None:
        out = torch.zeros(n, dtype=torch.float32, device=w0.device)
    assert out.numel() == n
    out[: flat.numel()].copy_(flat)
    return out
