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


def _pack_indices_bx3(n_out_tiles, n_k_tiles, first_layer):
    """(row, col) gather indices of ONE term of the bf16x3 weight tensor [KT, (3,) TN, 2, 64, 8]: lane l supplies
    A[i = l & 31][k = 8 (l >> 5) + e] of a 32x32x16 MFMA; k half q of the first layer contracts features
    32 kt + 16 q + 8 h + e, of the later layers the rows (32 kt + row_of(8 q + e, h)) the previous layer's accumulators
    hold in register 8 q + e."""
    kt, tn, q, lane, e = np.meshgrid(np.arange(n_k_tiles), np.arange(n_out_tiles), np.arange(2), np.arange(64),
                                     np.arange(8), indexing="ij")
    h = lane >> 5
    rows = tn * 32 + (lane & 31)
    cols = 32 * kt + (16 * q + 8 * h + e if first_layer else _row_of(8 * q + e, h))
    return rows, cols


def split_bf16x3(w):
    """float32 tensor -> [3, ...] bfloat16 terms hi, mid, lo with hi + mid + lo = w up to 2^-24 |w| (each term is the
    round-to-nearest bf16 of what the previous ones left; the subtractions are exact in float32)"""
    w = w.detach().float()
    hi = w.to(torch.bfloat16)
    r1 = w - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return torch.stack([hi, mid, lo])


def _bias_indices(n_tiles):
    tn, h, s = np.meshgrid(np.arange(n_tiles), np.arange(2), np.arange(16), indexing="ij")
    return tn * 32 + _row_of(s, h)


def parameter_versions(model):
    """the in-place version counter of every parameter of `model`, in `parameters()` order"""
    return tuple(int(p._version) for p in model.parameters())


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

    ARITHMETICS = ("float32", "bf16x3")

    def __init__(self, function_manager, model, obs_size, arithmetic="float32"):
        """arithmetic: "float32" = v_mfma_f32_32x32x2_f32 (an fmaf chain); "bf16x3" = every float32 product as six bf16
        partial products on the bf16 matrix cores (2.7 x the matrix rate, error of the size of float32 rounding: see
        csrc/kernels/policy_mlp.hip) -- same gates, same interface."""
        assert self.supports(model, obs_size), "unsupported policy shape for the fused forward"
        assert arithmetic in self.ARITHMETICS
        self.model = model
        self.arithmetic = arithmetic
        self.bx3 = arithmetic == "bf16x3"
        self.kernel_tag = "Bx3" if self.bx3 else ""
        self.F = int(obs_size)
        self.H = model.fc["0"][0].out_features
        self.kt1 = (self.F + 31) // 32
        self.heads = [int(a) for a in model.head_sizes]
        name = f"HipPolicyMlp{self.kernel_tag}_{self.H}x{self.H}_k{self.kt1}"
        function_manager.initialize_functions([name])
        self.fn = function_manager.get_function(name)
        # float32: two weight buffers of one k-tile (4 KB per output tile); bf16x3: three (6 KB per output tile); reused
        # at the end for one [32][65] output tile (+32 row ids) per wavefront
        self.lds_bytes = max((3 * 6144 if self.bx3 else 2 * 4096) * (self.H // 32), 4 * (32 * 65 + 32) * 4)
        dev = next(model.parameters()).device
        tn = self.H // 32
        as_idx = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        indices = _pack_indices_bx3 if self.bx3 else _pack_indices
        self._idx = [tuple(as_idx(x) for x in indices(tn, self.kt1, True)),
                     tuple(as_idx(x) for x in indices(tn, tn, False)),
                     tuple(as_idx(x) for x in indices(_OUT_TILES, tn, False))]
        self._bidx = [as_idx(_bias_indices(tn)), as_idx(_bias_indices(tn)), as_idx(_bias_indices(_OUT_TILES))]
        self._pads = [(tn * 32, self.kt1 * 32), (tn * 32, tn * 32), (_OUT_TILES * 32, tn * 32)]
        self.packed = None
        self.packed_versions = None   # version counters of the model's parameters the packed copy was made from
        self._range_cache = {}
        self.pack()

    @torch.no_grad()
    def pack(self):
        m = self.model
        # what the kernel will read is a COPY: remember which state of the parameters it is a copy of (every in-place
        # change of a parameter -- optimizer step, load_state_dict, a manual edit -- advances its version counter), so
        # that users of the kernel's results can tell whether they still belong to the model's current weights
        self.packed_versions = parameter_versions(m)
        w3 = torch.cat([h.weight for h in m.policy_head] + [m.vf_head.weight], dim=0)
        b3 = torch.cat([h.bias for h in m.policy_head] + [m.vf_head.bias], dim=0)
        layers = [(m.fc["0"][0].weight, m.fc["0"][0].bias), (m.fc["1"][0].weight, m.fc["1"][0].bias), (w3, b3)]
        packed = []
        for (w, b), (rows, cols), bidx, (pr, pc) in zip(layers, self._idx, self._bidx, self._pads):
            wp = torch.zeros((pr, pc), dtype=torch.float32, device=w.device)
            wp[:w.shape[0], :w.shape[1]] = w.detach().float()
            bp = torch.zeros((pr,), dtype=torch.float32, device=w.device)
            bp[:b.shape[0]] = b.detach().float()
            if self.bx3:  # [KT, 3 terms, TN, 2 k halves, 64 lanes, 8] bfloat16
                wpk = split_bf16x3(wp)[:, rows, cols].transpose(0, 1).contiguous()
            else:         # [KT, TN, 4, 64 lanes, 4] float32
                wpk = wp[rows, cols].contiguous()
            packed += [wpk, bp[bidx].contiguous()]
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


class FusedRolloutTick:
    """The policy side of a rollout tick as TWO launches around the env's step (csrc/kernels/policy_mlp.hip):

      forward()  HipPolicyMlpAct_*: the forward of ALL policies (one or two `FusedPolicyForward`s of the same shape) in
                 one launch, the actions of both heads drawn in its epilogue -- same Philox counters and the same
                 inverse-CDF search as the env's fused tick, on the probabilities the kernel would have written -- into
                 the env's `sampled_actions` and row t of every policy's action batch; the observation rows into row t
                 of the observation batches.  The probabilities never reach HBM.
      (the env's `TickA` entry: step + reset of finished replicas on those actions -- RolloutEngine(presampled_actions=True))
      record()   HipRolloutRecord: rewards / done into row t of the batches, the episodic-reward bookkeeping, t += 1.

    `batch_row` is the device counter t, one copy per replica (int64 [E], all equal: each replica's record block advances
    its own, so no kernel hands anything over between blocks); everything is enqueued on torch's current stream and is free of
    host-side indices, so a tick can be captured in a hipGraph."""

    def __init__(self, function_manager, forwards, agent_ids, obs, actions, rewards, done, rng_state, stream_tag,
                 batch_row, obs_batches, action_batches, reward_batches, done_batch, ep_rewards, ep_sums, ep_count,
                 stored=None):
        """stored (optional): per policy None or (h1 [T, E, n_pol, H], h2 [T, E, n_pol, H], out [T, E, n_pol, A0 + A1 + 1])
        float32 -- the forward launch also writes row t of the hidden activations and of the outputs, which is all the
        update's forward pass would recompute (bf16x3 arithmetic only)."""
        assert 1 <= len(forwards) <= 2
        f0 = forwards[0]
        assert all((f.H, f.kt1, f.heads, f.F) == (f0.H, f0.kt1, f0.heads, f0.F) for f in forwards), \
            "one launch serves policies of ONE network shape"
        assert len(f0.heads) == 2, "the epilogue draws the two heads of a MultiDiscrete([A0, A1]) action"
        E, N, F = obs.shape
        assert obs.is_contiguous() and obs.dtype == torch.float32 and F == f0.F
        assert actions.dtype == torch.int32 and actions.is_contiguous() and tuple(actions.shape) == (E, N, 2)
        assert all(f.arithmetic == f0.arithmetic for f in forwards)
        self.fwd_name = f"HipPolicyMlpAct{f0.kernel_tag}_{f0.H}x{f0.H}_k{f0.kt1}"
        function_manager.initialize_functions([self.fwd_name, "HipRolloutRecord"])
        self.fwd, self.rec = function_manager.get_function(self.fwd_name), function_manager.get_function("HipRolloutRecord")
        self.forwards, self.lds_bytes = forwards, f0.lds_bytes
        dev, null = obs.device, np.uint64(0)
        block_rows = f0.WAVES_PER_BLOCK * _WAVE_ROWS
        slot = np.full(N, -1, dtype=np.int64)
        per_policy, blocks = [], []
        self._id_tables = []  # (the launch arguments hold raw addresses: keep the id tensors alive)
        for k, (f, ids) in enumerate(zip(forwards, agent_ids)):
            ids_host = np.asarray(ids.cpu().numpy(), dtype=np.int64)
            slot[ids_host] = k * 65536 + np.arange(len(ids_host))
            n_pol = len(ids_host)
            contiguous = bool((np.diff(ids_host) == 1).all()) if n_pol > 1 else True
            ids32 = ids.to(torch.int32).contiguous()
            n_rows = E * n_pol
            blocks.append((n_rows + block_rows - 1) // block_rows)
            for t, shape in ((obs_batches[k], (E, n_pol, F)), (action_batches[k], (E, n_pol, 2)), (reward_batches[k], (E, n_pol))):
                assert t.is_contiguous() and tuple(t.shape[1:]) == shape, (tuple(t.shape), shape)
            assert action_batches[k].dtype == torch.int32
            extra = [null, null, null]
            if stored is not None and stored[k] is not None:
                assert f.bx3, "the activations are stored by the bf16x3 kernel"
                h1, h2, out = stored[k]
                W = sum(f.heads) + 1
                for t, shape in ((h1, (E, n_pol, f.H)), (h2, (E, n_pol, f.H)), (out, (E, n_pol, W))):
                    assert t.is_contiguous() and t.dtype == torch.float32 and tuple(t.shape[1:]) == shape, (tuple(t.shape), shape)
                extra = [h1, h2, out]
            per_policy.append([null if contiguous else ids32, np.int32(ids_host[0]), np.int32(n_pol), np.int32(n_rows),
                               *f.packed, obs_batches[k], action_batches[k], *extra])
            self._id_tables.append(ids32)
        assert (slot >= 0).all(), "every agent belongs to exactly one policy of the launch"
        if len(forwards) == 1:
            per_policy.append([null, np.int32(0), np.int32(1), np.int32(0)] + [null] * 11)
        self.slot = torch.from_numpy(slot.astype(np.int32)).to(dev)
        assert batch_row.dtype == torch.int64 and batch_row.numel() == E
        self.fwd_args = [obs, np.int32(F), np.int32(N), np.int32(f0.heads[0]), np.int32(f0.heads[1]), null, null, batch_row,
                         rng_state, actions, np.int32(stream_tag), np.int32(blocks[0] if len(forwards) == 2 else 2 ** 30),
                         *per_policy[0], *per_policy[1]]
        self.fwd_grid, self.fwd_block = (sum(blocks), 1), (64 * f0.WAVES_PER_BLOCK, 1, 1)
        second = len(forwards) == 2
        self.rec_args = [rewards, done, np.int32(N), np.int32(E), self.slot, batch_row, done_batch, ep_count,
                         reward_batches[0], ep_rewards[0], ep_sums[0], np.int32(per_policy[0][2]),
                         reward_batches[1] if second else null, ep_rewards[1] if second else null,
                         ep_sums[1] if second else null, np.int32(per_policy[1][2]) if second else np.int32(0)]
        self.rec_grid, self.rec_block, self.rec_lds = (E, 1), (min(1024, (N + 63) // 64 * 64), 1, 1), 4 * N

    def forward(self):
        self.fwd(*self.fwd_args, block=self.fwd_block, grid=self.fwd_grid, shared=self.lds_bytes)

    def record(self):
        self.rec(*self.rec_args, block=self.rec_block, grid=self.rec_grid, shared=self.rec_lds)


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
    out[flat.numel():].zero_()  # (the padding is never read)
    return out
