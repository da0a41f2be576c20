"""Policy / value network of the A2C / PPO trainer.

Mirror of reference warp_drive/training/models/{model_base,fully_connected}.py: an MLP
trunk, one softmax head per discrete action dimension and a scalar value head.  The
observation tensor the step kernel writes is consumed IN PLACE (a gather of this policy's
agent rows, no host copy; model_base.py:133-186)."""
import logging

import numpy as np
import torch
from torch import nn

from warp_drive_amd.utils.spaces import Box, Dict, Discrete, MultiDiscrete


def action_head_sizes(action_space):
    if isinstance(action_space, Discrete):
        return [int(action_space.n)]
    if isinstance(action_space, MultiDiscrete):
        return [int(v) for v in action_space.nvec]
    raise NotImplementedError("the A2C/PPO trainer drives Discrete / MultiDiscrete action spaces")


def flattened_obs_size(observation_space):
    if isinstance(observation_space, Box):
        return int(np.prod(observation_space.shape))
    if isinstance(observation_space, Dict):
        return int(sum(np.prod(v.shape) for k, v in observation_space.items() if k != "action_mask"))
    raise NotImplementedError("Observation space must be of Box or Dict type")


class FullyConnected(nn.Module):
    name = "torch_fully_connected"

    def __init__(self, obs_size, head_sizes, fc_dims=(256, 256)):
        super().__init__()
        dims = [int(obs_size)] + [int(d) for d in fc_dims]
        self.fc = nn.ModuleDict({
            str(i): nn.Sequential(nn.Linear(dims[i], dims[i + 1]), nn.ReLU()) for i in range(len(dims) - 1)})
        self.policy_head = nn.ModuleList([nn.Linear(dims[-1], int(a)) for a in head_sizes])
        self.vf_head = nn.Linear(dims[-1], 1)
        self.head_sizes = [int(a) for a in head_sizes]
        # the trainer's hand-written update kernels for THIS model (training/update_kernels.py::UpdateKernels, bound to the
        # function manager -- hence the device -- of the trainer that owns the model) or None: the framework's operations.
        # Carried by the model and handed to the autograd nodes below as an argument: there is no process-wide switch
        self.update_kernels = None
        self._inference_cache = {}  # dtype -> {trunk [(w, b)], head (w, b), versions}; see refresh_inference_cache

    def refresh_inference_cache(self):
        """bring the cast / concatenated copies of the weights that forward_inference keeps up to date with the
        parameters, IN PLACE (call after every optimizer step or checkpoint load, as FusedPolicyForward.pack() is).
        In place because a rollout tick captured in a hipGraph keeps reading the copies at the addresses they had at
        capture time: replacing the tensors (what this did until round 6) left a replayed tick on the head weights
        of the capture -- the trunk aliases the live parameters -- i.e. a behaviour policy that is neither the old nor
        the new network (found by tests/test_gpu_learning.py: Cartpole on the per-tick path un-learned)."""
        for dtype in list(self._inference_cache):
            self._fill_inference_cache(dtype)

    @torch.no_grad()
    def _fill_inference_cache(self, dtype):
        cast = (lambda t: t.detach()) if dtype is None else (lambda t: t.detach().to(dtype))
        trunk = [(cast(self.fc[str(i)][0].weight), cast(self.fc[str(i)][0].bias)) for i in range(len(self.fc))]
        w = cast(torch.cat([h.weight for h in self.policy_head] + [self.vf_head.weight], dim=0))
        b = cast(torch.cat([h.bias for h in self.policy_head] + [self.vf_head.bias], dim=0))
        entry = self._inference_cache.get(dtype)
        if entry is None or entry["head"][0].shape != w.shape or entry["head"][0].device != w.device:
            entry = self._inference_cache[dtype] = {"trunk": trunk, "head": (w, b)}
        else:
            if dtype is not None:  # (dtype None: the trunk entries ARE the parameters)
                for (dw, db), (sw, sb) in zip(entry["trunk"], trunk):
                    dw.copy_(sw)
                    db.copy_(sb)
            entry["head"][0].copy_(w)
            entry["head"][1].copy_(b)
        entry["versions"] = tuple(int(p._version) for p in self.parameters())
        return entry

    @torch.no_grad()
    def _inference_weights(self, dtype):
        # checked against the parameters' version counters too: an in-place update (optimizer step, checkpoint
        # load) that nobody announced still refreshes the copies (in place) the next time the host runs this
        entry = self._inference_cache.get(dtype)
        if entry is None or entry["versions"] != tuple(int(p._version) for p in self.parameters()):
            entry = self._fill_inference_cache(dtype)
        return entry["trunk"], entry["head"]

    def forward(self, obs):
        """obs [..., obs_size] -> ([probs per head, each [..., A_h]], values [...]).
        The layers run through `_Affine` (bias + ReLU in the GEMM epilogue; a backward without reduction
        kernels over the ~1e7 rows of a training batch) and all heads through ONE GEMM over the
        concatenated head weights, like forward_inference; parameters and their names are nn.Linear's."""
        x = obs
        for i in range(len(self.fc)):
            lin = self.fc[str(i)][0]
            x = _Affine.apply(x, lin.weight, lin.bias, True, self.update_kernels)
        w = torch.cat([h.weight for h in self.policy_head] + [self.vf_head.weight], dim=0)
        b = torch.cat([h.bias for h in self.policy_head] + [self.vf_head.bias], dim=0)
        out = _Affine.apply(x, w, b, False, self.update_kernels)
        probs, start = [], 0
        for a in self.head_sizes:
            probs.append(torch.softmax(out[..., start:start + a], dim=-1))
            start += a
        return probs, out[..., start]

    def forward_logits(self, obs):
        """obs [..., obs_size] -> [..., sum(head_sizes) + 1]: the logits of every head, then the value -- what `forward`
        turns into probabilities; the fused objective (training/update_kernels.py) works on this tensor directly"""
        x, k = obs, self.update_kernels
        w = torch.cat([h.weight for h in self.policy_head] + [self.vf_head.weight], dim=0)
        b = torch.cat([h.bias for h in self.policy_head] + [self.vf_head.bias], dim=0)
        # (a network without hidden layers -- `fc_dims: []` -- is the output layer alone)
        fused_tail = (not torch.is_autocast_enabled(obs.device.type) and obs.dtype == torch.float32 and len(self.fc) >= 1)
        if fused_tail and len(self.fc) == 2 and not obs.requires_grad:  # the usual network: one node, one planned backward
            l1, l2 = self.fc["0"][0], self.fc["1"][0]
            return _MlpTwoHidden.apply(obs, l1.weight, l1.bias, l2.weight, l2.bias, w, b, None, None, None, k)
        for i in range(len(self.fc) - (1 if fused_tail else 0)):
            lin = self.fc[str(i)][0]
            x = _Affine.apply(x, lin.weight, lin.bias, True, k)
        if fused_tail:  # the last hidden layer + the output layer: one node, one backward pass over h2
            last = self.fc[str(len(self.fc) - 1)][0]
            return _TailHead.apply(x, last.weight, last.bias, w, b, None, None, k)
        return _Affine.apply(x, w, b, False, k)

    def forward_logits_stored(self, obs, h1, h2, out):
        """`forward_logits` without arithmetic: the rollout's forward kernel stored the hidden activations and the outputs
        of exactly these rows under exactly these weights (on-policy: nothing changed them since), so the forward pass is
        a read and only the backward runs (training/policy_kernel.py::FusedRolloutTick `stored`).  Two hidden layers."""
        assert len(self.fc) == 2
        l1, l2 = self.fc["0"][0], self.fc["1"][0]
        w = torch.cat([h.weight for h in self.policy_head] + [self.vf_head.weight], dim=0)
        b = torch.cat([h.bias for h in self.policy_head] + [self.vf_head.bias], dim=0)
        return _MlpTwoHidden.apply(obs, l1.weight, l1.bias, l2.weight, l2.bias, w, b, h1, h2, out, self.update_kernels)

    @torch.no_grad()
    def forward_inference(self, obs, dtype=None):
        """The same network for the ROLLOUT (no autograd): bias + ReLU fused into the trunk GEMMs'
        epilogue (hipBLASLt through torch._addmm_activation: bit-identical to Linear followed by ReLU,
        212 -> 137 us on the 200 000 x 71 x 256 layer) and all heads evaluated by ONE GEMM over the
        concatenated head weights (3 x 88 us -> 95 us).  `dtype=torch.bfloat16` runs the GEMMs on the
        bf16 matrix cores.  Returns float32 probabilities per head and float32 values."""
        lead = obs.shape[:-1]
        x = obs.reshape(-1, obs.shape[-1])
        if dtype is not None:
            x = x.to(dtype)
        trunk, (w, b) = self._inference_weights(dtype)  # cast / concatenated once per optimizer step
        for wi, bi in trunk:
            x = _linear_relu(x, wi, bi)
        out = torch.nn.functional.linear(x, w, b).float()
        probs, start = [], 0
        for a in self.head_sizes:
            probs.append(torch.softmax(out[:, start:start + a], dim=-1).reshape(*lead, a))
            start += a
        return probs, out[:, start].reshape(*lead)


def _column_sums(g):
    """float32 column sums of a [rows, features] gradient with ~1e7 rows.  A plain `g.sum(0)` launches 512
    blocks for the whole tensor (38 ms per 256-wide layer on MI355X) and a GEMM with a row of ones is as
    slow (a 1 x rows x features GEMM): two stages -- [R, rows / R, features] summed over the middle
    dimension (R x features independent outputs: the chip is full and the reads are coalesced), then
    over R."""
    rows = g.shape[0]
    for r in (2000, 2048, 1024, 1000, 800, 512, 500, 400, 256, 250, 200, 128, 100, 64):
        if rows % r == 0 and rows // r >= 16:
            return g.reshape(r, rows // r, g.shape[1]).sum(dim=1, dtype=torch.float32).sum(dim=0)  # (reshape: g may be non-contiguous)
    return g.sum(dim=0, dtype=torch.float32)


def _weight_grad(g, x, kernels=None):
    """g^T @ x for [rows, out] x [rows, in] with ~1e7 rows: one GEMM whose contraction is 1e7 long and
    whose result is 256 x 256 leaves most of the chip idle (5-10 ms at 1 TB/s on MI355X); as a batch of
    S independent slices of the rows followed by a sum over S it streams both operands at memory speed.
    float32 operands of the shapes HipWeightGradBx3_* covers go there instead (training/update_kernels.py::weight_grad:
    the batched GEMM already runs at the f32 matrix peak; bf16x3 arithmetic is what is left)."""
    rows = g.shape[0]
    if kernels is not None and kernels.supports_weight_grad(g, x):
        return kernels.weight_grad(g, x)[0]
    if rows >= (1 << 20):
        for sl in (250, 256, 200, 128, 125, 100, 64, 50, 32):
            if rows % sl == 0:
                part = torch.bmm(g.reshape(sl, rows // sl, g.shape[1]).transpose(1, 2),
                                 x.reshape(sl, rows // sl, x.shape[1]))
                return part.sum(dim=0, dtype=torch.float32)
    return (g.t() @ x).float()


class _Affine(torch.autograd.Function):
    """y = [relu](x @ W^T + b) over the last dimension of x.

    Why not nn.Linear + nn.ReLU: a training batch of configs[2] is ~1e7 rows, and autograd's bias gradient
    is a column reduction of the [rows, features] gradient that PyTorch runs with 512 blocks -- 38 ms per
    256-wide layer on MI355X, 37 % of the whole update (profiles/r03_update_kernels.txt).  Here the bias
    gradient is a two-stage sum that fills the chip (`_column_sums`),
    bias + ReLU ride in the forward GEMM's epilogue, the ReLU mask is applied in one pass, and the input
    gradient is skipped where nobody needs it (the first layer).  Under autocast the GEMMs run in the
    autocast dtype (bf16 matrix cores) with float32 parameters and float32 parameter gradients."""

    @staticmethod
    def forward(ctx, x, w, b, relu, kernels=None):
        dev = x.device.type
        dt = torch.get_autocast_dtype(dev) if torch.is_autocast_enabled(dev) else x.dtype
        x2 = x.reshape(-1, x.shape[-1]).to(dt)
        wc, bc = w.to(dt), b.to(dt)
        with torch.autocast(device_type=dev, enabled=False):
            y = _linear_relu(x2, wc, bc) if relu else torch.addmm(bc, x2, wc.t())
        ctx.save_for_backward(x2, wc, y if relu else x2.new_empty(0))
        ctx.relu, ctx.in_shape, ctx.kernels = bool(relu), x.shape, kernels
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, g):
        x2, wc, y = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1]).to(x2.dtype)
        gb = None
        kernels = ctx.kernels
        if ctx.relu:
            if kernels is not None and kernels.supports_relu_backward(g2, y):
                g2, gb = kernels.relu_backward_colsum(g2, y)  # mask + bias gradient in one pass over the gradient
            else:
                g2 = torch.ops.aten.threshold_backward(g2, y, 0)
        with torch.autocast(device_type=g.device.type, enabled=False):
            gx = (g2 @ wc).reshape(ctx.in_shape) if ctx.needs_input_grad[0] else None
            gw = _weight_grad(g2, x2, kernels)
            if gb is None:
                gb = _column_sums(g2)
        return gx, gw, gb, None, None


class _TailHead(torch.autograd.Function):
    """The last hidden layer and the output layer as ONE autograd node: h2 = relu(h1 @ W2^T + b2), out = h2 @ W3^T + b3
    (W3 = all heads' rows, then the value's).  One node so that its backward can run the output layer's backward, the
    hidden layer's ReLU mask, its bias gradient and the output layer's weight gradient as ONE pass over h2
    (training/update_kernels.py::head_backward) instead of a GEMM, a mask + column-sum pass and a skinny GEMM that read
    h2 three times.  `h2_stored` / `out_stored`: the forward results are already known (the rollout stored them):
    nothing is computed going forward.  float32; the autocast update takes the per-layer `_Affine` path."""

    @staticmethod
    def forward(ctx, h1, w2, b2, w3, b3, h2_stored, out_stored, kernels=None):
        h1_2d = h1.reshape(-1, h1.shape[-1])
        if h2_stored is None:
            h2 = _linear_relu(h1_2d, w2, b2)
            out = torch.addmm(b3, h2, w3.t())
        else:
            h2, out = h2_stored.reshape(-1, w2.shape[0]), out_stored.reshape(-1, w3.shape[0])
        ctx.save_for_backward(h1_2d, w2, w3, h2)
        ctx.in_shape, ctx.kernels = h1.shape, kernels
        return out.view(*h1.shape[:-1], w3.shape[0])

    @staticmethod
    def backward(ctx, g):
        h1_2d, w2, w3, h2 = ctx.saved_tensors
        g3 = g.reshape(-1, g.shape[-1])
        kernels = ctx.kernels
        gb3 = None
        if kernels is not None and kernels.supports_head_backward(g3, w3, h2):
            g2, gb2, gw3, gb3 = kernels.head_backward(g3, w3, h2)
        else:
            g2 = torch.ops.aten.threshold_backward(g3 @ w3, h2, 0)
            gb2, gw3 = _column_sums(g2), _weight_grad(g3, h2, kernels)
        if gb3 is None:
            gb3 = _column_sums(g3)
        gw2 = _weight_grad(g2, h1_2d, kernels)
        gh1 = (g2 @ w2).reshape(ctx.in_shape) if ctx.needs_input_grad[0] else None
        return gh1, gw2, gb2, gw3, gb3, None, None, None


class _MlpTwoHidden(torch.autograd.Function):
    """A network of two hidden ReLU layers and the output layer as ONE autograd node (float32), so that its backward is
    the sequence this trainer wants rather than what per-layer nodes compose to:
        g3 -> HipHeadBackward (g2 masked, bias gradient of layer 2, weight gradient of the output layer: one pass over h2)
           -> weight gradient of layer 2 (GEMM)
           -> HipLinearMaskBackwardBx3 (g1 = [h1 > 0] * (g2 . W2): the square product on the bf16 matrix cores at float32
              accuracy with the mask applied to its accumulators -- a GEMM and a mask pass before)
           -> bias and weight gradient of layer 1
    each step falling back to the framework's operations where a kernel does not cover the shape.  `stored` = (h1, h2, out)
    already known from the rollout (nothing is computed going forward) or None."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, h1_stored, h2_stored, out_stored, kernels=None):
        x2 = x.reshape(-1, x.shape[-1])
        if h1_stored is None:
            h1 = _linear_relu(x2, w1, b1)
            h2 = _linear_relu(h1, w2, b2)
            out = torch.addmm(b3, h2, w3.t())
        else:
            h1, h2 = h1_stored.reshape(-1, w1.shape[0]), h2_stored.reshape(-1, w2.shape[0])
            out = out_stored.reshape(-1, w3.shape[0])
        ctx.save_for_backward(x2, w2, w3, h1, h2)
        ctx.kernels = kernels
        return out.view(*x.shape[:-1], w3.shape[0])

    @staticmethod
    def backward(ctx, g):
        x2, w2, w3, h1, h2 = ctx.saved_tensors
        g3 = g.reshape(-1, g.shape[-1])
        kernels = ctx.kernels
        gb3 = None
        if kernels is not None and kernels.supports_head_backward(g3, w3, h2):
            g2, gb2, gw3, gb3 = kernels.head_backward(g3, w3, h2)
        else:
            g2 = torch.ops.aten.threshold_backward(g3 @ w3, h2, 0)
            gb2, gw3 = _column_sums(g2), _weight_grad(g3, h2, kernels)
        if gb3 is None:
            gb3 = _column_sums(g3)
        gw2 = _weight_grad(g2, h1, kernels)
        if kernels is not None and kernels.supports_linear_mask_backward(g2, w2, h1):
            g1 = kernels.linear_mask_backward(g2, w2, h1)
        else:
            g1 = torch.ops.aten.threshold_backward(g2 @ w2, h1, 0)
        if kernels is not None and kernels.supports_weight_grad(g1, x2, with_bias=True):
            gw1, gb1 = kernels.weight_grad(g1, x2, with_bias=True)  # (the bias gradient rides as a column of ones)
        else:
            gb1, gw1 = _column_sums(g1), _weight_grad(g1, x2, kernels)
        return None, gw1, gb1, gw2, gb2, gw3, gb3, None, None, None, None


_FUSED_EPILOGUE = {"ok": hasattr(torch, "_addmm_activation")}  # decided once: a private torch entry point


def _linear_relu(x, weight, bias):
    """relu(x @ weight.T + bias), with the bias + ReLU epilogue fused into the GEMM where the backend
    offers it (CUDA/ROCm tensors); plain Linear + ReLU otherwise.  `torch._addmm_activation` is private:
    any failure (signature / dtype support changed) switches to the plain form for the rest of the run."""
    if x.is_cuda and _FUSED_EPILOGUE["ok"]:
        try:
            return torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
        except Exception as err:  # noqa: BLE001 -- TypeError / NotImplementedError / RuntimeError alike
            _FUSED_EPILOGUE["ok"] = False
            logging.warning(f"torch._addmm_activation failed ({type(err).__name__}: {err}); Linear + ReLU run as two "
                            "operations from here on (tests/test_trainer_cpu.py pins the entry point's signature)")
    return torch.relu(torch.nn.functional.linear(x, weight, bias))
