"""Policy / value network of the A2C / PPO trainer.

Mirror of reference warp_drive/training/models/{model_base,fully_connected}.py: an MLP
trunk, one softmax head per discrete action dimension and a scalar value head.  The
observation tensor the step kernel writes is consumed IN PLACE (a gather of this policy's
agent rows, no host copy; model_base.py:133-186)."""
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

    def forward(self, obs):
        """obs [..., obs_size] -> ([probs per head, each [..., A_h]], values [...])"""
        x = obs
        for i in range(len(self.fc)):
            x = self.fc[str(i)](x)
        probs = [torch.softmax(head(x), dim=-1) for head in self.policy_head]
        return probs, self.vf_head(x)[..., 0]

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
        for i in range(len(self.fc)):
            lin = self.fc[str(i)][0]
            w, b = (lin.weight, lin.bias) if dtype is None else (lin.weight.to(dtype), lin.bias.to(dtype))
            x = _linear_relu(x, w, b)
        w = torch.cat([h.weight for h in self.policy_head] + [self.vf_head.weight], dim=0)
        b = torch.cat([h.bias for h in self.policy_head] + [self.vf_head.bias], dim=0)
        if dtype is not None:
            w, b = w.to(dtype), b.to(dtype)
        out = torch.nn.functional.linear(x, w, b).float()
        probs, start = [], 0
        for a in self.head_sizes:
            probs.append(torch.softmax(out[:, start:start + a], dim=-1).reshape(*lead, a))
            start += a
        return probs, out[:, start].reshape(*lead)


def _linear_relu(x, weight, bias):
    """relu(x @ weight.T + bias), with the bias + ReLU epilogue fused into the GEMM where the backend
    offers it (CUDA/ROCm tensors); plain Linear + ReLU otherwise."""
    if x.is_cuda and hasattr(torch, "_addmm_activation"):
        try:
            return torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
        except RuntimeError:
            pass
    return torch.relu(torch.nn.functional.linear(x, weight, bias))
