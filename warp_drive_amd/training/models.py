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
