"""A2C and PPO objectives (reference warp_drive/training/algorithms/policygradient/
a2c.py:40-194 and ppo.py:42-228): bootstrapped discounted returns that stop at done flags,
MSE value loss, (clipped) policy-gradient loss, entropy bonus, optional return / advantage
normalisation across the (env, agent) dimensions.

Batches are [T, E, n(, heads)] tensors that live on the device; nothing here touches the host
unless `perform_logging` asks for metric scalars."""
import numpy as np
import torch
from torch import nn
from torch.distributions import Categorical

from warp_drive_amd.training.param_scheduler import ParamScheduler

_EPSILON = 1.0e-10


def discounted_returns(rewards, done_flags, values_detached, gamma):
    """R_T = done ? r : V_T ; R_t = r_t + (1 - done_t) * gamma * R_{t+1}   (a2c.py:80-95)"""
    done = (done_flags > 0).to(rewards.dtype)[..., None]  # [T, E, 1]
    returns = torch.zeros_like(rewards)
    returns[-1] = done[-1] * rewards[-1] + (1 - done[-1]) * values_detached[-1]
    for t in range(rewards.shape[0] - 2, -1, -1):
        returns[t] = rewards[t] + (1 - done[t]) * gamma * returns[t + 1]
    return returns


def _normalise(x):
    return (x - x.mean(dim=(1, 2), keepdim=True)) / (x.std(dim=(1, 2), keepdim=True) + _EPSILON)


class A2C:
    clip_param = None  # PPO sets it

    def __init__(self, discount_factor_gamma=1.0, normalize_advantage=False, normalize_return=False,
                 vf_loss_coeff=0.01, entropy_coeff=0.01):
        assert 0 <= discount_factor_gamma <= 1
        self.discount_factor_gamma = discount_factor_gamma
        self.normalize_advantage = normalize_advantage
        self.normalize_return = normalize_return
        self.vf_loss_coeff_schedule = ParamScheduler(vf_loss_coeff)
        self.entropy_coeff_schedule = ParamScheduler(entropy_coeff)

    def _policy_loss(self, log_prob, advantages):
        return (-log_prob * advantages).mean()

    @staticmethod
    def _sample_positive_negative_env_ids(done_flags_batch, negative_positive_ratio):
        """replicas that ended by reaching the goal (done == 2) are "positives"; keep all of them and a
        random `ratio` x as many of the others (a2c.py:196-220)"""
        positives = torch.any(done_flags_batch == 2, dim=0)
        positive_env_ids = positives.nonzero(as_tuple=True)[0].tolist()
        negative_env_ids = (~positives).nonzero(as_tuple=True)[0].tolist()
        pos_size = len(positive_env_ids)
        if pos_size > 0:
            neg_size = int(pos_size * negative_positive_ratio)
            if pos_size + neg_size < done_flags_batch.shape[1]:
                chosen = np.random.choice(negative_env_ids, size=neg_size, replace=False).tolist()
                return positive_env_ids, chosen, True
        return positive_env_ids, negative_env_ids, False

    def compute_loss_and_metrics(self, timestep=None, actions_batch=None, rewards_batch=None,
                                 done_flags_batch=None, action_probabilities_batch=None,
                                 value_functions_batch=None, perform_logging=False, negative_positive_ratio=-1):
        assert timestep is not None and actions_batch is not None and rewards_batch is not None
        assert done_flags_batch is not None and action_probabilities_batch is not None
        assert value_functions_batch is not None
        pos_env_ids = neg_env_ids = None
        if negative_positive_ratio > 0:
            pos_env_ids, neg_env_ids, need_downsample = self._sample_positive_negative_env_ids(
                done_flags_batch, negative_positive_ratio)
            if need_downsample:
                keep = pos_env_ids + neg_env_ids
                actions_batch, rewards_batch = actions_batch[:, keep], rewards_batch[:, keep]
                done_flags_batch, value_functions_batch = done_flags_batch[:, keep], value_functions_batch[:, keep]
                action_probabilities_batch = [p[:, keep] for p in action_probabilities_batch]
        values_detached = value_functions_batch.detach()
        returns = discounted_returns(rewards_batch, done_flags_batch, values_detached, self.discount_factor_gamma)
        norm_returns = _normalise(returns) if self.normalize_return else returns
        vf_loss = nn.functional.mse_loss(value_functions_batch, norm_returns)
        advantages = norm_returns - values_detached
        norm_adv = _normalise(advantages) if self.normalize_advantage else advantages
        log_prob, mean_entropy = 0.0, 0.0
        for h, probs in enumerate(action_probabilities_batch):
            dist = Categorical(probs)
            mean_entropy = mean_entropy + dist.entropy().mean()
            log_prob = log_prob + dist.log_prob(actions_batch[..., h])
        policy_loss = self._policy_loss(log_prob, norm_adv)
        vf_c = self.vf_loss_coeff_schedule.get_param_value(timestep)
        ent_c = self.entropy_coeff_schedule.get_param_value(timestep)
        loss = policy_loss + vf_c * vf_loss - ent_c * mean_entropy
        metrics = {}
        if perform_logging:
            var_explained = torch.clamp(1 - norm_adv.detach().var() / (norm_returns.detach().var() + _EPSILON), min=-1.0)
            metrics = {
                "VF loss coefficient": vf_c, "Entropy coefficient": ent_c, "Total loss": loss.item(),
                "Policy loss": policy_loss.item(), "Value function loss": vf_loss.item(),
                "Mean rewards": rewards_batch.mean().item(), "Max. rewards": rewards_batch.max().item(),
                "Min. rewards": rewards_batch.min().item(),
                "Mean value function": value_functions_batch.mean().item(),
                "Mean advantages": advantages.mean().item(),
                "Mean (norm.) advantages": norm_adv.mean().item(),
                "Mean (discounted) returns": returns.mean().item(),
                "Mean normalized returns": norm_returns.mean().item(), "Mean entropy": mean_entropy.item(),
                "Variance explained by the value function": var_explained.item(),
            }
            # mean of the standard deviation of the sampled actions (a2c.py:160-180)
            af = actions_batch.float()
            over_agents, over_time, over_envs = (af.std(dim=d).mean(dim=(0, 1)) for d in (2, 0, 1))
            for h in range(af.shape[-1]):
                metrics[f"Std. of action_{h} over agents"] = over_agents[h].item()
                metrics[f"Std. of action_{h} over envs"] = over_envs[h].item()
                metrics[f"Std. of action_{h} over time"] = over_time[h].item()
            if negative_positive_ratio > 0:
                metrics["Num of Positive Sampled Envs"] = len(pos_env_ids)
                metrics["Num of Negative Sampled Envs"] = len(neg_env_ids)
        return loss, metrics

    def compute_loss_and_metrics_from_logits(self, timestep, out, actions_batch, rewards_batch, done_flags_batch, head_sizes,
                                             perform_logging, kernels=None):
        """`compute_loss_and_metrics` on the network's raw output `out` [T, E, n, W] (logits of every head, then the value)
        with the objective and its gradient formed by ONE kernel (training/update_kernels.py::FusedObjective): same loss,
        same gradient, same metric names.  The returns / advantages -- small [T, E, n] tensors -- are the code above.
        `kernels`: the UpdateKernels of the caller's device (required: this entry IS the kernel path)."""
        from warp_drive_amd.training.update_kernels import FusedObjective

        if kernels is None:
            raise ValueError("compute_loss_and_metrics_from_logits needs the caller's UpdateKernels (kernels=...); "
                             "the framework objective is compute_loss_and_metrics")
        values_detached = out[..., -1].detach()
        out_d = out.detach()
        if kernels.supports_discounted_returns(rewards_batch, done_flags_batch, out_d):
            # (one kernel instead of a Python loop over T of four small ones: launch-bound, ~1.3 ms per policy)
            returns, advantages = kernels.discounted_returns(rewards_batch, done_flags_batch, out_d, self.discount_factor_gamma)
        else:
            returns = discounted_returns(rewards_batch, done_flags_batch, values_detached, self.discount_factor_gamma)
            advantages = None
        norm_returns = _normalise(returns) if self.normalize_return else returns
        if advantages is None or self.normalize_return:
            advantages = norm_returns - values_detached
        norm_adv = _normalise(advantages) if self.normalize_advantage else advantages
        vf_c = self.vf_loss_coeff_schedule.get_param_value(timestep)
        ent_c = self.entropy_coeff_schedule.get_param_value(timestep)
        W = out.shape[-1]
        loss, terms = FusedObjective.apply(kernels, out.reshape(-1, W), actions_batch.reshape(-1, actions_batch.shape[-1]).to(torch.int32).contiguous(),
                                           norm_adv.reshape(-1).contiguous(), norm_returns.reshape(-1).contiguous(),
                                           tuple(int(a) for a in head_sizes), float(ent_c), float(vf_c), self.clip_param is not None)
        metrics = {}
        if perform_logging:
            policy_loss, vf_loss, mean_entropy = (float(v) for v in terms.tolist())
            var_explained = torch.clamp(1 - norm_adv.var() / (norm_returns.var() + _EPSILON), min=-1.0)
            metrics = {
                "VF loss coefficient": vf_c, "Entropy coefficient": ent_c, "Total loss": loss.item(),
                "Policy loss": policy_loss, "Value function loss": vf_loss,
                "Mean rewards": rewards_batch.mean().item(), "Max. rewards": rewards_batch.max().item(),
                "Min. rewards": rewards_batch.min().item(), "Mean value function": values_detached.mean().item(),
                "Mean advantages": advantages.mean().item(), "Mean (norm.) advantages": norm_adv.mean().item(),
                "Mean (discounted) returns": returns.mean().item(), "Mean normalized returns": norm_returns.mean().item(),
                "Mean entropy": mean_entropy, "Variance explained by the value function": var_explained.item(),
            }
            af = actions_batch.float()
            over_agents, over_time, over_envs = (af.std(dim=d).mean(dim=(0, 1)) for d in (2, 0, 1))
            for h in range(af.shape[-1]):
                metrics[f"Std. of action_{h} over agents"] = over_agents[h].item()
                metrics[f"Std. of action_{h} over envs"] = over_envs[h].item()
                metrics[f"Std. of action_{h} over time"] = over_time[h].item()
        return loss, metrics


class PPO(A2C):
    def __init__(self, clip_param=0.1, **kwargs):
        super().__init__(**kwargs)
        assert 0 <= clip_param <= 1
        self.clip_param = clip_param

    def _policy_loss(self, log_prob, advantages):
        # single-epoch PPO exactly as the reference: ratio against the detached same-batch log-prob
        ratio = torch.exp(log_prob - log_prob.detach())
        surr1 = ratio * advantages
        surr2 = torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param) * advantages
        return -torch.minimum(surr1, surr2).mean()
