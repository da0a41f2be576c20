"""The SHIPPED update path, composed, at the size where it is selected.

At BASELINE configs[2] the update of a [256, 256] float32 policy is

    HipPolicyGradientHead -> HipHeadBackwardBx3_W43 -> HipWeightGradBx3_256x256 -> HipLinearMaskBackwardBx3_256
                          -> HipWeightGradBx3_256x96

on activations the rollout stored (training/models.py::_MlpTwoHidden.backward).  The two weight-gradient kernels and the
matrix-core output-layer pass only engage at >= 65 536 rows, so every test here runs ABOVE that threshold with a ragged row
count (R % 32 != 0: the host-side tail), asserts through the driver's launch counters that all of them actually ran, and
compares every parameter gradient with float64 autograd of the same network on the same batch -- what the reference's
update is (stock PyTorch: trainer_a2c.py:159-339, a2c.py:40-194, ppo.py:150-228), at the precision that can judge a
float32 result."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

_MATRIX_CORE_UPDATE = ("HipPolicyGradientHead", "HipHeadBackwardBx3_W43", "HipWeightGradBx3_256x256",
                       "HipLinearMaskBackwardBx3_256", "HipWeightGradBx3_256x96")


def _launch_counts():
    from warp_drive_amd.managers import hip_driver as drv

    return {k: drv.LAUNCH_COUNTS[k] for k in _MATRIX_CORE_UPDATE}


def _float64_reference(model, obs, actions, rewards, done, objective, heads):
    """float64 autograd of the same network / objective on the same batch: {parameter name: gradient}, loss"""
    ref = copy.deepcopy(model).double()
    ref.update_kernels = None
    probs, values = ref(obs.double())
    loss, _ = objective.compute_loss_and_metrics(
        timestep=0, actions_batch=actions.long(), rewards_batch=rewards.double(), done_flags_batch=done,
        action_probabilities_batch=probs, value_functions_batch=values, perform_logging=False)
    loss.backward()
    return {n: p.grad.detach().clone() for n, p in ref.named_parameters()}, float(loss)


def _float32_framework(model, obs, actions, rewards, done, objective):
    """the framework's own float32 autograd (no kernels of this repository): the yardstick for float32 rounding"""
    ref = copy.deepcopy(model)
    ref.update_kernels = None
    probs, values = ref(obs)
    loss, _ = objective.compute_loss_and_metrics(
        timestep=0, actions_batch=actions.long(), rewards_batch=rewards, done_flags_batch=done,
        action_probabilities_batch=probs, value_functions_batch=values, perform_logging=False)
    loss.backward()
    return {n: p.grad.detach().clone() for n, p in ref.named_parameters()}


@pytest.mark.parametrize("algo,normalise", [("A2C", False), ("PPO", True)])
def test_composed_backward_on_stored_activations_vs_float64_autograd(algo, normalise):
    """forward_logits_stored + the fused objective + _MlpTwoHidden.backward for ONE [256, 256] policy at 200 replicas x 105
    agents x 7 ticks = 147 000 rows (147 000 % 32 = 24): all four matrix-core kernels + the objective kernel launched, every
    parameter gradient within 4 x the framework's own float32 error of float64 autograd (floor: 2e-6 of the largest
    entry), the loss to 1e-6."""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers.function_manager import HIPFunctionManager
    from warp_drive_amd.training.losses import A2C, PPO
    from warp_drive_amd.training.models import FullyConnected
    from warp_drive_amd.training.update_kernels import UpdateKernels

    require_gpu()
    dev = torch.device("cuda:0")
    fm = HIPFunctionManager(num_agents=1, num_envs=1)
    fm.load_hip_from_binary_file()
    kernels = UpdateKernels(fm)
    torch.manual_seed(11 + normalise)
    T, E, n, F, heads = 7, 200, 105, 71, [21, 21]
    R = T * E * n
    assert R >= 131072 and R % 32 != 0
    model = FullyConnected(F, heads, (256, 256)).to(dev)
    model.update_kernels = kernels
    obs = torch.randn(T, E, n, F, device=dev) * (torch.rand(T, E, n, F, device=dev) < 0.8)  # (zeros, as real rows have)
    actions = torch.stack([torch.randint(0, a, (T, E, n), device=dev) for a in heads], dim=-1).to(torch.int32)
    rewards = torch.randn(T, E, n, device=dev) * (torch.rand(T, E, n, device=dev) < 0.2)
    done = (torch.rand(T, E, device=dev) < 0.1).to(torch.int32)
    kw = dict(discount_factor_gamma=0.98, normalize_advantage=normalise, normalize_return=normalise, vf_loss_coeff=0.5,
              entropy_coeff=0.03)
    objective = A2C(**kw) if algo == "A2C" else PPO(clip_param=0.1, **kw)
    plan = kernels.update_plan(model, R)
    assert sorted(v for v in plan.values()) == sorted(_MATRIX_CORE_UPDATE), plan   # what the trainer would log at start

    # what the rollout's forward kernel leaves behind: the post-ReLU activations of both layers and the outputs
    with torch.no_grad():
        x = obs.reshape(-1, F)
        h1 = torch.relu(model.fc["0"][0](x))
        h2 = torch.relu(model.fc["1"][0](h1))
        w3 = torch.cat([h.weight for h in model.policy_head] + [model.vf_head.weight], dim=0)
        b3 = torch.cat([h.bias for h in model.policy_head] + [model.vf_head.bias], dim=0)
        out = torch.addmm(b3, h2, w3.t())
    before = _launch_counts()
    logits = model.forward_logits_stored(obs, h1.view(T, E, n, 256), h2.view(T, E, n, 256), out.view(T, E, n, -1))
    loss, _ = objective.compute_loss_and_metrics_from_logits(0, logits, actions, rewards, done, heads, False, kernels=kernels)
    loss.backward()
    torch.cuda.synchronize()
    after = _launch_counts()
    for name in _MATRIX_CORE_UPDATE:
        assert after[name] == before[name] + 1, (name, before[name], after[name])
    got = {name: p.grad.detach().clone() for name, p in model.named_parameters()}

    want, loss64 = _float64_reference(model, obs, actions, rewards, done, objective, heads)
    f32 = _float32_framework(model, obs, actions, rewards, done, objective)
    assert abs(float(loss) - loss64) <= 1e-6 * max(1.0, abs(loss64)), (float(loss), loss64)
    report = {}
    for name in want:
        scale = float(want[name].abs().max())
        err = float((got[name].double() - want[name]).abs().max())
        err_f32 = float((f32[name].double() - want[name]).abs().max())
        report[name] = (err / scale, err_f32 / scale)
        assert err <= max(4.0 * err_f32, 2e-6 * scale), (name, err, err_f32, scale)
    print("relative error (kernels, framework float32) per parameter:", {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in report.items()})


def test_trainer_update_over_the_threshold_stored_vs_recomputed_vs_framework(tmp_path):
    """`Trainer._update_model_params` with both policies over 65 536 rows and ragged (tagger: 66 x 201 x 5 = 66 330 rows,
    66 330 % 32 = 26; runner: 1 326 600 rows, % 32 = 8): THREE trainers in one process on the same rollout --

      stored      the shipped default: the update reads the activations the rollout's forward kernel stored
      recomputed  `reuse_rollout_activations: False`: same kernels, forward pass recomputed by the framework
      framework   `fused_update: False`: stock autograd, no kernel of training/update_kernels.py (possible side by side
                  since the kernel handle lives on the trainer, not in a process global)

    same batch (the sampled actions do not depend on the switches), every matrix-core kernel launched once per policy in
    the first two, none in the third; judged by float64 autograd of the same networks on the same batch: losses to 1e-5,
    every parameter gradient of `stored` and `recomputed` within 4 x the framework update's own float32 error (floor:
    2e-6 of the largest entry)."""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers import hip_driver as drv
    from warp_drive_amd.training.scripts.train import setup_trainer

    require_gpu()
    E, T = 201, 66
    trainers = {}
    for name, switches in (("stored", {}), ("recomputed", {"reuse_rollout_activations": False}), ("framework", {"fused_update": False})):
        ov = {"trainer": {"num_envs": E, "train_batch_size": E * T, "num_episodes": 4000, "seed": 5,
                          "fused_policy_forward_min_rows": 0, **switches},
              "env": {"episode_length": 40}, "saving": {"metrics_log_freq": 1, "model_params_save_freq": 0}}
        torch.manual_seed(0)
        trainers[name] = setup_trainer("tag_continuous", ov, results_dir=str(tmp_path / name), verbose=False)
    a, b, c = trainers["stored"], trainers["recomputed"], trainers["framework"]
    assert a._stored is not None and b._stored is None and c._stored is None
    assert a._update_kernels is not None and c._update_kernels is None and not c._fused_update
    assert all(m.update_kernels is None for m in c.models.values())
    for pol in a.policies:   # on record at start: the whole update on this repository's kernels
        assert not [v for v in a.update_plan[pol].values() if v.startswith("framework")], a.update_plan[pol]
        assert sorted(a.update_plan[pol].values()) == sorted(_MATRIX_CORE_UPDATE)
    for tr in trainers.values():
        tr._generate_rollout_batch()
    torch.cuda.synchronize()
    for pol in a.policies:
        rows = a.batch[pol]["obs"][:T].numel() // a.batch[pol]["obs"].shape[-1]
        assert rows >= 65536 and rows % 32 != 0, rows
        for tr in (b, c):
            assert torch.equal(a.batch[pol]["obs"], tr.batch[pol]["obs"]) and torch.equal(a.batch[pol]["actions"], tr.batch[pol]["actions"])
    # float64 autograd of the same networks on the same batch, before any optimizer steps: the judge of float32 results
    want = {}
    for pol in a.policies:
        batch = c.batch[pol]
        want[pol] = _float64_reference(c.models[pol], batch["obs"][:T], batch["actions"][:T], batch["rewards"][:T],
                                       c.done_batch[:T], c.trainers[pol], a.head_sizes)
    grads, losses = {}, {}
    for name, tr in trainers.items():
        before = _launch_counts()
        valid = {pol: tr._stored_activations_valid(pol) for pol in tr.policies}
        assert all(valid.values()) == (name == "stored") and any(valid.values()) == (name == "stored")
        tr.grad_bucket.zero()
        metrics = tr._update_model_params(0, True)
        torch.cuda.synchronize()
        after = _launch_counts()
        for k in _MATRIX_CORE_UPDATE:
            assert after[k] - before[k] == (0 if name == "framework" else len(tr.policies)), (name, k, before[k], after[k])
        # (the optimizer has stepped inside _update_model_params: .grad is what it stepped with)
        grads[name] = {pol: {n: p.grad.detach().clone() for n, p in tr.models[pol].named_parameters()} for pol in tr.policies}
        losses[name] = {pol: metrics[pol]["Total loss"] for pol in tr.policies}
    report = {}
    for pol in a.policies:
        ref, loss64 = want[pol]
        for name in trainers:
            assert abs(losses[name][pol] - loss64) <= 1e-5 * max(1.0, abs(loss64)), (name, pol, losses[name][pol], loss64)
        for n in ref:
            scale = float(ref[n].abs().max())
            err = {name: float((grads[name][pol][n].double() - ref[n]).abs().max()) for name in trainers}
            report[(pol, n)] = {k: f"{v / scale:.1e}" for k, v in err.items()}
            for name in ("stored", "recomputed"):
                assert err[name] <= max(4.0 * err["framework"], 2e-6 * scale), (name, pol, n, err, scale)
    print("gradient error relative to the largest entry of the float64 gradient:", report)
    for tr in trainers.values():
        tr.graceful_close()
    assert drv.LAUNCH_COUNTS["HipPolicyGradientHead"] > 0


def test_stale_stored_activations_are_not_differentiated(tmp_path):
    """The stored activations belong to the weights the rollout's forward kernel read.  A change of the parameters between
    rollout and update that nobody announced -- `load_state_dict`, a manual in-place edit -- must make the update recompute
    its forward pass (parameter version counters, recorded when the weights were packed), not differentiate stale
    activations against the new weights: the gradients then equal those of a trainer that never stored anything."""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.training.scripts.train import setup_trainer

    require_gpu()
    trainers = {}
    for reuse in (True, False):
        ov = {"trainer": {"num_envs": 29, "train_batch_size": 29 * 10, "num_episodes": 4000, "seed": 3,
                          "reuse_rollout_activations": reuse, "fused_policy_forward_min_rows": 0},
              "env": {"num_runners": 40, "episode_length": 8, "num_other_agents_observed": 10},
              "saving": {"metrics_log_freq": 1, "model_params_save_freq": 0}}
        torch.manual_seed(0)
        trainers[reuse] = setup_trainer("tag_continuous", ov, results_dir=str(tmp_path / f"s{int(reuse)}"), verbose=False)
    a, b = trainers[True], trainers[False]
    assert a._stored is not None and a._fast_tick is not None and b._stored is None
    for tr in (a, b):
        tr._generate_rollout_batch()
    assert all(a._stored_activations_valid(pol) for pol in a.policies)
    # the same unannounced change in both trainers: new weights through load_state_dict for one policy, an in-place edit
    # of one bias for the other
    torch.manual_seed(99)
    new_state = {k: v + 0.05 * torch.randn_like(v) for k, v in a.models["runner"].state_dict().items()}
    for tr in (a, b):
        tr.models["runner"].load_state_dict(new_state)
        with torch.no_grad():
            tr.models["tagger"].fc["1"][0].bias.add_(0.1)
    assert not any(a._stored_activations_valid(pol) for pol in a.policies)
    grads = {}
    for name, tr in (("a", a), ("b", b)):
        tr.grad_bucket.zero()
        tr._update_model_params(0, False)
        grads[name] = {pol: [p.grad.detach().clone() for p in tr.models[pol].parameters()] for pol in tr.policies}
    for pol in a.policies:
        for ga, gb in zip(grads["a"][pol], grads["b"][pol]):
            # the same recomputing path on the same inputs
            assert float((ga - gb).abs().max()) <= 1e-6 * max(float(gb.abs().max()), 1e-12), pol
    # and the next rollout stores again for the weights it ran with
    a._generate_rollout_batch()
    assert all(a._stored_activations_valid(pol) for pol in a.policies)
    for tr in (a, b):
        tr.graceful_close()


def test_graph_capture_leaves_no_trace(tmp_path):
    """Capturing the tick in a hipGraph runs three real warm-up ticks; they must not be visible afterwards: a trainer
    that replays the graph and one that runs eager ticks, same seed, produce the same first batch (observations, actions,
    rewards, done flags) and the same episodic counters."""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.training.scripts.train import setup_trainer

    require_gpu()
    trainers = {}
    for graph in (True, False):
        ov = {"trainer": {"num_envs": 40, "train_batch_size": 40 * 12, "num_episodes": 400, "seed": 21, "graph_rollout": graph},
              "env": {"num_runners": 20, "episode_length": 9, "num_other_agents_observed": 6},
              "saving": {"metrics_log_freq": 1, "model_params_save_freq": 0}}
        torch.manual_seed(0)
        trainers[graph] = setup_trainer("tag_continuous", ov, results_dir=str(tmp_path / f"g{int(graph)}"), verbose=False)
    g, e = trainers[True], trainers[False]
    for tr in (g, e):
        tr._generate_rollout_batch()
    torch.cuda.synchronize()
    assert g._tick_graph is not None and e._tick_graph is None
    for pol in g.policies:
        for key in ("obs", "actions", "rewards"):
            assert torch.equal(g.batch[pol][key], e.batch[pol][key]), (pol, key)
        assert torch.equal(g._ep_sum[pol], e._ep_sum[pol]) and torch.equal(g._ep_reward[pol], e._ep_reward[pol])
    assert torch.equal(g.done_batch, e.done_batch) and torch.equal(g._ep_cnt, e._ep_cnt)
    assert float(e._ep_cnt.sum()) >= 40.0   # 12 ticks of 9-tick episodes: every replica finished (at least) one
    for tr in (g, e):
        tr.graceful_close()
