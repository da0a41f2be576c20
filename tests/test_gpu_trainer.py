"""The device-resident A2C / PPO training loop end to end on the MI355X (mirror of the reference's
tests/wd_training/pycuda_tests/test_env_training.py:56-92, shrunk to seconds)."""
import glob
import os

import pytest
import numpy as np
import torch

pytestmark = pytest.mark.gpu


def _train(env_name, overrides, tmp_path, iters=3):
    from tests.hip_harness import require_gpu
    from warp_drive_amd.training.scripts.train import setup_trainer

    require_gpu()
    trainer = setup_trainer(env_name, overrides, results_dir=str(tmp_path), verbose=False)
    before = {p: [w.detach().clone() for w in m.parameters()] for p, m in trainer.models.items()}
    metrics = trainer.train(iters)
    trainer.graceful_close()
    for pol, m in trainer.models.items():
        assert any((a != b).any() for a, b in zip(before[pol], m.parameters())), f"{pol} did not learn"
        assert all(torch.isfinite(w).all() for w in m.parameters())
        assert torch.isfinite(torch.tensor(metrics[pol]["Total loss"]))
    return trainer, metrics


def test_train_tag_continuous(tmp_path):
    ov = {"trainer": {"num_envs": 64, "train_batch_size": 64 * 20, "num_episodes": 200, "graph_rollout": True},
          "env": {"num_runners": 20, "episode_length": 30, "num_other_agents_observed": 6},
          "saving": {"metrics_log_freq": 1, "model_params_save_freq": 2}}
    trainer, metrics = _train("tag_continuous", ov, tmp_path, iters=3)
    assert trainer.num_iters == 200 * 30 // (64 * 20)  # total steps // batch (trainer_base.py:268-269)
    assert trainer.engine.fused  # the rollout is the single fused tick kernel
    assert trainer._tick_graph is not None  # ... replayed, with the policy forward, from a hipGraph
    assert set(metrics) == {"runner", "tagger"}
    assert trainer.perf_stats.get_perf_stats()["Mean steps per sec (rollout)"] > 0
    ckpts = sorted(glob.glob(os.path.join(str(tmp_path), "*.state_dict")))
    assert any(os.path.basename(c).startswith("runner_") for c in ckpts)
    # resume: weights and the timestep parsed from the file name
    last = [c for c in ckpts if os.path.basename(c).startswith("runner_")][-1]
    trainer.load_model_checkpoint({"runner": last})
    assert trainer.current_timestep["runner"] == int(os.path.basename(last).split(".state_dict")[0].split("_")[-1])
    assert os.path.exists(os.path.join(str(tmp_path), "results.json"))
    # episodes of length 30 with 20-tick batches: every replica finishes an episode inside the third
    # batch (tick 60), so the statistic of the last logged iteration is defined
    import math

    for pol in ("runner", "tagger"):
        assert math.isfinite(metrics[pol]["Mean episodic reward"]), metrics[pol]


def test_train_gridworld_and_cartpole(tmp_path):
    ov = {"trainer": {"num_envs": 50, "train_batch_size": 50 * 25, "num_episodes": 50},
          "saving": {"metrics_log_freq": 1, "model_params_save_freq": 0}}
    trainer, metrics = _train("tag_gridworld", ov, tmp_path / "gw")
    assert trainer.engine.fused and set(metrics) == {"runner", "tagger"}
    # the same through separate sampler / step / reset launches (envs without a tick kernel)
    ov["trainer"]["fused_rollout"] = False
    trainer, metrics = _train("tag_gridworld", ov, tmp_path / "gw_unfused", iters=4)
    assert not trainer.engine.fused and len(trainer.engine.entry_names) >= 2
    # the unfused path counts finished episodes from the flags as they were BEFORE the reset launch
    # clears them: 100-tick episodes, 4 x 25 ticks -> the last batch ends on the episode end
    import math

    assert math.isfinite(metrics["runner"]["Mean episodic reward"]), metrics["runner"]
    ov["trainer"].pop("fused_rollout")
    ov["policy"] = {"runner": {"algorithm": "PPO", "to_train": True, "lr": 0.01, "vf_loss_coeff": 1,
                               "model": {"fc_dims": [32]}},
                    "tagger": {"algorithm": "PPO", "to_train": True, "lr": 0.01, "vf_loss_coeff": 1,
                               "model": {"fc_dims": [32]}}}
    _train("tag_gridworld", ov, tmp_path / "gw_ppo")
    ov2 = {"trainer": {"num_envs": 300, "train_batch_size": 300 * 30, "num_episodes": 500},
           "env": {"episode_length": 40}, "saving": {"metrics_log_freq": 1, "model_params_save_freq": 0}}
    _train("single_cartpole", ov2, tmp_path / "cp")


def test_update_in_bf16_and_downsampling_keys(tmp_path):
    """`trainer.update_dtype: bfloat16` runs the update's GEMMs under autocast (float32 master weights, loss and
    optimizer): it must train, and its first loss must agree with the float32 update of the same batch to the
    stated tolerance (5e-2 relative on the total loss: bf16 keeps 8 mantissa bits through two 256-wide
    layers).  `trainer.neg_pos_env_ratio` reaches the objective (trainer_base.py:210): the two counters of
    a2c.py:196-220 are logged."""
    losses = {}
    for dt in ("float32", "bfloat16"):
        ov = {"trainer": {"num_envs": 32, "train_batch_size": 32 * 10, "num_episodes": 100, "update_dtype": dt,
                          "neg_pos_env_ratio": 1, "seed": 7},
              "env": {"num_runners": 20, "episode_length": 30, "num_other_agents_observed": 6},
              "saving": {"metrics_log_freq": 1, "model_params_save_freq": 0}}
        torch.manual_seed(1234)  # the same initial weights in both runs (the rollouts are seeded by the config)
        trainer, metrics = _train("tag_continuous", ov, tmp_path / dt, iters=1)
        assert trainer.neg_pos_env_ratio == 1
        for pol in ("runner", "tagger"):
            assert "Num of Positive Sampled Envs" in metrics[pol] and "Num of Negative Sampled Envs" in metrics[pol]
        assert trainer.grad_bucket is not None and trainer.grad_bucket.attached()
        losses[dt] = {pol: metrics[pol]["Total loss"] for pol in metrics}
    for pol in losses["float32"]:
        a, b = losses["float32"][pol], losses["bfloat16"][pol]
        assert abs(a - b) <= 5e-2 * max(abs(a), 1e-3), (pol, a, b)


def test_cartpole_trains_with_the_whole_batch_rollout_in_one_launch(tmp_path):
    """single_cartpole's [32, 32] policy is evaluated inside the env's rollout kernel: one launch per training
    batch.  It must train (finite loss, changing weights), keep the episodic-reward bookkeeping of the per-tick
    path (Cartpole pays 1 per tick: the mean episodic reward is the mean episode length, between 8 and the
    episode length), and the per-tick path stays available (`fused_rollout_policy: False`)."""
    ov = {"trainer": {"num_envs": 200, "train_batch_size": 200 * 40, "num_episodes": 400},
          "env": {"episode_length": 60}, "saving": {"metrics_log_freq": 1, "model_params_save_freq": 0}}
    trainer, metrics = _train("single_cartpole", ov, tmp_path / "one_launch", iters=3)
    assert trainer._batch_rollout is not None
    assert trainer.engine.step_kernel_name == "HipClassicControlCartPoleEnvRollout_H32"
    assert 8.0 <= metrics["shared"]["Mean episodic reward"] <= 60.0, metrics["shared"]["Mean episodic reward"]
    ov["trainer"]["fused_rollout_policy"] = False
    trainer2, metrics2 = _train("single_cartpole", ov, tmp_path / "per_tick", iters=3)
    assert trainer2._batch_rollout is None
    assert 8.0 <= metrics2["shared"]["Mean episodic reward"] <= 60.0
    # same env, same random policy at the start: the two rollouts see episodes of the same length on average
    assert abs(metrics["shared"]["Mean episodic reward"] - metrics2["shared"]["Mean episodic reward"]) < 8.0


def test_gridworld_trains_with_the_whole_batch_rollout_in_one_launch(tmp_path):
    """tag_gridworld with [32, 32] "tagger" / "runner" policies: both networks are evaluated inside the env's rollout
    kernel (HipTagGridWorldRollout_N5_H32), one launch per training batch; the env-level rows it records are scattered
    into the per-policy batches.  It must train, keep the per-tick path's episodic-reward bookkeeping (same random
    start policies: episodic rewards of the same size), and `fused_rollout_policy: False` keeps the per-tick path; the
    shipped [256, 256] policies take the per-tick path as well."""
    small = {"to_train": True, "algorithm": "A2C", "lr": 0.005, "vf_loss_coeff": 1, "model": {"fc_dims": [32, 32]}}
    ov = {"trainer": {"num_envs": 240, "train_batch_size": 240 * 25, "num_episodes": 400},
          "env": {"episode_length": 50}, "policy": {"runner": dict(small), "tagger": dict(small)},
          "saving": {"metrics_log_freq": 1, "model_params_save_freq": 0}}
    torch.manual_seed(3)
    trainer, metrics = _train("tag_gridworld", ov, tmp_path / "gw_one_launch", iters=4)
    assert trainer._batch_rollout is not None and trainer._batch_rollout["split"] is not None
    assert trainer.engine.step_kernel_name == "HipTagGridWorldRollout_N5_H32"
    b = trainer.batch
    assert b["tagger"]["obs"].shape[2] == 4 and b["runner"]["obs"].shape[2] == 1
    # what the kernel recorded reached the per-policy batches: actions in range, observation rows that carry the
    # agent's own "is me" one-hot (columns 15 .. 19 of a full observation row)
    for pol, ids in (("tagger", [0, 1, 2, 3]), ("runner", [4])):
        a = b[pol]["actions"][: trainer.batch_len]
        assert int(a.min()) >= 0 and int(a.max()) <= 4
        me = b[pol]["obs"][: trainer.batch_len, :, :, 15:20].argmax(dim=-1)
        assert (me == torch.tensor(ids, device=me.device)[None, None, :]).all()
    import math

    for pol in ("tagger", "runner"):
        assert math.isfinite(metrics[pol]["Total loss"]) and math.isfinite(metrics[pol]["Mean episodic reward"])
    ov["trainer"]["fused_rollout_policy"] = False
    torch.manual_seed(3)
    trainer2, metrics2 = _train("tag_gridworld", ov, tmp_path / "gw_per_tick", iters=4)
    assert trainer2._batch_rollout is None
    for pol in ("tagger", "runner"):
        a, c = metrics[pol]["Mean episodic reward"], metrics2[pol]["Mean episodic reward"]
        assert abs(a - c) <= 0.35 * max(abs(a), abs(c), 1.0), (pol, a, c)
    ov["trainer"]["fused_rollout_policy"] = True
    ov["policy"] = {}
    trainer3, _ = _train("tag_gridworld", ov, tmp_path / "gw_256", iters=1)
    assert trainer3._batch_rollout is None   # [256, 256]: the per-tick path


def test_graph_and_eager_rollouts_agree(tmp_path):
    """the hipGraph replay of a rollout tick must produce exactly what the eager tick produces"""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.training.scripts.train import setup_trainer

    require_gpu()
    ov = {"trainer": {"num_envs": 32, "train_batch_size": 32 * 10, "num_episodes": 40, "seed": 7},
          "env": {"num_runners": 12, "episode_length": 16, "num_other_agents_observed": 4},
          "saving": {"metrics_log_freq": 100, "model_params_save_freq": 0}}
    outs = []
    for graph in (True, False):
        torch.manual_seed(0)
        ov["trainer"]["graph_rollout"] = graph
        trainer = setup_trainer("tag_continuous", ov, results_dir=str(tmp_path / f"g{int(graph)}"), verbose=False)
        # (the capture's warm-up ticks are undone before the first replay -- env state, generator, episodic counters:
        # both runs start from the same state; tests/test_gpu_update_composed.py::test_graph_capture_leaves_no_trace)
        trainer._generate_rollout_batch()
        assert (trainer._tick_graph is not None) == graph
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in (("obs", trainer.batch["runner"]["obs"]),
                                               ("act", trainer.batch["runner"]["actions"]),
                                               ("rew", trainer.batch["tagger"]["rewards"]),
                                               ("done", trainer.done_batch))})
        trainer.graceful_close()
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_fused_policy_forward_in_the_rollout(tmp_path):
    """The rollout with the policy forward as one kernel (training/policy_kernel.py, the default for
    float32 rollouts of a supported shape) against the framework path: same observation rows in the
    batch, probabilities equal to summation order, and -- the probabilities being that close -- the same
    sampled actions except where a uniform draw falls within 1e-5 of a CDF step; the kernel keeps
    following the weights through an update."""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.training.scripts.train import setup_trainer

    require_gpu()
    ov = {"trainer": {"num_envs": 40, "train_batch_size": 40 * 8, "num_episodes": 40, "seed": 11},
          "env": {"num_runners": 30, "episode_length": 25, "num_other_agents_observed": 10},
          "saving": {"metrics_log_freq": 100, "model_params_save_freq": 0}}
    res = {}
    for fused in (True, False):
        torch.manual_seed(0)
        ov["trainer"]["fused_policy_forward"] = fused
        ov["trainer"]["fused_policy_forward_min_rows"] = 0  # (the default leaves policies with few rows to the framework)
        ov["trainer"]["fused_tick"] = False  # (this test compares the probability tensors: the three-launch tick never writes them)
        tr = setup_trainer("tag_continuous", ov, results_dir=str(tmp_path / f"f{int(fused)}"), verbose=False)
        assert all((f is not None) == fused for f in tr._fused_forward.values())
        tr._b_rows.zero_()
        tr._tick()  # one tick from identical states and identical initial weights
        torch.cuda.synchronize()
        res[fused] = dict(probs=[p.clone() for p in tr.probs], obs={p: tr.batch[p]["obs"][0].clone() for p in tr.policies},
                          act=tr.actions.clone())
        if fused:
            tr.train(2)  # updates re-pack the weights: the kernel must agree with the updated network
            flat = tr.obs.reshape(tr.num_envs, tr.w.n_agents, -1)
            for pol in tr.policies:
                want, _ = tr._inference_model(pol)(flat.index_select(1, tr.ids[pol]))
                tr._fused_forward[pol](flat, tr._ids32[pol], tr.probs)
                torch.cuda.synchronize()
                for h, wnt in enumerate(want):
                    got = tr.probs[h].index_select(1, tr.ids[pol])
                    assert torch.allclose(got, wnt, rtol=2e-5, atol=2e-6), (pol, h)
        tr.graceful_close()
    for a, b in zip(res[True]["probs"], res[False]["probs"]):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-6)
    for pol in res[True]["obs"]:
        assert torch.equal(res[True]["obs"][pol], res[False]["obs"][pol])
    assert (res[True]["act"] != res[False]["act"]).float().mean().item() < 1e-3


def test_fetch_episode_states_matches_the_oracle(tmp_path):
    """f4: Trainer.fetch_episode_states (reference trainer_base.py:689-792) -- one replica's states,
    actions and rewards for a whole episode, logged on the device by HIPLogController and pulled once.
    The logged actions are replayed through the oracle from the same seeded start state: positions,
    flags and rewards must match bit-exactly at every tick, and the log must stop at the episode end."""
    import numpy as np
    import yaml

    from oracle.tag_continuous_np import TagContinuousOracle
    from tests.hip_harness import require_gpu
    from warp_drive_amd.training.scripts import train as train_script
    from warp_drive_amd.training.scripts.train import setup_trainer

    require_gpu()
    ov = {"trainer": {"num_envs": 16, "train_batch_size": 16 * 10, "num_episodes": 40, "seed": 11},
          "env": {"num_runners": 14, "episode_length": 25, "num_other_agents_observed": 5, "tagging_distance": 0.3},
          "saving": {"metrics_log_freq": 100, "model_params_save_freq": 0}}
    trainer = setup_trainer("tag_continuous", ov, results_dir=str(tmp_path), verbose=False)
    trainer.train(1)
    names = ["loc_x", "loc_y", "still_in_the_game"]
    env_id = 5
    states, actions, rewards, probs = trainer.fetch_episode_states(
        names, env_id=env_id, include_rewards_actions=True, include_probabilities=True)
    T = trainer.w.episode_length
    assert set(states) == set(names) and states["loc_x"].shape == (T + 1, trainer.w.n_agents)
    assert actions.shape == (T, trainer.w.n_agents, 2) and rewards.shape == (T, trainer.w.n_agents)
    assert states["loc_x"].dtype == np.float64
    # replay through the oracle
    cfg = yaml.safe_load(open(os.path.join(train_script._CONFIGS, "tag_continuous.yaml")))["env"]
    cfg.update(ov["env"])
    orc = TagContinuousOracle(num_envs=1, **cfg)
    np.testing.assert_array_equal(states["loc_x"][0], orc.loc_x[0].astype(np.float64))
    end = None
    for t in range(T):
        orc.step(actions[t][None].astype(np.int32))
        np.testing.assert_array_equal(states["loc_x"][t + 1], orc.loc_x[0], err_msg=f"t={t}")
        np.testing.assert_array_equal(states["loc_y"][t + 1], orc.loc_y[0])
        np.testing.assert_array_equal(states["still_in_the_game"][t + 1], orc.sig[0])
        np.testing.assert_array_equal(rewards[t], orc.rewards[0])
        assert set(probs[t]) == {"runner", "tagger"} and len(probs[t]["runner"]) == 2
        if orc.done[0]:
            end = t + 1
            break
    assert end is not None and end <= T
    assert np.isnan(states["loc_x"][end + 1:]).all() and (rewards[end:] == 0).all()
    assert max(probs) == end - 1
    # a second call restarts from the seeded state: same start row
    again = trainer.fetch_episode_states(["loc_x"], env_id=0)
    np.testing.assert_array_equal(again["loc_x"][0], states["loc_x"][0])
    trainer.graceful_close()


def test_inference_forward_on_device():
    """fused-epilogue rollout forward vs the training forward on the GPU, float32 and bf16"""
    import numpy as np

    from tests.hip_harness import require_gpu
    from warp_drive_amd.training.models import FullyConnected

    require_gpu()
    torch.manual_seed(3)
    model = FullyConnected(71, [21, 21], [256, 256]).cuda()
    obs = torch.randn(64, 105, 71, device="cuda")
    probs, vals = model(obs)
    probs_i, vals_i = model.forward_inference(obs)
    for a, b in zip(probs, probs_i):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(vals.detach().cpu().numpy(), vals_i.cpu().numpy(), rtol=1e-4, atol=1e-5)
    probs_b, _ = model.forward_inference(obs, dtype=torch.bfloat16)
    for a, b in zip(probs, probs_b):
        assert b.dtype == torch.float32 and float((a.detach() - b).abs().max()) < 2e-2
        np.testing.assert_allclose(b.sum(-1).cpu().numpy(), 1.0, rtol=1e-5)


def test_three_launch_tick_equals_the_per_op_tick(tmp_path):
    """`trainer.fused_tick` (all policies' forward in one launch with the actions drawn in its epilogue -> the env's
    step + reset entry on given actions -> one bookkeeping kernel) against the path it replaces (a forward launch per
    policy writing probabilities, the env's sampling tick, ~20 framework ops): the SAME probabilities go through the
    SAME Philox counters and the same inverse-CDF search, so the two trainers must stay bit-identical -- sampled
    actions, every batch row (observations, actions, rewards, done), the generator's epochs, the env state -- through
    two whole rollouts that include episode ends, and their episodic-reward metric must agree."""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers import hip_driver as drv
    from warp_drive_amd.training.scripts.train import setup_trainer

    require_gpu()
    ov = {"trainer": {"num_envs": 37, "train_batch_size": 37 * 12, "num_episodes": 4000, "seed": 11,
                      "fused_policy_forward_min_rows": 0},
          "env": {"num_runners": 59, "episode_length": 9, "num_other_agents_observed": 10},
          "saving": {"metrics_log_freq": 100, "model_params_save_freq": 0}}
    trainers = {}
    for fast in (True, False):
        torch.manual_seed(0)
        ov["trainer"]["fused_tick"] = fast
        tr = setup_trainer("tag_continuous", ov, results_dir=str(tmp_path / f"t{int(fast)}"), verbose=False)
        assert (tr._fast_tick is not None) == fast
        trainers[fast] = tr
    a, b = trainers[True], trainers[False]
    assert a._presampled_engine.step_kernel_name.startswith("HipTagContinuousTickA_K10")
    assert b.engine.step_kernel_name.startswith("HipTagContinuousTick_K10")
    def words(tr):
        out = np.zeros(4 + tr.num_envs * tr.w.n_agents, dtype=np.uint32)
        drv.memcpy_dtoh(out, tr.sampler.rng_state)
        torch.cuda.synchronize()
        return out

    for rollout in range(2):
        for tr in (a, b):
            tr._generate_rollout_batch()
        torch.cuda.synchronize()
        assert int(a._b_idx.item()) == a.batch_len == int(b._b_idx.item())
        np.testing.assert_array_equal(words(a), words(b))
        assert torch.equal(a.actions, b.actions) and torch.equal(a.obs, b.obs) and torch.equal(a.done_batch, b.done_batch)
        assert int((a.done_batch > 0).sum()) >= a.num_envs  # episodes did end inside the batch
        for pol in a.policies:
            for key in ("obs", "actions", "rewards"):
                assert torch.equal(a.batch[pol][key][: a.batch_len], b.batch[pol][key][: b.batch_len]), (rollout, pol, key)
            assert torch.equal(a._ep_reward[pol], b._ep_reward[pol])
            assert torch.allclose(a._ep_sum[pol].sum(), b._ep_sum[pol].sum(), rtol=1e-5)
        assert torch.equal(a._ep_cnt, b._ep_cnt) and float(a._ep_cnt.sum()) > 0
    for tr in (a, b):  # and it trains
        m = tr.train(2)
        assert all(np.isfinite(m[pol]["Total loss"]) for pol in tr.policies)
        tr.graceful_close()


@pytest.mark.parametrize("algo,heads,norm", [("A2C", [21, 21], False), ("PPO", [21, 21], True), ("A2C", [5], True)])
def test_fused_objective_equals_autograd(algo, heads, norm):
    """HipPolicyGradientHead (objective + gradient with respect to the network's output in one kernel) against the
    framework path it replaces -- softmax, Categorical log-probability / entropy, MSE, autograd -- on the same
    output tensor: loss and every logged term to 1e-6 relative, the gradient to 1e-6 of its largest entry.
    (The framework path itself is pinned to the reference's objectives by tests/test_trainer_cpu.py.)"""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers.function_manager import HIPFunctionManager
    from warp_drive_amd.training import update_kernels
    from warp_drive_amd.training.losses import A2C, PPO

    require_gpu()
    fm = HIPFunctionManager(num_agents=1, num_envs=1)
    fm.load_hip_from_binary_file()
    kernels = update_kernels.UpdateKernels(fm)
    torch.manual_seed(len(heads) + norm)
    T, E, n, W = 7, 61, 9, sum(heads) + 1
    dev = torch.device("cuda:0")
    out = (torch.randn(T, E, n, W, device=dev) * 2.0).requires_grad_(True)
    actions = torch.stack([torch.randint(0, a, (T, E, n), device=dev) for a in heads], dim=-1).to(torch.int32)
    rewards = torch.randn(T, E, n, device=dev)
    done = (torch.rand(T, E, device=dev) < 0.1).to(torch.int32)
    kw = dict(discount_factor_gamma=0.97, normalize_advantage=norm, normalize_return=norm, vf_loss_coeff=0.7,
              entropy_coeff=0.03)
    obj = A2C(**kw) if algo == "A2C" else PPO(clip_param=0.2, **kw)
    loss_f, m_f = obj.compute_loss_and_metrics_from_logits(0, out, actions, rewards, done, heads, True, kernels=kernels)
    (g_f,) = torch.autograd.grad(loss_f, out)
    probs, start = [], 0
    for a in heads:
        probs.append(torch.softmax(out[..., start:start + a], dim=-1))
        start += a
    loss_r, m_r = obj.compute_loss_and_metrics(timestep=0, actions_batch=actions.long(), rewards_batch=rewards,
                                               done_flags_batch=done, action_probabilities_batch=probs,
                                               value_functions_batch=out[..., start], perform_logging=True)
    (g_r,) = torch.autograd.grad(loss_r, out)
    assert abs(float(loss_f) - float(loss_r)) <= 1e-6 * max(1.0, abs(float(loss_r))), (float(loss_f), float(loss_r))
    assert float((g_f - g_r).abs().max()) <= 1e-6 * float(g_r.abs().max()), float((g_f - g_r).abs().max())
    assert set(m_f) == set(m_r)
    for k in m_r:
        assert abs(m_f[k] - m_r[k]) <= 2e-6 * max(1.0, abs(m_r[k])), (k, m_f[k], m_r[k])
    with pytest.raises(ValueError):   # the kernel path without a handle is an error, not a silent framework run
        obj.compute_loss_and_metrics_from_logits(0, out, actions, rewards, done, heads, True)


@pytest.mark.parametrize("R,C", [(10007, 256), (4096, 64), (333, 128)])
def test_relu_backward_with_column_sums(R, C):
    """HipReluBackwardColumnSums: mask + bias gradient in one pass = threshold_backward followed by a column sum"""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers.function_manager import HIPFunctionManager
    from warp_drive_amd.training.update_kernels import UpdateKernels

    require_gpu()
    fm = HIPFunctionManager(num_agents=1, num_envs=1)
    fm.load_hip_from_binary_file()
    k = UpdateKernels(fm)
    torch.manual_seed(R)
    g = torch.randn(R, C, device="cuda")
    y = torch.relu(torch.randn(R, C, device="cuda"))
    assert k.supports_relu_backward(g, y)
    got, sums = k.relu_backward_colsum(g, y)
    want = torch.ops.aten.threshold_backward(g, y, 0)
    assert torch.equal(got, want)
    assert torch.allclose(sums, want.sum(0), rtol=1e-5, atol=1e-4)


def test_update_from_stored_rollout_activations(tmp_path):
    """`trainer.reuse_rollout_activations`: the rollout's forward kernel stores the hidden activations and the outputs of
    every batch row and the update's forward pass is a read of them.  Against the update that recomputes its forward
    pass (framework GEMMs) on the SAME batch and weights: the stored tensors equal the recomputed ones to float32
    rounding (the outputs up to the per-head shift, which softmax does not see), the loss to 1e-5 relative and every
    parameter's gradient to 1e-4 of its largest entry; and after two training iterations of both trainers the weights
    still agree closely (same rollouts: the sampled actions do not depend on the switch)."""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.training.scripts.train import setup_trainer

    require_gpu()
    ov = {"trainer": {"num_envs": 29, "train_batch_size": 29 * 10, "num_episodes": 4000, "seed": 3,
                      "fused_policy_forward_min_rows": 0},
          "env": {"num_runners": 40, "episode_length": 8, "num_other_agents_observed": 10},
          "saving": {"metrics_log_freq": 1, "model_params_save_freq": 0}}
    trainers = {}
    for reuse in (True, False):
        torch.manual_seed(0)
        ov["trainer"]["reuse_rollout_activations"] = reuse
        tr = setup_trainer("tag_continuous", ov, results_dir=str(tmp_path / f"s{int(reuse)}"), verbose=False)
        assert (tr._stored is not None) == reuse and tr._fast_tick is not None
        trainers[reuse] = tr
    a, b = trainers[True], trainers[False]
    for tr in (a, b):
        tr._generate_rollout_batch()
    torch.cuda.synchronize()
    assert a._rollout_filled_stored and not b._rollout_filled_stored
    for pol in a.policies:
        assert torch.equal(a.batch[pol]["obs"], b.batch[pol]["obs"]) and torch.equal(a.batch[pol]["actions"], b.batch[pol]["actions"])
        h1, h2, out = a._stored[pol]
        model = b.models[pol]
        with torch.no_grad():
            x = b.batch[pol]["obs"][: b.batch_len]
            r1 = torch.relu(model.fc["0"][0](x))
            r2 = torch.relu(model.fc["1"][0](r1))
            ref = model.forward_logits(x)
        assert torch.allclose(h1, r1, rtol=1e-5, atol=2e-6) and torch.allclose(h2, r2, rtol=1e-5, atol=2e-6)
        start = 0
        for A in a.head_sizes:  # per head: stored = logits - max(logits)
            z = ref[..., start:start + A]
            assert torch.allclose(out[..., start:start + A], z - z.max(dim=-1, keepdim=True).values, rtol=1e-5, atol=5e-6)
            start += A
        assert torch.allclose(out[..., start], ref[..., start], rtol=1e-5, atol=5e-6)
    grads = {}
    for name, tr in (("stored", a), ("recomputed", b)):
        tr.grad_bucket.zero()
        metrics = tr._update_model_params(0, True)
        grads[name] = ({pol: [p.grad.detach().clone() for p in tr.models[pol].parameters()] for pol in tr.policies}, metrics)
    for pol in a.policies:
        la, lb = grads["stored"][1][pol]["Total loss"], grads["recomputed"][1][pol]["Total loss"]
        assert abs(la - lb) <= 1e-5 * max(1.0, abs(lb)), (pol, la, lb)
        # (the optimizer has stepped inside _update_model_params: compare what it stepped with)
        for ga, gb in zip(grads["stored"][0][pol], grads["recomputed"][0][pol]):
            assert float((ga - gb).abs().max()) <= 1e-4 * max(float(gb.abs().max()), 1e-6), pol
    for tr in (a, b):
        tr.train(2)
    for pol in a.policies:
        for pa, pb in zip(a.models[pol].parameters(), b.models[pol].parameters()):
            assert torch.allclose(pa, pb, rtol=1e-2, atol=2e-3), pol
    for tr in (a, b):
        tr.graceful_close()


@pytest.mark.parametrize("R,W,C", [(10007, 43, 256), (5000, 6, 64), (777, 3, 128), (200007, 43, 256), (70000, 6, 256), (65536 + 31, 3, 256)])
def test_head_backward_kernel(R, W, C):
    """HipHeadBackward_W<w> (vector units) and, for 256 hidden units and >= 65536 rows, HipHeadBackwardBx3_W<w> (bf16 matrix
    cores, bf16x3): the output layer's backward, the hidden layer's ReLU mask + bias gradient and the output layer's weight
    gradient in one pass, against the three framework operations it replaces (ragged row counts: the last R % 32 rows)"""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers.function_manager import HIPFunctionManager
    from warp_drive_amd.training.update_kernels import UpdateKernels

    require_gpu()
    fm = HIPFunctionManager(num_agents=1, num_envs=1)
    fm.load_hip_from_binary_file()
    k = UpdateKernels(fm)
    torch.manual_seed(R)
    g3 = torch.randn(R, W, device="cuda")
    w3 = torch.randn(W, C, device="cuda") * 0.2
    h2 = torch.relu(torch.randn(R, C, device="cuda"))
    assert k.supports_head_backward(g3, w3, h2)
    g2, db2, dw3, db3 = k.head_backward(g3, w3, h2)
    assert (db3 is not None) == (C == 256 and R >= 65536)
    if db3 is not None:
        assert torch.allclose(db3, g3.double().sum(0).float(), rtol=1e-4, atol=1e-2)
    ref_g2 = torch.ops.aten.threshold_backward((g3.double() @ w3.double()).float(), h2, 0)
    assert torch.equal(g2 != 0, ref_g2 != 0) or float(((g2 != 0) != (ref_g2 != 0)).float().mean()) < 1e-6
    assert torch.allclose(g2, ref_g2, rtol=1e-5, atol=1e-5)
    assert torch.allclose(db2, ref_g2.double().sum(0).float(), rtol=1e-4, atol=1e-2)
    assert torch.allclose(dw3, (g3.double().t() @ h2.double()).float(), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("R,C", [(10007, 256), (300, 128), (4097, 64)])
def test_linear_mask_backward_kernel(R, C):
    """HipLinearMaskBackwardBx3_<C>: g_out = [h > 0] * (g_in . W) with the product in bf16x3 arithmetic -- against the
    float64 product: float32-accurate (relative error of the size of a float32 GEMM's own), mask exact, ragged R"""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers.function_manager import HIPFunctionManager
    from warp_drive_amd.training.update_kernels import UpdateKernels

    require_gpu()
    fm = HIPFunctionManager(num_agents=1, num_envs=1)
    fm.load_hip_from_binary_file()
    k = UpdateKernels(fm)
    torch.manual_seed(R + C)
    g = torch.randn(R, C, device="cuda")
    w = torch.randn(C, C, device="cuda") / C ** 0.5
    h = torch.relu(torch.randn(R, C, device="cuda"))
    assert k.supports_linear_mask_backward(g, w, h)
    got = k.linear_mask_backward(g, w, h)
    exact = (g.double() @ w.double()) * (h > 0)
    f32 = (g @ w) * (h > 0)  # the framework's float32 GEMM, for scale
    err, err_f32 = float((got.double() - exact).abs().max()), float((f32.double() - exact).abs().max())
    assert err <= max(4.0 * err_f32, 2e-6), (err, err_f32)
    assert torch.equal(got == 0, exact == 0) or float(((got == 0) != (exact == 0)).float().mean()) < 1e-6


@pytest.mark.parametrize("R,ci,bias", [(70001, 256, False), (65536 + 4, 71, True), (1 << 20, 71, True), (300004, 33, False),
                                       (1 << 20, 256, False)])
def test_weight_grad_kernel(R, ci, bias, monkeypatch):
    """HipWeightGradBx3_256x{256,96}: g^T @ x over the batch in bf16x3 arithmetic (+ the bias gradient as a column of ones)
    -- against the float64 product: float32-accurate (error of the size of the framework's float32 GEMM's own), ragged
    row counts (a last step of fewer than 16 rows, blocks with nothing to do), narrow inputs padded with zero columns"""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers.function_manager import HIPFunctionManager
    from warp_drive_amd.training.update_kernels import UpdateKernels

    require_gpu()
    fm = HIPFunctionManager(num_agents=1, num_envs=1)
    fm.load_hip_from_binary_file()
    k = UpdateKernels(fm)
    torch.manual_seed(R + ci)
    g = torch.randn(R, 256, device="cuda") * (torch.rand(R, 1, device="cuda") < 0.7)   # (whole zero rows, as masked gradients have)
    x = torch.relu(torch.randn(R, ci, device="cuda")) + 0.25
    assert k.supports_weight_grad(g, x, with_bias=bias)
    gw, gb = k.weight_grad(g, x, with_bias=bias)
    exact = g.double().t() @ x.double()
    f32 = g.t() @ x
    err, err_f32 = float((gw.double() - exact).abs().max()), float((f32.double() - exact).abs().max())
    scale = float(exact.abs().max())
    assert gw.shape == (256, ci) and err <= max(4.0 * err_f32, 2e-6 * scale), (err, err_f32, scale)
    if bias:
        exact_b = g.double().sum(0)
        assert float((gb.double() - exact_b).abs().max()) <= 2e-5 * float(exact_b.abs().max()) + 1e-3
    else:
        assert gb is None
    # rows below the threshold and other widths stay with the framework
    assert not k.supports_weight_grad(g[:1000], x[:1000])
    assert not k.supports_weight_grad(g[:, :128].contiguous(), x)


@pytest.mark.parametrize("T,E,n,W", [(50, 200, 100, 43), (7, 61, 9, 6), (1, 3, 5, 3)])
def test_discounted_returns_kernel_is_bit_identical(T, E, n, W):
    """HipDiscountedReturns against losses.discounted_returns (the reference's recursion, a2c.py:80-95): same float32
    operations in the same order -- returns and advantages bit for bit, done flags of 0 / 1 / 2"""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers.function_manager import HIPFunctionManager
    from warp_drive_amd.training.losses import discounted_returns
    from warp_drive_amd.training.update_kernels import UpdateKernels

    require_gpu()
    fm = HIPFunctionManager(num_agents=1, num_envs=1)
    fm.load_hip_from_binary_file()
    k = UpdateKernels(fm)
    torch.manual_seed(T + E)
    out = torch.randn(T, E, n, W, device="cuda")
    rewards = torch.randn(T, E, n, device="cuda") * 3.0
    done = torch.randint(0, 3, (T, E), device="cuda", dtype=torch.int32) * (torch.rand(T, E, device="cuda") < 0.2).to(torch.int32)
    assert k.supports_discounted_returns(rewards, done, out)
    for gamma in (1.0, 0.97):
        got, adv = k.discounted_returns(rewards, done, out, gamma)
        want = discounted_returns(rewards, done, out[..., -1], gamma)
        assert torch.equal(got, want)
        assert torch.equal(adv, want - out[..., -1])


def test_unit_gradient_shortcut_is_loss_backward():
    """`torch.autograd.backward(loss, grad_tensors=unit_gradient(...))` (what the trainer calls) = `loss.backward()`: same
    gradient bit for bit, and FusedObjective.backward recognised the unit tensor (no multiplication pass); a scaled loss
    still takes the multiplication"""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers.function_manager import HIPFunctionManager
    from warp_drive_amd.training import update_kernels
    from warp_drive_amd.training.losses import A2C

    require_gpu()
    fm = HIPFunctionManager(num_agents=1, num_envs=1)
    fm.load_hip_from_binary_file()
    kernels = update_kernels.UpdateKernels(fm)
    torch.manual_seed(5)
    T, E, n, heads = 6, 40, 7, (5, 3)
    W = sum(heads) + 1
    dev = torch.device("cuda:0")
    base = torch.randn(T, E, n, W, device=dev)
    actions = torch.stack([torch.randint(0, a, (T, E, n), device=dev) for a in heads], dim=-1).to(torch.int32)
    rewards = torch.randn(T, E, n, device=dev)
    done = (torch.rand(T, E, device=dev) < 0.1).to(torch.int32)
    obj = A2C(discount_factor_gamma=0.98, vf_loss_coeff=0.5, entropy_coeff=0.02)
    grads = []
    for mode in ("backward", "unit", "scaled"):
        out = base.clone().requires_grad_(True)
        loss, _ = obj.compute_loss_and_metrics_from_logits(0, out, actions, rewards, done, heads, False, kernels=kernels)
        hits = update_kernels.STATS["unit_gradient_hits"]
        if mode == "backward":
            loss.backward()
        elif mode == "unit":
            torch.autograd.backward(loss, grad_tensors=update_kernels.unit_gradient(dev))
            assert update_kernels.STATS["unit_gradient_hits"] == hits + 1
        else:
            (2.0 * loss).backward()
            assert update_kernels.STATS["unit_gradient_hits"] == hits
        grads.append(out.grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert torch.equal(grads[2], 2.0 * grads[0])
