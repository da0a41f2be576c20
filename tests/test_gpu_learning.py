"""Does the trainer LEARN?  (reference README.md:60 shows its Cartpole curve; its tests stop at "finite loss",
tests/wd_training/pycuda_tests/test_env_training.py:56-76 -- but the reference's update is stock PyTorch, here the rollout
and the update are hand-written kernels, and a wrong gradient or a stale behaviour policy trains "finitely" too.)

Seeded, fixed iteration budgets, thresholds chosen from scripts/learning_curves.py runs with a wide margin
(profiles/r06_learning_curves.txt)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _curve(env, overrides, policy, iters, tmp_path):
    from tests.hip_harness import require_gpu
    from warp_drive_amd.training.scripts.train import setup_trainer

    require_gpu()
    overrides = json.loads(json.dumps(overrides))
    overrides["saving"] = {"metrics_log_freq": 1, "model_params_save_freq": 0}
    torch.manual_seed(0)
    tr = setup_trainer(env, overrides, results_dir=str(tmp_path), verbose=False)
    tr.train(iters)
    tr.graceful_close()
    curve = [json.loads(line)[policy]["Mean episodic reward"] for line in open(os.path.join(str(tmp_path), "results.json"))]
    assert len(curve) == iters and all(np.isfinite(curve))
    return tr, curve


_CARTPOLE = {"trainer": {"num_envs": 1000, "train_batch_size": 1000 * 50, "num_episodes": 10 ** 6, "seed": 7},
             "env": {"episode_length": 200}}


@pytest.mark.parametrize("path", ["one launch per batch", "per tick (hipGraph)", "per tick (eager)"])
def test_cartpole_learns(path, tmp_path):
    """Cartpole pays 1 per tick: "Mean episodic reward" is the mean episode length (~19-22 ticks under the random initial
    policy, 200 at most here).  A2C on the [32, 32] policy of run_configs/single_cartpole.yaml must at least TRIPLE it
    within 300 iterations of 50 ticks x 1000 replicas (seeds 0-3 of
    scripts/learning_curves.py reach 4-7 x after 200) -- with the policy evaluated inside the env's rollout kernel (one
    launch per batch) and on the per-tick path (framework forward -> fused tick kernel), replayed from a hipGraph and
    eager."""
    ov = json.loads(json.dumps(_CARTPOLE))
    if path != "one launch per batch":
        ov["trainer"]["fused_rollout_policy"] = False
        ov["trainer"]["graph_rollout"] = path == "per tick (hipGraph)"
    tr, curve = _curve("single_cartpole", ov, "shared", 300, tmp_path)
    assert (tr._batch_rollout is not None) == (path == "one launch per batch")
    assert (tr._tick_graph is not None) == (path == "per tick (hipGraph)")
    first, last = float(np.mean(curve[:3])), float(np.mean(curve[-10:]))
    print(f"cartpole, {path}: mean episode length {first:.1f} -> {last:.1f}")
    assert 15.0 <= first <= 30.0, first
    assert last >= 3.0 * first, (first, last, curve[::10])


@pytest.mark.parametrize("fc_dims,path", [([32, 32], "one launch per batch"), ([256, 256], "per tick")])
def test_gridworld_taggers_learn_to_catch_a_random_runner(fc_dims, path, tmp_path):
    """TagGridWorld, 4 taggers + 1 runner on a 20 x 20 grid (run_configs/tag_gridworld.yaml's rewards: the taggers earn 10
    for a tag and pay 0.01 per tick; random taggers rarely find a random runner on a grid this size within 100 ticks:
    "Mean episodic reward" starts near 0).  With the runner kept at its random initial policy (`to_train: False`) the
    taggers' reward over the last 10 of 60 iterations of 100 ticks x 600 replicas must exceed the first iterations' by 3
    (measured: -0.3 -> 6.2 and 1.1 -> 7.0) -- with the [32, 32] networks evaluated inside the rollout kernel and with the
    [256, 256] networks on the per-tick path (fused forward kernel; the update's matrix-core kernels below their row
    threshold, i.e. partly framework GEMMs -- the plan logged at start says which)."""
    lr = 0.001
    pol = {p: {"to_train": p == "tagger", "algorithm": "A2C", "vf_loss_coeff": 1, "entropy_coeff": 0.05, "gamma": 0.98, "lr": lr,
               "model": {"type": "fully_connected", "fc_dims": fc_dims, "model_ckpt_filepath": ""}} for p in ("runner", "tagger")}
    ov = {"trainer": {"num_envs": 600, "train_batch_size": 600 * 100, "num_episodes": 10 ** 6, "seed": 7}, "policy": pol,
          "env": {"grid_length": 20}}
    tr, curve = _curve("tag_gridworld", ov, "tagger", 60, tmp_path)
    assert (tr._batch_rollout is not None) == (path == "one launch per batch")
    first, last = float(np.mean(curve[:3])), float(np.mean(curve[-10:]))
    print(f"gridworld taggers, {path}: mean episodic reward {first:.2f} -> {last:.2f}")
    assert last >= first + 3.0, (first, last, curve[::5])


@pytest.mark.parametrize("graph", [True, False])
def test_cartpole_per_tick_batch_rows_obey_the_dynamics(graph, tmp_path):
    """What the per-tick path records IS a trajectory: for every replica and every tick t whose step did not end the
    episode, the Euler step (oracle/cartpole_np.py, the restatement of cartpole_step_numba.py:42-78) of the recorded
    observation under the recorded action is the next recorded observation, bit for bit; after an episode's end the next
    row is the start state; rewards are 1.  Over THREE training iterations (the behaviour policy changes in between), with the
    tick replayed from a hipGraph and eager; and the actions follow the network the update is about to differentiate: the
    frequency of action 1 over the batch matches the mean probability the CURRENT model gives it on the recorded rows."""
    from oracle.cartpole_np import CartPoleOracle
    from tests.hip_harness import require_gpu
    from warp_drive_amd.training.scripts.train import setup_trainer

    require_gpu()
    E, T = 500, 40
    ov = {"trainer": {"num_envs": E, "train_batch_size": E * T, "num_episodes": 10 ** 6, "seed": 11, "fused_rollout_policy": False,
                      "graph_rollout": graph},
          "env": {"episode_length": 60}, "saving": {"metrics_log_freq": 1, "model_params_save_freq": 0}}
    torch.manual_seed(0)
    tr = setup_trainer("single_cartpole", ov, results_dir=str(tmp_path), verbose=False)
    assert tr._batch_rollout is None
    start = None
    for it in range(3):
        tr._generate_rollout_batch()
        torch.cuda.synchronize()
        assert (tr._tick_graph is not None) == graph
        b = tr.batch["shared"]
        obs = b["obs"][:T].cpu().numpy().reshape(T, E, 4)
        act = b["actions"][:T].cpu().numpy().reshape(T, E)
        rew = b["rewards"][:T].cpu().numpy().reshape(T, E)
        done = tr.done_batch[:T].cpu().numpy().reshape(T, E)
        if start is None:
            start = obs[0, 0].copy()
            assert (obs[0] == start).all()
        assert (rew == 1.0).all()
        orc = CartPoleOracle(E, episode_length=10 ** 9)
        checked = 0
        for t in range(T - 1):
            orc.state = obs[t].copy()
            orc.done[:] = 0
            nxt, _, _ = orc.step(act[t])
            cont = done[t] == 0
            assert np.array_equal(nxt[cont], obs[t + 1][cont]), (it, t)
            assert (obs[t + 1][~cont] == start).all(), (it, t)
            checked += int(cont.sum())
        assert checked > 0.8 * E * (T - 1)
        # on-policy: the recorded actions were drawn from the network as it is NOW (before this iteration's update)
        with torch.no_grad():
            probs, _ = tr.models["shared"](b["obs"][:T])
        p1 = float(probs[0][..., 1].mean())
        f1 = float(act.mean())
        assert abs(p1 - f1) < 4.0 * 0.5 / np.sqrt(E * T) + 1e-3, (it, p1, f1)
        tr._update_model_params(it, False)
    tr.graceful_close()
