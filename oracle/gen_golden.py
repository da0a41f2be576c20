#!/usr/bin/env python3
"""Generate golden fixtures from the REAL reference (test infrastructure only).

Runs only in the build container, where /root/reference exists: it imports the
reference's own CPU environments (through the 4-class gym.spaces shim in
oracle/gym_shim) and records their inputs/outputs as small .npz fixtures under
tests/golden/.  /root/reference does not exist on the GPU box, so nothing at
test/bench time reads it -- the committed fixtures travel instead.

What is recorded
  * gw_kat.npz        -- the reference's own known-answer vectors for the
                          TagGridWorld CPU step, parsed (ast.literal_eval) out of
                          reference tests/example_envs/pycuda_tests/
                          test_tag_gridworld_step_python.py:32-463, together with
                          what the reference env actually returns for them here.
  * gw_traj_*.npz     -- TagGridWorld lock-step trajectories (full / partial obs),
                          reference tag_gridworld.py:291-317.
  * tc_traj_*.npz     -- TagContinuous trajectories for the four scenarios of
                          reference tests/example_envs/pycuda_tests/
                          test_tag_continuous.py:15-80 and for the 5x100 K=10
                          benchmark shape, reference tag_continuous.py:796-887
                          (bench5x100_ep: the same shape with 15-tick episodes, so
                          episode ends and restarts are recorded from the reference too;
                          big5x250: a 255-agent replica, the sizes the `_N512` entries serve).
  * loss_fixtures.npz -- the reference's A2C / PPO `compute_loss_and_metrics`
                          (training/algorithms/policygradient/a2c.py:40-194, ppo.py:42-228) on
                          seeded random batches: inputs, loss, every logged metric and the
                          gradients w.r.t. logits and values.
  * ref_checkpoint_io.npz -- the reference's shipped policies (tutorials/assets/
                          tag_continuous_training/{runner,tagger}_1000010000.state_dict) loaded into
                          the reference's own FullyConnected (models/fully_connected.py:19-93):
                          tensor names / shapes, file hashes, and the outputs for a seeded input.
  * cp_traj.npz       -- Cartpole: the reference's OWN device kernel source,
                          example_envs/single_agent/classic_control/cartpole/
                          cartpole_step_numba.py:5-83, executed here under a minimal
                          `numba.cuda` stand-in (jit = identity, blockIdx/threadIdx set by
                          the driver loop) on float32 numpy arrays.  gym (the reference's
                          CPU step) is absent, so this is the only reference-run pin the
                          path has; it runs with Python/numpy scalar semantics, i.e. cos/sin
                          in float64 rounded to float32 where Numba would call cosf/sinf.

Usage:  python oracle/gen_golden.py          (from the repo root)
"""
import ast
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "gym_shim"))
sys.path.insert(0, REF)

from example_envs.tag_continuous.tag_continuous import TagContinuous  # noqa: E402
from example_envs.tag_gridworld.tag_gridworld import TagGridWorld  # noqa: E402
from warp_drive.env_wrapper import EnvWrapper  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def _obs_to_array(obs, n_agents):
    return np.stack([np.asarray(obs[a], dtype=np.float64) for a in range(n_agents)])


def _rew_to_array(rew, n_agents):
    return np.array([float(rew[a]) for a in range(n_agents)], dtype=np.float64)


# --------------------------------------------------------------------------
# TagGridWorld known-answer vectors from the reference's own test file
# --------------------------------------------------------------------------
def _parse_gridworld_kat():
    path = os.path.join(
        REF, "tests/example_envs/pycuda_tests/test_tag_gridworld_step_python.py"
    )
    tree = ast.parse(open(path).read())
    cases = []
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name.startswith("test_step_case"):
            env_kwargs, steps = None, []
            cur = {}
            for sub in ast.walk(node):
                pass
            # walk statements in order
            for stmt in node.body:
                for sub in ast.walk(stmt):
                    if (
                        isinstance(sub, ast.Call)
                        and isinstance(sub.func, ast.Attribute)
                        and sub.func.attr == "TagGridWorld"
                    ):
                        env_kwargs = {}
                        for kw in sub.keywords:
                            v = kw.value
                            if isinstance(v, ast.Call):  # np.array([...])
                                env_kwargs[kw.arg] = ast.literal_eval(v.args[0])
                            else:
                                env_kwargs[kw.arg] = ast.literal_eval(v)
                if isinstance(stmt, ast.Assign) and isinstance(stmt.targets[0], ast.Name):
                    name = stmt.targets[0].id
                    if name == "actions":
                        cur = {"actions": ast.literal_eval(stmt.value)}
                    elif name == "ref_rewards":
                        cur["rewards"] = ast.literal_eval(stmt.value.args[0])
                    elif name == "ref_observations":
                        cur["obs_x_grid"] = ast.literal_eval(stmt.value.args[0])
                if isinstance(stmt, ast.Expr) and isinstance(stmt.value, ast.Call):
                    f = stmt.value.func
                    if isinstance(f, ast.Attribute) and f.attr == "assertEqual":
                        a0 = stmt.value.args[0]
                        if isinstance(a0, ast.Name) and a0.id == "done_update":
                            cur["done"] = ast.literal_eval(stmt.value.args[1])
                            steps.append(cur)
                            cur = {}
            cases.append((node.name, env_kwargs, steps))
    return cases


def gen_gridworld_kat():
    out = {}
    meta = []
    for ci, (name, kw, steps) in enumerate(_parse_gridworld_kat()):
        kwargs = dict(kw)
        for k in ("starting_location_x", "starting_location_y"):
            kwargs[k] = np.array(kwargs[k])
        env = TagGridWorld(**kwargs)
        env.reset()
        n = env.num_agents
        meta.append(
            {
                "name": name,
                "n_steps": len(steps),
                "kwargs": {k: (v if not isinstance(v, list) else v) for k, v in kw.items()},
            }
        )
        for si, st in enumerate(steps):
            obs, rew, done, _ = env.step(dict(st["actions"]))
            p = f"c{ci}_s{si}_"
            out[p + "actions"] = np.array([st["actions"][a] for a in range(n)], dtype=np.int32)
            # expected values as written in the reference test (obs is scaled by
            # grid_length=4 there: test_tag_gridworld_step_python.py:169-172)
            out[p + "kat_rewards"] = np.array(st["rewards"], dtype=np.float64)
            out[p + "kat_obs_x_grid"] = np.array(st["obs_x_grid"], dtype=np.float64)
            out[p + "kat_done"] = np.array(bool(st["done"]))
            # what the reference returns here
            out[p + "ref_rewards"] = _rew_to_array(rew, n)
            out[p + "ref_obs"] = _obs_to_array(obs, n)
            out[p + "ref_done"] = np.array(bool(done["__all__"]))
            out[p + "ref_loc_x"] = env.global_state["loc_x"][env.timestep].copy()
            out[p + "ref_loc_y"] = env.global_state["loc_y"][env.timestep].copy()
            # the reference's own assertion, re-checked at generation time
            assert np.abs(out[p + "ref_rewards"] - out[p + "kat_rewards"]).max() < 1e-5
            g = kw["grid_length"]
            assert np.abs(out[p + "ref_obs"] * g - out[p + "kat_obs_x_grid"]).max() < 1e-5
            assert bool(done["__all__"]) == bool(st["done"])
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "gw_kat.npz"), **out)
    print("gw_kat.npz:", len(meta), "cases")


# --------------------------------------------------------------------------
# Lock-step trajectories
# --------------------------------------------------------------------------
def _jsonable(cfg):
    return {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in cfg.items()}


def gen_gridworld_traj(tag, cfg, num_envs, num_ticks, action_seed):
    envs = [EnvWrapper(env_obj=TagGridWorld(**cfg), env_backend="cpu") for _ in range(num_envs)]
    n = envs[0].env.num_agents
    obs0 = np.stack([_obs_to_array(e.reset(), n) for e in envs])
    rngs = [np.random.RandomState(action_seed + i) for i in range(num_envs)]
    rec = {k: [] for k in ("actions", "loc_x", "loc_y", "rewards", "obs", "done", "timestep")}
    for _ in range(num_ticks):
        acts = np.stack([r.randint(0, 5, size=n).astype(np.int32) for r in rngs])
        lx, ly, rw, ob, dn, ts = [], [], [], [], [], []
        for i, e in enumerate(envs):
            obs, rew, done, _ = e.step({a: int(acts[i, a]) for a in range(n)})
            t = e.env.timestep
            lx.append(e.env.global_state["loc_x"][t].copy())
            ly.append(e.env.global_state["loc_y"][t].copy())
            rw.append(_rew_to_array(rew, n))
            ob.append(_obs_to_array(obs, n))
            dn.append(bool(done["__all__"]))
            ts.append(t)
            if done["__all__"]:
                e.reset()
        rec["actions"].append(acts)
        rec["loc_x"].append(np.stack(lx))
        rec["loc_y"].append(np.stack(ly))
        rec["rewards"].append(np.stack(rw))
        rec["obs"].append(np.stack(ob))
        rec["done"].append(np.array(dn))
        rec["timestep"].append(np.array(ts, dtype=np.int32))
    out = {k: np.stack(v) for k, v in rec.items()}
    out["obs_at_reset"] = obs0
    out["start_x"] = np.asarray(envs[0].env.starting_location_x)
    out["start_y"] = np.asarray(envs[0].env.starting_location_y)
    out["config"] = np.array(json.dumps(_jsonable(cfg)))
    np.savez_compressed(os.path.join(OUT, f"gw_traj_{tag}.npz"), **out)
    print(f"gw_traj_{tag}.npz: E={num_envs} ticks={num_ticks} dones={int(out['done'].sum())}")


_TC_STATE = ("loc_x", "loc_y", "speed", "direction", "acceleration")


def gen_tag_continuous_traj(tag, cfg, num_envs, num_ticks, action_seed, obs_dtype=None):
    envs = [EnvWrapper(env_obj=TagContinuous(**cfg), env_backend="cpu") for _ in range(num_envs)]
    e0 = envs[0].env
    n = e0.num_agents
    obs0 = np.stack([_obs_to_array(e.reset(), n) for e in envs])
    rngs = [np.random.RandomState(action_seed + i) for i in range(num_envs)]
    na = len(e0.acceleration_actions)
    nt = len(e0.turn_actions)
    keys = _TC_STATE + (
        "still_in_the_game", "edge_hit_reward_penalty", "actions", "rewards", "obs",
        "done", "timestep", "num_runners",
    )
    rec = {k: [] for k in keys}
    for _ in range(num_ticks):
        acts = np.stack(
            [
                np.stack([r.randint(0, na, size=n), r.randint(0, nt, size=n)], axis=1).astype(np.int32)
                for r in rngs
            ]
        )
        cur = {k: [] for k in keys if k != "actions"}
        for i, e in enumerate(envs):
            obs, rew, done, _ = e.step({a: acts[i, a] for a in range(n)})
            t = e.env.timestep
            for k in _TC_STATE:
                cur[k].append(e.env.global_state[k][t].copy())
            cur["still_in_the_game"].append(e.env.still_in_the_game.copy())
            cur["edge_hit_reward_penalty"].append(
                np.asarray(e.env.edge_hit_reward_penalty, dtype=np.float32).copy()
            )
            cur["rewards"].append(_rew_to_array(rew, n))
            cur["obs"].append(_obs_to_array(obs, n))
            cur["done"].append(bool(done["__all__"]))
            cur["timestep"].append(t)
            cur["num_runners"].append(e.env.num_runners)
            if done["__all__"]:
                e.reset()
        rec["actions"].append(acts)
        for k, v in cur.items():
            rec[k].append(np.stack([np.asarray(x) for x in v]))
    out = {k: np.stack(v) for k, v in rec.items()}
    out["obs"] = out["obs"]  # float64 as returned by the reference (numpy 2 promotion)
    out["obs_at_reset"] = obs0
    if obs_dtype is not None:
        out["obs"], out["obs_at_reset"] = out["obs"].astype(obs_dtype), obs0.astype(obs_dtype)
    out["agent_types"] = np.array([e0.agent_type[a] for a in range(n)], dtype=np.int32)
    out["start_x"] = np.asarray(e0.starting_location_x, dtype=np.float64)
    out["start_y"] = np.asarray(e0.starting_location_y, dtype=np.float64)
    out["start_dir"] = np.asarray(e0.starting_directions, dtype=np.float64)
    out["skill_levels"] = np.asarray(e0.skill_levels, dtype=np.float32)
    out["step_rewards"] = np.asarray(e0.step_rewards, dtype=np.float32)
    out["acceleration_actions"] = np.asarray(e0.acceleration_actions, dtype=np.float32)
    out["turn_actions"] = np.asarray(e0.turn_actions, dtype=np.float32)
    out["distance_margin_for_reward"] = np.float32(e0.distance_margin_for_reward)
    out["config"] = np.array(json.dumps(_jsonable(cfg)))
    np.savez_compressed(os.path.join(OUT, f"tc_traj_{tag}.npz"), **out)
    print(
        f"tc_traj_{tag}.npz: E={num_envs} N={n} ticks={num_ticks} dones={int(out['done'].sum())} "
        f"tagged_out={int((out['still_in_the_game'] == 0).sum())}"
    )


# --------------------------------------------------------------------------
# Cartpole: run the reference's Numba kernel SOURCE under a numba.cuda stand-in
# --------------------------------------------------------------------------
def gen_cartpole_traj(num_envs=48, num_ticks=130, episode_length=45, action_seed=5000):
    import importlib.util
    import types

    class _Idx:
        x = 0

    stub_cuda = types.ModuleType("numba.cuda")
    stub_cuda.jit = lambda f=None, **kw: f if f is not None else (lambda g: g)
    stub_cuda.blockIdx, stub_cuda.threadIdx = _Idx(), _Idx()
    stub = types.ModuleType("numba")
    stub.cuda = stub_cuda
    saved = {k: sys.modules.get(k) for k in ("numba", "numba.cuda")}
    sys.modules["numba"], sys.modules["numba.cuda"] = stub, stub_cuda
    try:
        path = os.path.join(REF, "example_envs/single_agent/classic_control/cartpole/cartpole_step_numba.py")
        spec = importlib.util.spec_from_file_location("ref_cartpole_step_numba", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    step = mod.NumbaClassicControlCartPoleEnvStep
    f32 = np.float32
    # gym's CartPoleEnv constants (gym is absent: stated here as this repo's configuration; the
    # reference reads them off the gym object, cartpole.py:71-79) -- narrowed to float32 exactly as
    # the reference's data manager narrows every scalar (data_manager.py:348-351)
    masscart, masspole, length = 1.0, 0.1, 0.5
    consts = dict(gravity=f32(9.8), masspole=f32(masspole), total_mass=f32(masspole + masscart), length=f32(length),
                  polemass_length=f32(masspole * length), force_mag=f32(10.0), tau=f32(0.02),
                  theta_threshold_radians=f32(12 * 2 * np.pi / 360), x_threshold=f32(2.4))
    E, T = num_envs, episode_length
    initial = np.array([0.013, -0.021, 0.034, 0.027], dtype=f32)
    state = np.tile(initial, (E, 1, 1)).astype(f32)          # [E, 1, 4]
    action = np.zeros((E, 1, 1), np.int32)
    done = np.zeros(E, np.int32)
    reward = np.zeros((E, 1), f32)
    obs = np.zeros((E, 1, 4), f32)
    timestep = np.zeros(E, np.int32)
    rng = np.random.RandomState(action_seed)
    rec = {k: [] for k in ("actions", "state", "obs", "rewards", "done", "timestep")}
    for _ in range(num_ticks):
        action[:] = rng.randint(0, 2, size=(E, 1, 1))
        for e in range(E):                                   # grid = (E,), block = (1,)
            stub_cuda.blockIdx.x, stub_cuda.threadIdx.x = e, 0
            step(state, action, done, reward, obs, consts["gravity"], consts["masspole"], consts["total_mass"],
                 consts["length"], consts["polemass_length"], consts["force_mag"], consts["tau"],
                 consts["theta_threshold_radians"], consts["x_threshold"], timestep, T)
        for k, v in (("actions", action), ("state", state), ("obs", obs), ("rewards", reward), ("done", done),
                     ("timestep", timestep)):
            rec[k].append(v.copy())
        # reset_when_done (reset.cu:9-75 semantics): finished replicas restart from the initial state
        m = done > 0
        state[m] = initial
        obs[m] = initial
        timestep[m] = 0
        done[m] = 0
    out = {k: np.stack(v) for k, v in rec.items()}
    out["initial_state"] = initial
    out["episode_length"] = np.int32(T)
    out["constants"] = np.array(json.dumps({k: float(v) for k, v in consts.items()}))
    np.savez_compressed(os.path.join(OUT, "cp_traj.npz"), **out)
    print(f"cp_traj.npz: E={E} ticks={num_ticks} dones={int(out['done'].sum())} "
          f"(timeouts {int((out['timestep'] == T).sum())})")


# --------------------------------------------------------------------------
# A2C / PPO objectives: the reference's compute_loss_and_metrics on random batches
# --------------------------------------------------------------------------
def gen_loss_fixtures():
    import torch
    from warp_drive.training.algorithms.policygradient.a2c import A2C
    from warp_drive.training.algorithms.policygradient.ppo import PPO

    T, E, n, heads = 12, 6, 3, (5, 7)
    cases = {
        "a2c_plain": ("A2C", dict(discount_factor_gamma=0.98, vf_loss_coeff=0.01, entropy_coeff=0.05), 0, -1),
        "a2c_norm_sched": ("A2C", dict(discount_factor_gamma=0.95, normalize_advantage=True, normalize_return=True,
                                       vf_loss_coeff=[[0, 1.0], [1000, 0.1]],
                                       entropy_coeff=[[0, 0.5], [500, 0.05], [2000, 0.01]]), 750, -1),
        "ppo_plain": ("PPO", dict(discount_factor_gamma=1.0, clip_param=0.1, vf_loss_coeff=1.0, entropy_coeff=0.0), 10, -1),
        "ppo_norm": ("PPO", dict(discount_factor_gamma=0.9, clip_param=0.3, normalize_advantage=True,
                                 normalize_return=True, vf_loss_coeff=0.5, entropy_coeff=[[0, 0.1], [100, 0.0]]), 40, -1),
        "a2c_posneg": ("A2C", dict(discount_factor_gamma=0.99, vf_loss_coeff=0.2, entropy_coeff=0.01), 5, 1),
    }
    out = {}
    meta = {}
    for ci, (name, (algo, kw, timestep, ratio)) in enumerate(cases.items()):
        g = torch.Generator().manual_seed(9000 + ci)
        logits = [torch.randn(T, E, n, a, generator=g, dtype=torch.float32) for a in heads]
        values = torch.randn(T, E, n, generator=g, dtype=torch.float32)
        actions = torch.stack([torch.randint(0, a, (T, E, n), generator=g) for a in heads], dim=-1)
        rewards = torch.randn(T, E, n, generator=g, dtype=torch.float32) * 2.0
        done = (torch.rand(T, E, generator=g) < 0.15).to(torch.int32)
        done[-1, ::2] = 1                       # some replicas finish on the last row, some bootstrap
        if ratio > 0:
            done[3, 1] = 2                      # one replica "reached the goal"
            done[7, 4] = 2
        for x in logits:
            x.requires_grad_(True)
        values.requires_grad_(True)
        probs = [torch.softmax(x, dim=-1) for x in logits]
        trainer = (A2C if algo == "A2C" else PPO)(**kw)
        np.random.seed(1234 + ci)               # the positive/negative down-sampling draws from np.random
        loss, metrics = trainer.compute_loss_and_metrics(
            timestep=timestep, actions_batch=actions, rewards_batch=rewards, done_flags_batch=done,
            action_probabilities_batch=probs, value_functions_batch=values, perform_logging=True,
            negative_positive_ratio=ratio)
        loss.backward()
        for h, x in enumerate(logits):
            out[f"{name}.logits{h}"] = x.detach().numpy()
            out[f"{name}.grad_logits{h}"] = x.grad.numpy()
        out[f"{name}.values"] = values.detach().numpy()
        out[f"{name}.grad_values"] = values.grad.numpy()
        out[f"{name}.actions"] = actions.numpy()
        out[f"{name}.rewards"] = rewards.numpy()
        out[f"{name}.done"] = done.numpy()
        out[f"{name}.loss"] = np.float64(loss.item())
        meta[name] = {"algo": algo, "kwargs": kw, "timestep": timestep, "negative_positive_ratio": ratio,
                      "np_seed": 1234 + ci, "metrics": {k: float(v) for k, v in metrics.items()}}
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "loss_fixtures.npz"), **out)
    print(f"loss_fixtures.npz: {len(cases)} cases from the reference's A2C / PPO compute_loss_and_metrics")


# --------------------------------------------------------------------------
# Checkpoint wire format: the reference's shipped TagContinuous policies through ITS model class
# --------------------------------------------------------------------------
def gen_checkpoint_fixture():
    import hashlib
    import types

    import torch
    import yaml
    from warp_drive.training.models.fully_connected import FullyConnected

    assets = os.path.join(REF, "tutorials/assets/tag_continuous_training")
    cfg = yaml.safe_load(open(os.path.join(assets, "run_config.yaml")))
    env = TagContinuous(**cfg["env"])
    wrapper = EnvWrapper(env_obj=env, env_backend="cpu")        # fills env.observation_space
    policy_map = {"tagger": sorted(env.taggers), "runner": sorted(env.runners)}
    # the model constructor only asks the data manager for the batch placeholder's leading dimension
    wrapper.cuda_data_manager = types.SimpleNamespace(get_shape=lambda name: (1,))
    out, manifest = {}, {}
    g = torch.Generator().manual_seed(77)
    obs = torch.randn(3, 6, 71, generator=g, dtype=torch.float32)
    out["obs"] = obs.numpy()
    for pol in ("runner", "tagger"):
        path = os.path.join(assets, f"{pol}_1000010000.state_dict")
        sd = torch.load(path, map_location="cpu")
        model = FullyConnected(wrapper, cfg["policy"][pol]["model"], pol, policy_map)
        model.load_state_dict(sd)                                # strict: the reference's own format
        model.eval()
        with torch.no_grad():
            probs, vals = model(obs)
        for h, p in enumerate(probs):
            out[f"{pol}.probs{h}"] = p.numpy()
        out[f"{pol}.values"] = vals.numpy()
        manifest[pol] = {"file": f"tutorials/assets/tag_continuous_training/{pol}_1000010000.state_dict",
                         "sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(),
                         "tensors": {k: list(v.shape) for k, v in sd.items()},
                         "fc_dims": cfg["policy"][pol]["model"]["fc_dims"],
                         "head_sizes": [int(p.shape[-1]) for p in probs], "obs_size": 71}
    out["manifest"] = np.array(json.dumps(manifest))
    np.savez_compressed(os.path.join(OUT, "ref_checkpoint_io.npz"), **out)
    print("ref_checkpoint_io.npz: runner/tagger_1000010000.state_dict through the reference's FullyConnected")


def main():
    gen_gridworld_kat()
    gen_cartpole_traj()
    gen_loss_fixtures()
    gen_checkpoint_fixture()

    gw = dict(num_taggers=4, grid_length=4, episode_length=20, seed=27, wall_hit_penalty=0.1,
              tag_reward_for_tagger=10.0, tag_penalty_for_runner=2.0, step_cost_for_tagger=0.01)
    # reference tests/example_envs/pycuda_tests/test_tag_gridworld.py:13-38 (2 envs, 2 episodes)
    gen_gridworld_traj("full", dict(gw, use_full_observation=True), 2, 40, 100)
    gen_gridworld_traj("partial", dict(gw, use_full_observation=False), 2, 40, 200)
    # BASELINE config[0]/[1] shapes (6x6 plumbing, 10x10)
    gen_gridworld_traj("g6", dict(gw, grid_length=6, use_full_observation=True), 2, 45, 300)
    gen_gridworld_traj("g10", dict(gw, grid_length=10, episode_length=100,
                                   use_full_observation=True), 8, 210, 400)

    # reference tests/example_envs/pycuda_tests/test_tag_continuous.py:15-80
    tc = {
        "test1": dict(num_taggers=2, num_runners=3, max_acceleration=1, max_turn=np.pi / 4,
                      num_acceleration_levels=3, num_turn_levels=3, grid_length=10,
                      episode_length=100, seed=274880, skill_level_runner=1, skill_level_tagger=1,
                      use_full_observation=True, runner_exits_game_after_tagged=True,
                      tagging_distance=0.0),
        "test2": dict(num_taggers=4, num_runners=1, max_acceleration=0.05, max_turn=np.pi / 4,
                      num_acceleration_levels=3, num_turn_levels=3, grid_length=10,
                      episode_length=100, step_penalty_for_tagger=-0.1, seed=428096,
                      skill_level_runner=1, skill_level_tagger=2, use_full_observation=False,
                      runner_exits_game_after_tagged=False, tagging_distance=0.25),
        "test3": dict(num_taggers=1, num_runners=4, max_acceleration=2, max_turn=np.pi / 2,
                      num_acceleration_levels=3, num_turn_levels=3, grid_length=10,
                      episode_length=100, step_reward_for_runner=0.1, seed=654208,
                      skill_level_runner=1, skill_level_tagger=0.5, use_full_observation=False,
                      runner_exits_game_after_tagged=True),
        "test4": dict(num_taggers=3, num_runners=2, max_acceleration=0.05, max_turn=np.pi,
                      num_acceleration_levels=3, num_turn_levels=3, grid_length=10,
                      episode_length=100, seed=121024, skill_level_runner=0.5, skill_level_tagger=1,
                      use_full_observation=True, runner_exits_game_after_tagged=False),
    }
    for i, (name, cfg) in enumerate(tc.items()):
        gen_tag_continuous_traj(name, cfg, 2, 200, 1000 + 10 * i)

    # a tagging-heavy small case (large tagging distance, exits) so that tag /
    # exit / num_runners==0 / end-of-game paths are all exercised
    gen_tag_continuous_traj(
        "tagheavy",
        dict(num_taggers=3, num_runners=6, grid_length=6.0, episode_length=40, seed=7,
             max_acceleration=0.5, min_acceleration=-0.5, max_turn=np.pi / 2, min_turn=-np.pi / 2,
             num_acceleration_levels=5, num_turn_levels=5, edge_hit_penalty=-0.5,
             use_full_observation=False, num_other_agents_observed=4, tagging_distance=0.15,
             tag_reward_for_tagger=10.0, tag_penalty_for_runner=-10.0, step_penalty_for_tagger=-0.01,
             step_reward_for_runner=0.02, end_of_game_reward_for_runner=1.0,
             runner_exits_game_after_tagged=True),
        4, 120, 2000,
    )
    # BASELINE config[2] shape: 5 taggers x 100 runners, K=10 (run_configs/tag_continuous.yaml:11-34)
    bench_cfg = dict(num_taggers=5, num_runners=100, grid_length=20.0, episode_length=500,
                     max_acceleration=0.1, min_acceleration=-0.1, max_turn=2.356, min_turn=-2.356,
                     num_acceleration_levels=20, num_turn_levels=20, skill_level_runner=1.0,
                     skill_level_tagger=1.0, max_speed=1.0, seed=274880,
                     use_full_observation=False, num_other_agents_observed=10,
                     tagging_distance=0.02, tag_reward_for_tagger=10.0,
                     tag_penalty_for_runner=-10.0, step_penalty_for_tagger=-0.0,
                     step_reward_for_runner=0.0, edge_hit_penalty=-0.0,
                     end_of_game_reward_for_runner=1.0, runner_exits_game_after_tagged=True)
    gen_tag_continuous_traj("bench5x100", bench_cfg, 2, 12, 3000)
    gen_tag_continuous_traj("bench5x100_full", dict(bench_cfg, use_full_observation=True), 1, 3, 3100)
    # the bench shape through episode ends and restarts (15-tick episodes, 3 episodes)
    gen_tag_continuous_traj("bench5x100_ep", dict(bench_cfg, episode_length=15), 2, 47, 3200)
    gen_big_replica()


def gen_big_replica():
    """a replica of MORE THAN 128 AGENTS from the reference itself (255 agents: three wavefronts of searchers, 9 id
    bits in the search keys, the prefiltered neighbour search), through an episode end and restart (8-tick episodes)"""
    bench_cfg = dict(num_taggers=5, num_runners=250, grid_length=20.0, episode_length=8,
                     max_acceleration=0.1, min_acceleration=-0.1, max_turn=2.356, min_turn=-2.356,
                     num_acceleration_levels=20, num_turn_levels=20, skill_level_runner=1.0,
                     skill_level_tagger=1.0, max_speed=1.0, seed=274880,
                     use_full_observation=False, num_other_agents_observed=10,
                     tagging_distance=0.02, tag_reward_for_tagger=10.0,
                     tag_penalty_for_runner=-10.0, step_penalty_for_tagger=-0.0,
                     step_reward_for_runner=0.0, edge_hit_penalty=-0.0,
                     end_of_game_reward_for_runner=1.0, runner_exits_game_after_tagged=True)
    gen_tag_continuous_traj("big5x250", bench_cfg, 1, 13, 3300)
    # ... and of more than 512 (1005 agents: sixteen wavefronts, 10 id bits: the `_N1024` entries); the observations are
    # stored as float32 (the cast every comparison applies to the reference's float64 anyway) to keep the file small
    gen_tag_continuous_traj("big5x1000", dict(bench_cfg, num_runners=1000, episode_length=3), 1, 5, 3400, obs_dtype=np.float32)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":  # only the fixture added in round 4
        gen_big_replica()
    else:
        main()
