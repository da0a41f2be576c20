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
                          benchmark shape, reference tag_continuous.py:796-887.

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


def gen_tag_continuous_traj(tag, cfg, num_envs, num_ticks, action_seed):
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


def main():
    gen_gridworld_kat()

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


if __name__ == "__main__":
    main()
