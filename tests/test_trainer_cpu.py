"""Host-side pieces of the A2C/PPO trainer (no GPU): objectives, schedules, model, config merge,
and the 2-rank gradient all-reduce path (DistributedDataParallel over gloo)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from warp_drive_amd.training.losses import A2C, PPO, discounted_returns
from warp_drive_amd.training.models import FullyConnected
from warp_drive_amd.training.param_scheduler import ParamScheduler


def test_discounted_returns_stop_at_done():
    T, E, n, g = 6, 3, 2, 0.9
    rng = np.random.RandomState(0)
    r = torch.tensor(rng.randn(T, E, n), dtype=torch.float32)
    v = torch.tensor(rng.randn(T, E, n), dtype=torch.float32)
    done = torch.zeros(T, E, dtype=torch.int32)
    done[2, 1] = 1
    done[5, 2] = 1
    got = discounted_returns(r, done, v, g).numpy()
    want = np.zeros((T, E, n), np.float32)
    for e in range(E):
        for a in range(n):
            nxt = float(r[-1, e, a]) if done[-1, e] else float(v[-1, e, a])
            want[-1, e, a] = nxt
            for t in range(T - 2, -1, -1):
                nxt = float(r[t, e, a]) + (0.0 if done[t, e] else g * nxt)
                want[t, e, a] = nxt
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_objectives_and_model():
    torch.manual_seed(0)
    T, E, n, F = 5, 4, 3, 11
    model = FullyConnected(F, [4, 6], [16, 16])
    obs = torch.randn(T, E, n, F)
    probs, vals = model(obs)
    assert [tuple(p.shape) for p in probs] == [(T, E, n, 4), (T, E, n, 6)] and tuple(vals.shape) == (T, E, n)
    for p in probs:
        np.testing.assert_allclose(p.sum(-1).detach().numpy(), 1.0, rtol=1e-5)
    actions = torch.stack([torch.randint(0, 4, (T, E, n)), torch.randint(0, 6, (T, E, n))], dim=-1)
    rewards = torch.randn(T, E, n)
    done = torch.zeros(T, E, dtype=torch.int32)
    done[-1] = 1
    for algo in (A2C(discount_factor_gamma=0.98, vf_loss_coeff=1.0, entropy_coeff=[[0, 0.5], [100, 0.05]]),
                 PPO(clip_param=0.1, discount_factor_gamma=0.98, normalize_advantage=True, normalize_return=True)):
        loss, metrics = algo.compute_loss_and_metrics(timestep=50, actions_batch=actions, rewards_batch=rewards,
                                                      done_flags_batch=done, action_probabilities_batch=probs,
                                                      value_functions_batch=vals, perform_logging=True)
        assert torch.isfinite(loss) and "Total loss" in metrics and "Mean entropy" in metrics
        model.zero_grad()
        loss.backward(retain_graph=True)
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    sched = ParamScheduler([[0, 0.5], [100, 0.05]])
    assert abs(sched.get_param_value(50) - 0.275) < 1e-9 and sched.get_param_value(1e9) == 0.05
    assert ParamScheduler(0.3).get_param_value(7) == 0.3


def test_inference_forward_equals_training_forward():
    """the rollout's forward (fused epilogues, one GEMM for all heads) computes what forward() computes"""
    torch.manual_seed(1)
    model = FullyConnected(13, [4, 6, 3], [32, 16])
    obs = torch.randn(5, 7, 13)
    probs, vals = model(obs)
    probs_i, vals_i = model.forward_inference(obs)
    assert [tuple(p.shape) for p in probs_i] == [(5, 7, 4), (5, 7, 6), (5, 7, 3)] and tuple(vals_i.shape) == (5, 7)
    for a, b in zip(probs, probs_i):
        np.testing.assert_allclose(a.detach().numpy(), b.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(vals.detach().numpy(), vals_i.numpy(), rtol=1e-5, atol=1e-6)
    assert not any(p.requires_grad for p in probs_i)


def test_config_merge_and_yaml_files():
    import yaml

    from warp_drive_amd.training.trainer import recursive_merge_config_dicts

    base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "warp_drive_amd", "training",
                        "run_configs")
    default = yaml.safe_load(open(os.path.join(base, "default_configs.yaml")))
    for name in ("tag_continuous", "tag_gridworld", "single_cartpole"):
        cfg = yaml.safe_load(open(os.path.join(base, f"{name}.yaml")))
        for pol in cfg["policy"]:
            merged = recursive_merge_config_dicts(cfg["policy"][pol], default["policy"])
            assert merged["max_grad_norm"] == 0.5 and "fc_dims" in merged["model"]
        assert cfg["trainer"]["train_batch_size"] % cfg["trainer"]["num_envs"] == 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from warp_drive_amd import distributed as wdd

    wdd.init_process_group(backend="gloo")
    torch.manual_seed(0)  # identical initial weights, as DDP broadcasts rank 0's anyway
    model = torch.nn.parallel.DistributedDataParallel(FullyConnected(7, [3], [8]))
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    algo = A2C(discount_factor_gamma=0.9, vf_loss_coeff=1.0, entropy_coeff=0.01)
    g = torch.Generator().manual_seed(100 + rank)  # every rank rolls out its OWN replicas
    obs = torch.randn(4, 5, 2, 7, generator=g)
    actions = torch.randint(0, 3, (4, 5, 2, 1), generator=g)
    rewards = torch.randn(4, 5, 2, generator=g)
    done = torch.zeros(4, 5, dtype=torch.int32)
    probs, vals = model(obs)
    loss, _ = algo.compute_loss_and_metrics(timestep=0, actions_batch=actions, rewards_batch=rewards,
                                            done_flags_batch=done, action_probabilities_batch=probs,
                                            value_functions_batch=vals)
    opt.zero_grad()
    loss.backward()  # gradient all-reduce (average) across the 2 ranks
    opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    np.save(os.path.join(out_dir, f"w_{rank}.npy"), flat.numpy())
    wdd.shutdown()


def test_ddp_gradient_allreduce_two_ranks(tmp_path):
    mp.spawn(_ddp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    w0, w1 = np.load(tmp_path / "w_0.npy"), np.load(tmp_path / "w_1.npy")
    np.testing.assert_array_equal(w0, w1)  # different data, identical parameters after the step


def _bucket_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from warp_drive_amd import distributed as wdd
    from warp_drive_amd.training.grad_bucket import GradientBucket

    wdd.init_process_group(backend="gloo")
    torch.manual_seed(10 + rank)  # DIFFERENT initial weights per rank: the bucket broadcasts rank 0's
    models = [FullyConnected(7, [3, 4], [8]), FullyConnected(5, [2], [6, 6])]  # two policies, one bucket
    bucket = GradientBucket(models)
    bucket.broadcast_parameters(src=0)
    opts = [torch.optim.SGD(m.parameters(), lr=0.1) for m in models]
    algo = A2C(discount_factor_gamma=0.9, vf_loss_coeff=1.0, entropy_coeff=0.01)
    for it in range(2):  # two iterations: the views must survive an optimizer step
        bucket.zero()
        for k, (m, obs_dim, heads) in enumerate(zip(models, (7, 5), ((3, 4), (2,)))):
            g = torch.Generator().manual_seed(1000 * it + 100 * k + rank)  # every rank rolls out its OWN replicas
            obs = torch.randn(4, 5, 2, obs_dim, generator=g)
            actions = torch.stack([torch.randint(0, a, (4, 5, 2), generator=g) for a in heads], dim=-1)
            rewards = torch.randn(4, 5, 2, generator=g)
            probs, vals = m(obs)
            loss, _ = algo.compute_loss_and_metrics(timestep=0, actions_batch=actions, rewards_batch=rewards,
                                                    done_flags_batch=torch.zeros(4, 5, dtype=torch.int32),
                                                    action_probabilities_batch=probs, value_functions_batch=vals)
            loss.backward()
        local = bucket.flat.clone()
        bucket.all_reduce_mean()
        assert bucket.attached()
        np.save(os.path.join(out_dir, f"local_{it}_{rank}.npy"), local.numpy())
        np.save(os.path.join(out_dir, f"avg_{it}_{rank}.npy"), bucket.flat.numpy())
        for o in opts:
            o.step()
    flat = torch.cat([p.detach().reshape(-1) for m in models for p in m.parameters()])
    np.save(os.path.join(out_dir, f"w_{rank}.npy"), flat.numpy())
    open(os.path.join(out_dir, f"n_{rank}.txt"), "w").write(str(bucket.collectives))
    wdd.shutdown()


def test_gradient_bucket_one_allreduce_for_all_policies(tmp_path):
    """both policies' gradients are ONE flat buffer and ONE all-reduce per iteration (the
    reference wraps each policy in its own DDP, trainer_a2c.py:137-146)."""
    mp.spawn(_bucket_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "n_0.txt").read() == "2" and open(tmp_path / "n_1.txt").read() == "2"  # 2 iterations
    for it in range(2):
        l0, l1 = np.load(tmp_path / f"local_{it}_0.npy"), np.load(tmp_path / f"local_{it}_1.npy")
        assert np.abs(l0 - l1).max() > 1e-4  # different replicas, different local gradients ...
        for r in range(2):                    # ... the same average on both ranks
            np.testing.assert_allclose(np.load(tmp_path / f"avg_{it}_{r}.npy"), 0.5 * (l0 + l1), rtol=0, atol=1e-6)
    np.testing.assert_array_equal(np.load(tmp_path / "w_0.npy"), np.load(tmp_path / "w_1.npy"))


def test_launcher_command_line(capsys):
    """f2: the launcher starts one rank per GPU through torch.distributed.run on localhost"""
    from warp_drive_amd.training.scripts import launch

    cmd = launch.build_command("tag_continuous", 8, 29512, ["--iters", "3"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-5:] == ["warp_drive_amd.training.scripts.train", "--env", "tag_continuous", "--iters", "3"]
    assert launch.child_environment({})["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert launch.main(["--env", "tag_gridworld", "--num_gpus", "2", "--dry_run", "--iters", "1"]) == 0
    out = capsys.readouterr().out
    assert "--nproc-per-node=2" in out and out.strip().endswith("--env tag_gridworld --iters 1")
    port = launch.free_port()
    assert 1024 < port < 65536


# ----------------------------------------------------------------- parity with the reference (f1 / f4)
_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_objectives_match_the_reference_fixtures():
    """tests/golden/loss_fixtures.npz = the reference's A2C / PPO compute_loss_and_metrics
    (a2c.py:40-194, ppo.py:42-228) on seeded random batches incl. done flags 0/1/2, return / advantage
    normalisation, piecewise-linear coefficient schedules and the positive/negative replica
    down-sampling (oracle/gen_golden.py::gen_loss_fixtures).  Loss, every logged metric and the
    gradients w.r.t. logits and values must agree to 1e-6."""
    import json

    g = np.load(os.path.join(_GOLDEN, "loss_fixtures.npz"))
    meta = json.loads(str(g["meta"]))
    assert set(meta) == {"a2c_plain", "a2c_norm_sched", "ppo_plain", "ppo_norm", "a2c_posneg"}
    for name, m in meta.items():
        logits = [torch.tensor(g[f"{name}.logits{h}"], requires_grad=True) for h in range(2)]
        values = torch.tensor(g[f"{name}.values"], requires_grad=True)
        algo = (A2C if m["algo"] == "A2C" else PPO)(**m["kwargs"])
        np.random.seed(m["np_seed"])
        loss, metrics = algo.compute_loss_and_metrics(
            timestep=m["timestep"], actions_batch=torch.tensor(g[f"{name}.actions"]),
            rewards_batch=torch.tensor(g[f"{name}.rewards"]), done_flags_batch=torch.tensor(g[f"{name}.done"]),
            action_probabilities_batch=[torch.softmax(x, dim=-1) for x in logits], value_functions_batch=values,
            perform_logging=True, negative_positive_ratio=m["negative_positive_ratio"])
        assert abs(loss.item() - float(g[f"{name}.loss"])) <= 1e-6 * max(1.0, abs(float(g[f"{name}.loss"]))), name
        assert set(metrics) == set(m["metrics"]), (name, set(metrics) ^ set(m["metrics"]))
        for k, want in m["metrics"].items():
            assert abs(float(metrics[k]) - want) <= 1e-6 * max(1.0, abs(want)), (name, k, metrics[k], want)
        loss.backward()
        for h in range(2):
            np.testing.assert_allclose(logits[h].grad.numpy(), g[f"{name}.grad_logits{h}"], rtol=1e-5, atol=1e-7,
                                       err_msg=f"{name} head {h}")
        np.testing.assert_allclose(values.grad.numpy(), g[f"{name}.grad_values"], rtol=1e-5, atol=1e-7, err_msg=name)


def test_reference_checkpoint_wire_format():
    """f4: `{policy}_{timestep}.state_dict` files written by the reference load into this repo's
    FullyConnected unchanged.  tests/golden/ref_checkpoint_io.npz holds the tensor names / shapes of the
    reference's shipped TagContinuous policies and what the REFERENCE's model class computes with them
    for a seeded input (oracle/gen_golden.py::gen_checkpoint_fixture).  The name / shape contract is
    checked everywhere; the numerical replay needs the checkpoint files, i.e. /root/reference."""
    import hashlib
    import json

    g = np.load(os.path.join(_GOLDEN, "ref_checkpoint_io.npz"))
    manifest = json.loads(str(g["manifest"]))
    obs = torch.tensor(g["obs"])
    for pol, m in manifest.items():
        model = FullyConnected(m["obs_size"], m["head_sizes"], m["fc_dims"])
        ours = {k: list(v.shape) for k, v in model.state_dict().items()}
        assert ours == m["tensors"], f"{pol}: state_dict layout differs from the reference checkpoint"
        path = os.path.join("/root/reference", m["file"])
        if not os.path.isfile(path):
            continue  # (GPU box: no reference tree; the layout check above still ran)
        assert hashlib.sha256(open(path, "rb").read()).hexdigest() == m["sha256"]
        model.load_state_dict(torch.load(path, map_location="cpu"))  # strict
        model.eval()
        with torch.no_grad():
            probs, vals = model(obs)
        for h, p in enumerate(probs):
            np.testing.assert_allclose(p.numpy(), g[f"{pol}.probs{h}"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(vals.numpy(), g[f"{pol}.values"], rtol=1e-6, atol=1e-6)


def test_update_plan_names_what_runs_and_what_falls_back():
    """`UpdateKernels.update_plan` (logged once at trainer start): a float32 [256, 256] policy at a BASELINE-sized batch is
    served end to end by the hand-written kernels; below 65 536 rows, for other widths / depths and under autocast the steps
    that fall back to the framework say so.  (Predicates only: no GPU, nothing allocated.)"""
    from warp_drive_amd.training.update_kernels import UpdateKernels, _ShapeOnly

    k = UpdateKernels.__new__(UpdateKernels)   # (the predicates do not touch the function manager)
    orig = UpdateKernels.update_plan

    def plan(model, rows, autocast=False):   # stand in for a CUDA model: the predicates ask `is_cuda` of their tensors
        import warp_drive_amd.training.update_kernels as uk

        saved = uk._ShapeOnly
        try:
            uk._ShapeOnly = lambda shape, dev: saved(shape, "cuda:0")
            return orig(k, model, rows, autocast)
        finally:
            uk._ShapeOnly = saved

    big = plan(FullyConnected(71, [21, 21], (256, 256)), 250 * 2000 * 100)
    assert big == {"objective": "HipPolicyGradientHead", "output layer backward + mask + bias": "HipHeadBackwardBx3_W43",
                   "dW2": "HipWeightGradBx3_256x256", "input gradient of layer 2 + mask": "HipLinearMaskBackwardBx3_256",
                   "dW1 + db1": "HipWeightGradBx3_256x96"}
    small = plan(FullyConnected(71, [21, 21], (256, 256)), 13050)
    assert small["output layer backward + mask + bias"] == "HipHeadBackward_W43" and small["dW2"] == "framework" \
        and small["dW1 + db1"] == "framework" and small["input gradient of layer 2 + mask"] == "HipLinearMaskBackwardBx3_256"
    other = plan(FullyConnected(21, [5], (32, 32)), 10 ** 6)
    assert other["output layer backward + mask + bias"] == "framework" and other["dW2"] == "framework"
    assert "framework" in plan(FullyConnected(71, [21, 21], (256, 256)), 10 ** 7, autocast=True)["backward"]
    assert "framework" in plan(FullyConnected(71, [21, 21], (256,)), 10 ** 7)["hidden layers below the last"]
    assert _ShapeOnly((3, 4), "cpu").is_contiguous() and _ShapeOnly((3, 4), "cpu").dim() == 2


def test_policy_without_hidden_layers_and_the_private_gemm_epilogue():
    """`fc_dims: []` is the output layer alone on both entry points (forward_logits raised KeyError '-1' once); and the one
    private torch entry point on the recomputing path, `torch._addmm_activation(bias, x, w.T, use_gelu=False)`, is pinned:
    if its signature or meaning changes this fails loudly here instead of `_linear_relu` silently switching form."""
    from warp_drive_amd.training import models

    torch.manual_seed(3)
    m = FullyConnected(9, [4, 3], [])
    x = torch.randn(5, 2, 9)
    probs, values = m(x)
    logits = m.forward_logits(x)
    assert logits.shape == (5, 2, 8)
    np.testing.assert_allclose(torch.softmax(logits[..., :4], -1).detach().numpy(), probs[0].detach().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(logits[..., 7].detach().numpy(), values.detach().numpy(), rtol=1e-6, atol=1e-7)
    assert hasattr(torch, "_addmm_activation"), "torch._addmm_activation is gone: give models._linear_relu a new fused form"
    w, b, xx = torch.randn(6, 9), torch.randn(6), torch.randn(11, 9)
    fused = torch._addmm_activation(b, xx, w.t(), use_gelu=False)
    assert torch.allclose(fused, torch.relu(torch.nn.functional.linear(xx, w, b)), rtol=1e-6, atol=1e-6)
    assert torch.equal(models._linear_relu(xx, w, b), torch.relu(torch.nn.functional.linear(xx, w, b)))  # (CPU: plain form)


def test_parameter_versions_see_every_unannounced_change():
    """what the stored-activation guard of the trainer relies on: an optimizer step, `load_state_dict` and a manual
    in-place edit each advance a parameter's version counter; reading the parameters does not"""
    from warp_drive_amd.training.policy_kernel import parameter_versions

    m = FullyConnected(7, [3], (8, 8))
    v0 = parameter_versions(m)
    m(torch.randn(4, 7))[1].sum().backward()
    assert parameter_versions(m) == v0
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    opt.step()
    v1 = parameter_versions(m)
    assert all(b > a for a, b in zip(v0, v1))
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    v2 = parameter_versions(m)
    assert all(b > a for a, b in zip(v1, v2))
    with torch.no_grad():
        m.fc["1"][0].bias.add_(1.0)
    v3 = parameter_versions(m)
    assert sum(b > a for a, b in zip(v2, v3)) == 1


def test_inference_copies_are_refreshed_in_place():
    """`forward_inference` keeps cast / concatenated copies of the weights; a rollout tick captured in a hipGraph reads them
    at fixed addresses, so `refresh_inference_cache` (called after every optimizer step) must update the SAME tensors --
    replacing them left a replayed tick on stale head weights (round 6: Cartpole on the per-tick path un-learned)."""
    torch.manual_seed(1)
    m = FullyConnected(4, [2], (32, 32))
    x = torch.randn(6, 1, 4)
    for dtype in (None, torch.bfloat16):
        m.forward_inference(x, dtype=dtype)
    heads = {dt: m._inference_cache[dt]["head"] for dt in (None, torch.bfloat16)}
    ptrs = {dt: (w.data_ptr(), b.data_ptr()) for dt, (w, b) in heads.items()}
    opt = torch.optim.SGD(m.parameters(), lr=0.5)
    m.forward_logits(x).square().sum().backward()
    opt.step()
    m.refresh_inference_cache()
    for dt in (None, torch.bfloat16):
        w, b = m._inference_cache[dt]["head"]
        assert (w.data_ptr(), b.data_ptr()) == ptrs[dt] and w is heads[dt][0]
    want = torch.cat([h.weight for h in m.policy_head] + [m.vf_head.weight], dim=0)
    assert torch.equal(heads[None][0], want) and torch.equal(heads[torch.bfloat16][0], want.to(torch.bfloat16))
    probs_i, vals_i = m.forward_inference(x)
    probs, vals = m(x)
    assert torch.allclose(probs_i[0], probs[0], atol=1e-6) and torch.allclose(vals_i, vals, atol=1e-6)
    # an unannounced in-place change is still seen (version counters) the next time the host runs the forward
    with torch.no_grad():
        m.vf_head.bias.add_(1.0)
    assert torch.allclose(m.forward_inference(x)[1], vals + 1.0, atol=1e-6)
    assert m._inference_cache[None]["head"][0].data_ptr() == ptrs[None][0]
