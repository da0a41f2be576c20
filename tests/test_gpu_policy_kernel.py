"""Fused policy forward (csrc/kernels/policy_mlp.hip) against the plain PyTorch float32 network of
the same weights: probabilities per head, values, the batch copy of the observation rows.  float32
MFMA is an fmaf chain, so the only difference to the framework's GEMMs is summation order: the
tolerance is 2e-6 absolute on probabilities (logits of magnitude ~1, K <= 256)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


def _fm():
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers.function_manager import HIPFunctionManager

    require_gpu()

    fm = HIPFunctionManager(num_agents=1, num_envs=1)
    fm.load_hip_from_binary_file()
    return fm


@pytest.mark.parametrize("hidden,F,heads,E,N,ids", [
    (256, 71, [21, 21], 37, 13, [2, 3, 5, 7, 8, 9, 10, 11, 12]),   # TagContinuous shape, rows not a multiple of 32
    (256, 71, [21, 21], 64, 105, list(range(5, 105))),             # the runners of the bench shape
    (256, 71, [21, 21], 64, 105, [0, 1, 2, 3, 4]),                 # the taggers
    (128, 40, [5], 50, 6, [0, 1, 2, 3, 4, 5]),                      # one head, two k-tiles
    (64, 4, [2], 300, 1, [0]),                                      # Cartpole-like: tiny rows, tiny head
    (256, 96, [30, 33], 9, 4, [3, 1]),                              # widest supported row, 64 output rows, ids out of order
])
@pytest.mark.parametrize("arithmetic", ["float32", "bf16x3"])
def test_fused_forward_matches_torch(hidden, F, heads, E, N, ids, arithmetic):
    from warp_drive_amd.training.models import FullyConnected
    from warp_drive_amd.training.policy_kernel import FusedPolicyForward

    torch.manual_seed(hidden + F + E)
    dev = torch.device("cuda:0")
    model = FullyConnected(F, heads, fc_dims=(hidden, hidden)).to(dev)
    with torch.no_grad():  # logits of a useful spread, biases that matter
        for p in model.parameters():
            p.mul_(3.0) if p.dim() == 2 else p.normal_(0.0, 0.5)
    assert FusedPolicyForward.supports(model, F)
    fused = FusedPolicyForward(_fm(), model, F, arithmetic=arithmetic)  # (the SAME gates for both arithmetics)
    obs = torch.randn(E, N, F, device=dev)
    ids_t = torch.tensor(ids, dtype=torch.int32, device=dev)
    n_pol = len(ids)
    probs = [torch.full((E, N, a), -7.0, device=dev) for a in heads]
    values = torch.full((E, n_pol), -7.0, device=dev)
    T = 3
    obs_out = torch.full((T, E, n_pol, F), -7.0, device=dev)
    row = torch.tensor(2, dtype=torch.int64, device=dev)
    fused(obs, ids_t, probs, values=values, obs_out=obs_out, batch_row=row)
    torch.cuda.synchronize()
    with torch.no_grad():
        obs_p = obs.index_select(1, ids_t.long())
        want_probs, want_values = model(obs_p)
    for h, (got, want) in enumerate(zip(probs, want_probs)):
        sel = got.index_select(1, ids_t.long())
        np.testing.assert_allclose(sel.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=f"head {h}")
        np.testing.assert_allclose(sel.sum(-1).cpu().numpy(), 1.0, atol=1e-5)
        others = [a for a in range(N) if a not in ids]
        if others:  # rows of other policies' agents are not touched
            assert (got[:, others] == -7.0).all()
    np.testing.assert_allclose(values.cpu().numpy(), want_values.cpu().numpy(), rtol=2e-5, atol=2e-5)
    assert torch.equal(obs_out[2], obs_p) and (obs_out[0] == -7.0).all() and (obs_out[1] == -7.0).all()


def test_pack_follows_weight_updates():
    from warp_drive_amd.training.models import FullyConnected
    from warp_drive_amd.training.policy_kernel import FusedPolicyForward

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = FullyConnected(71, [21, 21], fc_dims=(256, 256)).to(dev)
    fused = FusedPolicyForward(_fm(), model, 71)
    obs = torch.randn(8, 5, 71, device=dev)
    ids = torch.arange(5, dtype=torch.int32, device=dev)
    probs = [torch.zeros(8, 5, 21, device=dev) for _ in range(2)]
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    fused.pack()
    fused(obs, ids, probs)
    torch.cuda.synchronize()
    with torch.no_grad():
        want, _ = model(obs)
    for got, w in zip(probs, want):
        np.testing.assert_allclose(got.cpu().numpy(), w.cpu().numpy(), rtol=2e-5, atol=2e-6)
