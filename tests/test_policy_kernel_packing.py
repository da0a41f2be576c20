"""Host side of the fused policy forward (training/policy_kernel.py): the packed weight layout is
checked by replaying the kernel's contraction in numpy -- lane l of a wavefront supplies
A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31] to a 32x32x2 MFMA and receives rows
(s & 3) + 8 (s >> 2) + 4 (l >> 5) of column l & 31 in accumulator register s -- against the PyTorch
network of the same weights.  (The device run is tests/test_gpu_policy_kernel.py.)"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


class _NoDevice:
    def initialize_functions(self, names):
        self.names = names

    def get_function(self, name):
        return None


def _emulate(packed, F, H, x):
    from warp_drive_amd.training.policy_kernel import _row_of

    w1p, b1p, w2p, b2p, w3p, b3p = [t.numpy() for t in packed]
    tn, kt1 = H // 32, (F + 31) // 32
    lanes = [(l & 31, l >> 5) for l in range(64)]
    feat = np.zeros((kt1, 64, 16), np.float64)
    for l, (j, h) in enumerate(lanes):
        for kt in range(kt1):
            for s in range(16):
                f = 32 * kt + 16 * h + s
                feat[kt, l, s] = x[j, f] if f < F else 0.0

    def layer(wp, bp, n_out, n_k, bfrag, relu):
        acc = np.zeros((n_out, 64, 16), np.float64)
        for t in range(n_out):
            for l, (j, h) in enumerate(lanes):
                acc[t, l, :] = bp[t, h, :]
        for kt in range(n_k):
            for s in range(16):
                A = np.zeros((n_out, 32, 2))
                B = np.zeros((2, 32))
                for l in range(64):
                    A[:, l & 31, l >> 5] = wp[kt, :, s // 4, l, s % 4]
                    B[l >> 5, l & 31] = bfrag[kt, l, s]
                D = A @ B  # [tile, row, column]
                for l, (j, h) in enumerate(lanes):
                    for r in range(16):
                        acc[:, l, r] += D[:, _row_of(r, h), j]
        return np.maximum(acc, 0.0) if relu else acc

    a1 = layer(w1p, b1p, tn, kt1, feat, True)
    a2 = layer(w2p, b2p, tn, tn, a1, True)
    a3 = layer(w3p, b3p, 2, tn, a2, False)
    out = np.zeros((32, 64))
    for t in range(2):
        for l, (j, h) in enumerate(lanes):
            for r in range(16):
                out[j, 32 * t + _row_of(r, h)] = a3[t, l, r]
    return out


@pytest.mark.parametrize("H,F,heads", [(64, 40, [5, 3]), (64, 71, [21, 21]), (128, 7, [2])])
def test_packed_layout_reproduces_the_network(H, F, heads):
    from warp_drive_amd.training.models import FullyConnected
    from warp_drive_amd.training.policy_kernel import FusedPolicyForward

    torch.manual_seed(H + F)
    model = FullyConnected(F, heads, fc_dims=(H, H))
    assert FusedPolicyForward.supports(model, F)
    fused = FusedPolicyForward(_NoDevice(), model, F)
    x = torch.randn(32, F)
    out = _emulate(fused.packed, F, H, x.numpy().astype(np.float64))
    with torch.no_grad():
        h = x
        for i in range(2):
            h = model.fc[str(i)](h)
        want = torch.cat([hd(h) for hd in model.policy_head] + [model.vf_head(h)], dim=1).numpy()
    np.testing.assert_allclose(out[:, :want.shape[1]], want, rtol=1e-5, atol=1e-5)
    assert np.all(out[:, want.shape[1]:] == 0.0)  # padded output rows: zero weights, zero bias
    # re-packing after a weight update goes into the same tensors (a captured graph keeps their addresses)
    ptrs = [t.data_ptr() for t in fused.packed]
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.1)
    fused.pack()
    assert ptrs == [t.data_ptr() for t in fused.packed]
    out2 = _emulate(fused.packed, F, H, x.numpy().astype(np.float64))
    assert np.abs(out2[:, :want.shape[1]] - want).max() > 1e-3


def test_unsupported_shapes_take_the_framework_path():
    from warp_drive_amd.training.models import FullyConnected
    from warp_drive_amd.training.policy_kernel import FusedPolicyForward

    assert not FusedPolicyForward.supports(FullyConnected(71, [21, 21], fc_dims=(256, 128)), 71)   # unequal widths
    assert not FusedPolicyForward.supports(FullyConnected(71, [21, 21], fc_dims=(256,)), 71)       # one hidden layer
    assert not FusedPolicyForward.supports(FullyConnected(729, [21, 21], fc_dims=(256, 256)), 729)  # full observations
    assert not FusedPolicyForward.supports(FullyConnected(71, [40, 40], fc_dims=(256, 256)), 71)   # > 63 output rows
    assert not FusedPolicyForward.supports(FullyConnected(71, [5, 5, 5], fc_dims=(64, 64)), 71)    # three heads
    assert FusedPolicyForward.supports(FullyConnected(4, [2], fc_dims=(64, 64)), 4)


def _emulate_bx3(packed, F, H, x):
    """the bf16x3 kernel's contraction replayed in numpy: lane l supplies A[i = l & 31][k = 8 (l >> 5) + e] and
    B[k = 8 (l >> 5) + e][j = l & 31] of a 32x32x16 MFMA (C/D layout as above); six partial products per k half, the
    activations split into three bf16 terms after every ReLU"""
    from warp_drive_amd.training.policy_kernel import _row_of, split_bf16x3

    w1p, b1p, w2p, b2p, w3p, b3p = packed
    tn, kt1 = H // 32, (F + 31) // 32
    lanes = [(l & 31, l >> 5) for l in range(64)]
    feat = np.zeros((kt1, 64, 16), np.float32)
    for l, (j, h) in enumerate(lanes):
        for kt in range(kt1):
            for q in range(2):
                for e in range(8):
                    f = 32 * kt + 16 * q + 8 * h + e
                    feat[kt, l, 8 * q + e] = x[j, f] if f < F else 0.0

    def layer(wp, bp, n_out, n_k, acts, relu):
        wp = wp.float().numpy().astype(np.float64)                       # [KT, 3, TN, 2, 64, 8]
        xs = split_bf16x3(torch.from_numpy(acts)).float().numpy().astype(np.float64)  # [3, KT, 64, 16]
        bp = bp.numpy()
        acc = np.zeros((n_out, 64, 16), np.float64)
        for t in range(n_out):
            for l, (j, h) in enumerate(lanes):
                acc[t, l, :] = bp[t, h, :]
        for kt in range(n_k):
            for q in range(2):
                for wt, xt in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
                    A = np.zeros((n_out, 32, 16))
                    B = np.zeros((16, 32))
                    for l in range(64):
                        A[:, l & 31, 8 * (l >> 5): 8 * (l >> 5) + 8] = wp[kt, wt, :, q, l, :]
                        B[8 * (l >> 5): 8 * (l >> 5) + 8, l & 31] = xs[xt, kt, l, 8 * q: 8 * q + 8]
                    D = A @ B
                    for l, (j, h) in enumerate(lanes):
                        for r in range(16):
                            acc[:, l, r] += D[:, _row_of(r, h), j]
        return (np.maximum(acc, 0.0) if relu else acc).astype(np.float32)

    a1 = layer(w1p, b1p, tn, kt1, feat, True)
    a2 = layer(w2p, b2p, tn, tn, a1, True)
    a3 = layer(w3p, b3p, 2, tn, a2, False)
    out = np.zeros((32, 64))
    for t in range(2):
        for l, (j, h) in enumerate(lanes):
            for r in range(16):
                out[j, 32 * t + _row_of(r, h)] = a3[t, l, r]
    return out


@pytest.mark.parametrize("H,F,heads", [(64, 71, [21, 21]), (128, 7, [2])])
def test_bf16x3_packed_layout_reproduces_the_network(H, F, heads):
    """the three-term split is float32-accurate: the emulated kernel agrees with the float32 network to 2e-6 on logits
    of magnitude ~1 (the float32 layout's own test uses 1e-5), and the terms sum back to the weights to 2^-24"""
    from warp_drive_amd.training.models import FullyConnected
    from warp_drive_amd.training.policy_kernel import FusedPolicyForward, split_bf16x3

    torch.manual_seed(H + F)
    model = FullyConnected(F, heads, fc_dims=(H, H))
    fused = FusedPolicyForward(_NoDevice(), model, F, arithmetic="bf16x3")
    assert fused.packed[0].dtype == torch.bfloat16 and tuple(fused.packed[0].shape) == ((F + 31) // 32, 3, H // 32, 2, 64, 8)
    assert fused.lds_bytes == 3 * 6144 * (H // 32)
    w = model.fc["1"][0].weight.detach()
    terms = split_bf16x3(w).double()
    assert (terms.sum(0) - w.double()).abs().max() <= 2.0 ** -24 * w.abs().max()
    x = torch.randn(32, F)
    out = _emulate_bx3(fused.packed, F, H, x.numpy())
    with torch.no_grad():
        h = x.double()
        dm = FullyConnected(F, heads, fc_dims=(H, H)).double()
        dm.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
        for i in range(2):
            h = dm.fc[str(i)](h)
        want = torch.cat([hd(h) for hd in dm.policy_head] + [dm.vf_head(h)], dim=1).numpy()
    np.testing.assert_allclose(out[:, :want.shape[1]], want, rtol=0, atol=2e-6)
    assert np.all(out[:, want.shape[1]:] == 0.0)
