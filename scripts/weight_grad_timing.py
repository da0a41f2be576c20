import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, time
from warp_drive_amd.managers.function_manager import HIPFunctionManager
from warp_drive_amd.training.update_kernels import UpdateKernels
from warp_drive_amd.training import models
fm = HIPFunctionManager(num_agents=1, num_envs=1); fm.load_hip_from_binary_file()
k = UpdateKernels(fm)
R = 10_000_000
g = torch.randn(R, 256, device="cuda"); h = torch.randn(R, 256, device="cuda"); x = torch.randn(R, 71, device="cuda")
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("dW2 kernel %.2f ms" % t(lambda: k.weight_grad(g, h)))
print("dW1+db1 kernel %.2f ms" % t(lambda: k.weight_grad(g, x, with_bias=True)))
w2 = torch.randn(256, 256, device="cuda") / 16
hh = torch.relu(h)
print("mask backward %.2f ms" % t(lambda: k.linear_mask_backward(g, w2, hh)))
g3 = torch.randn(R, 43, device="cuda"); w3 = torch.randn(43, 256, device="cuda") * 0.2
print("head backward %.2f ms" % t(lambda: k.head_backward(g3, w3, hh)))
out43 = torch.randn(R, 43, device="cuda"); acts = torch.randint(0, 21, (R, 2), device="cuda", dtype=torch.int32)
adv = torch.randn(R, device="cuda"); ret = torch.randn(R, device="cuda")
print("objective %.2f ms" % t(lambda: k.policy_gradient_head(out43, acts, adv, ret, (21, 21), 0.05, 1.0)))
if len(sys.argv) > 1:
    sys.exit(0)
def bmm(a, b):
    return torch.bmm(a.reshape(250, R // 250, a.shape[1]).transpose(1, 2), b.reshape(250, R // 250, b.shape[1])).sum(0)
print("dW2 bmm %.2f ms" % t(lambda: bmm(g, h)))
print("dW1 bmm %.2f ms + colsum %.2f ms" % (t(lambda: bmm(g, x)), t(lambda: models._column_sums(g))))
