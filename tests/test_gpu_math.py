"""Device restatements of numpy's float32 routines, bit-for-bit (they are what makes the
TagContinuous float path bit-exact instead of 'within 1e-5')."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_device_math_is_numpy_exact():
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers import hip_driver as drv

    require_gpu()
    mod = drv.Module(drv.code_object_of("wd_test_math"))  # the test-only code object
    fn = mod.get_function("wd_test_math")
    rng = np.random.RandomState(0)
    n = 1 << 20
    from tests.test_oracle_golden import _tie_neighbourhoods

    ties = _tie_neighbourhoods()
    ties = ties[ties > 0]
    special = np.array([np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi, 6.2831855, 5 * np.pi / 4], np.float32)
    a = np.concatenate([(rng.rand(n - len(ties) - len(special)) * 2 * np.pi + 1e-3).astype(np.float32),
                        ties, special])
    b = (rng.rand(n) * 20 + 0.01).astype(np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    outs = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(5)]
    fn(ta, tb, *outs, np.int32(n), block=(256, 1, 1), grid=(1024, 1))
    torch.cuda.synchronize()
    s, c, rem, sq, dv = (o.cpu().numpy() for o in outs)
    np.testing.assert_array_equal(s.view(np.uint32), np.sin(a).view(np.uint32))
    np.testing.assert_array_equal(c.view(np.uint32), np.cos(a).view(np.uint32))
    np.testing.assert_array_equal(rem.view(np.uint32), np.remainder(a, b).view(np.uint32))
    np.testing.assert_array_equal(sq.view(np.uint32), np.sqrt(a * a + b * b).view(np.uint32))
    np.testing.assert_array_equal(dv.view(np.uint32), (a / b).view(np.uint32))
    # negative dividends: numpy's remainder takes the divisor's sign
    a2 = (-a).astype(np.float32)
    fn(torch.from_numpy(a2).cuda(), tb, *outs, np.int32(n), block=(256, 1, 1), grid=(1024, 1))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(outs[2].cpu().numpy().view(np.uint32), np.remainder(a2, b).view(np.uint32))
