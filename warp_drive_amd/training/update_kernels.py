"""The trainer's update as hand-written kernels (csrc/kernels/policy_mlp.hip, the trainer's code object).

At configs[2] a training batch is ~1e7 rows.  For a float32 network of two 256-wide hidden layers the whole update runs
here (training/models.py::_MlpTwoHidden, training/losses.py); other shapes use the kernels that cover them and the
framework's operations for the rest (profiles/r05_update_kernels.txt, r05_update_timeline.txt):

  discounted_returns     HipDiscountedReturns: bootstrapped returns + advantages of a batch, one thread per (replica, agent)
                         (reference algorithms/policygradient/a2c.py:80-95; bit-identical to the framework loop over T)
  policy_gradient_head   HipPolicyGradientHead: A2C / PPO objective + its gradient with respect to the network's output
                         in ONE pass (softmax of both heads, log-probability of the taken actions, entropies, value
                         loss; reference a2c.py:97-194, ppo.py:150-228)
  head_backward          HipHeadBackwardBx3_W<w> (256 hidden units, bf16 matrix cores) / HipHeadBackward_W<w> (vector units):
                         the output layer's backward, the last hidden layer's ReLU mask + bias gradient and the output
                         layer's weight gradient in one pass over the hidden activations
  linear_mask_backward   HipLinearMaskBackwardBx3_<C>: a hidden layer's input gradient with the ReLU mask of the layer
                         under it applied to the accumulators
  weight_grad            HipWeightGradBx3_256x{256,96}: a layer's weight gradient g^T . x over the batch (+ its bias gradient
                         as a column of ones)
  relu_backward_colsum   HipReluBackwardColumnSums: the ReLU mask of a hidden layer's backward and that layer's bias
                         gradient in one pass (the per-layer path of other network shapes)

"Bx3" = bf16x3 arithmetic: every float32 operand is split exactly into three bf16 terms and a product is the float32 sum
of the six partial products that reach 2^-24 of it -- float32-accurate at 2.7 x the float32 matrix rate.

An `UpdateKernels` object belongs to ONE function manager (one device) and is handed around explicitly: the trainer
stores it on each of its models (`FullyConnected.update_kernels`, from where it reaches the autograd nodes as an argument)
and passes it to the objective (`compute_loss_and_metrics_from_logits(..., kernels=)`).  There is no process-wide switch: a
trainer built with `fused_update: False`, or on another device, next to one that uses the kernels is unaffected; without a
handle (CPU tests) both modules run their framework paths."""
import logging

import numpy as np
import torch


class UpdateKernels:
    ROWS_PER_BLOCK = 4096   # rows per block of the passes that leave per-block partial sums (reduced by a framework sum)
    HEAD_WIDTHS = (43, 6, 3)  # output widths HipHeadBackward_W<w> exists for (21 + 21 + 1, 5 + 1, 2 + 1)

    def __init__(self, function_manager):
        assert function_manager is not None
        names = ["HipPolicyGradientHead", "HipReluBackwardColumnSums"]
        function_manager.initialize_functions(names)
        self.head_fn, self.relu_fn = (function_manager.get_function(n) for n in names)
        self._fm, self._head_backward_fns = function_manager, {}

    # ------------------------------------------------------------------------------------------------ what will run
    def update_plan(self, model, rows, autocast=False):
        """{step of the update: the kernel of this module that runs it, or "framework"} for one policy network and `rows`
        batch rows -- decided by the same `supports_*` predicates the backward consults, on empty stand-in tensors.
        The trainer logs this once at start: a shape the kernels do not cover falls back to the framework's GEMMs
        step by step, and that is then on record rather than silent."""
        dev = next(model.parameters()).device
        W, widths = sum(model.head_sizes) + 1, [model.fc[str(i)][0].out_features for i in range(len(model.fc))]
        obs = model.fc["0"][0].in_features if widths else model.vf_head.in_features
        f32 = lambda *shape: _ShapeOnly(shape, dev)  # noqa: E731
        plan = {"objective": "HipPolicyGradientHead" if self.supports_head(f32(rows, W), model.head_sizes) else "framework"}
        if autocast or not widths:
            plan["backward"] = "framework GEMMs" + (" + HipReluBackwardColumnSums" if widths and all(
                self.supports_relu_backward(f32(rows, c), f32(rows, c)) for c in widths) else "")
            return plan
        C = widths[-1]
        if self.supports_head_backward(f32(rows, W), f32(W, C), f32(rows, C)):
            plan["output layer backward + mask + bias"] = (f"HipHeadBackwardBx3_W{W}" if C == 256 and rows >= self.WEIGHT_GRAD_MIN_ROWS
                                                           else f"HipHeadBackward_W{W}")
        else:
            plan["output layer backward + mask + bias"] = "framework"
        if len(widths) == 2:
            plan["dW2"] = "HipWeightGradBx3_256x256" if self.supports_weight_grad(f32(rows, C), f32(rows, widths[0])) else "framework"
            plan["input gradient of layer 2 + mask"] = (f"HipLinearMaskBackwardBx3_{C}" if self.supports_linear_mask_backward(
                f32(rows, C), f32(C, C), f32(rows, widths[0])) and widths[0] == C else "framework")
            plan["dW1 + db1"] = ("HipWeightGradBx3_256x96" if self.supports_weight_grad(f32(rows, widths[0]), f32(rows, obs), with_bias=True)
                                 else "framework")
        else:
            plan["hidden layers below the last"] = "framework GEMMs + HipReluBackwardColumnSums"
        return plan

    @staticmethod
    def log_update_plan(policy, plan):
        fallen_back = [k for k, v in plan.items() if v.startswith("framework")]
        logging.info(f"update of policy '{policy}': " + "; ".join(f"{k}: {v}" for k, v in plan.items()))
        if fallen_back:
            logging.warning(f"update of policy '{policy}': the network's shape is outside the hand-written kernels for "
                            f"{', '.join(fallen_back)} -- those steps run as the framework's operations")

    # ------------------------------------------------------------------------------------------------ objective
    @staticmethod
    def supports_head(out, head_sizes):
        W = sum(head_sizes) + 1
        return (out.is_cuda and out.dtype == torch.float32 and 1 <= len(head_sizes) <= 2 and W <= 64
                and out.shape[-1] == W)

    def policy_gradient_head(self, out, actions, adv, ret, head_sizes, ent_coeff, vf_coeff):
        """out [R, W] float32, actions [R, n_heads] int32, adv / ret [R] float32 ->
        (grad [R, W] = d loss / d out, sums float64 [4] = sum logp * adv, sum of entropies, sum (v - ret)^2, sum adv)"""
        R, W = out.shape
        assert out.is_contiguous() and actions.is_contiguous() and adv.is_contiguous() and ret.is_contiguous()
        assert actions.dtype == torch.int32 and tuple(actions.shape) == (R, len(head_sizes)) and adv.numel() == R == ret.numel()
        grad = torch.empty_like(out)
        blocks = (R + 255) // 256
        sums = torch.empty((blocks, 4), dtype=torch.float32, device=out.device)
        a1 = int(head_sizes[1]) if len(head_sizes) > 1 else 0
        self.head_fn(out, actions, adv, ret, grad, sums, np.int32(R), np.int32(head_sizes[0]), np.int32(a1),
                     np.float32(1.0 / R), np.float32(ent_coeff), np.float32(vf_coeff),
                     block=(256, 1, 1), grid=(blocks, 1), shared=4 * 256 * W)
        return grad, sums.sum(dim=0, dtype=torch.float64)

    # --------------------------------------------------------------------------------------------- hidden layers
    @staticmethod
    def supports_relu_backward(g, y):
        C = g.shape[-1]
        return (g.is_cuda and g.dtype == torch.float32 and y.dtype == torch.float32 and g.is_contiguous()
                and y.is_contiguous() and g.shape == y.shape and C % 4 == 0 and 16 <= C <= 256 and 256 % (C // 4) == 0)

    def relu_backward_colsum(self, g, y):
        """g, y [R, C] float32 -> (g * [y > 0] as a new tensor, its column sums float32 [C])"""
        R, C = g.shape
        out = torch.empty_like(g)
        blocks = (R + self.ROWS_PER_BLOCK - 1) // self.ROWS_PER_BLOCK
        partial = torch.empty((blocks, C), dtype=torch.float32, device=g.device)
        self.relu_fn(g, y, out, partial, np.int64(R), np.int32(C), np.int32(self.ROWS_PER_BLOCK),
                     block=(256, 1, 1), grid=(blocks, 1), shared=0)
        return out, partial.sum(dim=0)


    # ------------------------------------------------------------------------------------ output layer's backward
    def supports_head_backward(self, g3, w3, h2):
        W, C = w3.shape
        return (g3.is_cuda and g3.dtype == w3.dtype == h2.dtype == torch.float32 and W in self.HEAD_WIDTHS
                and C in (64, 128, 256) and g3.shape[-1] == W and h2.shape[-1] == C and h2.is_contiguous())

    def head_backward(self, g3, w3, h2):
        """g3 [R, W] = d loss / d out, w3 [W, C] (all heads' rows, then the value's), h2 [R, C] = relu(...) ->
        (g2 [R, C] = (g3 @ w3) * [h2 > 0], db2 [C] = its column sums, dw3 [W, C] = g3^T @ h2, db3 [W] = the column sums of
        g3 or None where the kernel does not produce them) in one pass"""
        R, C = h2.shape
        W = w3.shape[0]
        if C == 256 and R >= self.WEIGHT_GRAD_MIN_ROWS:
            return self._head_backward_bx3(g3.contiguous(), w3, h2)
        fn = self._head_backward_fns.get(W)
        if fn is None:
            self._fm.initialize_functions([f"HipHeadBackward_W{W}"])
            fn = self._head_backward_fns[W] = self._fm.get_function(f"HipHeadBackward_W{W}")
        g3, w3 = g3.contiguous(), w3.contiguous()
        g2 = torch.empty_like(h2)
        blocks = (R + self.ROWS_PER_BLOCK - 1) // self.ROWS_PER_BLOCK
        db2 = torch.empty((blocks, C), dtype=torch.float32, device=h2.device)
        dw3 = torch.empty((blocks, W, C), dtype=torch.float32, device=h2.device)
        fn(g3, w3, h2, g2, db2, dw3, np.int64(R), np.int32(self.ROWS_PER_BLOCK), block=(C, 1, 1), grid=(blocks, 1), shared=0)
        return g2, db2.sum(dim=0), dw3.sum(dim=0), None


    HEAD_BACKWARD_STAGES = 3   # WD_HEAD_BACKWARD_STAGES of the kernel source

    def _head_backward_bx3(self, g3, w3, h2):
        """`head_backward` for 256 hidden units on the bf16 matrix cores (HipHeadBackwardBx3_W<W>: bf16x3 arithmetic,
        float32-accurate); the last R % 32 rows go through the framework"""
        from warp_drive_amd.training.policy_kernel import split_bf16x3

        R, C = h2.shape
        W = w3.shape[0]
        ks = (W + 15) // 16
        key = ("head_backward_bx3", W, str(h2.device))
        if key not in self._head_backward_fns:
            name = f"HipHeadBackwardBx3_W{W}"
            self._fm.initialize_functions([name])
            cus = torch.cuda.get_device_properties(h2.device).multi_processor_count
            self._head_backward_fns[key] = (self._fm.get_function(name), int(cus))
        fn, blocks = self._head_backward_fns[key]
        # W3^T as the kernel's A operand, in register-image order [wave][tile][k step][term][lane = 32 kg + i][e]:
        # element = W3[k = 16 ks + 8 kg + e][unit = 64 wave + 32 tile + i], zero for k >= W
        wt = torch.zeros((C, 16 * ks), dtype=torch.float32, device=h2.device)
        wt[:, :W] = w3.detach().t()
        w3pk = (split_bf16x3(wt).reshape(3, 4, 2, 32, ks, 2, 8)     # [term][wave][tile][i][k step][kg][e]
                .permute(1, 2, 4, 0, 5, 3, 6).contiguous())          # [wave][tile][k step][term][kg][i][e]
        main = R - R % 32
        rows_per_block = -(-main // (32 * blocks)) * 32
        blocks = -(-main // rows_per_block)
        g2 = torch.empty_like(h2)
        db2 = torch.empty((blocks, C), dtype=torch.float32, device=h2.device)
        dw3 = torch.empty((blocks, W, C), dtype=torch.float32, device=h2.device)
        db3 = torch.empty((blocks * 4, W), dtype=torch.float32, device=h2.device)
        g3_max = 31 * W + max(32 * ((W + 31) // 32), 16 * ks) - 1
        g3_floats = 1024 * ((g3_max // 256 + 1 + 3) // 4)
        fn(g3, w3pk, h2, g2, db2, dw3, db3, np.int64(main), np.int64(rows_per_block), block=(256, 1, 1), grid=(blocks, 1),
           shared=self.HEAD_BACKWARD_STAGES * 4 * (g3_floats + 32 * 260))
        db2, dw3, db3 = db2.sum(dim=0), dw3.sum(dim=0), db3.sum(dim=0)
        if main < R:
            tail = torch.ops.aten.threshold_backward(g3[main:] @ w3, h2[main:], 0)
            g2[main:] = tail
            db2 = db2 + tail.sum(dim=0)
            dw3 = dw3 + g3[main:].t() @ h2[main:]
            db3 = db3 + g3[main:].sum(dim=0)
        return g2, db2, dw3, db3

    # ------------------------------------------------ a hidden layer's input gradient + the mask of the layer under it
    def supports_linear_mask_backward(self, g_in, w, h):
        C = w.shape[0]
        return (g_in.is_cuda and g_in.dtype == w.dtype == h.dtype == torch.float32 and w.shape == (C, C)
                and C in (64, 128, 256) and g_in.shape == h.shape and g_in.shape[-1] == C and g_in.is_contiguous()
                and h.is_contiguous())

    def linear_mask_backward(self, g_in, w, h):
        """g_in [R, C] (gradient with respect to a C x C layer's pre-activations), w [C out, C in], h [R, C] = the post-ReLU
        activations of the layer under it -> (g_in @ w) * [h > 0], the product in bf16x3 arithmetic (float32-accurate: six
        bf16 partial products per float32 product, as the rollout's forward kernel)"""
        from warp_drive_amd.training.policy_kernel import _pack_indices_bx3, split_bf16x3

        R, C = h.shape
        tn = C // 32
        key = ("mask_backward", C, str(h.device))
        if key not in self._head_backward_fns:
            name = f"HipLinearMaskBackwardBx3_{C}"
            self._fm.initialize_functions([name])
            rows, cols = _pack_indices_bx3(tn, tn, True)
            self._head_backward_fns[key] = (self._fm.get_function(name), torch.from_numpy(rows).to(h.device),
                                            torch.from_numpy(cols).to(h.device))
        fn, rows, cols = self._head_backward_fns[key]
        # A operand of G_out^T = W^T . G_in^T: rows = this layer's INPUT units, contraction over its output units
        wpk = split_bf16x3(w.detach().t().contiguous())[:, rows, cols].transpose(0, 1).contiguous()
        g_out = torch.empty_like(h)
        # a wavefront = 32 rows; 8 per block where a weight chunk's 6 * tn KB divide over 8 wavefronts (C = 128, 256): the
        # chunk fetched into LDS then serves twice the rows (9.75 -> 9.2 ms at configs[2])
        waves = 8 if (6 * tn) % 8 == 0 else 4
        fn(g_in, wpk, h, g_out, np.int64(R), block=(64 * waves, 1, 1), grid=((R + 32 * waves - 1) // (32 * waves), 1),
           shared=3 * tn * 6144)
        return g_out

    # ------------------------------------------------------------------ discounted returns of a batch
    def discounted_returns(self, rewards, done_flags, out, gamma):
        """rewards [T, E, n] float32, done_flags [T, E], out [T, E, n, W] (the network's output rows: the value is the last
        column) -> (returns [T, E, n], returns - values) as ONE kernel, bit-identical to losses.discounted_returns"""
        T, E, n = rewards.shape
        W = out.shape[-1]
        if "returns" not in self._head_backward_fns:
            self._fm.initialize_functions(["HipDiscountedReturns"])
            self._head_backward_fns["returns"] = self._fm.get_function("HipDiscountedReturns")
        returns, adv = torch.empty_like(rewards), torch.empty_like(rewards)
        self._head_backward_fns["returns"](rewards, done_flags, out, np.int32(W), np.int32(W - 1), np.float32(gamma),
                                           np.int32(T), np.int32(E), np.int32(n), returns, adv,
                                           block=(256, 1, 1), grid=((E * n + 255) // 256, 1), shared=0)
        return returns, adv

    @staticmethod
    def supports_discounted_returns(rewards, done_flags, out):
        return (rewards.is_cuda and rewards.dtype == out.dtype == torch.float32 and rewards.dim() == 3 and out.dim() == 4
                and rewards.is_contiguous() and out.is_contiguous() and done_flags.is_contiguous()
                and done_flags.dtype == torch.int32 and tuple(done_flags.shape) == tuple(rewards.shape[:2])
                and tuple(out.shape[:3]) == tuple(rewards.shape))

    # ------------------------------------------------------------------ a layer's weight (and bias) gradient over the batch
    WEIGHT_GRAD_MIN_ROWS = 1 << 16   # below this the framework GEMM is as good (launch-bound either way)
    WEIGHT_GRAD_STAGES = 4           # WD_WEIGHT_GRAD_STAGES of the kernel source

    @staticmethod
    def supports_weight_grad(g, x, with_bias=False):
        ci = x.shape[-1]
        return (g.is_cuda and g.dtype == x.dtype == torch.float32 and g.dim() == x.dim() == 2 and g.shape[0] == x.shape[0]
                and g.shape[0] >= UpdateKernels.WEIGHT_GRAD_MIN_ROWS and g.shape[1] == 256 and g.is_contiguous()
                and x.is_contiguous() and (ci < 96 or (ci == 256 and not with_bias)))

    def weight_grad(self, g, x, with_bias=False):
        """g [R, 256] (gradient with respect to a layer's pre-activations), x [R, ci] (the layer's input), ci = 256 or < 96 ->
        (g^T @ x [256, ci], column sums of g [256] or None) in bf16x3 arithmetic (HipWeightGradBx3_*: six bf16 partial products
        per float32 product, float32 accumulation); with_bias: the bias gradient rides as one more input column of ones"""
        R, ci = x.shape
        cip = 256 if ci == 256 else 96
        key = ("weight_grad", cip, str(g.device))
        if key not in self._head_backward_fns:
            name = f"HipWeightGradBx3_256x{cip}"
            self._fm.initialize_functions([name])
            cus = torch.cuda.get_device_properties(g.device).multi_processor_count
            self._head_backward_fns[key] = (self._fm.get_function(name), int(cus))
        fn, blocks = self._head_backward_fns[key]
        main = R - R % 32   # the kernel takes pairs of whole 16-row steps; the last R % 32 rows are added below
        rows_per_block = -(-main // (32 * blocks)) * 32
        blocks = -(-main // rows_per_block)
        partial = torch.empty((blocks, 256, cip), dtype=torch.float32, device=g.device)
        fn(g, x, partial, np.int64(main), np.int32(ci), np.int32(ci if with_bias else -1), np.int64(rows_per_block),
           block=(256, 1, 1), grid=(blocks, 1), shared=self.WEIGHT_GRAD_STAGES * 4 * (16 * 260 + (16 * 260 if cip == 256 else 2048)))
        total = partial.sum(dim=0)
        gw, gb = total[:, :ci], (total[:, ci] if with_bias else None)
        if main < R:
            gw = gw + g[main:].t() @ x[main:]
            gb = gb + g[main:].sum(dim=0) if with_bias else None
        return gw, gb


class FusedObjective(torch.autograd.Function):
    """loss = policy_loss + vf_coeff * vf_loss - ent_coeff * mean_entropy as ONE kernel that also produces d loss / d
    out; `terms` (float64 [3], detached: policy loss, value loss, mean entropy) is returned beside it for the metrics."""

    @staticmethod
    def forward(ctx, k, out, actions, adv, ret, head_sizes, ent_coeff, vf_coeff, ppo):
        R = out.shape[0]
        grad, sums = k.policy_gradient_head(out, actions, adv, ret, head_sizes, ent_coeff, vf_coeff)
        policy_loss = -(sums[3] if ppo else sums[0]) / R  # PPO at ratio 1: min(ratio * A, clamp(ratio) * A) = A
        vf_loss, entropy = sums[2] / R, sums[1] / R
        loss = policy_loss + vf_coeff * vf_loss - ent_coeff * entropy
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(terms := torch.stack([policy_loss, vf_loss, entropy]))
        return loss.to(torch.float32), terms

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        (grad,) = ctx.saved_tensors
        if g_loss.data_ptr() == unit_gradient(grad.device).data_ptr():
            STATS["unit_gradient_hits"] += 1
            return None, grad, None, None, None, None, None, None, None   # d loss / d loss = 1, known without reading it
        return None, grad * g_loss.to(grad.dtype), None, None, None, None, None, None, None


class _ShapeOnly:
    """what the `supports_*` predicates read of a tensor (device, dtype, shape, contiguity), answered for a contiguous
    float32 tensor of `shape` on `device` that is never allocated (a training batch's activations are gigabytes)"""

    def __init__(self, shape, device):
        self.device, self.dtype, self.shape = torch.device(device), torch.float32, torch.Size(shape)
        self.is_cuda = self.device.type == "cuda"

    def is_contiguous(self):
        return True

    def dim(self):
        return len(self.shape)


_UNIT = {}
STATS = {"unit_gradient_hits": 0}   # (tests: the shortcut above was taken)


def unit_gradient(device):
    """THE float32 scalar 1.0 of `device`: `torch.autograd.backward(loss, grad_tensors=unit_gradient(loss.device))` is
    `loss.backward()`, except that FusedObjective.backward recognises the tensor by its address and skips the pass that
    would multiply the [rows, outputs] gradient by it (0.55 ms at configs[2]; it cannot look at the VALUE of an incoming
    gradient without a host synchronisation).  Any other incoming gradient -- a scaled loss, a sum of losses -- takes the
    multiplication."""
    key = str(device)
    if key not in _UNIT:
        _UNIT[key] = torch.ones((), dtype=torch.float32, device=device)
    return _UNIT[key]
