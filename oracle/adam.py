"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the optimizer step of the reference's training loop.

The reference builds four `torch.optim.Adam(betas=(beta1, beta2), eps=eps)` + `MultiStepLR` (src/modules/model.py:54-99)
and steps them in src/trainer/trainer.py:155-165.  The arithmetic lives in a third-party dependency that IS installed in
this image (torch 2.10.0, torch/optim/adam.py::_single_tensor_adam, the default non-foreach/non-fused/non-capturable path
with weight_decay=0, amsgrad=False, maximize=False); this file restates it element-wise in numpy fp32 and is pinned
against the live `torch.optim.Adam` on CPU by tests/test_oracle_golden.py::test_oracle_adam_matches_torch (bit-exact but for <1e-5 of the
elements, 1 ulp, where ATen's scalar tail loop rounds differently from its vector body).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module."""
import math

import numpy as np


def multistep_lr(base_lr, gamma, milestones, epoch):
    """torch.optim.lr_scheduler.MultiStepLR closed form: base_lr * gamma ** (number of milestones <= epoch)."""
    return base_lr * gamma ** sum(1 for m in milestones if m <= epoch)


class AdamState:
    def __init__(self):
        self.step = 0
        self.exp_avg = None
        self.exp_avg_sq = None


def adam_step(param, grad, st, lr, beta1, beta2, eps):
    """In-place update of `param` (float32 ndarray) following _single_tensor_adam, operation by operation in fp32."""
    f = np.float32
    if st.exp_avg is None:
        st.exp_avg = np.zeros_like(param)
        st.exp_avg_sq = np.zeros_like(param)
    st.step += 1
    step = float(st.step)
    w = f(1.0 - beta1)
    # exp_avg.lerp_(grad, 1 - beta1): ATen lerp = (w < 0.5) ? a + w*(b - a) : b - (b - a)*(1 - w)
    d = grad - st.exp_avg
    st.exp_avg[...] = st.exp_avg + w * d if w < f(0.5) else grad - d * (f(1.0) - w)
    # exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2): ATen's vectorized addcmul is one fused
    # multiply-add, fma(alpha*b, c, a) with alpha cast to fp32 (emulated in float64: the product of two fp32 is exact
    # there, so only the final rounding differs from a true fma in astronomically rare double-rounding cases)
    st.exp_avg_sq[...] = st.exp_avg_sq * f(beta2)
    st.exp_avg_sq[...] = (st.exp_avg_sq.astype(np.float64) +
                          (f(1.0 - beta2) * grad).astype(np.float64) * grad.astype(np.float64)).astype(f)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    denom = np.sqrt(st.exp_avg_sq) / f(math.sqrt(bc2)) + f(eps)
    # param.addcdiv_(exp_avg, denom, value=-step_size): ATen evaluates self + alpha * t1 / t2 left to right = (alpha*t1)/t2
    param[...] = param + (f(-step_size) * st.exp_avg) / denom
    return param
