"""Test infrastructure: the RNN-T loss as an explicit fp64 log-sum-exp lattice, differentiated by torch.autograd.

An INDEPENDENT checker -- no gradient formula is written down here, only the forward recursion of
docs/rnnt_notes.tex:60-100 / include/detail/cpu_rnnt.h:181-209 -- for the one case where the reference's two
locations disagree and the CPU oracle therefore cannot be the judge: a label that EQUALS the blank symbol.
  * GPU reference (include/detail/gpu_rnnt_kernel.h:161-174): the blank and the label corrections are independent `if`s,
    so both are subtracted from the blank column -- which IS the derivative of the loss (log p(blank | t,u) then feeds the
    blank transition AND the label transition of cell (t,u)); autograd gives exactly that.
  * CPU reference (include/detail/cpu_rnnt.h:253-267): the label term is ASSIGNED after the blank term and overwrites it.
CPU only, fp64, a Python loop over the anti-diagonals: for the small shapes of the tests.
"""
import numpy as np
import torch


def _sample_loss(z, lab, blank):
    """-log p(labels | z) of one sample: z (T, U, A) fp64 logits (a leaf or a view of one), lab (U-1,) ints."""
    T, U, _ = z.shape
    lp = torch.log_softmax(z, -1)
    pb = lp[:, :, blank]                                                  # (T, U)  log p(blank | t, u)
    if U > 1:
        idx = torch.as_tensor(np.asarray(lab[:U - 1], dtype=np.int64)).view(1, U - 1, 1).expand(T, U - 1, 1)
        pl = lp[:, :U - 1].gather(2, idx).squeeze(2)                       # (T, U-1) log p(label_u | t, u)
    # alpha over anti-diagonals d = t + u; diagonal d holds the cells t in [lo, hi], u = d - t
    prev_lo, prev = 0, torch.zeros(1, dtype=z.dtype)
    for d in range(1, T + U - 1):
        lo, hi = max(0, d - (U - 1)), min(d, T - 1)
        t = torch.arange(lo, hi + 1)
        u = d - t
        neg = torch.full((hi - lo + 1,), -float("inf"), dtype=z.dtype)
        # from (t-1, u) through the blank: needs t >= 1
        m = t >= 1
        top = neg.clone()
        if m.any():
            top[m] = prev[(t[m] - 1) - prev_lo] + pb[t[m] - 1, u[m]]
        # from (t, u-1) through label u-1: needs u >= 1
        m = u >= 1
        left = neg.clone()
        if m.any():
            left[m] = prev[t[m] - prev_lo] + pl[t[m], u[m] - 1]
        prev_lo, prev = lo, torch.logaddexp(top, left)
    return -(prev[(T - 1) - prev_lo] + pb[T - 1, U - 1])


def rnnt_autograd(acts, labels, act_lens, label_lens, blank=0, weights=None):
    """costs (N,) and d(sum_b w_b cost_b)/d(acts) (N,T,U,A) in fp64; the padding's gradient is zero by construction."""
    x = torch.tensor(np.asarray(acts, dtype=np.float64), requires_grad=True)
    N = x.shape[0]
    labels = np.asarray(labels).reshape(N, -1)
    costs = [_sample_loss(x[b, :int(act_lens[b]), :int(label_lens[b]) + 1], labels[b], blank) for b in range(N)]
    w = np.ones(N) if weights is None else np.asarray(weights, dtype=np.float64)
    sum(c * float(w[b]) for b, c in enumerate(costs)).backward()
    return np.array([c.item() for c in costs]), x.grad.numpy()


def rnnt_add_autograd(f, g, labels, act_lens, label_lens, blank=0):
    """The additive joint z[b,t,u,:] = f[b,t,:] + g[b,u,:]: costs, d(sum cost)/df, d(sum cost)/dg in fp64."""
    tf = torch.tensor(np.asarray(f, dtype=np.float64), requires_grad=True)
    tg = torch.tensor(np.asarray(g, dtype=np.float64), requires_grad=True)
    N = tf.shape[0]
    labels = np.asarray(labels).reshape(N, -1)
    costs = []
    for b in range(N):
        T, U = int(act_lens[b]), int(label_lens[b]) + 1
        costs.append(_sample_loss(tf[b, :T, None, :] + tg[b, None, :U, :], labels[b], blank))
    sum(costs).backward()
    return np.array([c.item() for c in costs]), tf.grad.numpy(), tg.grad.numpy()
