"""Restatement of ChamferDistancePytorch@719b0f1 `dist_chamfer.chamferDist`.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The CUDA extension is absent (pinned only by
commit hash in /root/reference/README.md:7); call sites /root/reference/global_optimization.py
:292-294 and :349-353.  Algorithm: SURVEY.md Appendix A.4 -- squared L2 nearest neighbour by
direct differences in both directions, argmin = lowest index among ties (ascending scan with a
strict `<`), backward 2*g*(a_i - b_idx) scattered to both inputs.  Parity unpinned against the
real extension.  Also restates /root/reference/chamfer_python.py:4-15 for the equal-size check.
"""
import torch


def nn_direct(query: torch.Tensor, target: torch.Tensor, chunk: int = 8192):
    """query [n,3], target [m,3] -> (dist [n] squared, idx [n] int64); direct-difference form,
    accumulation order x, y, z; ties -> lowest index."""
    n = query.shape[0]
    m = target.shape[0]
    best = torch.full((n,), float("inf"), dtype=query.dtype)
    best_i = torch.zeros((n,), dtype=torch.long)
    for s in range(0, m, chunk):
        t = target[s:s + chunk]
        dx = query[:, None, 0] - t[None, :, 0]
        dy = query[:, None, 1] - t[None, :, 1]
        dz = query[:, None, 2] - t[None, :, 2]
        d = dx * dx + dy * dy + dz * dz
        i = torch.argmin(d, dim=1)
        v = d.gather(1, i[:, None])[:, 0]
        upd = v < best
        best = torch.where(upd, v, best)
        best_i = torch.where(upd, i + s, best_i)
    return best, best_i


class _ChamferFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, one_direction):
        B, n, _ = xyz1.shape
        m = xyz2.shape[1]
        dist1 = torch.empty(B, n, dtype=xyz1.dtype)
        idx1 = torch.empty(B, n, dtype=torch.long)
        dist2 = torch.zeros(B, m, dtype=xyz1.dtype)
        idx2 = torch.zeros(B, m, dtype=torch.long)
        for b in range(B):
            dist1[b], idx1[b] = nn_direct(xyz1[b], xyz2[b])
            if not one_direction:
                dist2[b], idx2[b] = nn_direct(xyz2[b], xyz1[b])
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.one_direction = one_direction
        return dist1, dist2

    @staticmethod
    def backward(ctx, g1, g2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        grad1 = torch.zeros_like(xyz1)
        grad2 = torch.zeros_like(xyz2)
        B = xyz1.shape[0]
        for b in range(B):
            nb = xyz2[b][idx1[b]]
            t = 2.0 * g1[b][:, None] * (xyz1[b] - nb)
            grad1[b] += t
            grad2[b].index_add_(0, idx1[b], -t)
            if not ctx.one_direction:
                na = xyz1[b][idx2[b]]
                u = 2.0 * g2[b][:, None] * (xyz2[b] - na)
                grad2[b] += u
                grad1[b].index_add_(0, idx2[b], -u)
        return grad1, grad2, None


class chamferDist(torch.nn.Module):
    """`chamferDist()(xyz1[B,n,3], xyz2[B,m,3]) -> (dist1[B,n], dist2[B,m])`, squared."""

    def __init__(self, one_direction: bool = False):
        super().__init__()
        self.one_direction = one_direction

    def forward(self, input1, input2):
        return _ChamferFunction.apply(input1.contiguous(), input2, self.one_direction)


def pairwise_dist(x, y):
    """chamfer_python.py:4-9 (expansion form, equal sizes only)."""
    xx, yy, zz = torch.mm(x, x.t()), torch.mm(y, y.t()), torch.mm(x, y.t())
    rx = xx.diag().unsqueeze(0).expand_as(xx)
    ry = yy.diag().unsqueeze(0).expand_as(yy)
    return rx.t() + ry - 2 * zz


def NN_loss(x, y, dim=0):
    """chamfer_python.py:12-15."""
    dist = pairwise_dist(x, y)
    values, _ = dist.min(dim=dim)
    return values.mean()


def distChamfer(a, b):
    """chamfer_python.py:18-28, batched expansion form, equal sizes only (x's point count indexes both diagonals, :24-26).
    Return order as the reference's (:28): min over the x index first -- per-y distances [bs,Ny] -- then per-x [bs,Nx],
    then the two argmins: (y->x, x->y, idx of x per y, idx of y per x), the OPPOSITE of the CUDA extension's
    (dist1 = x->y, dist2 = y->x).  Device-free (the reference's `torch.cuda.LongTensor` only builds the diagonal index)."""
    x, y = a, b
    bs, num_points, _ = x.size()
    xx = torch.bmm(x, x.transpose(2, 1))
    yy = torch.bmm(y, y.transpose(2, 1))
    zz = torch.bmm(x, y.transpose(2, 1))
    diag_ind = torch.arange(0, num_points)
    rx = xx[:, diag_ind, diag_ind].unsqueeze(1).expand_as(xx)
    ry = yy[:, diag_ind, diag_ind].unsqueeze(1).expand_as(yy)
    P = rx.transpose(2, 1) + ry - 2 * zz
    m1, m2 = torch.min(P, 1), torch.min(P, 2)
    return m1[0], m2[0], m1[1], m2[1]
