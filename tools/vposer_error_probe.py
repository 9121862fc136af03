"""Measured error of the fused VPoser decoder (forward, and latent gradient through ops.VPoser's autograd) against a float64 evaluation
of the same network; FDCAP_LIB / FDCAP_GEMM_SPLIT3 select the build / the form."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdcap_amd  # noqa: F401,E402
from fdcap_amd import capi, ops, synth  # noqa: E402


def main():
    vp = synth.make_vposer(seed=11)
    bm = synth.make_body_model(300, seed=0)
    ctx = capi.Context(bm, vp)
    rng = np.random.default_rng(3)
    for name, zs in (("z ~ N(0,1)", 1.0), ("z ~ 1e-4 N(0,1)", 1e-4), ("z ~ 30 N(0,1)", 30.0)):
        z = (rng.standard_normal((1024, 32)) * zs).astype(np.float32)
        w1 = torch.tensor(vp.fc1_w, dtype=torch.float64); b1 = torch.tensor(vp.fc1_b, dtype=torch.float64)
        w2 = torch.tensor(vp.fc2_w, dtype=torch.float64); b2 = torch.tensor(vp.fc2_b, dtype=torch.float64)
        w3 = torch.tensor(vp.out_w, dtype=torch.float64); b3 = torch.tensor(vp.out_b, dtype=torch.float64)
        z64 = torch.tensor(z, dtype=torch.float64, requires_grad=True)
        h1 = torch.nn.functional.leaky_relu(z64 @ w1.T + b1, 0.2)
        h2 = torch.nn.functional.leaky_relu(h1 @ w2.T + b2, 0.2)
        x = (h2 @ w3.T + b3).reshape(-1, 3, 2)                       # the continuous 6D representation -> rotation matrices (Gram-Schmidt)
        c0 = torch.nn.functional.normalize(x[:, :, 0], dim=1)
        c1 = torch.nn.functional.normalize(x[:, :, 1] - (c0 * x[:, :, 1]).sum(1, keepdim=True) * c0, dim=1)
        o64 = torch.stack([c0, c1, torch.cross(c0, c1, dim=1)], dim=-1).reshape(z.shape[0], -1)
        gw = torch.tensor(rng.standard_normal(o64.shape))
        (o64 * gw).sum().backward()
        zd = torch.tensor(z).cuda().requires_grad_(True)
        od = ops.VPoser(ctx).decode(zd, "matrot").reshape(z.shape[0], -1)
        (od * gw.float().cuda()).sum().backward()
        eo = (od.detach().cpu().double() - o64.detach()).abs().max().item() / o64.detach().abs().max().item()
        eg = (zd.grad.cpu().double() - z64.grad).abs().max().item() / z64.grad.abs().max().item()
        print("%-18s forward max err / max|out| %.2e   latent gradient max err / max|grad| %.2e" % (name, eo, eg))
    ctx.close()


if __name__ == "__main__":
    main()
