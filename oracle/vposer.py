"""Restatement of the VPoser v1.0 decoder (human_body_prior, `load_vposer(..., 'snapshot')`).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The package is absent; call site
/root/reference/global_optimization.py:153 (load) and :270-271 (decode).  Algorithm: SURVEY.md
Appendix A.2.  Parity unpinned against the real package.
"""
import torch
import torch.nn.functional as F

from . import rotrepr


class VPoserDecoder(torch.nn.Module):
    """Same call surface the reference uses: `.decode(z, output_type='aa') -> [B,1,21,3]`."""

    def __init__(self, fc1_w, fc1_b, fc2_w, fc2_b, out_w, out_b, dtype=torch.float32):
        super().__init__()
        t = lambda a: torch.as_tensor(a).to(dtype).clone()
        self.register_buffer("fc1_w", t(fc1_w)); self.register_buffer("fc1_b", t(fc1_b))
        self.register_buffer("fc2_w", t(fc2_w)); self.register_buffer("fc2_b", t(fc2_b))
        self.register_buffer("out_w", t(out_w)); self.register_buffer("out_b", t(out_b))
        self.eval()

    @classmethod
    def from_data(cls, vp, dtype=torch.float32):
        return cls(vp.fc1_w, vp.fc1_b, vp.fc2_w, vp.fc2_b, vp.out_w, vp.out_b, dtype=dtype)

    def decode_matrot(self, z: torch.Tensor) -> torch.Tensor:
        h = F.leaky_relu(F.linear(z, self.fc1_w, self.fc1_b), negative_slope=0.2)
        # dropout(p=.25) is the identity in eval mode (load_vposer calls .eval())
        h = F.leaky_relu(F.linear(h, self.fc2_w, self.fc2_b), negative_slope=0.2)
        o = F.linear(h, self.out_w, self.out_b)                       # [B,126]
        return rotrepr.decode_6d(o).view(-1, 1, 21, 9)

    def decode(self, z: torch.Tensor, output_type: str = "matrot") -> torch.Tensor:
        m = self.decode_matrot(z)
        if output_type == "aa":
            b = m.shape[0]
            return rotrepr.matrot2aa(m).view(b, 1, -1, 3).contiguous()
        return m
