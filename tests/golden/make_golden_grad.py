"""Gradient fixture of the imported reference's Self_Attn block (layers/self_attn.py:46-89 with layers/spectral_norm.py:39-89): pins the
GRADIENT semantics of the oracle's restatement -- in particular that spectral norm's power iteration runs under torch.no_grad() (u, v are
constants of the graph, sigma = u^T W v is differentiated through W only).  The forward fixtures cannot see that; the float64 gradients the
GPU gradient tests compare against depend on it (tests/test_gpu_grad.py found the restatement differentiating through the iteration).

Runs ONLY in the build container.    python tests/golden/make_golden_grad.py      # rewrites tests/golden/grad.npz  (< 100 KB)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                      # noqa: E402


def main():
    torch.manual_seed(0)
    R = G.import_reference()
    sa = R.Self_Attn(64)
    sd = G.synth.synth_state_dict({k: tuple(v.shape) for k, v in sa.state_dict().items()}, seed=21)
    sa.load_state_dict(sd)
    sa = sa.double().train()
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.normal(size=(2, 64, 6, 6))).requires_grad_()
    r = torch.from_numpy(rng.normal(size=(2, 64, 6, 6)))
    out = sa(x)
    out = out[0] if isinstance(out, tuple) else out
    (out * r).sum().backward()
    ed = dict(x=x.detach().numpy(), r=r.numpy(), dx=x.grad.numpy())
    for k, p in sa.named_parameters():
        ed['grad.' + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'grad.npz'), **ed)
    print({k: float(np.abs(v).max()) for k, v in ed.items() if k.startswith('grad.')})


if __name__ == '__main__':
    main()
