"""B = 32 end-to-end record of the imported reference: the BENCHMARKED batch.

The B = 4 records of e2e.npz run a different kernel mix on the GPU (``gssd/ops.py::x6_wanted`` needs M >= 4096: at B = 4 conv6 / conv7 /
fuse_21 / the 19 x 19 Self_Attn GEMMs stay on the fp32-MFMA kernels); this file ties the plan bench.py times -- GSSD and GSSD++ at
batch 32 -- to the reference itself (models/ssd_multiphase_custom_group.py:217-400, layers/modules/multibox_loss.py:46-120), not only to
the oracle.  Same weights / images as tests/test_gpu_parity.py::test_full_size_parity's first seed pair (weights 1111, images 11).

Runs ONLY in the build container (imports /root/reference through make_golden.import_reference()); ~1-2 min of CPU.

    python tests/golden/make_golden_b32.py       # rewrites tests/golden/e2e_b32.npz  (< 100 KB)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                      # noqa: E402  (puts the repo + package on sys.path, imports oracle + synth)

B, WSEED, XSEED = 32, 1111, 11
NETS = {
    'gssd': (True, 4, 4, 1, True, False, False, 0, 1, False, False, 1),
    'gssdpp': (True, 4, 4, 1, True, True, True, 1, 4, True, False, 1),
}


def main():
    torch.manual_seed(0)
    R = G.import_reference()
    x = G.synth.synth_images(B, seed=XSEED)
    tg = G.synth.synth_targets(B, seed=XSEED)
    crit = R.MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, False)
    ed = dict(batch=np.int64(B), wseed=np.int64(WSEED), xseed=np.int64(XSEED))
    for name, args in NETS.items():
        net = R.mg.build_ssd('train', 300, 2, *args)
        shapes = {k: v.shape for k, v in net.state_dict().items()}
        net.load_state_dict(G.synth.synth_state_dict(shapes, seed=WSEED))
        net.train()
        with torch.no_grad():
            loc, conf, pr = net(x)
            ll, lc = crit((loc, conf, pr), tg)
        after = net.state_dict()
        loc, conf = loc.numpy(), conf.numpy()
        ed[f'{name}.loc_sha'] = np.frombuffer(bytes.fromhex(G.sha(loc)), dtype=np.uint8)
        ed[f'{name}.conf_sha'] = np.frombuffer(bytes.fromhex(G.sha(conf)), dtype=np.uint8)
        si = G.sample_idx(loc.size, 2048, seed=9)
        ed[f'{name}.loc_idx'], ed[f'{name}.loc_val'] = si, loc.reshape(-1)[si]
        si = G.sample_idx(conf.size, 2048, seed=10)
        ed[f'{name}.conf_idx'], ed[f'{name}.conf_val'] = si, conf.reshape(-1)[si]
        # every image's four priors of the 1 x 1 map (8728 .. 8731): the worst-conditioned outputs of the graph
        ed[f'{name}.loc_1x1'], ed[f'{name}.conf_1x1'] = loc[:, 8728:].copy(), conf[:, 8728:].copy()
        ed[f'{name}.loss'] = np.array([ll.item(), lc.item()], np.float64)
        ed[f'{name}.loc_absmax'] = np.float64(np.abs(loc).max())
        ed[f'{name}.conf_absmax'] = np.float64(np.abs(conf).max())
        for k in ('vgg.1.running_mean', 'vgg.41.running_var', 'bn_fuse_11.running_mean', 'extras.15.running_var'):
            ed[f'{name}.after.{k}'] = after[k].numpy().copy()
        print(name, 'B', B, 'loss', ed[f'{name}.loss'], 'absmax', ed[f'{name}.loc_absmax'], ed[f'{name}.conf_absmax'])
    np.savez_compressed(os.path.join(HERE, 'e2e_b32.npz'), **ed)


if __name__ == '__main__':
    main()
