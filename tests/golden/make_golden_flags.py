"""Golden vectors for the driver-reachable NON-default constructor flags of ``build_ssd`` (train_lesion_multiphase_v2.py:49-77,
142-145): ``--use_fuseconv False``, ``--batch_norm False``, ``--max_pool_factor 2 / 3``, ``--feature_scale 2`` (every channel width doubled).

    python tests/golden/make_golden_flags.py     # rewrites tests/golden/flags.npz  (build container only: imports /root/reference)
    python tests/golden/make_golden_flags.py g1 g2pp g4e1     # computes only the named nets and merges them into flags.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, synth, sample_idx      # noqa: E402

# positional arguments of build_ssd after (phase, size, num_classes): batch_norm, groups_vgg, groups_extra, feature_scale,
# use_fuseconv, use_self_attention, use_self_attention_base, num_dcn_layers, groups_dcn, dcn_cat_sab, detach_sab, max_pool_factor
FLAG_NETS = {
    'nofuse': (True, 4, 4, 1, False, True, True, 1, 4, True, False, 1),
    'nobn': (False, 4, 4, 1, True, True, True, 1, 4, True, False, 1),
    'nobn_plain': (False, 4, 4, 1, False, False, False, 0, 1, False, False, 1),
    'mpf2': (True, 4, 4, 1, True, True, True, 1, 4, True, False, 2),
    'mpf3_sa': (True, 4, 4, 1, True, True, True, 0, 1, False, False, 3),
    'fs2': (True, 4, 4, 2, True, False, False, 0, 1, False, False, 1),
    'fs2pp': (True, 4, 4, 2, True, True, True, 1, 4, True, False, 1),           # round 4: GSSD++ with every channel width doubled
    'dcn2_detach': (True, 4, 4, 1, True, True, True, 2, 1, True, True, 1),      # two DCN layers, one deformable group, detached SAB
    'dcn_nocat': (True, 4, 4, 1, True, False, False, 1, 4, False, False, 1),     # DCN on the plain conv4_3 map (512 -> 512)
    # round 3: --groups_vgg / --groups_extra other than 4 (train_lesion_multiphase_v2.py:47-48)
    'g1': (True, 1, 1, 1, True, False, False, 0, 1, False, False, 1),            # ungrouped trunk and extras on the 12-channel input
    'g2pp': (True, 2, 2, 1, True, True, True, 1, 4, True, False, 1),             # GSSD++ with two groups (slice_and_cat over 2 groups)
    'g4e1': (True, 4, 1, 1, True, False, False, 0, 1, False, False, 1),          # grouped trunk, ungrouped extras
}
EB = 2


def main():
    R = import_reference()
    torch.manual_seed(0)
    crit = R.MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, False)
    x = synth.synth_images(EB, seed=5)
    tg = synth.synth_targets(EB, seed=5)
    d = {}
    only = sys.argv[1:]
    if only:                                     # keep every other entry of the existing file byte for byte
        with np.load(os.path.join(HERE, 'flags.npz'), allow_pickle=False) as old:
            d = {k: old[k] for k in old.files}
    for name, args in FLAG_NETS.items():
        if only and name not in only:
            continue
        net = R.mg.build_ssd('train', 300, 2, *args)
        shapes = {k: v.shape for k, v in net.state_dict().items()}
        sd = synth.synth_state_dict(shapes, seed=1111)
        net.load_state_dict(sd)
        net.train()
        with torch.no_grad():
            loc, conf, pr = net(x)
            ll, lc = crit((loc, conf, pr), tg)
        after = {k: v.clone() for k, v in net.state_dict().items()}
        d[f'{name}.keys'] = np.array(sorted(shapes.keys()))
        d[f'{name}.shapes'] = np.array([str(tuple(shapes[k])) for k in sorted(shapes.keys())])
        si = sample_idx(loc.numel(), 2048, seed=9)
        d[f'{name}.loc_idx'], d[f'{name}.loc_val'] = si, loc.numpy().reshape(-1)[si]
        si = sample_idx(conf.numel(), 2048, seed=10)
        d[f'{name}.conf_idx'], d[f'{name}.conf_val'] = si, conf.numpy().reshape(-1)[si]
        d[f'{name}.loss'] = np.array([ll.item(), lc.item()], np.float64)
        d[f'{name}.loc_absmax'] = np.float64(loc.abs().max().item())
        d[f'{name}.conf_absmax'] = np.float64(conf.abs().max().item())
        if args[0]:
            for k in ('vgg.1.running_mean', 'bn_fuse_11.running_var' if args[4] else 'vgg.31.running_var'):
                d[f'{name}.after.{k}'] = after[k].numpy().copy()
        if args[5]:
            k = 'self_attn_list.1.snconv1x1_phi.weight_u'
            d[f'{name}.after.{k}'] = after[k].numpy().copy()
        print(name, 'params', sum(p.numel() for p in net.parameters()), 'loss', d[f'{name}.loss'], 'absmax',
              d[f'{name}.loc_absmax'], d[f'{name}.conf_absmax'])
    np.savez_compressed(os.path.join(HERE, 'flags.npz'), **d)


if __name__ == '__main__':
    main()
