"""Generates tests/golden/pixellink.npz from the IMPORTED reference (run in the build container only; /root/reference does not
travel):  python tests/golden/make_pixellink_golden.py

  model_*   PixelLink(cascade_fuse, fuseconv, BN, SA, SA-base, 1 DCN layer, dcn_cat_sab) forward (train mode, B = 1) on the seeded
            synthetic weights / image of gssd.synth -- reference ``pixel_link/model.py`` with cv2 as an empty import stub (not on a
            numeric path) and dcn_v2 as the oracle's DCN restatement (wiring pinned, DCN arithmetic unpinned -- as for GSSD++);
            plus the no-cascade / no-DCN variant.
  loss_*    ``criterion.PixelLinkLoss`` on seeded logits / masks (includes an image with zero positive area -> the 10000 branch).
  dec_*     ``postprocess.func`` label maps on seeded logits (thresholded exactly like mask_to_box).
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                     # noqa: E402  (import_reference: stubs + sys.path handling)
from oracle import gssd_oracle as O          # noqa: E402,F401
from gssd import synth                       # noqa: E402

warnings.filterwarnings('ignore')


def loss_inputs(seed, B=3, H=75):
    rng = np.random.default_rng(seed)
    out_1 = rng.normal(0, 2.0, size=(B, 2, H, H)).astype(np.float32)
    out_2 = rng.normal(0, 2.0, size=(B, 16, H, H)).astype(np.float32)
    pix = np.zeros((B, H, H), np.int64)
    for b in range(B - 1):                                   # the last image has no positive pixel (r_pos_area == 0 branch)
        for _ in range(3 + b):
            y, x, h, w = rng.integers(5, 55), rng.integers(5, 55), rng.integers(3, 14), rng.integers(3, 14)
            pix[b, y:y + h, x:x + w] = 1
    neg = ((pix == 0) & (rng.random((B, H, H)) > 0.1)).astype(np.uint8)
    posw = (pix * rng.uniform(0.5, 2.0, size=(B, H, H))).astype(np.float32)
    link = (rng.random((B, 8, H, H)) > 0.4).astype(np.int64) * pix[:, None]
    return out_1, out_2, pix, neg, posw, link


def main():
    R = MG.import_reference()
    from pixel_link import model as ref_model, criterion as ref_crit, postprocess as ref_post
    out = {}
    # ---- model ---------------------------------------------------------------------------------------
    for tag, kw in (('full', dict(cascade_fuse=True, use_fuseconv=True, batch_norm=True, use_self_attention=True,
                                  use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True, detach_sab=False)),
                    ('plain', dict(cascade_fuse=False, use_fuseconv=True, batch_norm=True, use_self_attention=False,
                                   use_self_attention_base=False, num_dcn_layers=0, groups_dcn=1, dcn_cat_sab=False, detach_sab=False))):
        torch.manual_seed(7)
        net = ref_model.PixelLink(**kw)
        sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=2222)
        net.load_state_dict(sd)
        net.train()
        x = synth.synth_images(1, seed=300)
        with torch.no_grad():
            o1, o2 = net(x)
        out[f'model_{tag}_out1'], out[f'model_{tag}_out2'] = o1.numpy(), o2.numpy()
        after = net.state_dict()
        out[f'model_{tag}_keys'] = np.array([f'{k}:{"x".join(map(str, v.shape))}' for k, v in after.items()])
        out[f'model_{tag}_bn_fuse3_rm'] = after['bn_fuse3.running_mean'].numpy()
        if kw['use_self_attention']:
            out[f'model_{tag}_sa0_u'] = after['self_attn_list.0.snconv1x1_theta.weight_u'].numpy()
        print(tag, o1.shape, o2.shape, float(o1.abs().max()), float(o2.abs().max()))
    # ---- criterion -----------------------------------------------------------------------------------
    o1, o2, pix, neg, posw, link = loss_inputs(11)
    crit = ref_crit.PixelLinkLoss()
    pp, pn = crit.pixel_loss(torch.from_numpy(o1), torch.from_numpy(pix), torch.from_numpy(neg), torch.from_numpy(posw))
    lp, ln = crit.link_loss(torch.from_numpy(o2), torch.from_numpy(link))
    out.update(loss_out1=o1, loss_out2=o2, loss_pix=pix, loss_neg=neg, loss_posw=posw, loss_link=link,
               loss_vals=np.array([float(pp), float(pn), float(lp), float(ln)], np.float64),
               loss_neg_weight=crit.neg_pixel_weight.numpy().astype(np.uint8), loss_neg_area=crit.neg_area.numpy())
    print('loss', out['loss_vals'], out['loss_neg_area'])
    # ---- decoding ------------------------------------------------------------------------------------
    rng = np.random.default_rng(21)
    B, H = 3, 75
    d1 = rng.normal(0, 1.0, size=(B, 2, H, H)).astype(np.float32)
    d1[:, 1] += np.where(rng.random((B, H, H)) > 0.55, 2.0, -2.0).astype(np.float32)     # blobs of positives
    d2 = rng.normal(0, 2.5, size=(B, 16, H, H)).astype(np.float32)
    t1, t2 = torch.from_numpy(d1), torch.from_numpy(d2)
    pixc = torch.softmax(t1, 1)[:, 1] > 0.2
    labels = []
    for b in range(B):
        ln_ = torch.stack([(torch.softmax(t2[b:b + 1, 2 * n:2 * n + 2], 1)[0, 1] > 0.8) & pixc[b] for n in range(8)]).to(torch.uint8)
        labels.append(ref_post.func(pixc[b].to(torch.uint8), ln_).astype(np.int32))      # NB: uint8 in the reference (wraps at 256)
    out.update(dec_out1=d1, dec_out2=d2, dec_labels=np.stack(labels))
    print('decode components', [int(l.max()) for l in labels])
    np.savez_compressed(os.path.join(HERE, 'pixellink.npz'), **out)


if __name__ == '__main__':
    main()
