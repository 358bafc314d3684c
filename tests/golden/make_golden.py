"""Generate the golden fixtures in this directory by importing the reference.

Runs ONLY in the build container (``/root/reference`` does not exist on the GPU box).  The
reference's Python is imported from where it lies, with empty stubs for the three modules that
are not installed and not on a numeric path (``cv2``, ``torchvision[.transforms]``) and a
``dcn_v2`` stub whose ``_DCNv2.apply`` is the oracle's restatement (DCN parity is unpinned --
see ``oracle/gssd_oracle.py``).  Nothing from the reference is copied: the fixtures are inputs
and the outputs the reference computed for them.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/ssd_liverdet'
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))

from oracle import gssd_oracle as O          # noqa: E402
from gssd import synth                       # noqa: E402


def import_reference():
    for name in ('cv2', 'torchvision', 'torchvision.transforms', 'tensorboardX'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    dcn_stub = types.ModuleType('dcn_v2')

    class _DCNv2:
        @staticmethod
        def apply(inp, offset, mask, weight, bias, stride, padding, dilation, dg):
            return O.dcn_v2_conv(inp, offset, mask, weight, bias, stride[0], padding[0], dilation[0], dg)
    dcn_stub._DCNv2 = _DCNv2
    sys.modules['dcn_v2'] = dcn_stub
    # our own drop-in packages share the names models/layers/data: make sure the reference wins here
    for k in [k for k in sys.modules if k.split('.')[0] in ('models', 'layers', 'data', 'utils')]:
        del sys.modules[k]
    # the reference's `models` is a namespace package (no __init__.py) and would lose to ours: drop our path
    pkg = os.path.join(ROOT, 'grouped-ssd-pytorch_amd')
    sys.path[:] = [p for p in sys.path if os.path.abspath(p) != pkg]
    sys.path.insert(0, REF)
    import layers                                            # noqa: F401
    from layers import box_utils
    from layers.functions import Detect, PriorBox
    from layers.modules import MultiBoxLoss, L2Norm
    from layers.self_attn import Self_Attn
    from layers.dcn_v2_custom import DCN
    from models import ssd_multiphase_custom_group as mg
    from models import ssd as mv
    import data
    return types.SimpleNamespace(box_utils=box_utils, Detect=Detect, PriorBox=PriorBox, MultiBoxLoss=MultiBoxLoss,
                                 L2Norm=L2Norm, Self_Attn=Self_Attn, DCN=DCN, mg=mg, mv=mv, data=data)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sample_idx(n, count=512, seed=3):
    return np.random.default_rng(seed).choice(n, size=min(count, n), replace=False).astype(np.int64)


def closed_form_detect_inputs(P, B=2):
    """Deterministic loc / logits from closed forms."""
    i = np.arange(P, dtype=np.float64)
    loc, logit = [], []
    for b in range(B):
        s = 1.0 + 0.37 * b
        loc.append(np.stack([np.sin(.37 * s * i), np.cos(.11 * s * i), .5 * np.sin(.05 * s * i),
                             .5 * np.cos(.23 * s * i)], 1))
        logit.append(np.stack([np.zeros(P), 4 * np.sin(.7 * s * i) - 3], 1))
    return np.stack(loc).astype(np.float32), np.stack(logit).astype(np.float32)


def random_targets(rng, n):
    cxy = rng.uniform(0.15, 0.85, size=(n, 2))
    wh = rng.uniform(0.02, 0.4, size=(n, 2))
    box = np.concatenate([cxy - wh / 2, cxy + wh / 2], 1).clip(0, 1)
    return np.concatenate([box, np.zeros((n, 1))], 1).astype(np.float32)


EB = 4   # batch of the end-to-end fixtures: >= 4 keeps the 1x1-map train-mode BatchNorm well conditioned


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = import_reference()
    out = {}

    # ---- priors (a11) ------------------------------------------------------------------
    pri = {}
    for name in ('v2', 'v2_512'):
        p = R.PriorBox(getattr(R.data, name)).forward().numpy()
        pri[name] = p
        print(name, p.shape, sha(p)[:16], float(p.astype(np.float64).sum()))
    np.savez_compressed(os.path.join(HERE, 'priors.npz'), v2=pri['v2'],
                        v2_512_sha=np.frombuffer(bytes.fromhex(sha(pri['v2_512'])), dtype=np.uint8),
                        v2_512_sample_idx=sample_idx(pri['v2_512'].shape[0]),
                        v2_512_sample=pri['v2_512'][sample_idx(pri['v2_512'].shape[0])])
    priors = torch.from_numpy(pri['v2'])
    P = priors.shape[0]

    # ---- match (a12) ----------------------------------------------------------------------
    rng = np.random.default_rng(42)
    cases = [np.array([[.1, .1, .4, .5, 0]], np.float32),
             np.array([[.3, .3, .6, .7, 0], [.5, .5, .9, .9, 0]], np.float32),
             # two GTs sharing their best prior (identical boxes): later GT wins (box_utils.py:100-101)
             np.array([[.2, .2, .5, .5, 0], [.2, .2, .5, .5, 0], [.6, .1, .9, .3, 0]], np.float32),
             # degenerate tiny boxes
             np.array([[.5, .5, .5004, .5004, 0], [.01, .01, .015, .02, 0]], np.float32)]
    cases += [random_targets(rng, int(rng.integers(1, 6))) for _ in range(8)]
    md = {}
    for ci, t in enumerate(cases):
        loc_t = torch.zeros(1, P, 4)
        conf_t = torch.zeros(1, P, dtype=torch.long)
        tt = torch.from_numpy(t)
        R.box_utils.match(0.5, tt[:, :-1], priors, [0.1, 0.2], tt[:, -1], loc_t, conf_t, 0)
        md[f't{ci}'] = t
        md[f'conf{ci}'] = conf_t[0].numpy().astype(np.int8)
        pos = conf_t[0].numpy() > 0
        md[f'locpos{ci}'] = loc_t[0].numpy()[pos]
    md['n'] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, 'match.npz'), **md)
    print('match: n_pos', [int((md[f"conf{i}"] > 0).sum()) for i in range(len(cases))])

    # ---- MultiBoxLoss (a13) ---------------------------------------------------------------
    crit = R.MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, False)
    ld = {}
    targets2 = [torch.from_numpy(cases[0]), torch.from_numpy(cases[1])]
    l0, c0 = crit((torch.zeros(2, P, 4), torch.zeros(2, P, 2), priors), targets2)
    ld['zero_loss'] = np.array([l0.item(), c0.item()], np.float64)
    print('zero-input loss', ld['zero_loss'])
    for ci in range(3):
        B = 4
        g = np.random.default_rng(100 + ci)
        loc = g.normal(0, 1.0, size=(B, P, 4)).astype(np.float32)
        conf = g.normal(0, 2.0, size=(B, P, 2)).astype(np.float32)
        tg = [random_targets(g, int(g.integers(1, 5))) for _ in range(B)]
        loc_v = torch.from_numpy(loc).requires_grad_()
        conf_v = torch.from_numpy(conf).requires_grad_()
        ll, lc = crit((loc_v, conf_v, priors), [torch.from_numpy(t) for t in tg])
        (ll + lc).backward()
        # recompute the reference's neg mask with its own ops (multibox_loss.py:91-106)
        conf_t = torch.zeros(B, P, dtype=torch.long)
        loc_t = torch.zeros(B, P, 4)
        for b in range(B):
            tt = torch.from_numpy(tg[b])
            R.box_utils.match(0.5, tt[:, :-1], priors, [0.1, 0.2], tt[:, -1], loc_t, conf_t, b)
        bc = torch.from_numpy(conf).view(-1, 2)
        lca = R.box_utils.log_sum_exp(bc) - bc.gather(1, conf_t.view(-1, 1))
        pos = conf_t > 0
        lca[pos.view(-1, 1)] = 0
        lca = lca.view(B, -1)
        _, li = lca.sort(1, descending=True)
        _, rk = li.sort(1)
        nneg = torch.clamp(3 * pos.long().sum(1, keepdim=True), max=P - 1)
        neg = rk < nneg
        ld[f'seed{ci}'] = np.int64(100 + ci)
        for b in range(B):
            ld[f'tg{ci}_{b}'] = tg[b]
        ld[f'loss{ci}'] = np.array([ll.item(), lc.item()], np.float64)
        ld[f'neg{ci}'] = np.packbits(neg.numpy())
        ld[f'pos{ci}'] = np.packbits(pos.numpy())
        ld[f'lca_sample{ci}'] = lca.numpy().reshape(-1)[sample_idx(B * P)]
        ld[f'gloc_sha{ci}'] = np.frombuffer(bytes.fromhex(sha(loc_v.grad.numpy())), dtype=np.uint8)
        ld[f'gloc_sample{ci}'] = loc_v.grad.numpy().reshape(-1)[sample_idx(B * P * 4)]
        ld[f'gconf_sample{ci}'] = conf_v.grad.numpy().reshape(-1)[sample_idx(B * P * 2)]
        print('loss case', ci, ld[f'loss{ci}'], 'n_neg', int(neg.sum()))
    np.savez_compressed(os.path.join(HERE, 'loss.npz'), **ld)

    # ---- Detect / NMS (a14, a15) ----------------------------------------------------------
    loc, logit = closed_form_detect_inputs(P, 2)
    conf_sm = torch.softmax(torch.from_numpy(logit), dim=-1)
    det = R.Detect.apply(2, 0, 200, 0.01, 0.45, torch.from_numpy(loc), conf_sm, priors).numpy()
    kept = [(det[b, 1, :, 0] > 0).sum() for b in range(2)]
    print('detect kept', kept, sha(det)[:16])
    # kept prior indices: redo nms per image through the reference's own functions
    keep_idx = []
    for b in range(2):
        boxes = R.box_utils.decode(torch.from_numpy(loc[b]), priors, [0.1, 0.2])
        sc = conf_sm[b, :, 1]
        m = sc.gt(0.01)
        ids, cnt = R.box_utils.nms(boxes[m.unsqueeze(1).expand_as(boxes)].view(-1, 4), sc[m], 0.45, 200)
        keep_idx.append(torch.nonzero(m).view(-1)[ids[:cnt]].numpy())
    dd = dict(out=det, keep0=keep_idx[0], keep1=keep_idx[1], conf_sm=conf_sm.numpy())
    # a random-logit case where far more than 200 candidates pass the threshold
    g = np.random.default_rng(5)
    loc_r = g.normal(0, 0.5, size=(2, P, 4)).astype(np.float32)
    logit_r = g.normal(0, 1.5, size=(2, P, 2)).astype(np.float32)
    sm_r = torch.softmax(torch.from_numpy(logit_r), -1)
    det_r = R.Detect.apply(2, 0, 200, 0.01, 0.45, torch.from_numpy(loc_r), sm_r, priors).numpy()
    dd['out_rand'] = det_r
    dd['conf_sm_rand'] = sm_r.numpy()
    print('detect rand kept', [(det_r[b, 1, :, 0] > 0).sum() for b in range(2)])
    np.savez_compressed(os.path.join(HERE, 'detect.npz'), **dd)

    # ---- small ops: L2Norm, Self_Attn, DCN wrapper, slice_and_cat ---------------------------
    od = {}
    g = np.random.default_rng(11)
    x = torch.from_numpy(g.normal(0, 1, size=(2, 32, 7, 7)).astype(np.float32))
    l2 = R.L2Norm(32, 20)
    l2.weight.data = torch.from_numpy(g.uniform(15, 25, size=32).astype(np.float32))
    od['l2_x'], od['l2_w'], od['l2_y'] = x.numpy(), l2.weight.data.numpy(), l2(x).detach().numpy()
    for mode in ('eval', 'train'):
        sa = R.Self_Attn(64)
        shapes = {k: v.shape for k, v in sa.state_dict().items()}
        sd = synth.synth_state_dict(shapes, seed=21)
        sa.load_state_dict(sd)
        sa.train(mode == 'train')
        xs = torch.from_numpy(g.normal(0, 1, size=(2, 64, 6, 6)).astype(np.float32))
        o, ag, at = sa(xs, True)
        od[f'sa_{mode}_x'] = xs.numpy()
        od[f'sa_{mode}_out'], od[f'sa_{mode}_ag'], od[f'sa_{mode}_attn'] = (o.detach().numpy(), ag.detach().numpy(),
                                                                         at.detach().numpy())
        if mode == 'train':
            for k, v in sa.state_dict().items():
                if k.endswith('_u') or k.endswith('_v'):
                    od[f'sa_train_after.{k}'] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'ops.npz'), **od)

    # ---- end-to-end nets --------------------------------------------------------------------
    nets = {
        'gssd': dict(args=(True, 4, 4, 1, True, False, False, 0, 1, False, False, 1)),
        'gssd_sa': dict(args=(True, 4, 4, 1, True, True, True, 0, 1, False, False, 1)),
        'gssdpp': dict(args=(True, 4, 4, 1, True, True, True, 1, 4, True, False, 1)),
    }
    x = synth.synth_images(EB, seed=5)
    tg = synth.synth_targets(EB, seed=5)
    ed = {}
    for name, spec in nets.items():
        net = R.mg.build_ssd('train', 300, 2, *spec['args'])
        shapes = {k: v.shape for k, v in net.state_dict().items()}
        sd = synth.synth_state_dict(shapes, seed=1111)
        net.load_state_dict(sd)
        net.train()
        with torch.no_grad():
            loc, conf, pr = net(x)
            ll, lc = crit((loc, conf, pr), tg)
        after = {k: v.clone() for k, v in net.state_dict().items()}
        ed[f'{name}.keys'] = np.array(sorted(shapes.keys()))
        ed[f'{name}.shapes'] = np.array([str(tuple(shapes[k])) for k in sorted(shapes.keys())])
        ed[f'{name}.loc_sha'] = np.frombuffer(bytes.fromhex(sha(loc.numpy())), dtype=np.uint8)
        si = sample_idx(loc.numel(), 2048, seed=9)
        ed[f'{name}.loc_idx'], ed[f'{name}.loc_val'] = si, loc.numpy().reshape(-1)[si]
        si = sample_idx(conf.numel(), 2048, seed=10)
        ed[f'{name}.conf_idx'], ed[f'{name}.conf_val'] = si, conf.numpy().reshape(-1)[si]
        ed[f'{name}.loss'] = np.array([ll.item(), lc.item()], np.float64)
        ed[f'{name}.loc_absmax'] = np.float64(loc.abs().max().item())
        ed[f'{name}.conf_absmax'] = np.float64(conf.abs().max().item())
        for k in ('vgg.1.running_mean', 'vgg.1.running_var', 'bn_fuse_11.running_mean', 'extras.15.running_var'):
            ed[f'{name}.after.{k}'] = after[k].numpy().copy()
        if name != 'gssd':
            k = 'self_attn_list.0.snconv1x1_theta.weight_u'
            ed[f'{name}.after.{k}'] = after[k].numpy().copy()
        # test-phase twin shares the state dict (strict load) and runs Detect.  The synthetic running statistics
        # do not describe these activations (eval-mode BN would blow the scale up layer by layer), so first run one
        # more training forward with momentum 1.0: running stats := this batch's statistics.
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.momentum = 1.0
        with torch.no_grad():
            net(x)
        after = net.state_dict()
        net_t = R.mg.build_ssd('test', 300, 2, *spec['args'])
        net_t.load_state_dict(after)
        net_t.eval()
        with torch.no_grad():
            det = net_t(x)
        ed[f'{name}.det'] = det.numpy()
        print(name, 'params', sum(p.numel() for p in net.parameters()), 'loss', ed[f'{name}.loss'],
              'det kept', [(det[b, 1, :, 0] > 0).sum().item() for b in range(EB)])
    # vanilla SSD300 (BASELINE.json configs[0])
    net = R.mv.build_ssd('train', 300, 2)
    shapes = {k: v.shape for k, v in net.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=1111)
    net.load_state_dict(sd)
    xv = synth.synth_images(2, seed=6, channels=3)
    loc, conf, pr = net(xv)
    ll, lc = crit((loc, conf, pr), tg[:2])
    (ll + lc).backward()
    gfin = all(torch.isfinite(p.grad).all().item() for p in net.parameters() if p.grad is not None)
    ed['ssd.keys'] = np.array(sorted(shapes.keys()))
    ed['ssd.shapes'] = np.array([str(tuple(shapes[k])) for k in sorted(shapes.keys())])
    si = sample_idx(loc.numel(), 2048, seed=9)
    ed['ssd.loc_idx'], ed['ssd.loc_val'] = si, loc.detach().numpy().reshape(-1)[si]
    si = sample_idx(conf.numel(), 2048, seed=10)
    ed['ssd.conf_idx'], ed['ssd.conf_val'] = si, conf.detach().numpy().reshape(-1)[si]
    ed['ssd.loss'] = np.array([ll.item(), lc.item()], np.float64)
    ed['ssd.grad_finite'] = np.bool_(gfin)
    print('vanilla ssd loss', ed['ssd.loss'], 'grad finite', gfin)
    np.savez_compressed(os.path.join(HERE, 'e2e.npz'), **ed)


if __name__ == '__main__':
    main()
