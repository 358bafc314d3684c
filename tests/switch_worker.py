"""Subprocess of tests/test_gpu_switches.py: one GSSD++ training step (forward + MultiBoxLoss + backward) at batch 4 (or argv[2]: the
size-gated kernels -- conv_wino_x6, conv_x6 on the 19 x 19 maps, the 256 x 128 bf16 tiles -- take launches only at larger batches) in the dtype
given on the command line, under whatever GSSD_* ablation switches the parent put into the environment (they are read once per
process).  Prints one JSON line: sampled outputs, losses, gradient norms and samples, and which kernel instances the plan launched."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')):
    sys.path.insert(0, p)
import numpy as np          # noqa: E402
import torch                # noqa: E402


def main():
    from gssd import synth
    from layers.modules import MultiBoxLoss
    from models.ssd_multiphase_custom_group import build_ssd
    dtype = sys.argv[1]
    dev = torch.device('cuda:0')
    net = build_ssd('train', 300, 2, True, 4, 4, 1, True, True, True, 1, 4, True, False, 1)
    net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
    net = net.to(dev).train()
    net.compute_dtype = dtype
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    x = synth.synth_images(B, seed=5).to(dev)
    tg = synth.synth_targets(B, seed=5)
    crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
    with torch.no_grad():
        for _ in range(3):                      # the third forward replays from hipGraphs (unless GSSD_NO_GRAPH)
            net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
            loc0, conf0, _ = net(x)
    graphs = len(getattr(net._engine._last_plan, '_graphs', {}) or {})      # the no-backward plan of the three forwards above
    kernels0 = sorted({st.tag[0] for st in net._engine._last_plan.steps if st.tag is not None})
    # GSSD_BWD_GRAPH=1: the backward plan replays from hipGraphs from its third run on -- three training steps from the same state
    for _ in range(3 if os.environ.get('GSSD_BWD_GRAPH') == '1' else 1):
        net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
        net.zero_grad(set_to_none=True)
        loc, conf, pri = net(x)
        ll, lc = crit((loc, conf, pri), tg)
        (ll + lc).backward()
    torch.cuda.synchronize()
    bwd_graphs = len(getattr(net._engine._last_plan._bwd, '_graphs', {}) or {})
    bwd_fns = sorted({getattr(fn, '__name__', '') for fn, _ in net._engine._last_plan._bwd.steps} - {''})     # C-ABI entry points of the backward plan
    idx = np.random.default_rng(0).integers(0, loc.numel(), 256)
    named = dict(net.named_parameters())
    keys = ['vgg.0.weight', 'vgg.14.weight', 'vgg.31.weight', 'fuse_11.weight', 'loc.0.weight', 'dcn_list.0.weight',
            'self_attn_list.0.snconv1x1_g.weight_orig', 'extras.2.weight']
    plan = net._engine._last_plan
    out = dict(loc=loc.detach().reshape(-1)[idx].cpu().tolist(), loc_max=float(loc.abs().max()),
               graph_replay=float((loc0 - loc.detach()).abs().max()),
               loss=[float(ll), float(lc)],
               gnorm={k: float(named[k].grad.norm()) for k in keys},
               gsample={k: named[k].grad.reshape(-1)[:64].cpu().tolist() for k in keys},
               kernels=sorted({st.tag[0] for st in plan.steps if st.tag is not None} | set(kernels0)), graphs=graphs, bwd_graphs=bwd_graphs, bwd_fns=bwd_fns)
    print('SWITCHJSON ' + json.dumps(out))


if __name__ == '__main__':
    main()
