#!/usr/bin/env python3
"""bench.py -- images/s of forward + MultiBoxLoss (train-mode BatchNorm, no backward) of the GSSD++ detector on MI355X:
the metric BASELINE.json names, on its configs[2] workload (GSSD++: self-attention + self-attention-base + 1 deformable
conv layer with 4 deformable groups, 4-phase CT, batch 32 per GPU) -- the largest single-GPU configuration and the N = 1
point of configs[3]'s scaling curve.  configs[1] (plain GSSD) is measured the same way and reported as `secondary` (N = 1).

One process per GPU (torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE); the path shards by image with no
data-path collective, so N GPUs = N independent batch-32 shards ("scaling": "weak") and the only
communication is the timing barrier.  Inputs (resized 300x300 slices, targets, weights) are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0.

  value        : EXACTLY --steps timed steps between barrier + synchronize pairs, max over ranks.  `steady` repeats the
                 measurement over 100 more steps (>= 1 s of GPU time) so the short default region can be judged.
  roofline     : the dominant hand-written kernel instance (most time in two untimed survey passes that bracket every tagged
                 launch: that survey is the `kernels` breakdown).  achieved = algorithmic FLOPs per launch / average launch
                 duration, measured live with HIP events recorded on the launch stream around every launch of that
                 instance inside the timed region (only those: each event pair costs ~3 us of GPU idle).  fp32 runs are
                 bound by the fp32 matrix/vector peak (157.3 TFLOP/s), not HBM; bf16 runs by HBM.
                 traffic = HBM bytes per launch from the committed PMC passes (profiles/pmc_summary.json: FETCH_SIZE /
                 WRITE_SIZE in separate --pmc runs of this same command, scripts/pmc_traffic.sh).
  cpu_baseline : the oracle (CPU restatement of the reference graph, torch-CPU fp32, fastest thread count of a 2-image probe), 2 timed
                 passes over the same 32 images as rank 0's GPU batch (~20 s), mean; rank 0, N = 1 only;
                 gpu_vs_cpu_loss_rel = agreement of the two losses on that batch.
  bf16         : BASELINE configs[4] on the same workload (bf16 storage + bf16 MFMA, fp32 accumulate / BN statistics / loss / NMS);
                 its `roofline` = HBM roofline of the WHOLE grouped-conv backbone conv1_1 .. conv5_3 (trunk_roofline()).
  full_step    : BASELINE configs[3]: K training steps (fwd + loss + backward + RCCL gradient all-reduce + SGD), timed last.

N > 1: `python bench.py --gpus N` starts its own N ranks (self_launch(); nothing touches the GPU in the parent); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it uses the launcher's environment.
"""
import argparse
import json
import os
import statistics
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # before anything initialises the HIP runtime (RCCL / dmabuf IPC)

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'grouped-ssd-pytorch_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch                                                              # noqa: E402

CONFIGS = {
    # name: (build_ssd positional args after (phase, size, num_classes), oracle flags, GFLOP/img, MB/img train-mode fp32)
    'gssd': ((True, 4, 4, 1, True, False, False, 0, 1, False, False, 1), dict(), 17.47, 374.0),
    'gssdpp': ((True, 4, 4, 1, True, True, True, 1, 4, True, False, 1),
               dict(use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4,
                    dcn_cat_sab=True), 39.4, 402.0),
}
WORKLOAD_SHORT = {
    'gssd': 'gssd (BASELINE configs[1]): build_ssd(groups 4, BN, fuse) fwd + MultiBoxLoss, train-mode BN, batch 32, [B,12,300,300] slices',
    'gssdpp': 'gssdpp (BASELINE configs[2], GSSD++: self-attention + SA-base + 1 DCN layer x 4 deformable groups, dcn_cat_sab) fwd + '
              'MultiBoxLoss, train-mode BN, batch 32, [B,12,300,300] slices',
}
WORKLOAD = {
    'gssd': 'gssd (BASELINE configs[1]): build_ssd(groups 4, BN, fuse) forward + MultiBoxLoss, train-mode BN, '
            '[B,12,300,300] slices (4-phase 512x512 CT resized outside the timed region)',
    'gssdpp': 'gssdpp (BASELINE configs[2], GSSD++): build_ssd(groups 4, BN, fuse, self-attention + self-attention-base, '
              '1 DCN layer, 4 deformable groups, dcn_cat_sab) forward + MultiBoxLoss, train-mode BN, [B,12,300,300] slices '
              '(4-phase 512x512 CT resized outside the timed region)',
}
PEAK_F32_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix == fp32 vector peak
PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


LINE_LIMIT = 8192          # the driver keeps a bounded tail of stdout: r04's 21.6 KB line did not parse
ROOF_KEYS = ('bound', 'achieved', 'peak', 'unit', 'frac', 'useful_frac', 'traffic', 'traffic_over_alg', 'kernel', 'avg_launch_us', 'launches_timed', 'ms_per_step',
             'alg_flop_per_launch', 'alg_bytes_per_launch', 'fp32_equivalent_tflops', 'direct_conv_tflops')


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d} if isinstance(d, dict) else d


def _roof(r, note=True):
    if not isinstance(r, dict):
        return r
    out = _pick(r, ROOF_KEYS)
    if r.get('source'):
        out['measured'] = ('HIP events inside the timed region' if 'inside the timed region' in r['source'] else
                           'HIP event nodes inside the hipGraph, replays of their own behind the timed region' if 'of their own' in r['source'] else 'survey passes')
    if note and r.get('note'):
        out['note'] = str(r['note'])[:160]
    if r.get('alone'):
        out['alone'] = _pick(r['alone'], ('avg_launch_us', 'frac', 'useful_frac'))
    if r.get('next'):
        out['next'] = [_pick(q, ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'ms_per_step', 'avg_launch_us')) for q in r['next']]
    return out


def _fstep(f):
    return _pick(f, ('value', 'unit', 'steps', 'ms_per_step', 'host_enqueue_ms_per_step', 'host_cpu_ms_per_step', 'grad_elems', 'rccl_ranks',
                     'allreduce_exposed_ms', 'allreduce_overlapped', 'error'))


def compact_line(full, detail_path=None):
    """The ONE stdout line: the contract's fields + `roofline` + `cpu_baseline` + short summaries of the other legs, <= LINE_LIMIT
    bytes.  Everything else (per-kernel tables, per-layer trunk table, notes) lives in the detail file."""
    line = _pick(full, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                        'vs_baseline', 'dtype', 'data'))
    cfg = full.get('config') or {}
    line['config'] = dict(_pick(cfg, ('batch_per_gpu', 'global_batch', 'alg_gflop_per_img', 'alg_mb_per_img')),
                          workload=str(cfg.get('workload_short') or cfg.get('workload', ''))[:200])
    line.update(_pick(full, ('rccl_ranks', 'collective_backend', 'launcher', 'per_rank_ms_per_step', 'host_enqueue_ms_per_step',
                             'steady', 'loss', 'first_step_loss', 'whole_path', 'with_input_stage')))
    line['roofline'] = _roof(full.get('roofline'))
    tr = full.get('trunk')
    if isinstance(tr, dict):
        line['trunk_roofline'] = dict(_pick(tr, ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'ms_per_step', 'tflops')),
                                      worst_layer=_pick(tr.get('worst_layer'), ('layer', 'us', 'gbs', 'tflops', 'frac')))
    cb = full.get('cpu_baseline')
    if isinstance(cb, dict):
        line['cpu_baseline'] = dict(_pick(cb, ('value', 'unit', 'cores', 'kind', 'loss', 'passes_s', 'gpu_vs_cpu_loss_rel', 'host_threads', 'b2_img_s', 'twin')),
                                    sample=str(cb.get('sample', ''))[:240])
    else:
        line['cpu_baseline'] = cb
    b = full.get('bf16')
    if isinstance(b, dict):
        r = b.get('roofline')
        line['bf16'] = dict(_pick(b, ('value', 'unit', 'dtype', 'steps', 'ms_per_step', 'loss')),
                            roofline=(dict(_pick(r, ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'ms_per_step', 'tflops')),
                                           binding_frac=r.get('binding_frac'),
                                           worst_layer=_pick(r.get('worst_layer'), ('layer', 'us', 'gbs', 'tflops', 'frac')))
                                      if isinstance(r, dict) else r),
                            dominant_kernel=_roof(b.get('dominant_kernel'), note=False), full_step=_fstep(b.get('full_step')))
    sec = full.get('secondary')
    if isinstance(sec, dict):
        line['secondary'] = dict(_pick(sec, ('value', 'unit', 'ms_per_step', 'loss')), workload=str(sec.get('workload', ''))[:60],
                                 roofline=_roof(sec.get('roofline'), note=False))
    pl = full.get('pixellink')
    if isinstance(pl, dict):
        line['pixellink'] = dict(_pick(pl, ('value', 'unit', 'ms_per_step', 'batch')), full_step=_pick(pl.get('full_step'), ('ms_per_step', 'value')))
    line['full_step'] = _fstep(full.get('full_step'))
    line['input_stage'] = _pick(full.get('input_stage'), ('ms_per_batch', 'alg_gbs', 'frac_hbm_peak', 'dtype'))
    line['detail'] = detail_path
    # belt and braces: drop optional summaries until the line fits
    for k in ('pixellink', 'input_stage', 'secondary', 'trunk_roofline', 'first_step_loss', 'host_enqueue_ms_per_step'):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(k, None)
    if len(json.dumps(line)) > LINE_LIMIT:          # (8 ranks: per-rank lists are the only thing that grows)
        line['roofline'] = _roof(full.get('roofline'), note=False)
        if isinstance(line.get('bf16'), dict):
            line['bf16'] = _pick(line['bf16'], ('value', 'unit', 'ms_per_step', 'roofline'))
    if len(json.dumps(line)) > LINE_LIMIT:
        # last resort: never lose the line of a finished multi-minute run -- the contract's fields, the two required objects without
        # their notes, and the path of the detail file that holds everything else
        cb = line.get('cpu_baseline')
        line = dict(_pick(line, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                                 'dtype', 'data', 'config', 'rccl_ranks')),
                    roofline=_pick(line.get('roofline'), ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel')),
                    cpu_baseline=_pick(cb, ('value', 'unit', 'cores', 'kind')) if isinstance(cb, dict) else cb,
                    detail=detail_path, truncated=True)
    return line


def write_detail(full, path):
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(full, f, indent=1)
        return path
    except OSError as e:
        print(f'bench.py: could not write {path}: {e}', file=sys.stderr)
        return None


def cpu_baseline(cfg_name, sample_b, seed, twin=None):
    """Oracle forward + loss on the host cores (never the product path), per BASELINE.md section 4: torch.no_grad(), forward + MultiBoxLoss,
    the same images as rank 0's GPU batch, 1 warm-up + 3 timed passes, MEDIAN.  Threads: BASELINE.md asks for os.cpu_count() "or state why not":
    on the 256-thread GPU box oneDNN's grouped convs run 3x SLOWER with all hardware threads than with 16-32 (measured round 4/5), so the
    thread count is the fastest of {16, 32, 64} in a 2-image probe (itself 1 warm-up + 1 timed: the B = 2 figure BASELINE.md also asks for)
    and `cores` states it.  `twin` = the other config (GSSD for the GSSD++ headline), timed the same way with the chosen thread count."""
    from oracle import gssd_oracle as O
    from gssd import synth
    from models.ssd_multiphase_custom_group import build_ssd
    ncpu = os.cpu_count() or 1
    pri = O.prior_box()

    def runner(name):
        args, flags, _, _ = CONFIGS[name]
        net = build_ssd('train', 300, 2, *args)
        sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111)

        def one(b, s):
            x = synth.synth_images(b, seed=s)
            tg = [t.numpy() for t in synth.synth_targets(b, seed=s)]
            t0 = time.perf_counter()
            with torch.no_grad():
                loc, conf, _ = O.gssd_forward(sd, x, **flags)
            ll, lc = O.multibox_loss(loc.numpy(), conf.numpy(), pri, tg)[:2]
            return time.perf_counter() - t0, (float(ll), float(lc))
        return one
    one = runner(cfg_name)
    probes = {}
    for nt in sorted({min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}, reverse=True):
        torch.set_num_threads(nt)
        one(2, seed + 1)
        probes[nt] = one(2, seed + 1)[0]
    cores = min(probes, key=probes.get)
    torch.set_num_threads(cores)

    def timed(fn, b):
        fn(b, seed)                                             # warm-up pass at the full sample
        runs = [fn(b, seed) for _ in range(3)]
        dts = sorted(r[0] for r in runs)
        return dts[1], [round(r[0], 2) for r in runs], runs[0][1]
    med, passes, loss = timed(one, sample_b)
    out = dict(value=round(sample_b / med, 3), unit='img/s', cores=cores, kind='port', loss=[round(loss[0], 5), round(loss[1], 5)],
               passes_s=passes, host_threads=ncpu, b2_img_s={str(k): round(2 / v, 2) for k, v in probes.items()},
               sample=f'fwd+MultiBoxLoss over the same {sample_b} images as the GPU batch ({cfg_name}, fp32, train-mode BN, torch-CPU oracle): '
                      f'1 warm-up + 3 timed passes, median {med:.1f} s; {cores} of {ncpu} host threads (fastest of a B=2 probe over 16/32/64; '
                      f'all {ncpu} is ~3x slower for grouped convs)')
    if twin:
        tmed, tpasses, _ = timed(runner(twin), sample_b)
        out['twin'] = dict(config=twin, value=round(sample_b / tmed, 3), unit='img/s', cores=cores, passes_s=tpasses)
    return out


class EventList(list):
    only = None       # kernel instances bracketed by HIP events around EAGER launches (the plan's hipGraph is cut around them)
    nodes = None      # kernel instances bracketed by event-record NODES inside the replayed hipGraph (gssd/plan_exec.py::NodeEvent)


class _Ms:
    """a time stamp in ms with torch.cuda.Event's elapsed_time(): samples read from graph event nodes right after their replay"""

    def __init__(self, ms):
        self.ms = ms

    def elapsed_time(self, other):
        return other.ms - self.ms


def aggregate(evs):
    agg = {}
    for (name, flops, byts), e0, e1 in evs:
        r = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
        r[0] += 1
        r[1] += e0.elapsed_time(e1)
        r[2] += flops
        r[3] += byts
    return agg


# the grouped-conv backbone conv1_1 .. conv5_3 (+ pool4) as the engine names its launches (gssd/engine.py::Tag.layer)
TRUNK = {'vgg.0': 'conv1_1', 'vgg.3': 'conv1_2', 'vgg.7': 'conv2_1', 'vgg.10': 'conv2_2', 'vgg.14': 'conv3_1', 'vgg.17': 'conv3_2',
         'vgg.20': 'conv3_3', 'vgg.24': 'conv4_1', 'vgg.27': 'conv4_2', 'vgg.30': 'conv4_3', 'vgg.33': 'pool4', 'vgg.34': 'conv5_1',
         'vgg.37': 'conv5_2', 'vgg.40': 'conv5_3'}
# SURVEY.md 8(d)'s per-layer accounting (conv in + raw out, BN re-read + activated write), fp32 MB / image, conv1_1 .. conv5_3
SURVEY_TRUNK_MB_F32 = 73.4 + 74.9 + 40.3 + 37.4 + 20.2 + 23.0 + 18.8 + 10.4 + 11.8 + 11.8 + 3 * 3.0


def trunk_roofline(survey, n_pass, B, dtype):
    """HBM roofline of the whole grouped-conv backbone conv1_1 .. conv5_3: `achieved` = SURVEY.md 8(d)'s algorithmic bytes per image x
    images / the summed durations of every trunk launch (convs AND their BatchNorm / ReLU / pool passes; HIP events around every launch
    in the eager survey passes).  `as_built` gives the same with the compulsory bytes of the pass structure as built (input + output +
    weights of each launch that still exists): deleting a pass shrinks those, so that fraction cannot be raised by accounting."""
    lay = {}
    for tag, e0, e1 in survey:
        name = TRUNK.get(getattr(tag, 'layer', None))
        if name is None:
            continue
        r = lay.setdefault(name, [0.0, 0.0, 0.0, set()])
        r[0] += e0.elapsed_time(e1) / n_pass
        r[1] += tag[2] / n_pass
        r[2] += tag[1] / n_pass
        r[3].add(tag[0])
    if not lay:
        return None
    ms = sum(r[0] for r in lay.values())
    by = sum(r[1] for r in lay.values())
    fl = sum(r[2] for r in lay.values())
    per = {k: dict(us=round(1e3 * r[0], 1), alg_mb=round(r[1] / 1e6, 1), gbs=round(r[1] / r[0] / 1e6, 1),
                   tflops=round(r[2] / r[0] / 1e9, 1), kernels=sorted(r[3])) for k, r in lay.items()}
    # the BINDING roof per layer: max(bytes / t / HBM peak, FLOPs / t / matrix peak of the storage mode); the trunk's figure
    # is the time-weighted mean of the layers'
    pk = PEAK_BF16_TFLOPS if dtype == 'bf16' else PEAK_F32_TFLOPS
    for v in per.values():
        v['binding_frac'] = round(max(v['gbs'] / PEAK_HBM_GBS, v['tflops'] / pk), 4)
    binding = sum(per[k]['binding_frac'] * lay[k][0] for k in lay) / ms
    big = {k: v for k, v in per.items() if v['alg_mb'] >= 0.02 * by / 1e6}
    worst = min(big, key=lambda k: big[k]['gbs'])
    survey_mb = SURVEY_TRUNK_MB_F32 * (0.5 if dtype == 'bf16' else 1.0)
    eff = survey_mb * B / ms                          # MB / ms = GB/s
    built = by / ms / 1e6
    # `achieved` / `frac` are the bytes the pass structure AS BUILT moves (input + output + weights of every launch that still
    # exists) -- deleting a pass shrinks them, so the fraction cannot be raised by accounting.  The figure under SURVEY.md 8(d)'s
    # accounting (the reference's pass structure, deleted passes counted as moved) is kept beside it under its own name.
    return dict(bound='hbm', achieved=round(built, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(built / PEAK_HBM_GBS, 4),
                traffic=None, kernel='trunk conv1_1 .. conv5_3 (convs + BN/ReLU/pool passes)', ms_per_step=round(ms, 4),
                binding_frac=round(binding, 4),
                plan='nograd (torch.no_grad() forward: pooled trunk layers keep no full-resolution raw maps; a training step cannot use it)',
                alg_mb_per_img=round(by / B / 1e6, 2), alg_bytes_per_step=round(by), tflops=round(fl / ms / 1e9, 1),
                accounting='achieved = compulsory bytes of the launches as built (input + output + weights per launch) x images / '
                           'measured trunk time (every trunk launch -- convs AND BatchNorm / ReLU / pool passes -- bracketed by HIP events '
                           'in the eager survey passes)',
                as_built=dict(alg_mb_per_img=round(by / B / 1e6, 2), alg_bytes_per_step=round(by), gbs=round(built, 1),
                              frac=round(built / PEAK_HBM_GBS, 4)),
                effective_vs_reference_passes=dict(
                    alg_mb_per_img=round(survey_mb, 1), alg_bytes_per_step=round(survey_mb * 1e6 * B), gbs=round(eff, 1),
                    frac=round(eff / PEAK_HBM_GBS, 4),
                    note="SURVEY.md 8(d)'s accounting: every layer priced as conv in + raw out + BatchNorm re-read + activated write (bf16 = "
                         'half the fp32 figure); passes this build deleted count as moved -- an effective rate, NOT HBM utilisation'),
                worst_layer=dict(layer=worst, **per[worst], frac=round(per[worst]['gbs'] / PEAK_HBM_GBS, 4)), layers=per,
                note='eager survey passes, every trunk launch bracketed by HIP events on the launch stream')


def measure(cfg, a, dev, gd, rank, world, dtype='f32'):
    """Build the net of one config on this rank, time --steps forward + loss steps; returns the result dict."""
    from gssd import synth
    from layers.modules import MultiBoxLoss
    from models.ssd_multiphase_custom_group import build_ssd
    args, flags, gflop_img, mb_img = CONFIGS[cfg]
    net = build_ssd('train', 300, 2, *args)
    net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
    net = net.to(dev).train()
    if dtype == 'bf16':
        net.compute_dtype = 'bf16'
        mb_img = mb_img / 2
    crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
    B = a.batch
    x = synth.synth_images(B, seed=gd.shard_seed(100, rank)).to(dev)      # rank r owns its own 32 images
    tg = [t.to(dev) for t in synth.synth_targets(B, seed=gd.shard_seed(100, rank))]

    def step():
        with torch.no_grad():
            return crit(net(x), tg)

    def sync():
        gd.barrier(dev)

    # the first step starts from the same state as the CPU baseline (a training forward advances spectral norm's u / v and
    # the BN running statistics): its loss is the one compared against the oracle's
    ll, lc = step()
    first_loss = (float(ll), float(lc))
    for _ in range(a.warmup - 1):
        ll, lc = step()
    # Per-launch HIP events cost ~3 us of GPU idle each.  Two untimed passes bracket EVERY tagged launch (the `kernels`
    # breakdown and the choice of the dominant instance); in the timed region only the dominant instance's launches are
    # bracketed, live, on the launch stream -- that is where `roofline` comes from.
    sagg, events, trunk, runner_up = None, None, None, None
    if not a.no_events:
        survey = EventList()
        net.__dict__['_events'] = survey
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        net.__dict__['_events'] = None
        sagg = aggregate(survey)
        trunk = trunk_roofline(survey, 2, a.batch, dtype)
        if trunk is not None:
            try:        # HBM bytes of the trunk's kernels per step from the committed PMC passes (scripts/pmc_traffic.sh)
                tr = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_summary.json'))).get((cfg if dtype == 'f32' else f'{cfg}_{dtype}') + '_trunk')
                if tr:
                    trunk['traffic'] = tr['hbm_bytes_per_step']
                    trunk['as_built']['traffic_over_bytes'] = round(tr['hbm_bytes_per_step'] / trunk['as_built']['alg_bytes_per_step'], 3)
            except (OSError, ValueError, KeyError):
                pass
        events = EventList()
        # the dominant KERNEL: the template instance with the most time per step in the survey passes -- no launch-count cap, no
        # tie-break; the next two instances are reported beside it from the same survey passes (`roofline_next`)
        order = sorted(sagg, key=lambda k: -sagg[k][1])
        top = order[0]
        events.only = {top}
        runner_up = order[1:3]
    # Where the dominant instance's launches are bracketed: INSIDE the timed region when that costs at most MAX_INSIDE event pairs per step (each
    # pair is ~3 us of GPU idle and one cut of the step's hipGraph; the fp32 headline's dominant kernel is ONE launch per step); an instance
    # launched more often (bf16 mode: the generic 128 x 64 GEMM tile, 15 launches per step -- bracketing those inside the region cost the leg 0.6 of
    # 3.9 ms) is bracketed in a pass of its own of the same K steps right behind the timed region; `roofline.source` says which.
    MAX_INSIDE = 4
    inside = events is not None and sagg[top][0] // 2 <= MAX_INSIDE
    live = events if inside else None

    def primed(ev):
        # untimed steps in exactly the launch mode of the region that follows (hipGraph segments around the bracketed kernel): the graph
        # capture of that mode happens here, not inside the timed steps, whatever --warmup is
        for _ in range(3):
            prime = None
            if ev is not None:
                prime = EventList()
                prime.only, prime.nodes = ev.only, ev.nodes
            net.__dict__['_events'] = prime
            step()
        torch.cuda.synchronize()
        net.__dict__['_events'] = ev
    primed(live)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ll, lc = step()
    sync()
    dt = time.perf_counter() - t0
    net.__dict__['_events'] = None
    if events is not None and not inside:
        # event-record nodes INSIDE the step's hipGraph (round 6): the launches keep their place in the graph and their neighbours on the other
        # branches -- the same conditions as the timed region and as a rocprofv3 trace of it (bracketing them eagerly cut the graph around every
        # launch and timed each one ALONE: conv_x6<128> 97 us against 147 us in the trace).  The nodes hold the last replay's times: every replay
        # of this pass is followed by a sync and a read.
        events.only, events.nodes = None, {top}
        primed(events)
        samples = EventList()
        for _ in range(a.steps):
            ev = EventList()
            ev.nodes = {top}
            net.__dict__['_events'] = ev
            step()
            torch.cuda.synchronize()
            samples.extend((tag, _Ms(0.0), _Ms(e0.elapsed_time(e1))) for tag, e0, e1 in ev)
        net.__dict__['_events'] = None
        events = samples
    per_rank = [round(1e3 * t / a.steps, 3) for t in gd.gather_over_ranks(dt, dev)]
    loss = (float(ll), float(lc))
    if not all(map(lambda v: v == v and abs(v) != float('inf'), loss)):
        raise SystemExit(f'non-finite loss {loss}')
    dt = gd.max_over_ranks(dt, dev)
    # the same measurement over 100 more steps, no events
    sdt = 0.0
    if a.steady > 0:
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steady):
            step()
        sync()
        sdt = gd.max_over_ranks(time.perf_counter() - t0, dev)

    # host side of a step: wall time this rank's Python needs to ENQUEUE a step (graph replays + loss launches), nothing waited for.
    # With 8 ranks on one host this -- not the GPU -- is what weak scaling can lose; max / per-rank values are reported.
    sync()
    nh = max(1, min(a.steps, 20))
    t0 = time.perf_counter()
    for _ in range(nh):
        step()
    host_ms = 1e3 * (time.perf_counter() - t0) / nh
    sync()
    host_per_rank = [round(t, 3) for t in gd.gather_over_ranks(host_ms, dev)]

    roof, kernels = None, {}
    peak_t = PEAK_BF16_TFLOPS if dtype == 'bf16' else PEAK_F32_TFLOPS
    if events:
        for name, (n, ms, fl, by) in sagg.items():
            kernels[name] = dict(launches_per_step=n // 2, avg_us=round(1e3 * ms / n, 2),
                                 ms_per_step=round(ms / 2, 4), tflops=round(fl / (ms * 1e-3) / 1e12, 2),
                                 alg_gbs=round(by / (ms * 1e-3) / 1e9, 1))
        agg = aggregate(events)

        def make_roof(dom, row, source):
            n, ms, fl, by = row
            traffic = None
            pmc = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(cfg if dtype == 'f32' else f'{cfg}_{dtype}', {}).get(dom, {}).get('hbm_bytes_per_launch')
                except Exception:
                    traffic = None
            ach_t, ach_b = fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9
            wino = dom.startswith(('conv_wino', 'conv_thin_wino'))
            x6 = dom.startswith(("dcn_x6", "conv_x6", "conv_wino_x6", "conv_thin_x6", "flash_attn_x"))
            # matrix instructions per fp32-equivalent product of the forward launches (round 6): three fp16 ones (v_mfma_f32_16x16x32_f16, same
            # dense peak as bf16) over fp16 planes; six bf16 ones with GSSD_X6_F16=0 and in the attention core (whose fp16 form is opt-in)
            f16_form = os.environ.get('GSSD_X6_F16', '1') != '0' and (not dom.startswith('flash_attn_x') or os.environ.get('GSSD_FLASH_X6_F16') == '1')
            mm = 3 if f16_form else 6
            form = ('fp16 planes, three v_mfma_f32_16x16x32_f16 per product' if f16_form else 'three bf16 planes, six v_mfma_f32_16x16x32_bf16 per product')
            if dom.startswith('conv_wino_x6'):
                # Winograd F(2x2,3x3) with split operands: 2.25 x fewer products than the direct conv, `mm` MFMAs each -- `achieved` /
                # `frac` are the ISSUED 16-bit FLOPs (direct-conv FLOPs x mm / 2.25) against the bf16 / fp16 matrix peak
                iss = ach_t * mm / 2.25
                roof = dict(bound='mfma', achieved=round(iss, 2), peak=PEAK_BF16_TFLOPS, unit='TFLOP/s', frac=round(iss / PEAK_BF16_TFLOPS, 4),
                            useful_frac=round(ach_t / PEAK_BF16_TFLOPS, 4), direct_conv_tflops=round(ach_t, 2), fp32_equivalent_over_fp32_mfma_peak=round(ach_t / PEAK_F32_TFLOPS, 4))
                roof['mfma_per_product'] = mm
                note = f'Winograd F(2x2,3x3), fp32 operands as {form}: achieved = ISSUED 16-bit FLOPs'
            elif x6:
                # split-operand kernels: fp32-equivalent products as `mm` 16-bit MFMAs over operands split into planes -- `achieved` /
                # `frac` are the ISSUED 16-bit FLOPs (mm x algorithmic) against the bf16 / fp16 matrix peak; the algorithmic rate has its own name
                roof = dict(bound='mfma', achieved=round(mm * ach_t, 2), peak=PEAK_BF16_TFLOPS, unit='TFLOP/s',
                            frac=round(mm * ach_t / PEAK_BF16_TFLOPS, 4), useful_frac=round(ach_t / PEAK_BF16_TFLOPS, 4),
                            mfma_per_product=mm, fp32_equivalent_tflops=round(ach_t, 2),
                            fp32_equivalent_over_fp32_mfma_peak=round(ach_t / PEAK_F32_TFLOPS, 4))
                note = (f'fp32 operands as {form}, fp32 accumulate: achieved / frac = ISSUED 16-bit FLOPs '
                        '(pipe occupancy); useful_frac = algorithmic FLOPs / that peak')
            elif ach_b / PEAK_HBM_GBS > ach_t / peak_t:
                roof = dict(bound='hbm', achieved=round(ach_b, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(ach_b / PEAK_HBM_GBS, 4))
                note = 'algorithmic bytes (in + out + weights) / launch time'
            elif wino:
                # Winograd F(2x2,3x3) issues 2.25x fewer MFMA FLOPs than the direct convolution it computes: `achieved` / `frac` are the
                # ISSUED FLOPs against the matrix pipe (what a roofline fraction means); the direct-convolution rate is reported under its own name
                roof = dict(bound='mfma', achieved=round(ach_t / 2.25, 2), peak=peak_t, unit='TFLOP/s', frac=round(ach_t / 2.25 / peak_t, 4),
                            direct_conv_tflops=round(ach_t, 2), direct_conv_over_peak=round(ach_t / peak_t, 4))
                note = 'Winograd F(2x2,3x3) on fp32 MFMA: achieved = ISSUED FLOPs (direct-conv FLOPs / 2.25)'
            else:
                roof = dict(bound='mfma', achieved=round(ach_t, 2), peak=peak_t, unit='TFLOP/s', frac=round(ach_t / peak_t, 4))
                note = 'fp32 MFMA (v_mfma_f32_16x16x4_f32)' if dtype == 'f32' else 'bf16 MFMA, fp32 accumulate'
            if traffic and by:
                roof['traffic_over_alg'] = round(traffic / (by / n), 2)
            roof.update(traffic=traffic, kernel=dom, avg_launch_us=round(1e3 * ms / n, 2), launches_timed=n,
                        ms_per_step=round(ms / n * (sagg[dom][0] // 2), 4),
                        alg_flop_per_launch=round(fl / n), alg_bytes_per_launch=round(by / n), source=source, note=note)
            return roof
        dom = max(agg, key=lambda k: agg[k][1])
        roof = make_roof(dom, agg[dom], 'HIP events around every launch of this instance inside the timed region' if inside else
                         f'HIP event-record nodes around every launch of this instance INSIDE the replayed hipGraph, in {a.steps} steps of their own right '
                         f'behind the timed region ({sagg[dom][0] // 2} launches per step; each replay followed by a sync that reads the nodes)')
        if not inside and dom in sagg:
            # the same launches timed ALONE (eager survey passes: nothing else on the chip) beside the in-graph figure: inside the step's graph a
            # launch shares the CUs with the other branches' kernels, so its own duration -- what rocprofv3 reports too -- is longer
            alone = make_roof(dom, sagg[dom], 'untimed survey passes')
            roof['alone'] = dict(avg_launch_us=alone['avg_launch_us'], frac=alone['frac'], useful_frac=alone.get('useful_frac'),
                                 note='every launch bracketed eagerly, nothing else running')
        roof['next'] = [make_roof(k, sagg[k], 'untimed survey passes') for k in (runner_up or []) if k in sagg]
    # the HBM-side companion of `roofline`: the heaviest of the byte-bound trunk layers (the patch-staged thin kernels of conv1_1 ..
    # conv2_2: arithmetic intensity below the ridge in both storage modes), from the untimed survey passes' per-launch HIP events
    roof_hbm = None
    if events and sagg:
        thin = {k: v for k, v in sagg.items() if k.startswith('conv_thin')}
        if thin:
            k = max(thin, key=lambda q: thin[q][3] / max(thin[q][0], 1))          # most algorithmic bytes per launch
            n, ms, fl, by = thin[k]
            ach = by / (ms * 1e-3) / 1e9
            t2 = None
            try:
                t2 = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_summary.json'))).get(
                    cfg if dtype == 'f32' else f'{cfg}_{dtype}', {}).get(k, {}).get('hbm_bytes_per_launch')
            except (OSError, ValueError):
                pass
            roof_hbm = dict(bound='hbm', achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(ach / PEAK_HBM_GBS, 4),
                            traffic=t2, kernel=k, avg_launch_us=round(1e3 * ms / n, 2), launches_timed=n,
                            alg_bytes_per_launch=round(by / n), note='survey passes (eager, every tagged launch bracketed)')
    value = gd.aggregate_rate(world, B, a.steps, dt)
    res = dict(value=round(value, 2), ms_per_step=round(1e3 * dt / a.steps, 3), per_rank_ms_per_step=per_rank, loss=[round(loss[0], 5), round(loss[1], 5)],
               steady=(dict(steps=a.steady, ms_per_step=round(1e3 * sdt / a.steady, 3),
                            value=round(gd.aggregate_rate(world, B, a.steady, sdt), 2)) if a.steady > 0 else None),
               workload=WORKLOAD[cfg] + (', bf16 storage / bf16 MFMA / fp32 accumulate + BN statistics + loss' if dtype == 'bf16' else ''),
               alg_gflop_per_img=gflop_img, alg_mb_per_img=mb_img,
               whole_path=dict(tflops=round(value * gflop_img / 1e3, 2),
                               frac_mfma_peak=round(value * gflop_img / 1e3 / (peak_t * world), 4),        # fp32 mode: against the fp32 matrix peak (157 TFLOP/s) --
                               # above 1 since round 6: most of its products run as three fp16 matrix instructions, a pipe 16 x wider
                               frac_16bit_mfma_peak=round(value * gflop_img / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
                               alg_gbs=round(value * mb_img / 1e3, 1),
                               frac_hbm_peak=round(value * mb_img / 1e3 / (PEAK_HBM_GBS * world), 4)),
               host_enqueue_ms_per_step=host_per_rank,
               first_step_loss=[round(first_loss[0], 5), round(first_loss[1], 5)], roofline=roof, roofline_hbm_trunk=roof_hbm,
               trunk=trunk, kernels=kernels)
    return res, net, crit, x, tg, first_loss


def measure_pixellink(B, dev, steps=20):
    """PixelLink++ (cascade_fuse, fuse conv + BN, Self_Attn x 8, 1 DCN layer on slice_and_cat(x, SA-base)) forward
    (train-mode BN, spectral-norm power iteration) + PixelLinkLoss + link decoding, B images, synthetic weights / images / masks."""
    import numpy as np
    from gssd import synth
    from pixel_link.model import PixelLink
    from pixel_link.criterion import PixelLinkLoss
    from pixel_link import postprocess
    net = PixelLink(cascade_fuse=True, use_fuseconv=True, batch_norm=True, use_self_attention=True, use_self_attention_base=True,
                    num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True, detach_sab=False)
    net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=2222))
    net = net.to(dev).train()
    x = synth.synth_images(B, seed=300).to(dev)
    rng = np.random.default_rng(5)
    pix = np.zeros((B, 75, 75), np.int64)
    for b in range(B):
        y, xx = rng.integers(5, 55, size=2)
        pix[b, y:y + 12, xx:xx + 9] = 1
    t = lambda v: torch.from_numpy(v).to(dev)
    pix_t, neg_t = t(pix), t((pix == 0).astype(np.uint8))
    posw_t, link_t = t(pix.astype(np.float32)), t(np.repeat(pix[:, None], 8, 1))
    crit = PixelLinkLoss()

    def step():
        with torch.no_grad():
            o1, o2 = net(x)
            pp, pn = crit.pixel_loss(o1, pix_t, neg_t, posw_t, link=(o2, link_t))
            lp, ln = crit.link_loss(o2, link_t)
            postprocess.decode(o1, o2)
        return pp + pn, lp + ln
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pl, ll = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # training step of the row (train loop of ssd_liverdet/pixel_link: forward, PixelLinkLoss, loss.backward(), SGD): HIP forward plan,
    # HIP loss backward, HIP backward plan (gssd/backward.py::PixelLinkBackwardPlan)
    opt = torch.optim.SGD(net.parameters(), lr=1e-4, momentum=0.9)
    tsteps = max(2, steps // 4)

    def train_step():
        opt.zero_grad(set_to_none=True)
        o1, o2 = net(x)
        pp, pn = crit.pixel_loss(o1, pix_t, neg_t, posw_t, link=(o2, link_t))
        lp, ln = crit.link_loss(o2, link_t)
        (pp + pn + lp + ln).backward()
        opt.step()
    for _ in range(2):
        train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(tsteps):
        train_step()
    torch.cuda.synchronize()
    dtt = time.perf_counter() - t0
    return dict(metric='512x512 4-phase CT img/s (PixelLink++ fwd + loss + link decoding)', value=round(B * steps / dt, 2), unit='img/s',
                steps=steps, ms_per_step=round(1e3 * dt / steps, 3), batch=B, dtype='f32', loss=[round(float(pl), 5), round(float(ll), 5)],
                full_step=dict(metric='PixelLink++ training step (forward + PixelLinkLoss + backward + SGD)', steps=tsteps,
                               ms_per_step=round(1e3 * dtt / tsteps, 3), value=round(B * tsteps / dtt, 2), unit='img/s'),
                workload='pixellink++ cascade_fuse=1 fuseconv=1 bn=1 sa=1 sab=1 dcn=1x4 cat_sab=1, 300x300x12 -> [B,2,75,75] | [B,16,75,75]')


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python3 bench.py --gpus N` without a launcher: this process touches no GPU (device_count() does not initialise HIP on this
    image), starts one child per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set -- the environment torch.distributed.run would
    give them -- and exits with their return code.  Rank 0's JSON line is the only thing the children print on stdout."""
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < n and '--launch-probe' not in sys.argv and not os.environ.get('GSSD_DIST_SAME_DEVICE'):
        print(json.dumps({'error': f'--gpus {n} needs {n} visible GPUs, this host has {ndev}', 'n_gpus': n, 'devices_visible': ndev}))
        return 2
    import tempfile
    # rendezvous through a FileStore in a private directory: picking a free loopback port here and binding it seconds later in rank 0 is a
    # race with everything else on the host (the eight-rank test failed ~1 run in 5 with the port form).  MASTER_ADDR / MASTER_PORT are
    # still exported for code that reads them, but nothing binds that port.
    rdzv_dir = tempfile.mkdtemp(prefix='gssd_rdzv_')
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               GSSD_DIST_INIT_FILE=os.path.join(rdzv_dir, 'store'), HSA_ENABLE_IPC_MODE_LEGACY='0', GSSD_BENCH_SELF_LAUNCHED='1')
    # each rank gets its own block of host cores (its Python enqueue thread, torch's intra-op pool and RCCL's proxy thread stay off the
    # other ranks' cores); GSSD_BENCH_NO_PIN=1 leaves the affinity alone
    try:
        cores = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = []
    per = len(cores) // n if not os.environ.get('GSSD_BENCH_NO_PIN') else 0

    def pin(r):
        return (lambda: os.sched_setaffinity(0, cores[r * per:(r + 1) * per])) if per >= 1 else None
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], preexec_fn=pin(r),
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r), GSSD_BENCH_CORES=str(per))) for r in range(n)]
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            time.sleep(0.05)
            done = {r: procs[r].poll() for r in pending}
            done = {r: c for r, c in done.items() if c is not None}
            if not done:
                continue
            if any(c != 0 for c in done.values()) and rc == 0:
                # one rank died: the others would wait in a collective forever.  Ranks that die BECAUSE a peer went away (a gloo / RCCL
                # error -> exit code 1) can be seen in the same poll as the rank that caused it: give the rest a moment, then report
                # the most specific code (anything but the generic 1 first)
                # (a rank that exits on purpose still spends a second or two in interpreter teardown: the peer that failed BECAUSE of it can be
                # reaped first -- give the others up to 3 s to end on their own)
                t_end = time.perf_counter() + 3.0
                while time.perf_counter() < t_end and len(done) < len(pending):
                    time.sleep(0.05)
                    for r in pending - set(done):
                        c = procs[r].poll()
                        if c is not None:
                            done[r] = c
                bad = [c for _, c in sorted(done.items()) if c != 0]
                # an explicit exit code (> 1) is the cause; a signal (negative: gloo aborts when its peer closes) or the generic 1 is the consequence
                rc = next((c for c in bad if c > 1), next((c for c in bad if c < 0), bad[0]))
                for q in pending - set(done):
                    procs[q].terminate()
            pending -= set(done)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        import shutil
        shutil.rmtree(rdzv_dir, ignore_errors=True)
    return rc


def full_step_leg(a, net, crit, x, tg, dev, gd, world, B):
    """K full training steps (train_lesion_multiphase_v2.py:242-253 with DataParallel :593 replaced by one process per GPU):
    forward + MultiBoxLoss + backward + gradient all-reduce + SGD.  At world 1 no collective runs."""
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-4, momentum=0.9, weight_decay=5e-4)
    gd.broadcast_params(net)
    red = gd.OverlappedGradReducer(world)         # ranges of the flat gradient buffer are all-reduced under the backward
    ev = []

    def train_step(timed=False):
        opt.zero_grad(set_to_none=True)
        ll, lc = crit(net(x), tg)
        red.arm(net)
        (ll + lc).backward()
        if timed and world > 1:
            # what the all-reduce leaves EXPOSED: the main stream's time between "backward enqueued" and "every range reduced
            # and scaled" (finish() makes the main stream wait for RCCL's stream, then divides by the world size)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = red.finish()
            e1.record()
            ev.append((e0, e1))
        else:
            n = red.finish() if world > 1 else sum(p.grad.numel() for p in params if p.grad is not None)
        opt.step()
        return n
    for _ in range(4):                           # (the backward plan captures its hipGraphs on its third run)
        nred = train_step()
    gd.barrier(dev)
    t0, c0 = time.perf_counter(), time.process_time()
    for _ in range(a.full_step):
        train_step(True)
    host = time.perf_counter() - t0                 # every step enqueued, nothing waited for (beyond what the steps themselves wait for)
    host_cpu = time.process_time() - c0             # CPU seconds of ALL threads of this rank (Python, autograd worker, HIP runtime)
    gd.barrier(dev)
    fdt = gd.max_over_ranks(time.perf_counter() - t0, dev)
    host_per_rank = [round(1e3 * t / a.full_step, 3) for t in gd.gather_over_ranks(host, dev)]
    exposed = (sum(e0.elapsed_time(e1) for e0, e1 in ev) / len(ev)) if ev else 0.0
    exposed = gd.max_over_ranks(exposed, dev)
    return dict(value=round(gd.aggregate_rate(world, B, a.full_step, fdt), 2), unit='img/s', steps=a.full_step,
                ms_per_step=round(1e3 * fdt / a.full_step, 3), host_enqueue_ms_per_step=host_per_rank,
                host_cpu_ms_per_step=round(1e3 * host_cpu / a.full_step, 3), grad_elems=int(nred), rccl_ranks=world,
                allreduce_exposed_ms=round(exposed, 3), allreduce_overlapped=bool(red.overlapped_last),
                host_note='host_enqueue = wall time to enqueue the K steps: the GPU step is longer than the host side, so it includes the time the '
                          'HIP queues push back (scripts/host_profile.py at batch 2, where the host is the bound: 6.8 ms per step); host_cpu = CPU '
                          'seconds of all threads of the rank per step',
                note='fwd (HIP) + MultiBoxLoss (HIP fwd/bwd) + network backward (HIP: gssd/backward.py) + '
                     + (f'RCCL all-reduce of the flat fp32 gradient buffer over {world} ranks in 4 ranges started under the backward '
                        '(allreduce_exposed_ms = main-stream time from backward-enqueued to all ranges reduced + scaled, max over ranks)'
                        if world > 1 else 'NO collective (one rank: nothing to reduce)') + ' + SGD')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--steady', type=int, default=100, help='extra steps timed after the --steps region (reported as `steady`)')
    ap.add_argument('--config', default='gssdpp', choices=list(CONFIGS))
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--batch', type=int, default=32, help='images per GPU')
    ap.add_argument('--cpu-sample', type=int, default=32, help='images in the CPU baseline sample (0 = skip)')
    ap.add_argument('--no-events', action='store_true', help='skip the per-launch HIP events (roofline = null)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary (plain GSSD) measurement')
    ap.add_argument('--full-step', type=int, default=None, metavar='K',
                    help='additionally time K full training steps (fwd + loss + backward + gradient all-reduce + SGD); '
                         'reported as "full_step" beside the fwd+loss metric (BASELINE configs[3]); default: 8 in fp32, 0 in bf16')
    ap.add_argument('--no-input-stage', action='store_true', help='skip the separate timing of the device input stage')
    ap.add_argument('--no-bf16', action='store_true', help='skip the bf16 leg (BASELINE configs[4] on this workload)')
    ap.add_argument('--full-step-timeout', type=float, default=300.0, help='N > 1: seconds the full-step leg may take')
    ap.add_argument('--detail', default=os.path.join(ROOT, 'gpurun_out', 'bench_detail.json'),
                    help='where rank 0 writes the full record (kernels / trunk layers / notes); the stdout line stays <= 8 KB')
    ap.add_argument('--launch-probe', action='store_true',
                    help='rendezvous check only, no GPU: every rank joins a gloo group, rank 0 prints the ranks it saw '
                         '(tests/test_host_cpu.py drives the self-launcher with it on the CPU)')
    ap.add_argument('--replay-detail', metavar='JSON', help='no GPU: read a full record (a detail file) and print the stdout line a run '
                                                            'would print from it (tests/test_host_cpu.py checks its size)')
    a = ap.parse_args()
    if a.replay_detail:
        print(json.dumps(compact_line(json.load(open(a.replay_detail)), os.path.relpath(a.replay_detail, ROOT))), flush=True)
        return

    from gssd import dist as gd
    world, rank, local = gd.env_world()
    if a.gpus != world and world > 1:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    if a.gpus > 1 and world == 1:
        sys.exit(self_launch(a.gpus))      # before anything initialises the GPU in this process
    if a.launch_probe:
        gd.init('gloo')
        seen = gd.gather_over_ranks(rank)
        if os.environ.get('GSSD_PROBE_FAIL_RANK') == str(rank):
            os._exit(7)                    # a rank that dies (no interpreter teardown: gloo's destructor may abort there and replace the code)
        gd.barrier()
        if rank == 0:
            print(json.dumps({'probe': True, 'n_gpus': world, 'ranks': seen, 'rccl_ranks': gd.world_size(),
                              'launcher': 'self' if os.environ.get('GSSD_BENCH_SELF_LAUNCHED') else 'external'}))
        gd.finish()
        return
    if a.full_step is None:
        a.full_step = 8 if a.dtype == 'f32' else 4
    # GSSD_DIST_SAME_DEVICE=1 + GSSD_DIST_BACKEND=gloo: every rank on GPU 0 with gloo collectives -- the N > 1 code path (sharding,
    # barriers, per-rank gather, overlapped gradient reducer, watchdog) on a one-GPU box; tests/test_gpu_multi.py.  Not a measurement.
    backend = os.environ.get('GSSD_DIST_BACKEND', 'nccl')
    if os.environ.get('GSSD_DIST_SAME_DEVICE'):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    gd.init(backend, dev)         # 'nccl' = RCCL over xGMI; the fwd+loss metric uses it for the timing barrier only

    from gssd import synth
    import copy
    import threading
    B = a.batch
    res, net, crit, x, tg, loss = measure(a.config, a, dev, gd, rank, world, a.dtype)

    # BASELINE.json configs[4] (bf16 storage, bf16 MFMA, fp32 accumulate / BN statistics / loss / NMS) on the same workload: its
    # `roofline` is the HBM roofline of the WHOLE grouped-conv backbone (north_star's ">= 70 % HBM roofline on the backbone")
    bf16 = None
    if a.dtype == 'f32' and not a.no_bf16 and a.config == 'gssdpp':
        a2 = copy.copy(a)
        a2.steps, a2.warmup, a2.steady = min(a.steps, 50), min(a.warmup, 5), min(a.steady, 50)
        bres, bnet, bcrit, bx, btg, _ = measure('gssdpp', a2, dev, gd, rank, world, 'bf16')
        bfull = None
        if a.full_step > 0 and world == 1:
            # configs[4] as a training config: bf16 forward + loss, mixed-precision backward (fp32 gradients at the stored bf16
            # activations, bf16 operands: gssd/backward.py), SGD on the fp32 master weights
            a2.full_step = min(a.full_step, 8)
            try:
                bfull = full_step_leg(a2, bnet, bcrit, bx, btg, dev, gd, world, B)
                bfull['note'] = ('bf16 forward + loss, mixed-precision backward on bf16 operands (fp32 masters, gradients and accumulation; '
                                 'GSSD_BWD_BF16=0: the fp32 backward plan on fp32 copies of the stored maps), SGD on fp32 masters')
            except Exception as e:                                  # noqa: BLE001 -- reported in the line
                bfull = {'error': f'{type(e).__name__}: {e}'[:300]}
        del bnet, bcrit, bx, btg
        bf16 = dict(metric='512x512 4-phase CT img/s (fwd+loss)', unit='img/s', dtype='bf16', steps=a2.steps, warmup=a2.warmup,
                    value=bres['value'], ms_per_step=bres['ms_per_step'], per_rank_ms_per_step=bres['per_rank_ms_per_step'],
                    steady=bres['steady'], loss=bres['loss'], workload=bres['workload'], whole_path=bres['whole_path'],
                    roofline=bres['trunk'], dominant_kernel=bres['roofline'], full_step=bfull, kernels=bres['kernels'])
        torch.cuda.empty_cache()

    # Device-side input stage, timed on its own: raw uint8 [B,4,512,512,3] -> [B,12,300,300] fp32.  The
    # headline keeps the reference's split (resize in the loader, outside the timed region).
    stage_info, with_stage = None, None
    if not a.no_input_stage and world == 1:
        import numpy as np
        from gssd.input_stage import DeviceInputStage
        one = synth.synth_study_u8(gd.shard_seed(100, rank), 4, 512)
        raw = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(one, (B,) + one.shape))).to(dev)
        stage = DeviceInputStage(300, (49., 49., 49.), True)
        xs = stage(raw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            stage(raw, out=xs)
        e1.record()
        torch.cuda.synchronize()
        sms = e0.elapsed_time(e1) / 10
        sby = raw.numel() + xs.numel() * 4
        stage_info = dict(ms_per_batch=round(sms, 4), alg_bytes=sby, alg_gbs=round(sby / sms / 1e6, 1),
                          frac_hbm_peak=round(sby / sms / 1e6 / PEAK_HBM_GBS, 4), dtype='u8',
                          note='Pillow-exact 8-bit bicubic 512->300 + mean + min-max + [B,12,300,300] pack')
        # The metric's "512x512 4-phase CT in", literally: ONE timed region per step = uint8 studies [B,4,512,512,3] resident
        # in HBM -> device input stage -> forward -> MultiBoxLoss.  The headline `value` keeps SURVEY 8(d)'s split (resize outside the region).
        if a.dtype == 'f32':
            x_keep = x.clone()

            def step_in():
                stage(raw, out=x)                      # writes the net's own input buffer: the captured plan reads the same pointer
                with torch.no_grad():
                    return crit(net(x), tg)
            for _ in range(3):
                step_in()
            gd.barrier(dev)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                li, ci = step_in()
            gd.barrier(dev)
            dti = time.perf_counter() - t0
            with_stage = dict(value=round(B * a.steps / dti, 2), unit='img/s', ms_per_step=round(1e3 * dti / a.steps, 3), steps=a.steps,
                              loss=[round(float(li), 5), round(float(ci), 5)],
                              note='uint8 [B,4,512,512,3] in HBM -> input stage -> forward -> MultiBoxLoss in ONE timed region (one synthetic study x B)')
            x.copy_(x_keep)
            del x_keep
        del raw, xs

    secondary = None
    if world == 1 and not a.no_secondary and a.config == 'gssdpp':
        sres = measure('gssd', a, dev, gd, rank, world, a.dtype)[0]
        secondary = dict(metric='512x512 4-phase CT img/s (fwd+loss)', unit='img/s', **sres)
        torch.cuda.empty_cache()

    pixellink = None
    if world == 1 and not a.no_secondary and a.config == 'gssdpp' and a.dtype == 'f32':
        pixellink = measure_pixellink(B, dev)
        torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and a.cpu_sample > 0:
        cpu = cpu_baseline(a.config, a.cpu_sample, gd.shard_seed(100, rank), twin='gssd' if (a.config == 'gssdpp' and not a.no_secondary) else None)
        if a.cpu_sample == B:
            # same inputs and same starting state on both sides: the GPU's first step against the CPU oracle, at the full batch
            cpu['gpu_vs_cpu_loss_rel'] = [round(abs(loss[i] - cpu['loss'][i]) / max(abs(cpu['loss'][i]), 1e-12), 7) for i in (0, 1)]

    line = {
        'metric': '512x512 4-phase CT img/s (fwd+loss)', 'value': res['value'], 'unit': 'img/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': res['ms_per_step'],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
        'config': {'workload': res['workload'], 'workload_short': WORKLOAD_SHORT[a.config] + (', bf16 storage' if a.dtype == 'bf16' else ''),
                   'batch_per_gpu': B, 'global_batch': B * world, 'priors': 8732,
                   'alg_gflop_per_img': res['alg_gflop_per_img'], 'alg_mb_per_img': res['alg_mb_per_img']},
        'rccl_ranks': gd.world_size(), 'collective_backend': backend if world > 1 else None,
        'per_rank_ms_per_step': res['per_rank_ms_per_step'], 'host_enqueue_ms_per_step': res['host_enqueue_ms_per_step'],
        'launcher': ('self (python bench.py --gpus N)' if os.environ.get('GSSD_BENCH_SELF_LAUNCHED') else
                     'torch.distributed.run' if world > 1 else 'single process'),
        'whole_path': res['whole_path'], 'steady': res['steady'], 'loss': res['loss'],
        'first_step_loss': res['first_step_loss'],
        'roofline': res['roofline'], 'roofline_hbm_trunk': res['roofline_hbm_trunk'], 'trunk': res['trunk'], 'kernels': res['kernels'],
        'cpu_baseline': cpu, 'bf16': bf16, 'secondary': secondary, 'pixellink': pixellink,
        'full_step': None, 'input_stage': stage_info, 'with_input_stage': with_stage,
    }
    printed = threading.Lock()

    def emit(code=None, full_step=None):
        """Print the line once.  `full_step` (the watchdog's error object) is stored under the same lock that guards the print, so
        the main thread and the watchdog cannot overwrite each other's result."""
        if printed.acquire(blocking=False):
            if full_step is not None:
                line['full_step'] = full_step
            if rank == 0:
                # the full record (per-kernel tables, per-layer trunk table, notes) goes to the detail file; stdout carries ONE short line
                dp = write_detail(line, a.detail)
                print(json.dumps(compact_line(line, dp and os.path.relpath(dp, ROOT))), flush=True)
            if code is not None:
                os._exit(code)

    # BASELINE.json configs[3]: K full training steps, LAST (the only leg with a data-path collective).  A collective that hangs
    # must not cost the line its measured headline: past the deadline every rank's watchdog prints the line (with the error in
    # `full_step`) and exits NON-ZERO (3), which self_launch() / the launcher propagate -- a hung collective is not a success.
    if a.full_step > 0:
        wd = None
        if world > 1:
            def on_timeout():
                emit(3, {'error': f'full-step leg did not finish within {a.full_step_timeout} s (RCCL all-reduce over '
                                  f'{world} ranks); the fwd+loss metric above is complete; exit code 3'})
            wd = threading.Timer(a.full_step_timeout, on_timeout)
            wd.daemon = True
            wd.start()
        try:
            fs = full_step_leg(a, net, crit, x, tg, dev, gd, world, B)
        except Exception as e:                                  # noqa: BLE001 -- reported in the line, never swallowed silently
            fs = {'error': f'{type(e).__name__}: {e}'[:400]}
        if wd is not None:
            wd.cancel()
        with printed:                                           # (the watchdog may be printing right now: then it also exits)
            line['full_step'] = fs
    emit()
    gd.finish()


if __name__ == '__main__':
    main()
