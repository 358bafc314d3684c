"""Multi-GPU (one process per GPU, RCCL over xGMI) checks.  They need >= 2 MI355X in one box and skip cleanly on the 1-GPU test
boxes; what CAN run on one GPU (the launcher's refusal) does."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], capture_output=True, text=True, timeout=1500)
    return r.returncode, [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')], r.stderr[-2000:]


def test_bench_gpus_beyond_the_box_is_one_error_line():
    n = torch.cuda.device_count() + 1
    rc, lines, _ = _bench('--gpus', str(n))
    assert rc == 2 and len(lines) == 1 and 'error' in lines[0] and lines[0]['n_gpus'] == n


def test_bench_two_ranks_code_path_on_one_gpu():
    """The N > 1 path of bench.py on a ONE-GPU box: two self-launched ranks share GPU 0 and use gloo for the collectives (RCCL refuses
    two ranks on one device).  Everything but RCCL itself runs: rank-sharded inputs, barriers + max-over-ranks timing, the per-rank
    gather, broadcast of the weights, the overlapped gradient reducer inside the full-step leg (its all-reduces really run, on
    ranges of the flat gradient buffer, while the backward is still enqueueing), the watchdog.  Not a measurement."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--steady', '0',
                        '--no-bf16', '--full-step', '2', '--batch', '8'], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, GSSD_DIST_SAME_DEVICE='1', GSSD_DIST_BACKEND='gloo'))
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
    ln = lines[0]
    assert ln['n_gpus'] == 2 and ln['rccl_ranks'] == 2 and ln['collective_backend'] == 'gloo' and len(ln['per_rank_ms_per_step']) == 2
    assert ln['config']['global_batch'] == 16 and ln['launcher'].startswith('self')
    fs = ln['full_step']
    assert 'error' not in fs, fs
    assert fs['rccl_ranks'] == 2 and fs['allreduce_overlapped'] and 18488172 <= fs['grad_elems'] <= 18488172 + 4 * 400 and fs['allreduce_exposed_ms'] >= 0


def test_bench_eight_ranks_code_path_on_one_gpu():
    """BASELINE configs[3]'s launch shape -- EIGHT ranks -- on the one-GPU box: `python bench.py --gpus 8` starts its own
    eight processes, all on GPU 0 with gloo collectives (GSSD_DIST_SAME_DEVICE=1), batch 4 per rank.  Proves that eight ranks rendezvous,
    shard (eight different image shards), gather per-rank timings, broadcast the weights and reduce the 18.5 M gradients through the
    overlapped reducer inside a real backward; and measures what a rank's HOST side costs per step while seven other Python processes do
    the same on this host (host_enqueue_ms_per_step: fwd + loss graph replay, and the eager full training step) -- the quantity that
    bounds weak scaling on one node, since the data path has no collective (train_lesion_multiphase_v2.py:593's DataParallel replaced by
    one process per GPU).  Not a throughput measurement: the eight ranks share one GPU."""
    # (round 5 allowed a second attempt here: ~1 failure in 5 full-suite runs.  Round 6: the self-launcher's ranks meet through a FileStore in a
    # private directory instead of a loopback port picked ahead of rank 0's bind -- gssd/dist.py::init, bench.py::self_launch -- and there is
    # ONE attempt again, so a rendezvous failure is a test failure)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '2', '--steady', '0',
                        '--no-bf16', '--no-events', '--full-step', '2', '--batch', '4'], capture_output=True, text=True, timeout=2400,
                       env=dict(os.environ, GSSD_DIST_SAME_DEVICE='1', GSSD_DIST_BACKEND='gloo'))
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, f'rc {r.returncode}\n' + r.stderr[-4000:]
    ln = lines[0]
    assert ln['n_gpus'] == 8 and ln['rccl_ranks'] == 8 and ln['collective_backend'] == 'gloo' and len(ln['per_rank_ms_per_step']) == 8
    assert ln['config']['global_batch'] == 32 and ln['launcher'].startswith('self')
    assert len(ln['host_enqueue_ms_per_step']) == 8 and all(0 < t < 1e3 for t in ln['host_enqueue_ms_per_step'])
    fs = ln['full_step']
    assert 'error' not in fs, fs
    assert fs['rccl_ranks'] == 8 and fs['allreduce_overlapped'] and 18488172 <= fs['grad_elems'] <= 18488172 + 4 * 400
    assert len(fs['host_enqueue_ms_per_step']) == 8
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'eight_ranks_one_gpu.json'), 'w') as f:
            json.dump(dict(note='8 self-launched ranks of bench.py on ONE GPU over gloo, batch 4 per rank: code path + host-side cost per '
                                'rank with 8 concurrent Python processes; NOT a throughput measurement', host_cpus=os.cpu_count(),
                           fwd_loss_host_enqueue_ms_per_step=ln['host_enqueue_ms_per_step'], fwd_loss_ms_per_step=ln['per_rank_ms_per_step'],
                           full_step_host_enqueue_ms_per_step=fs['host_enqueue_ms_per_step'], full_step_ms_per_step=fs['ms_per_step'],
                           loss=ln['loss']), f, indent=1)


def test_global_normalizer_equals_single_batch(tmp_path):
    """VERDICT r5 item 7 / SURVEY.md 8e: MultiBoxLoss.global_normalizer all-reduces N (multibox_loss.py:117) over the ranks; two ranks on one GPU
    over gloo: the mean over ranks of the losses and of d(loss)/d(loc, conf) equals the single 2B-image batch."""
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE='2', RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT='29577',
                   GSSD_DIST_INIT_FILE=str(tmp_path / 'store'))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'global_n_worker.py')], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1][-2000:] + outs[1][1][-2000:]
    o = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith('GLOBALNJSON ')][-1][len('GLOBALNJSON '):])
    print(o)
    for i in (0, 1):
        assert abs(o['loss'][i] - o['ref'][i]) <= 2e-6 * abs(o['ref'][i])
    assert o['dloc_rel'] < 1e-6 and o['dconf_rel'] < 1e-6
    assert o['rank0_local'] != o['rank0_global']            # the option changes something: the two ranks' positive counts differ


def test_bench_two_gpus_self_launched_rccl():
    """`python bench.py --gpus 2` (no launcher): two ranks, RCCL barrier + the full-step leg's overlapped gradient all-reduce."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs in one box')
    rc, lines, err = _bench('--gpus', '2', '--steps', '3', '--warmup', '2', '--steady', '0', '--no-bf16', '--full-step', '2')
    assert rc == 0 and len(lines) == 1, err
    ln = lines[0]
    assert ln['n_gpus'] == 2 and ln['rccl_ranks'] == 2 and len(ln['per_rank_ms_per_step']) == 2 and ln['launcher'].startswith('self')
    fs = ln['full_step']
    assert 'error' not in fs and fs['rccl_ranks'] == 2 and fs['allreduce_overlapped'] and 18488172 <= fs['grad_elems'] <= 18488172 + 4 * 400


def test_overlapped_reducer_equals_flat_allreduce_on_rccl():
    """OverlappedGradReducer vs allreduce_grads vs the hand-computed mean, on real RCCL ranks, incl. the fallback for
    accumulated gradients."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs in one box')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', '29621', os.path.join(ROOT, 'tests', 'multi_gpu_worker.py')],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('[')][-1])
    for o in res:
        assert o['flat'] < 1e-6 and o['over'] < 1e-6 and o['acc'] < 1e-6 and not o['acc_overlapped'] and o['differs_from_local'] > 1e-3
