"""Per-queue timeline and critical path of ONE forward + loss step from a rocprofv3 --kernel-trace CSV.

usage: python3 scripts/critical_path.py <..._kernel_trace.csv> [label] [steps-from-the-end]

A step = everything between two consecutive launches of the input pack kernel (the first launch of the plan; the steps before the
timed ones include captures and eager passes, so the LAST complete steps of the trace are used and averaged).  hipGraph replays map
the plan's streams (trunk, six source branches, spectral norm) onto the device's hardware queues; the trace has no dependency
edges, so the critical path is reconstructed the usual way: walk back from the kernel that ends last; the predecessor of a kernel is
the kernel -- on any queue -- whose end is the latest one not after this kernel's start (the one whose completion released it, or
the previous kernel of its own queue).  Time between a predecessor's end and the successor's start is a gap (launch latency / graph
edge / barrier packet).  Output: wall time of the step, busy time and kernel count per queue, the critical path aggregated by kernel
name (ms and launches on the path, ms of gaps) and the time the chip spent with 1 / 2 / 3+ kernels in flight.
"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_]+)(<[^(]*>)?', name)
    if not m:
        return name[:60]
    base, targs = m.group(1), m.group(2) or ''
    targs = re.sub(r'\s+', '', targs)
    if len(targs) > 28:
        targs = targs[:28] + '..>'
    return base + targs


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append(dict(q=r['Queue_Id'], name=r['Kernel_Name'], s=int(r['Start_Timestamp']), e=int(r['End_Timestamp'])))
    rows.sort(key=lambda r: r['s'])
    return rows


def steps_of(rows):
    marks = [i for i, r in enumerate(rows) if 'pack_input' in r['name']]
    return [(rows[a:b]) for a, b in zip(marks[:-1], marks[1:])]


def analyse(step):
    t0 = min(r['s'] for r in step)
    t1 = max(r['e'] for r in step)
    wall = (t1 - t0) / 1e6
    perq = collections.OrderedDict()
    for r in step:
        q = perq.setdefault(r['q'], [0, 0.0])
        q[0] += 1
        q[1] += (r['e'] - r['s']) / 1e6
    # critical path
    by_end = sorted(step, key=lambda r: r['e'])
    ends = [r['e'] for r in by_end]
    import bisect
    cur = by_end[-1]
    path, gaps = [], 0.0
    while True:
        path.append(cur)
        i = bisect.bisect_right(ends, cur['s']) - 1
        # tolerate 2 us of overlap between a kernel's recorded end and its successor's start
        j = bisect.bisect_right(ends, cur['s'] + 2000) - 1
        cand = None
        for k in range(j, -1, -1):
            if by_end[k] is not cur and by_end[k]['s'] < cur['s']:
                cand = by_end[k]
                break
        if cand is None:
            break
        gaps += max(0, cur['s'] - cand['e']) / 1e6
        cur = cand
        del i
    agg = collections.OrderedDict()
    for r in path:
        a = agg.setdefault(short(r['name']), [0, 0.0])
        a[0] += 1
        a[1] += (r['e'] - r['s']) / 1e6
    # concurrency histogram
    ev = []
    for r in step:
        ev.append((r['s'], 1))
        ev.append((r['e'], -1))
    ev.sort()
    conc = collections.Counter()
    n, last = 0, ev[0][0]
    for t, d in ev:
        conc[min(n, 3)] += (t - last) / 1e6
        last = t
        n += d
    tot = collections.OrderedDict()
    for r in step:
        a = tot.setdefault(short(r['name']), [0, 0.0])
        a[0] += 1
        a[1] += (r['e'] - r['s']) / 1e6
    return dict(wall=wall, perq=perq, path=agg, gaps=gaps, conc=conc, n=len(step), tot=tot, path_len=len(path))


def main():
    path = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else path
    last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    rows = load(path)
    steps = steps_of(rows)
    if not steps:
        print('no input-pack kernel in the trace')
        return
    # the steady steps: the last `last_n` whose launch count equals the mode of the last 2 * last_n
    tail = steps[-2 * last_n:]
    mode = collections.Counter(len(s) for s in tail).most_common(1)[0][0]
    use = [s for s in tail if len(s) == mode][-last_n:]
    res = [analyse(s) for s in use]
    k = len(res)
    mean = lambda f: sum(f(r) for r in res) / k
    print(f'== {label}: {k} steady steps of {mode} kernel launches each (trace: {len(rows)} dispatches, {len(steps)} steps)')
    print(f'step wall time (first kernel start -> last kernel end)   {mean(lambda r: r["wall"]):8.3f} ms')
    ksum = mean(lambda r: sum(v[1] for v in r['tot'].values()))
    print(f'kernel time summed over all queues                       {ksum:8.3f} ms')
    pk = mean(lambda r: sum(v[1] for v in r['path'].values()))
    print(f'critical path: kernels {pk:8.3f} ms in {mean(lambda r: r["path_len"]):.0f} launches + gaps {mean(lambda r: r["gaps"]):6.3f} ms')
    print('chip time with N kernels in flight: ' + '  '.join(
        f'{n if n < 3 else "3+"}: {mean(lambda r, n=n: r["conc"].get(n, 0.0)):.3f} ms' for n in range(4)))
    print('per hardware queue (launches, busy ms):')
    qs = sorted({q for r in res for q in r['perq']})
    for q in qs:
        print(f'  queue {q}: {mean(lambda r, q=q: r["perq"].get(q, [0, 0])[0]):6.1f} launches  {mean(lambda r, q=q: r["perq"].get(q, [0, 0.0])[1]):8.3f} ms busy')
    print('critical path by kernel (launches on the path, ms on the path | launches per step, ms per step over all queues):')
    names = collections.OrderedDict()
    for r in res:
        for n_, v in r['path'].items():
            a = names.setdefault(n_, [0.0, 0.0])
            a[0] += v[0] / k
            a[1] += v[1] / k
    tot = collections.OrderedDict()
    for r in res:
        for n_, v in r['tot'].items():
            a = tot.setdefault(n_, [0.0, 0.0])
            a[0] += v[0] / k
            a[1] += v[1] / k
    for n_, v in sorted(names.items(), key=lambda kv: -kv[1][1]):
        t = tot.get(n_, [0, 0])
        print(f'  {n_:58s} {v[0]:6.1f} {v[1]:8.3f} | {t[0]:6.1f} {t[1]:8.3f}')
    off = [(n_, t) for n_, t in tot.items() if n_ not in names]
    if off:
        print('kernels never on the critical path (launches, ms per step):')
        for n_, t in sorted(off, key=lambda kv: -kv[1][1]):
            print(f'  {n_:58s} {t[0]:6.1f} {t[1]:8.3f}')


if __name__ == '__main__':
    main()
