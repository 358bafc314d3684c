"""Golden vectors for the AP / IoBB evaluator: the reference's own ``test_net``
(ssd_liverdet/test_ap_iobb.py:231-328) driven with a fake dataset / transform / network that replay seeded synthetic
Detect outputs, so the numbers stored are what the reference computes from them.

    python tests/golden/make_golden_eval.py     # rewrites tests/golden/eval.npz  (build container only)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, synth      # noqa: E402


def main():
    import_reference()
    import test_ap_iobb as T
    d = {}
    for case, (N, seed, S) in enumerate(((40, 5, 512), (7, 6, 300))):
        det, gts = synth.synth_detections(N, seed=seed, size=S)
        state = {'i': 0}

        class FakeSet:
            name = 'lesion_test_ap_synth'

            def __len__(self):
                return N

            def pull_image(self, idx):
                return np.zeros((4, S, S, 3), np.uint8)

            def pull_anno(self, idx):
                g = gts[idx]
                return np.concatenate([g, np.zeros((g.shape[0], 1))], axis=1)

        def transform(img):
            return (np.zeros((4, 8, 8, 3), np.float32),)

        def net(x):
            i = state['i']
            state['i'] += 1
            return torch.from_numpy(det[i:i + 1])

        d[f'c{case}_det'] = det
        d[f'c{case}_gt'] = np.concatenate(gts)
        d[f'c{case}_gt_n'] = np.array([g.shape[0] for g in gts])
        d[f'c{case}_size'] = np.array(S)
        for use07 in (True, False):
            state['i'] = 0
            ap, iobb = T.test_net(net, False, FakeSet(), transform, 300, thresh=0.05, mode='v2', use_07_metric=use07,
                                  ap_list=[0.1, 0.5], iobb_list=[0.1, 0.5])
            d[f'c{case}_ap_{int(use07)}'] = np.array(ap, np.float64)
            d[f'c{case}_iobb_{int(use07)}'] = np.array(iobb, np.float64)
            print(case, use07, ap, iobb)
    np.savez_compressed(os.path.join(HERE, 'eval.npz'), **d)


if __name__ == '__main__':
    main()
