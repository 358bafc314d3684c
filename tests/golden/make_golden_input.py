"""Golden vectors for the input stage: the reference's ``base_transform_fast`` / ``ResizeFast``
(data/__init__.py:33-54, utils/augmentations.py:506-516), which call Pillow, run here on seeded uint8 study slices.

    python tests/golden/make_golden_input.py     # rewrites tests/golden/input.npz  (build container only)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, sha, sample_idx, synth      # noqa: E402


def main():
    import PIL
    R = import_reference()
    from utils.augmentations import ResizeFast
    d = {'pillow_version': np.frombuffer(PIL.__version__.encode(), dtype=np.uint8)}
    mean = np.array([49., 49., 49.], np.float32)
    # small case stored in full
    small = synth.synth_study_u8(20260, 4, 64)
    d['small_in'] = small
    d['small_out_norm'] = R.data.base_transform_fast(small, 37, mean, use_normalize=True).astype(np.float32)
    d['small_out_raw'] = R.data.base_transform_fast(small, 37, mean, use_normalize=False).astype(np.float32)
    up = synth.synth_study_u8(20261, 4, 20)
    d['up_in'] = up
    d['up_out_norm'] = R.data.base_transform_fast(up, 33, mean, use_normalize=True).astype(np.float32)
    # the real geometry: 512 -> 300, stored as sha256 + samples (the input is regenerated from the seed by the test)
    big = synth.synth_study_u8(777, 4, 512)
    out = R.data.base_transform_fast(big, 300, mean, use_normalize=True).astype(np.float32)
    d['big_in_sha'] = np.frombuffer(bytes.fromhex(sha(big)), dtype=np.uint8)
    d['big_out_sha'] = np.frombuffer(bytes.fromhex(sha(out)), dtype=np.uint8)
    idx = sample_idx(out.size, 2048)
    d['big_sample_idx'] = idx
    d['big_sample'] = out.reshape(-1)[idx]
    # training-side ResizeFast on an already normalised float image
    xf = (small.astype(np.float32) - 49.)
    xf = (xf - xf.min()) / (xf.max() - xf.min())
    rz, _, _ = ResizeFast(37)(xf.copy(), None, None)
    d['resizefast_in'] = xf
    d['resizefast_out'] = rz.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'input.npz'), **d)
    print('input.npz written; pillow', PIL.__version__, 'big sha', sha(out)[:16])


if __name__ == '__main__':
    main()
