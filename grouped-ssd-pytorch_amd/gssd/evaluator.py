"""AP / IoBB evaluation on the device: consumes ``Detect`` outputs ``[B, 2, top_k, 5]`` as they come
off the test-phase network and reproduces ``test_net`` / ``voc_ap`` (ssd_liverdet/test_ap_iobb.py:231-328, :10-41).

    ev = DeviceEvaluator(thresh=0.05, ap_list=[0.5], iobb_list=[0.1], use_07_metric=True)
    for batch: ev.add_batch(det, scales, gts)          # det on the GPU; gts = list of [n_i, 4] pixel boxes
    ap, iobb = ev.result()                             # two lists of Python floats
"""
import numpy as np
import torch

from . import _lib
from ._lib import lib, check

MAX_GT = 256


class DeviceEvaluator:
    def __init__(self, thresh=0.05, ap_list=(0.5,), iobb_list=(0.1,), use_07_metric=True):
        self.thresh = float(thresh)
        self.ap_list, self.iobb_list = [float(t) for t in ap_list], [float(t) for t in iobb_list]
        self.use_07 = bool(use_07_metric)
        if not 0 < len(self.ap_list) + len(self.iobb_list) <= 8:
            raise ValueError('between 1 and 8 thresholds in ap_list + iobb_list')
        self._conf, self._flags, self.npos, self._thr = [], [], 0, None

    def add_batch(self, det, scales, gts):
        """``det``: CUDA fp32 ``[B, C, top_k, 5]`` (class 1 is evaluated, test_ap_iobb.py:128); ``scales``: ``[B, 4]`` =
        (W, H, W, H) of the original images; ``gts``: B arrays ``[n_i, 4]`` (pixels)."""
        if not (isinstance(det, torch.Tensor) and det.is_cuda):
            raise _lib.GssdError('evaluator: detections must be a CUDA tensor (no CPU fallback)')
        det = det.contiguous().float()
        B, Cc, K, five = det.shape
        assert five == 5 and Cc >= 2 and len(gts) == B
        dev = det.device
        counts = [int(np.asarray(g).reshape(-1, 4).shape[0]) for g in gts]
        if max(counts, default=0) > MAX_GT:
            raise _lib.GssdError(f'more than {MAX_GT} ground-truth boxes in one image')
        off = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=dev)
        flat = np.concatenate([np.asarray(g, np.float64).reshape(-1, 4) for g in gts]) if sum(counts) else np.zeros((1, 4))
        gt = torch.from_numpy(np.ascontiguousarray(flat)).to(dev)
        sc = torch.as_tensor(np.asarray(scales, np.float32).reshape(B, 4)).to(dev)
        if self._thr is None or self._thr.device != dev:
            self._thr = torch.tensor(self.ap_list + self.iobb_list, dtype=torch.float64, device=dev)
        nm = len(self.ap_list) + len(self.iobb_list)
        conf = torch.empty(B * K, dtype=torch.float32, device=dev)
        flags = torch.empty(nm, B * K, dtype=torch.uint8, device=dev)
        cls1 = det[:, 1]                                       # view: image stride C*K*5 floats
        check(lib.gssd_eval_match(cls1.data_ptr(), Cc * K * 5, B, K, sc.data_ptr(), gt.data_ptr(), off.data_ptr(),
                                  max(counts, default=0), self.thresh, self._thr.data_ptr(), len(self.ap_list),
                                  len(self.iobb_list), conf.data_ptr(), flags.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream))
        self._keep = (det, gt, off, sc)                        # alive until the launch has been ordered behind later work
        self._conf.append(conf)
        self._flags.append(flags)
        self.npos += sum(counts)
        return conf, flags

    def result(self):
        nm = len(self.ap_list) + len(self.iobb_list)
        if not self._conf:
            return [0.] * len(self.ap_list), [0.] * len(self.iobb_list)
        conf = torch.cat(self._conf)
        flags = torch.cat(self._flags, dim=1).contiguous()
        M = conf.numel()
        wb = int(lib.gssd_eval_workspace_bytes(M))
        work = torch.empty(wb, dtype=torch.uint8, device=conf.device)
        ap = torch.empty(nm, dtype=torch.float64, device=conf.device)
        check(lib.gssd_eval_ap(conf.data_ptr(), flags.data_ptr(), M, nm, float(self.npos), int(self.use_07), work.data_ptr(), wb,
                               ap.data_ptr(), torch.cuda.current_stream().cuda_stream))
        out = ap.cpu().tolist()
        return out[:len(self.ap_list)], out[len(self.ap_list):]
