"""Per-kernel timing of the device input stage at B = 32 (raw uint8 [32,4,512,512,3] -> [32,12,300,300] fp32)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'grouped-ssd-pytorch_amd'))
import numpy as np, torch
from gssd import synth
from gssd.input_stage import DeviceInputStage
dev = torch.device('cuda:0')
one = synth.synth_study_u8(1, 4, 512)
raw = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(one, (32,) + one.shape))).to(dev)
st = DeviceInputStage(300, (49., 49., 49.), True)
x = st(raw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): st(raw, out=x)
e1.record(); torch.cuda.synchronize()
print(f'input stage: {e0.elapsed_time(e1) / 20:.3f} ms per batch of 32')
