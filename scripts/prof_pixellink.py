import sys, os, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')); sys.path.insert(0, ROOT)
import torch, bench
print(json.dumps(bench.measure_pixellink(32, torch.device('cuda:0'), steps=10)))
