"""Phase stamps of the fused DIN forward (library built with -DDIN_TRACE): us since the sample's first stamp."""
import ctypes
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import _lib, ops
from recommendation_amd.din import DIN

dev = torch.device('cuda:0')
B, L, rows = 8192, 100, 8000000
m = DIN(rows, dev, max_len=L, seed=1)
rng = np.random.default_rng(0)
hist = torch.from_numpy(rng.integers(0, rows, size=(B, L))).to(dev)
qid = torch.from_numpy(rng.integers(0, rows, size=(B,))).to(dev)
lengths = torch.from_numpy(rng.integers(1, L + 1, size=(B,)).astype(np.int32)).to(dev)
labels = torch.from_numpy((rng.random(B) < 0.5).astype(np.float32)).to(dev)
for _ in range(3):
    m.train_step(hist, qid, lengths, labels)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 16))()
lib = _lib.load()
lib.mi_debug_din_trace.argtypes = [ctypes.c_void_p]
assert lib.mi_debug_din_trace(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(16, 4, 16).astype(np.float64) / 100.0  # 100 MHz -> us
names = ['top', 'tile built', 'barrier+prefetch issued', 'A copied', 'layer1 done', 'barrier', 'epi1+barrier', 'H1 copied', 'layer2', '...', 'end']
for wg in (0, 5):
    for it in range(4):
        r = a[wg, it]
        print("wg %d it %d: " % (wg, it) + " ".join("%.2f" % (r[k] - r[0]) for k in range(11)) + "  | since kernel start %.2f" % (r[0] - a[:, 0, 0].min()))
