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
for wg in (0, 5, 11):
    r = a[wg, 0]
    print("fwd wg %d: " % wg + " ".join("%.2f" % (r[k] - r[0]) for k in range(11)))
for wg in (0, 5, 11):
    r = a[wg, 1]
    print("bwd wg %d: " % wg + " ".join("%.2f" % (r[k] - r[0]) for k in range(10)))
print("bwd marks: 0 start | 1 loads issued | 2 da + H1 in LDS + barrier | 3 dZ2 + sums | 4 barrier, part out | 5 dH1 + mask |"
      " 6 barrier + dZ1 out + S | 7 dA products | 8 barrier + dk staging + barrier | 9 dE out + dq reduce + barrier")
