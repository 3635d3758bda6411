import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from syllable_detector_swift_amd.dist import pack_flags, unpack_flags
fl = (torch.rand((64, 127090), device="cuda") < 0.01).to(torch.uint8)
big = pack_flags((torch.rand((512, 127090), device="cuda") < 0.01).to(torch.uint8))
for _ in range(3): pack_flags(fl); unpack_flags(big, 127090)
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
e0.record()
for _ in range(20): b = pack_flags(fl)
e1.record()
for _ in range(20): u = unpack_flags(big, 127090)
e2.record(); torch.cuda.synchronize()
print("pack [64 x 127090] %.1f us   unpack [512 x 127090] %.1f us" % (e0.elapsed_time(e1) * 50, e1.elapsed_time(e2) * 50))
