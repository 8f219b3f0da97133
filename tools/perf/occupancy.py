import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from layout2img_amd import _lib
torch.zeros(1, device="cuda")
lib = _lib.load()
for which in (0, 10, 12):
    for lds in (32768, 49152, 65536, 70000, 79872, 81920, 98304, 131072):
        print(which, lds, lib.l2i_debug_occupancy(which, lds))
p = torch.cuda.get_device_properties(0)
print(p)
