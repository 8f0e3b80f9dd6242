import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpirt_amd import _lib
lib = _lib.lib()
print("lib devices", lib.gpirt_device_count())
_lib.check(lib.gpirt_set_device(0))
import torch
print("torch sees", torch.cuda.device_count(), torch.cuda.is_available())
torch.cuda.set_device(0)
print("ok", torch.zeros(3, device="cuda").sum().item())
