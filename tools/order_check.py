import sys; sys.path.insert(0, ".")
from harkdb_amd.engine import Engine
e = Engine(0)
import torch
print("torch sees GPU after libhark:", torch.cuda.is_available(), torch.cuda.device_count())
