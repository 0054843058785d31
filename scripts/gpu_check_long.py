"""Long-horizon parity + timing: python scripts/gpu_check_long.py [B] (MPCX_STAGEWISE selects the kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scripts.gpu_check import run
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
print(torch.cuda.get_device_name(0), "MPCX_STAGEWISE =", os.environ.get("MPCX_STAGEWISE"))
for name, T in (("dsr_ackermann", 30), ("dsr_diff", 30), ("diff_drive", 12), ("dsr_crab", 32), ("dsr_ackermann", 40), ("sim_ackermann", 8), ("diff_drive", 1)):
    run(name, B, T=T)
