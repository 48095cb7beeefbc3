"""Run named rows of tools/layers_isolated.py a few times (for rocprofv3 --pmc).  usage: run_micro_case.py <substring>..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import layers_isolated as li
for name, flop, nbytes, make in li.ROWS:
    if any(p in name for p in sys.argv[1:]):
        fn = make()
        for _ in range(5):
            fn()
torch.cuda.synchronize()
