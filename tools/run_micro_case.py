import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.microbench import CASES
for n in sys.argv[1:]:
    fn, flop = CASES[n]()
    for _ in range(4): fn()
torch.cuda.synchronize()
