#!/usr/bin/env python3
"""Two identically seeded agents run K pipelined updates each on the same batches (in-kernel Philox noise): every
parameter must end bit-identical.  A stale or torn read in the column-split scans' all-gathers (csrc/scan_cs.hip),
or any other race, shows up as a mismatch.  usage: determinism_soak.py [K] [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from repo_amd.algorithms.repo import RePo
from repo_amd.common.utils import set_gpu_mode

K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 50
set_gpu_mode(True)
batches = [tuple(torch.from_numpy(x).cuda() for x in bench.synthetic_batch(100 + i, B, 6)) for i in range(4)]
finals = []
for run in range(2):
    torch.manual_seed(0)
    agent = RePo(bench.config("repo", B), bench.Env(6), bench.Env(6), bench.NullLogger())
    agent.seed_noise(1234)
    for i in range(K):
        agent.update(batches[i % 4], join=False)
    agent.synchronize()
    torch.cuda.synchronize()
    finals.append([o.flat.clone() for o in (agent.model_optimizer, agent.actor_optimizer, agent.value_optimizer)]
                  + [agent.log_beta.clone().reshape(1)])
    print(f"run {run}: {K} updates at B={B}, last scalars finite: {all(v == v for v in agent.last_scalars.values())}", flush=True)
ok = all(torch.equal(a, b) for a, b in zip(*finals))
print("bit-identical parameters after both runs:", ok)
sys.exit(0 if ok else 1)
