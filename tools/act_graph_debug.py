"""Acting-path graph replays vs eager, with / without poisoned torch.empty: magnitude of the returned posterior state."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tests.conftest import _poisoned  # noqa: E402
from tests.test_update_gpu import make_agent  # noqa: E402

real = (torch.empty, torch.empty_like)
if os.environ.get("POISON", "1") == "1":
    torch.empty, torch.empty_like = _poisoned(torch.empty), _poisoned(torch.empty_like)

agent, cfg = make_agent("repo", 6, 3, 4, 6)
rs = np.random.RandomState(0)
frame = torch.from_numpy(rs.uniform(-0.5, 0.5, (1, 3, 64, 64)).astype(np.float32)).cuda()
zero = agent.init_latent_and_action()
nz = (torch.full((1, 200), 0.1, device="cuda"), torch.full((1, 30), 0.5, device="cuda"), torch.full((1, 6), 0.3, device="cuda"))
for explore in (False, True):
    for name, lat in (("zero", zero), ("nonzero", nz), ("zero", zero), ("nonzero", nz), ("nonzero", nz)):
        out = agent.update_latent_and_select_action(*lat, frame, explore)
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = agent._act_eager(*lat, frame, explore)
        torch.cuda.synchronize()
        print(f"explore={explore} in={name:8s} graph: belief max {float(out[0].abs().max()):.4g} state max {float(out[1].abs().max()):.4g}"
              f" | eager: belief max {float(ref[0].abs().max()):.4g} state max {float(ref[1].abs().max()):.4g}"
              f" | belief equal {bool(torch.equal(out[0], ref[0]))}", flush=True)
