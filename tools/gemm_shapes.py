"""List the dense GEMM calls of one update (shape, count) to see where the head time goes."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from repo_amd import ops, _lib

def main():
    from repo_amd.algorithms.repo.repo import RePo
    agent = RePo(bench.config("repo"), bench.Env(), bench.Env(), bench.NullLogger())
    batch = tuple(torch.from_numpy(x).cuda() for x in bench.synthetic_batch(1234))
    agent.update(batch)
    L = _lib.lib()
    cnt = collections.Counter()
    og, ow = L.repo_gemm, L.repo_gemm_wgrad
    def g(ta, tb, M, N, K, *a):
        cnt[("gemm", "t" if ta else "n", "t" if tb else "n", M, N, K)] += 1
        return og(ta, tb, M, N, K, *a)
    def w(M, N, K, *a):
        cnt[("wgrad", "", "", M, N, K)] += 1
        return ow(M, N, K, *a)
    L.repo_gemm, L.repo_gemm_wgrad = g, w
    agent.update(batch)
    torch.cuda.synchronize()
    tot = 0
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1] * kv[0][3] * kv[0][4] * kv[0][5]):
        fl = 2.0 * k[3] * k[4] * k[5] * v
        tot += fl
        print(f"{v:3d} x {k[0]:5s} {k[1]}{k[2]} M={k[3]:6d} N={k[4]:5d} K={k[5]:5d}  {fl/1e9:7.2f} GF")
    print("total GF", tot / 1e9)
main()
