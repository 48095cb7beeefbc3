#!/usr/bin/env python3
"""Where the world-model lane spends its time INSIDE the pipelined update (no profiler: HIP events recorded on the
stream each phase is enqueued on, before and after it): encoder forward, observe scan, decoder forward + NLL,
heads / KL, decoder backward (main-stream part), reverse scan (side stream), encoder backward, optimiser step --
start and end of each phase relative to the update's first kernel, averaged over K steady updates."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    from repo_amd import functional as Fn
    from repo_amd import ops
    from repo_amd.algorithms.repo import dreamer as D
    from repo_amd.algorithms.repo.repo import RePo

    torch.manual_seed(0)
    B = int(os.environ.get("PHASE_B", "50"))   # PHASE_B=7: one rank's shard of the strong-scaling job
    agent = RePo(bench.config("repo", B), bench.Env(), bench.Env(), bench.NullLogger())
    batch = tuple(torch.from_numpy(x).cuda() for x in bench.synthetic_batch(1234, B))
    marks = []

    def wrap(mod, name, label):
        orig = getattr(mod, name)

        def f(*a, **k):
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            r = orig(*a, **k)
            e1.record(torch.cuda.current_stream())
            marks.append((label, e0, e1))
            return r

        setattr(mod, name, f)

    wrap(Fn, "encoder_fwd", "encoder fwd")
    wrap(ops, "rssm_observe_fwd", "observe scan fwd")
    wrap(Fn, "decoder_fwd_nll", "decoder fwd + NLL")
    wrap(ops, "kl_balance", "KL")
    wrap(Fn, "decoder_bwd", "decoder bwd (main stream part, joined)")
    wrap(ops, "rssm_observe_bwd", "observe scan bwd (side stream)")
    wrap(Fn, "encoder_bwd", "encoder bwd (joined)")
    wrap(ops, "rssm_imagine_fwd", "AC: imagine fwd")
    wrap(ops, "rssm_imagine_bwd", "AC: imagine bwd")
    wrap(ops, "tanh_normal_entropy", "AC: entropy")
    wrap(agent.model_optimizer, "clip_and_step", "model clip+Adam")
    wrap(agent.actor_optimizer, "clip_and_step", "AC: actor clip+Adam")
    assert D.Fn is Fn and D.ops is ops
    for _ in range(6):
        agent.update(batch, join=False)
    agent.synchronize()
    torch.cuda.synchronize()
    marks.clear()
    K = 10
    for _ in range(K):
        agent.update(batch, join=False)
    agent.synchronize()
    torch.cuda.synchronize()
    # split into updates at each "encoder fwd"
    starts = [i for i, m in enumerate(marks) if m[0] == "encoder fwd"]
    acc = {}
    order = []
    for ui, i0 in enumerate(starts):
        i1 = starts[ui + 1] if ui + 1 < len(starts) else len(marks)
        t0 = marks[i0][1]
        for label, e0, e1 in marks[i0:i1]:
            a, b = t0.elapsed_time(e0), t0.elapsed_time(e1)
            if label not in acc:
                acc[label] = [0.0, 0.0, 0]
                order.append(label)
            acc[label][0] += a
            acc[label][1] += b
            acc[label][2] += 1
    period = marks[starts[1]][1].elapsed_time(marks[starts[-1]][1]) / (len(starts) - 2) if len(starts) > 2 else float("nan")
    print(f"# phases of the pipelined update, ms from the update's first kernel (mean of {K}); period {period:.3f} ms")
    print(f"# {'start':>7} {'end':>7} {'dur':>7}  phase")
    for label in order:
        a, b, n = acc[label]
        print(f"  {a / n:7.3f} {b / n:7.3f} {(b - a) / n:7.3f}  {label}")


main()
