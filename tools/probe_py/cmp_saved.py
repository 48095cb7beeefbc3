import torch, sys
a = torch.load(sys.argv[1]); b = torch.load(sys.argv[2])
names = ["emb", "w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4", "h1", "h2", "h3", "h4", "m1?", "m2", "m3"]
for n, x, y in zip(names, a, b):
    if x is None or y is None: continue
    if x.dtype == torch.uint8: print(n, "mask mismatches", (x != y).sum().item()); continue
    print(n, "rel l2 %.2e  max abs %.2e" % (((x - y).norm() / (y.norm() + 1e-30)).item(), (x - y).abs().max().item()))
