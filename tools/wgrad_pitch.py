#!/usr/bin/env python3
"""Bank conflicts of the weight-gradient kernel's B-fragment reads (csrc/dconv.h, dconv_wgrad_kernel) per layer and
candidate LDS pitches (BRP floats per row of `big`, BP per channel): lane li of a 32-column N tile reads
(cb - cbf) * BP + ky * BRP + kx with n = (cb, ky, kx) = n_tile + li; ds_read_b32 has 32 banks.  Prints the current
pitches' average / worst multiplicity over all N tiles and the smallest conflict-free candidates.  Round 3 tried them
(and an odd pitch for `small`): with the element-wise staging stores they need, dec3 552 -> 592 us; with NO staging
stores at all (ablation build) enc2 302 -> 291, dec3 505 -> 497, the rest unchanged -- the loop is not LDS-bound.
Not adopted."""
geos = {'enc1': (3, 32, 64, 4), 'enc2': (32, 64, 31, 4), 'enc3': (64, 128, 14, 4), 'enc4': (128, 256, 6, 4),
        'dec2': (64, 128, 13, 5), 'dec3': (32, 64, 30, 6), 'dec4': (3, 32, 64, 6)}
tiles = {'enc1': (64, 4), 'enc2': (128, 7), 'enc3': (128, 6), 'enc4': (128, 2), 'dec2': (128, 5), 'dec3': (128, 7), 'dec4': (128, 2)}  # BN, RB


def pitch4(n):
    p = (n + 3) & ~3
    return p if (p // 4) % 2 else p + 4


def ways(KS, BRP, BP, BN, NW):
    KK, tot, cnt, mx = KS * KS, 0, 0, 0
    for n0 in range(0, NW, BN):
        cbf = n0 // KK
        for t in range(BN // 32):
            banks = {}
            for li in range(32):
                n = min(n0 + t * 32 + li, NW - 1)
                cb, r = divmod(n, KK)
                a = (cb - cbf) * BP + (r // KS) * BRP + r % KS
                banks.setdefault(a % 32, set()).add(a)
            m = max(len(v) for v in banks.values())
            tot, cnt, mx = tot + m, cnt + 1, max(mx, m)
    return tot / cnt, mx


for g, (CB, CS, HB, KS) in geos.items():
    BN, RB = tiles[g]
    WB, BR, NW = HB, 2 * RB + KS - 2, CB * KS * KS
    brp0 = WB + 4 if WB % 64 == 0 else WB
    bp0 = pitch4(BR * brp0)
    cand = []
    for brp in range(WB, WB + 33):
        for pad in range(32):
            bp = ((BR * brp + 3) & ~3) + pad
            w = ways(KS, brp, bp, BN, NW)
            cand.append((round(w[0], 2), bp, brp, w[1]))
    cand.sort()
    print(f"{g}: memory pitch BRP {brp0} BP {bp0}: avg {ways(KS, brp0, bp0, BN, NW)[0]:.2f} worst {ways(KS, brp0, bp0, BN, NW)[1]}-way;"
          f" best (avg, BP, BRP, worst): {cand[:2]}")
