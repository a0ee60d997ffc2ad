#!/usr/bin/env python3
"""Offline score of the Gram kernel's tile orders (csrc/xtx.hip: xtx_tile_order): distinct 256-channel panels per
32-entry chunk (= what the 32 workgroups of one XCD stage together: every further use of a panel is an L2 hit) and per
round of 256 entries (= what the whole chip streams together: decides how much of the L2 misses the Infinity Cache can
serve).  usage: xtx_tile_order_eval.py [nt ...]   (nt = K / 256)"""
import sys


def order_r12(nt):
    tab = []
    for bi in range(0, nt, 16):
        for bj in range(0, min(bi + 16, nt), 16):
            for si in range(bi, min(bi + 16, nt), 4):
                for sj in range(bj, min(bj + 16, nt), 8):
                    for ti in range(si, min(si + 4, nt)):
                        for tj in range(sj, min(sj + 8, ti + 1)):
                            tab.append((ti, tj))
    return tab


def order_pairs(nt, m=8):
    nb = (nt + m - 1) // m
    tab = []

    def blk(a, b):
        r0, r1, c0, c1 = a * m, min(nt, a * m + m), b * m, min(nt, b * m + m)
        if a == b:
            tab.extend((ti, tj) for ti in range(r0, r1) for tj in range(c0, ti + 1))
        else:
            for si in range(r0, r1, 4):
                tab.extend((ti, tj) for ti in range(si, min(si + 4, r1)) for tj in range(c0, c1))

    for a in range(0, nb, 2):
        if a + 1 < nb:
            for b in range(0, a + 2):
                if b <= a:
                    blk(a, b)
                blk(a + 1, b)
        else:
            for b in range(0, a + 1):
                blk(a, b)
    return tab


def order_aligned(nt, m=8):
    """xtx_tile_order_aligned: every piece a multiple of 32 entries; 4 left-over tiles per diagonal triangle and a
    ragged last block row at the end."""
    nb = (nt + m - 1) // m
    ragged = nt % m != 0
    nfull = nb - 1 if ragged else nb
    tab, left = [], []

    def blk(a, b):
        r0, r1, c0, c1 = a * m, min(nt, a * m + m), b * m, min(nt, b * m + m)
        for si in range(r0, r1, 4):
            tab.extend((ti, tj) for ti in range(si, min(si + 4, r1)) for tj in range(c0, c1))

    def tri(a):
        r0, r1 = a * m, min(nt, a * m + m)
        t = [(ti, tj) for ti in range(r0, r1) for tj in range(r0, ti + 1)]
        if r1 - r0 == m:
            tab.extend(t[:32])
            left.extend(t[32:])
        else:
            left.extend(t)

    for a in range(0, nfull, 2):
        if a + 1 < nfull:
            for b in range(0, a):
                blk(a, b)
                blk(a + 1, b)
            blk(a + 1, a)
            tri(a)
            tri(a + 1)
        else:
            for b in range(0, a):
                blk(a, b)
            tri(a)
    if ragged:
        for b in range(0, nb - 1):
            blk(nb - 1, b)
        tri(nb - 1)
    return tab + left


def panels(tab, size):
    return sum(len({x for t in tab[c:c + size] for x in t}) for c in range(0, len(tab), size))


for nt in [int(x) for x in sys.argv[1:]] or [16, 32, 56, 112]:
    for name, tab in (("round 1/2", order_r12(nt)), ("pairs    ", order_pairs(nt)), ("aligned  ", order_aligned(nt))):
        assert sorted(tab) == [(i, j) for i in range(nt) for j in range(i + 1)]
        c = panels(tab, 32)
        print(f"nt={nt:4d} {name}: panels per 32-chunk, summed {c:5d} (best-case L2 hit {1 - c / (2 * len(tab)):.3f}); "
              f"per round of 256, summed {panels(tab, 256):5d}")
