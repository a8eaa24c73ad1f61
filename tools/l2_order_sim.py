#!/usr/bin/env python3
"""LRU model of one XCD's L2 under the tile kernel's reads of ONE frame, for candidate orders of the tile rows
(csrc/m1v_kernels.hip, tile_row_order_for).  No GPU needed.

A tile (8 strips x 4 macroblock rows) reads its luma region — rows [64R, 64R+64), 384 bytes per row — and, the reference's
chroma quirk (include/encoder.h:347-348: the full-resolution plane addressed with stride W/2), two 64-pixel x 16-row patches
of picture rows [16R, 16R+16): every byte of the picture's top quarter is read twice, once as luma and once as chroma, and the
second read is an L2 hit only if little traffic passed in between.  The model walks the tiles of a frame in the given order
through a fully associative LRU of `cap` 128-byte lines (4 MiB = 32,768) with the accesses of `window` consecutive tiles
interleaved (tiles in flight) and prints misses / distinct lines = the read part of "HBM traffic / algorithmic bytes".

    python tools/l2_order_sim.py [W H]

Measured beside it (rocprofv3 FETCH_SIZE, tools/pmc_traffic.sh, 3840x2160): top to bottom 1.22 (run kernel), depth first
1.069, parent in the middle 1.041 (reads alone 1.028 x the pixel bytes)."""
import collections
import itertools
import sys

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) >= 3 else (3840, 2160)
NS, NM = (W & ~15) // 16, (H & ~15) // 16
TC, TR = (NS + 7) // 8, (NM + 3) // 4
PITCH = W * 3


def tile_lines(tr, tc):
    lines = []
    x0, y0 = tc * 128, tr * 64
    for y in range(y0, min(y0 + 64, NM * 16)):                      # luma
        b0 = y * PITCH + x0 * 3
        b1 = b0 + min(128, W - x0) * 3
        lines.extend(range(b0 // 128, (b1 - 1) // 128 + 1))
    for m in range(4):                                               # chroma: plane offset (y/2+i)*(W/2) + x/2
        mb = tr * 4 + m
        if mb >= NM:
            continue
        for i in range(8):
            b0 = ((mb * 8 + i) * (W // 2) + x0 // 2) * 3
            b1 = b0 + min(64, (W - x0) // 2) * 3
            lines.extend(range(b0 // 128, (b1 - 1) // 128 + 1))
    return lines


def simulate(rows, cap, window):
    order = [(r, c) for r in rows for c in range(TC)]
    cache, miss = collections.OrderedDict(), 0
    for w0 in range(0, len(order), window):
        acc = [tile_lines(tr, tc) for tr, tc in order[w0:w0 + window]]
        for chunk in itertools.zip_longest(*acc):
            for ln in chunk:
                if ln is None:
                    continue
                if ln in cache:
                    cache.move_to_end(ln)
                else:
                    miss += 1
                    cache[ln] = 1
                    if len(cache) > cap:
                        cache.popitem(last=False)
    return miss


def children(r):
    return [c for c in range(4 * r, 4 * r + 4) if 0 < c < TR]


def top_to_bottom():
    return list(range(TR))


def depth_first():                       # round 3
    order, stack = [], [0]
    while stack:
        r = stack.pop()
        order.append(r)
        stack.extend(reversed(children(r)))
    return order


def parent_in_the_middle():              # round 4: tile_row_order_for
    size = {}

    def sz(r):
        size[r] = 1 + sum(sz(c) for c in children(r))
        return size[r]
    sz(0)

    def arrange(r):
        left, right = [], []
        for i, c in enumerate(sorted(children(r), key=lambda c: size[c])):
            a = arrange(c)
            if i == 0:
                left = left + a           # smallest subtree: directly in front of r
            elif i == 1:
                right = a + right         # next: directly behind r
            elif i == 2:
                left = a + left           # the larger ones outside
            else:
                right = right + a
        return left + [r] + right
    return arrange(0)


if __name__ == "__main__":
    uniq = W * H * 3 // 128
    print(f"{W}x{H}: {TR} tile rows x {TC} tile columns, {uniq} lines of pixels")
    for name, rows in (("top to bottom", top_to_bottom()), ("depth first", depth_first()), ("parent in the middle", parent_in_the_middle())):
        print(f"{name:22s} {rows}")
        for cap, window in ((32768, 1), (30000, 8), (28000, 16)):
            print(f"    LRU of {cap} lines, {window:2d} tiles interleaved: {simulate(rows, cap, window) / uniq:.4f} x the distinct lines")
