"""Bank analysis of k_group8m's tile updates. A tile update is a ds_read_b32 / ds_write_b32: the
LDS serves it in two groups of 32 lanes, {0-31} and {32-63}, on 32 banks of 4 bytes (MI355X_MICROARCH.md, LDS); an
extra distinct address on a busy bank costs one more LDS cycle. For both lane maps of pass B this prints the conflict
cycles per step round of the strides in use and every conflict-free (row stride mod 32, plane stride mod 32).
   python tools/lds_banks.py"""
from collections import defaultdict


def extra_cycles(addrs):
    banks = defaultdict(set)
    for a in addrs:
        banks[a % 32].add(a)
    return max(len(v) for v in banks.values()) - 1


def separable(rwp, plane):
    """lane = 16 * column g4 + 4 * plane spl + row si; pixels: rows si / 7 - si, columns g4 / 4 + g4 (k_group8m.h: poff)"""
    tot = 0
    for kk in range(4):
        for half in range(2):
            ad = []
            for lane in range(32 * half, 32 * half + 32):
                si, spl, g4 = lane & 3, (lane >> 2) & 3, lane >> 4
                r = (7 - si) if kk & 2 else si
                ad.append(spl * plane + r * rwp + 4 * (kk & 1) + g4)
            tot += extra_cycles(ad)
    return tot


def kronecker(rwp, plane):
    """lane = 16 * plane g4 + folded pixel (pi, pj); pixels: rows pi / 7 - pi, columns pj / 7 - pj"""
    tot = 0
    for kk in range(4):
        for half in range(2):
            ad = []
            for lane in range(32 * half, 32 * half + 32):
                lo, g4 = lane & 15, lane >> 4
                pi, pj = lo >> 2, lo & 3
                ad.append(g4 * plane + ((7 - pi) if kk & 2 else pi) * rwp + ((7 - pj) if kk & 1 else pj))
            tot += extra_cycles(ad)
    return tot


if __name__ == "__main__":
    print("round 5 (both lane maps on the Kronecker strides: row 28, plane 624): separable", separable(28, 624),
          "extra cycles per 8 accesses, Kronecker", kronecker(28, 624))
    print("round 6 (separable lane map on row 26, plane 584):", separable(26, 584))
    for name, f in (("separable", separable), ("Kronecker", kronecker)):
        print(f"{name}: conflict-free (row stride mod 32 -> plane strides mod 32)")
        for r in range(32):
            ps = [p for p in range(32) if f(r + 32, p + 640) == 0]
            if ps:
                print(f"   {r:2d} -> {ps}")
