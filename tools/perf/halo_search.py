# Search (pitch, swizzle) making the A-fragment ds_read_b128 of a halo-resident conv tile bank-conflict free.
import itertools
G = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]

def rows_of_tile(PW, PHs, tile_i):
    """pixel (sub-patch, pyl, px) of the 32 GEMM rows of MFMA tile `tile_i` (quad-major enumeration)."""
    out = []
    hw_shift = (PW // 2).bit_length() - 1
    for r in range(32):
        idx = tile_i * 32 + r
        q, s = idx >> 2, idx & 3
        qy, qx = q >> hw_shift, q & ((1 << hw_shift) - 1)
        py, px = 2 * qy + (s >> 1), 2 * qx + (s & 1)
        out.append((py // PHs, py % PHs, px))
    return out

def conflicts(PW, PHs, P, SUBH, swz, up2, ntiles):
    bad = 0
    for ti in range(ntiles):
        rows = rows_of_tile(PW, PHs, ti)
        for ky in range(3):
            for kx in range(3):
                for g in G:
                    hs = set()
                    for l in g:
                        sp, pyl, px = rows[l]
                        if up2:
                            hy, hx = ((pyl + ky - 1) >> 1) + 1, ((px + kx - 1) >> 1) + 1
                        else:
                            hy, hx = pyl + ky, px + kx
                        hs.add(sp * SUBH + hy * P + hx)
                    slots = {}
                    for h in hs:
                        key = (h & 1, swz(h, P))
                        slots[key] = slots.get(key, 0) + 1
                    bad += sum(v - 1 for v in slots.values())
    return bad

swzs = {
    "h>>1": lambda h, P: (h >> 1) & 7,
    "(h>>1)+row": lambda h, P: ((h >> 1) + h // P) & 7,
    "(h>>1)+2row": lambda h, P: ((h >> 1) + 2 * (h // P)) & 7,
    "(h>>1)+3row": lambda h, P: ((h >> 1) + 3 * (h // P)) & 7,
    "(h>>1)+4row": lambda h, P: ((h >> 1) + 4 * (h // P)) & 7,
    "(h>>1)^(h>>4)": lambda h, P: ((h >> 1) ^ (h >> 4)) & 7,
    "(h>>1)^(h>>5)": lambda h, P: ((h >> 1) ^ (h >> 5)) & 7,
    "(h>>1)+(h>>4)": lambda h, P: ((h >> 1) + (h >> 4)) & 7,
    "(h>>1)+(h>>5)": lambda h, P: ((h >> 1) + (h >> 5)) & 7,
}
# configurations: (name, PW, PHs(sub-patch height), up2, BM)
cfgs = [("w16", 16, 8, 0, 128), ("w16x16", 16, 16, 0, 256), ("w8", 8, 8, 0, 128), ("w4", 4, 4, 0, 128),
        ("w16u", 16, 8, 1, 128), ("w16x16u", 16, 16, 1, 256), ("w8u", 8, 8, 1, 128), ("w4u", 4, 4, 1, 128)]
for name, PW, PHs, up2, BM in cfgs:
    ntiles = BM // 32
    res = []
    base = (PW // 2 + 2) if up2 else (PW + 2)
    rowsn = (PHs // 2 + 2) if up2 else (PHs + 2)
    for P in range(base, base + 12):
        for extra in range(0, 9):
            SUBH = rowsn * P + extra
            for sn, sf in swzs.items():
                b = conflicts(PW, PHs, P, SUBH, sf, up2, ntiles)
                res.append((b, P, extra, sn))
    res.sort()
    print(name, "best:", res[:6])
print("---- all zero-conflict (P, extra, m) with swz = ((h>>1) + m*row) & 7")
for name, PW, PHs, up2, BM in cfgs:
    ntiles = BM // 32
    base = (PW // 2 + 2) if up2 else (PW + 2)
    rowsn = (PHs // 2 + 2) if up2 else (PHs + 2)
    ok = []
    for P in range(base, base + 16):
        for extra in (0, 4):
            for m in range(8):
                sf = lambda h, P_, m=m: ((h >> 1) + m * (h // P_)) & 7
                if conflicts(PW, PHs, P, rowsn * P + extra, sf, up2, ntiles) == 0:
                    ok.append((P, extra, m))
    print(name, ok[:24])
