# Verify the unified halo layout rule: physical row h = sp*SUBH + hy*P + hx (P, SUBH even), swizzle = ((hx>>1) + 4*hy + 2*sp) & 7
import os
import sys
sys.argv = sys.argv[:1]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "halo_search.py")).read().split('swzs = {')[0])
def conflicts2(PW, PHs, P, SUBH, up2, ntiles, rule):
    bad = 0
    for ti in range(ntiles):
        rows = rows_of_tile(PW, PHs, ti)
        for ky in range(3):
            for kx in range(3):
                for g in G:
                    seen = {}
                    for l in g:
                        sp, pyl, px = rows[l]
                        if up2:
                            hy, hx = ((pyl + ky - 1) >> 1) + 1, ((px + kx - 1) >> 1) + 1
                        else:
                            hy, hx = pyl + ky, px + kx
                        h = sp * SUBH + hy * P + hx
                        seen[h] = (h & 1, rule(sp, hy, hx))
                    vals = list(seen.values())
                    bad += len(vals) - len(set(vals))
    return bad
rule = lambda sp, hy, hx: ((hx >> 1) + 4 * hy + 2 * sp) & 7
for name, PW, PHs, up2, BM in [("w16", 16, 8, 0, 128), ("w16x16", 16, 16, 0, 256), ("w8", 8, 8, 0, 128), ("w8/256", 8, 8, 0, 256), ("w4", 4, 4, 0, 128), ("w4/256", 4, 4, 0, 256),
        ("w16u", 16, 8, 1, 128), ("w16x16u", 16, 16, 1, 256), ("w8u", 8, 8, 1, 128), ("w8u/256", 8, 8, 1, 256)]:
    wid = (PW // 2 + 2) if up2 else (PW + 2)
    hgt = (PHs // 2 + 2) if up2 else (PHs + 2)
    P = wid + (wid & 1)
    SUBH = hgt * P
    print(name, "P", P, "SUBH", SUBH, "halo rows", (BM // (PW * PHs)) * SUBH, "conflicts", conflicts2(PW, PHs, P, SUBH, up2, BM // 32, rule))
