# bank analysis of the tile updates of k_group8m (ds_read_b32 / ds_write_b32: two groups of 32 lanes, bank = dword mod 32)
import itertools
def conflicts(addrs):  # extra cycles for a group
    from collections import defaultdict
    b=defaultdict(set)
    for a in addrs: b[a%32].add(a)
    return max(len(v) for v in b.values())-1
def sepb(rwp,plane):
    tot=0
    for kk in range(4):
        for half in range(2):
            ad=[]
            for lane in range(32*half,32*half+32):
                si=lane&3; spl=(lane>>2)&3; g4=lane>>4
                r=(7-si) if kk&2 else si; c=4*(kk&1)+g4
                ad.append(spl*plane+r*rwp+c)
            tot+=conflicts(ad)
    return tot
def kron(rwp,plane):
    tot=0
    for kk in range(4):
        for half in range(2):
            ad=[]
            for lane in range(32*half,32*half+32):
                lo=lane&15; g4=lane>>4; pi=lo>>2; pj=lo&3
                r=(7-pi) if kk&2 else pi; c=(7-pj) if kk&1 else pj
                ad.append(g4*plane+r*rwp+c)
            tot+=conflicts(ad)
    return tot
print("current rwp=28 plane=624: sepb",sepb(28,624),"kron",kron(28,624))
for name,f in (("sepb",sepb),("kron",kron)):
    good=[(r,p) for r in range(32) for p in range(32) if f(r+32,p+640)==0]
    print(name,"conflict-free (rwp mod 32, plane mod 32):",len(good))
    print(sorted(set(r for r,p in good)))
    for r in (26,27,28,29,30): print(r,[p for rr,p in good if rr==r])
print("full sepb table")
for r in range(32):
    ps=[p for p in range(32) if sepb(r+32,p+640)==0]
    if ps: print(r,ps)
# G16 gray layout: rounds: lanes spl==t add image to plane 0, lanes spl==t+1 add weights to plane 1
def g16(rwp,plane):
    tot=0
    for t in range(4):
      for kk in range(4):
        for half in range(2):
            ad=[]
            for lane in range(32*half,32*half+32):
                si=lane&3; spl=(lane>>2)&3; g4=lane>>4
                if spl==t or spl==((t+1)&3):
                    r=(7-si) if kk&2 else si; c=4*(kk&1)+g4
                    ad.append((plane if spl!=t else 0)+r*rwp+c)
            tot+=conflicts(ad)
    return tot
print("g16 current (gray 1080p: rw_max 26->28, plane?)", g16(28,624))
