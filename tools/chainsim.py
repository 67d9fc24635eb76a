#!/usr/bin/env python3
"""tools/chainsim.py -- prices a TABLE-FREE parse on the CPU (EXPERIMENTS.md R4.4): instead of a 32 KiB hash table per
block, a parse-independent array prev[p] (nearest earlier position in p's bucket, computable for all positions at
once) + an 8 KiB bitmap of the positions the parse has inserted; a look-up walks prev[] to the nearest INSERTED
predecessor.  The walk's result is asserted equal to the hash table's at every probe of the reference parse
(SURVEY App. A); printed: how many hops the probes of a 64 KiB text block need."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import fuzzgen
KMUL=0x1e35a7bd
def parse_stats(d):
    n=len(d); assert n==65536
    shift=32-14
    a=np.frombuffer(d,dtype=np.uint8).astype(np.uint32)
    w=(a[:-3]|(a[1:-2]<<8)|(a[2:-1]<<16)|(a[3:]<<24)).astype(np.uint64)
    h=((w*KMUL)&0xffffffff)>>shift
    h=h.astype(np.int64)
    # prev chain: nearest earlier position with same bucket
    last=-np.ones(16384,dtype=np.int64); prev=-np.ones(n,dtype=np.int64)
    for p in range(len(h)):
        prev[p]=last[h[p]]; last[h[p]]=p
    inserted=np.zeros(n,dtype=bool)
    table=np.zeros(16384,dtype=np.int64)   # 0 = empty/position 0
    hops=[]; probes=0; empty=0
    def lookup(p):
        # chain walk to nearest inserted predecessor (or none)
        nonlocal probes, empty
        probes+=1
        q=prev[p]; k=0
        while q>=0 and not inserted[q]:
            q=prev[q]; k+=1
        hops.append(k)
        cand = q if q>=0 else 0
        assert cand==table[h[p]], (p,cand,table[h[p]])
        if q<0: empty+=1
        return cand
    def insert(p):
        inserted[p]=True; table[h[p]]=p
    ip=1; next_emit=0; ip_limit=n-15
    wd=lambda p: int(w[p])
    done=False
    while not done:
        skip=32; next_ip=ip
        while True:
            ip=next_ip; step=skip>>5; skip+=step; next_ip=ip+step
            if next_ip>ip_limit: done=True; break
            cand=lookup(ip); insert(ip)
            if wd(ip)==wd(cand): break
        if done: break
        while True:
            base=ip; m=4
            while ip+m<n and d[cand+m]==d[ip+m]: m+=1
            ip+=m; next_emit=ip
            if ip>=ip_limit: done=True; break
            insert(ip-1)
            cand=lookup(ip); insert(ip)
            if wd(ip)!=wd(cand): break
        if done: break
        ip+=1
    hops=np.array(hops)
    return probes, empty, hops, inserted.sum()
text=fuzzgen.corpus("mtg")
tot=[]
for i in range(6):
    o=i*400000
    pr,em,hops,ins=parse_stats(text[o:o+65536])
    print("block %d: probes %d  inserted %d  empty %d  hops mean %.2f  frac0 %.2f frac<=1 %.2f frac<=3 %.2f  max %d" % (i,pr,ins,em,hops.mean(),(hops==0).mean(),(hops<=1).mean(),(hops<=3).mean(),hops.max()))
