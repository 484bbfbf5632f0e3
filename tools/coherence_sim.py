#!/usr/bin/env python3
"""How coherent would the rank-select gathers of the encode chain be if a wave's lanes held rank-neighbours instead
of index-neighbours?  CPU simulation on one 8192-line block of the bench matrix (64 976 haplotypes, seed 43): the
LDS-array cycles of a wave64 ds_read_b64 gather (two groups of 32 lanes over 32 bank pairs, one cycle per distinct
address on the fullest bank pair: MI355X_MICROARCH.md "LDS") with the slots sorted by the PBWT order of a reference
line, as a function of the distance from that line.  Numbers quoted in DESIGN.md 5.11.  Minutes of numpy."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xsqueezeit_amd import synth
N=64976; L=8192
bits=synth.synth_bits(43, 3*8192, L, N)
cnt=bits.sum(1)
wah=[i for i in range(L) if min(cnt[i],N-cnt[i])>64]
print("wah lines", len(wah), flush=True)
def lds_cycles(entries):
    tot=0
    for half in (entries[:,:32], entries[:,32:]):
        bank=half%32
        mx=np.zeros(half.shape[0],dtype=np.int64)
        for b in range(32):
            m=(bank==b)
            vals=np.where(m, half, -1)
            vals.sort(axis=1)
            d=(np.diff(vals,axis=1)!=0).sum(1) + 1 - (vals[:,0]==-1)*1
            mx=np.maximum(mx,d)
        tot+=mx
    return tot.mean()
def step(rank, x):
    y=np.zeros(N,dtype=np.uint8); y[rank]=x
    ones_before=np.cumsum(y)-y
    Z=N-y.sum()
    ob=ones_before[rank]
    return np.where(x==1, Z+ob, rank-ob).astype(np.int64)
rank=np.arange(N)
snaps={}
probe=[0,50,100,200,400,800,1200,1600,2000,2400,2800,3200,3600,4000,4400,4800]
for t,i in enumerate(wah):
    if t in probe: snaps[t]=rank.copy()
    rank=step(rank,bits[i])
M=N//64*64
def layout_cost(order, r):
    # slots sorted by `order` (order[s] = hap in slot s); conflict-free lane layout: slot s -> (wave, e, lane) with
    # rank-consecutive slots 32 apart across lanes: emulate by evaluating entries of slots s = base + 32*lane + (e&31) ...
    rs=r[order][:M]
    # simple layout: lane-consecutive slots (64 consecutive slots per gather)
    return lds_cycles((rs.reshape(-1,64))>>5)
for t0 in (0, 2400):
    order=np.argsort(snaps[t0])
    print("slots sorted by the order at WAH line", t0, ":", " ".join("%d:%.1f"%(t,layout_cost(order,snaps[t])) for t in probe), flush=True)
