"""MFMA-bound time of candidate decode-order mappings of the hidden layer (H=64, W=128, G=48, 144 samples per launch): counts the
16x16x4 MFMAs each (groups per task GB, lane-class modulus M, row tiles per wave NT) needs per plane, incl. row-tile, K-block and
row-window padding, and prints the average / worst plane at 32 cycles per MFMA on 1024 SIMDs (DESIGN.md 4.1 b')."""
import math,sys
H,W,G=64,128,48
S=H+W-1; P=H+W+G-2
hidden=1
ntaps=[1,2,3,4,5,4,3,2,1]
def diag_range(s):
    if s<0 or s>=S: return None
    return (max(0,s-W+1), min(H-1,s))
def slots(GB,M,r0,c,valid_q):
    n=0
    for q in valid_q:
        e=c+q
        if e<0 or e>8: continue
        for kh in range(5):
            kw=e-kh
            if kw<0 or kw>4: continue
            for gid in range(4):
                i=gid*25+5*kh+kw
                if (i-q)%M==r0: n+=1
    return n
def run(GB,M,NT,ns=144):
    # NT = N-tiles per wave (interleaved), positions per wave = 16*NT ; samples packed per wave when short
    tot=0  # MFMA count weighted per launch (sum over planes), counting per-wave MFMAs * waves ... we count total MFMAs
    mac=0
    per_plane=[]
    for p in range(P):
        m=0.0
        for tb in range(0,G,GB):
            vq=[q for q in range(GB) if tb+q<G and diag_range(p-tb-q)]
            if not vq: continue
            lo=min(diag_range(p-tb-q)[0] for q in vq); hi=max(diag_range(p-tb-q)[1] for q in vq)
            lo_in=max(0,lo-2); hi_in=min(H-1,hi+2)
            ln=hi_in-lo_in+1
            # columns needed per tile = ceil(ln/NT); samples per wave-set = floor(16/cols); wave-sets per sample-span = ceil over 64/(16*NT) halves
            pos_per_wave=16*NT
            nw=math.ceil(ln/pos_per_wave)           # waves along positions for one sample
            if nw==1:
                cols=math.ceil(ln/NT)
                spw=max(1,16//cols)                  # samples per wave
                wave_sets=math.ceil(ns/spw)/ns       # per-sample fraction
            else:
                wave_sets=nw
            for r0 in range(M):
                for c in range(-(GB-1),9):
                    s=slots(GB,M,r0,c,vq)
                    if s==0: continue
                    L=min(G,tb+4+hidden-c)
                    if L<=0: continue
                    m+=((s+3)//4)*((L+3)//4)*NT*wave_sets
        per_plane.append(m)
    tot=sum(per_plane)
    return tot, max(per_plane)
clk=2.4e9
for GB,M,NT in ((4,4,2),(3,4,4),(3,8,4),(4,8,4),(4,4,4),(6,8,2),(8,8,2)):
    tot,mx=run(GB,M,NT)
    print(GB,M,NT,"avg us/launch %.1f  worst plane %.1f"%(tot/P*144*32/1024/clk*1e6, mx*144*32/1024/clk*1e6))
print("----")
for GB,M,NT in ((3,4,2),(4,4,2),(2,4,4),(2,4,2),(6,4,2),(6,4,1),(8,4,1)):
    tot,mx=run(GB,M,NT)
    print(GB,M,NT,"avg us/launch %.1f  worst plane %.1f"%(tot/P*144*32/1024/clk*1e6, mx*144*32/1024/clk*1e6))
# the production kernel (v_mfma_f32_4x4x1, tasks of 3 groups x 64 lanes, 25 MFMAs per wave and input group, ~10.7 cycles each)
tot = 0
for p in range(P):
    for g0 in range(0, G, 3):
        if any(diag_range(p - g0 - q) for q in range(3) if g0 + q < G):
            tot += min(G, g0 + 2 + 4 + hidden) * 75
print("4x4x1 kernel (3 groups x 64 lanes per task): avg us/launch %.1f" % (tot / P * 144 * 10.7 / 256 / clk * 1e6))
