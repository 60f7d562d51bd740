"""CPU simulation (no product code): at 30x coverage and 1 % substitutions, which share of the k-mer window instances sits in super-k-mer
slots that are identical to the error-free slot of their locus (what the fold of identical slots removes), in error-free slots cut short by a
read end or a neighbouring error (what a containment fold could remove on top), and in slots that contain an error. DESIGN.md section 7."""
import numpy as np, sys, collections
rng=np.random.default_rng(3)
G=300_000; L=150; k=55; m=16; cov=30; err=0.01
genome=rng.integers(0,4,G,dtype=np.uint8)
def rc(a): return (3-a)[::-1]
def mmer_keys(seq):
    # canonical m-mer values (2-bit packed into uint64 via rolling), then a mixing hash for the ordering
    n=len(seq)-m+1
    f=np.zeros(n,dtype=np.uint64); r=np.zeros(n,dtype=np.uint64)
    v=np.uint64(0); w=np.uint64(0); mask=np.uint64((1<<(2*m))-1)
    for i,c in enumerate(seq):
        v=((v<<np.uint64(2))|np.uint64(c))&mask
        w=(w>>np.uint64(2))|(np.uint64(3-c)<<np.uint64(2*(m-1)))
        if i>=m-1: f[i-m+1]=v; r[i-m+1]=w
    can=np.minimum(f,r)
    h=(can*np.uint64(0x9E3779B97F4A7C15))&np.uint64(0xFFFFFFFFFFFFFFFF)
    h^=h>>np.uint64(29)
    return h
def superkmers(seq):
    """list of (start_window, n_windows) runs of windows sharing the position of their minimizer"""
    h=mmer_keys(seq); nwin=len(seq)-k+1; wlen=k-m+1
    pos=np.empty(nwin,dtype=np.int64)
    for i in range(nwin):
        pos[i]=i+int(np.argmin(h[i:i+wlen]))
    runs=[]; s=0
    for i in range(1,nwin+1):
        if i==nwin or pos[i]!=pos[s]:
            runs.append((s,i-s)); s=i
    return runs
# reference runs from the genome (forward strand; strand-oriented slot = canonical of the run sequence)
def canon(seq):
    a=seq.tobytes(); b=rc(seq).tobytes()
    return min(a,b)
gruns=superkmers(genome)
full=set(canon(genome[s:s+n+k-1]) for s,n in gruns)
# all genome k-mers (canonical) for error-free test
gk=set()
for i in range(G-k+1):
    gk.add(canon(genome[i:i+k]))
nreads=G*cov//L
tot=ident=trunc=witherr=0
slots=ident_s=trunc_s=err_s=0
for _ in range(nreads):
    p=rng.integers(0,G-L+1)
    r=genome[p:p+L].copy()
    if rng.random()<0.5: r=rc(r)
    e=rng.random(L)<err
    r[e]=(r[e]+rng.integers(1,4,e.sum()))%4
    for s,n in superkmers(r):
        seq=r[s:s+n+k-1]
        tot+=n; slots+=1
        c=canon(seq)
        if c in full: ident+=n; ident_s+=1
        else:
            ok=all(canon(seq[j:j+k]) in gk for j in range(n))
            if ok: trunc+=n; trunc_s+=1
            else: witherr+=n; err_s+=1
print("instances per slot %.2f"%(tot/slots))
print("instances: identical-to-full %.3f  error-free-but-partial %.3f  with-errors %.3f"%(ident/tot,trunc/tot,witherr/tot))
print("slots:     identical-to-full %.3f  error-free-but-partial %.3f  with-errors %.3f"%(ident_s/slots,trunc_s/slots,err_s/slots))
