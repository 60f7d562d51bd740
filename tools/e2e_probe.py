import sys, os, time, subprocess, tempfile
sys.path.insert(0, '/root/repo')
import numpy as np
L=150
n=8_000_000
rng=np.random.default_rng(1)
G=rng.integers(0,4,100_000_000,dtype=np.uint8)
pos=rng.integers(0,len(G)-L,n)
td='/dev/shm/smx_e2e'; os.makedirs(td,exist_ok=True)
fq=os.path.join(td,'r.fq')
lut=np.frombuffer(b"ACGT",dtype=np.uint8)
with open(fq,'wb') as f:
    for c0 in range(0,n,1<<21):
        p=pos[c0:c0+(1<<21)]; m=len(p)
        codes=G[p[:,None]+np.arange(L)[None,:]]
        err=rng.random(codes.shape)<0.01
        codes=np.where(err,(codes+rng.integers(1,4,codes.shape))%4,codes).astype(np.uint8)
        rec=np.empty((m,12+L+3+L+1),dtype=np.uint8)
        rec[:,0],rec[:,1]=ord('@'),ord('r')
        rec[:,2:11]=np.frombuffer("".join(np.char.zfill(np.arange(c0,c0+m).astype(str),9)).encode(),dtype=np.uint8).reshape(m,9)
        rec[:,11]=10; rec[:,12:12+L]=lut[codes]; rec[:,12+L],rec[:,13+L],rec[:,14+L]=10,ord('+'),10
        rec[:,15+L:15+2*L]=ord('I'); rec[:,15+2*L]=10
        rec.tofile(f)
exe='/root/repo/spades_amd/tools/spades-gbuilder-mi355x'
for it in range(2):
    t0=time.time(); r=subprocess.run([exe,fq,os.path.join(td,'o.gfa'),'-k','55','-t','16','--gfa'],stdout=subprocess.DEVNULL,stderr=subprocess.PIPE,env=dict(os.environ,SMX_DEBUG='1')); dt=time.time()-t0
    print(dt, [l for l in r.stderr.decode().splitlines() if l.startswith('[tool]') or 'write_gfa' in l])
import shutil; shutil.rmtree(td)
