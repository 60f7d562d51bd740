"""The C restatement (oracle/) against the md5s of the REAL tools at 2 M reads (tests/golden/scale_2000k_g10000k_s77.json): final_kmers of
spades-kmercount and the GFA of spades-gbuilder -c. CPU only, ~30 minutes; round 4: both equal."""
import sys, time, json, hashlib
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, synth
from oracle import oracle
g=json.load(open('/root/repo/tests/golden/scale_2000k_g10000k_s77.json'))
codes=synth.synth_codes(g['seed'],g['genome_len'],g['n_reads'],g['err'],g['n_rate'])
assert hashlib.md5(codes.tobytes()).hexdigest()==g['codes_md5']
bases,off=synth.ascii_and_offsets(codes)
t=time.time()
rec,sizes=oracle.count_raw(bases.tobytes(), off, g['k'], "A", 16)
print("count s", time.time()-t, rec.shape, flush=True)
print("final_kmers md5 equal:", hashlib.md5(rec.tobytes()).hexdigest()==g['final_kmers_md5'], rec.nbytes, g['final_kmers_bytes'], flush=True)
del rec
lut=np.frombuffer(b"ACGTN",dtype=np.uint8)
reads=[lut[c].tobytes().decode() for c in codes]
t=time.time()
r=oracle.build_graph(reads,g['k'],10*g['effective_threads'],coverage=True)
print("graph s", time.time()-t, len(r['unitigs']), r['n_loops'], flush=True)
print("gfa -c md5 equal:", hashlib.md5(r['gfa'].encode()).hexdigest()==g['gfa_cov_md5'], flush=True)
