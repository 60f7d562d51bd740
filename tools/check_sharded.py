import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from bench import synth_reads_device
from spades_amd import KMerDiskCounter, ReadKMerSplitter
from spades_amd.kmercount import Context
from spades_amd import dist as smx_dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = int(float(sys.argv[1])) // 32 * 32
K, nb = 55, 16
words, start, ln, codes = synth_reads_device(1000, 50_000_000, n, dev); del codes
ctx = Context(0); sp = ReadKMerSplitter(K, "A", ctx)
sp.push_back_device(words.data_ptr(), words.numel() - 8, start.data_ptr(), ln.data_ptr(), n)
st = KMerDiskCounter(None, sp).Count(nb)
print("direct distinct", st.total_kmers(), "instances", st.kmer_instances())
eng = smx_dist.GpuEngine(ctx, "A")
nloc = eng.extract_count(K)
send = eng.alloc(nloc * 2, dev)
counts = eng.extract_partition(K, nb, 1, send, nloc)
res = eng.count_records(K, nb, send, nloc)
print("no-collective distinct", res["distinct"], counts)
res = smx_dist.sharded_count(eng, K, nb, 0, 1, dev)
print("sharded_count distinct", res["distinct"], "(must equal direct)")
assert res["distinct"] == st.total_kmers()
dist.destroy_process_group()
