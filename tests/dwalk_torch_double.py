"""Test infrastructure: the plumbing of the distributed walks restated on torch tensors — owner routing, pointer doubling, the
ragged fetch of the chains — between the four k-mer-specific primitives of an engine double (tests/test_dist_cpu.py: OracleWalkEngine). Until
round 5 this code WAS the product path (spades_amd/dist.py); since round 6 the product does all of it on the device behind ONE entry point of the
library (smx_shard_walks: csrc/smx_dwalk.hpp + kernels in csrc/smx_dwalk.hip) with the caller's collectives between the kernels, and this file
is what the CPU doubles answer that entry point with: an independent statement of the same algorithm, which the gloo tests check against the
oracle's unitigs at world 2-4 and the C++ host's shim tests drive through the host's own collectives (tests/test_mgpu_hosts_cpu.py).
Nothing under spades_amd/ imports it."""
import torch
import torch.distributed as dist

from spades_amd.dist import _a2a, _guarded, _sync

# the collectives the plumbing runs on: torch.distributed by default; the shim tests of the C++ host put the HOST's own callbacks here
# (tests/test_mgpu_hosts_cpu.py), so that its exchange code is what moves the double's data
_A2A = _a2a                      # (send, counts, rank, world, dev) -> (recv, counts received)
_AR = dist.all_reduce            # (tensor, op=...) in place
_GUARD = _guarded                # (dev, what, fn, *args): a rank-local step every rank hears the outcome of


def _by_owner(owner: torch.Tensor, world: int):
    """-> (stable order that groups by owner, elements per owner). The keys are ranks: sorted as 16-bit integers (two radix passes
    instead of the eight of an int64 sort), counted from the sorted keys."""
    skey, order = torch.sort(owner.to(torch.int16), stable=True)
    edges = torch.searchsorted(skey, torch.arange(world + 1, dtype=torch.int16, device=owner.device))
    return order, [int(c) for c in (edges[1:] - edges[:-1]).tolist()]


def _remote_rows(targets: torch.Tensor, owner: torch.Tensor, table: torch.Tensor, my_base: int, rank: int, world: int, dev):
    """table[:, targets - base of the owner] from the ranks that own them (table: one row per field, one column per local node): one
    all-to-all of the indices, one of the fields (both in the order of the requests, so nothing but indices and fields travels).
    -> (len(targets), fields)"""
    w = table.shape[0]
    order, counts = _by_owner(owner, world)
    q, rcounts = _A2A(targets[order].contiguous(), counts, rank, world, dev)
    nq = sum(rcounts)
    rows = table[:, q[:nq] - my_base].t().reshape(-1).contiguous()
    back, _ = _A2A(rows, [c * w for c in rcounts], rank, world, dev)
    out = torch.empty((targets.numel(), w), dtype=table.dtype, device=dev)
    out[order] = back[:targets.numel() * w].reshape(-1, w)
    return out


def _ragged(off: torch.Tensor, ln: torch.Tensor, dev):
    """indices off[i] .. off[i] + ln[i] of every i, concatenated"""
    tot = int(ln.sum().item()) if ln.numel() else 0
    if tot == 0:
        return torch.empty(0, dtype=torch.int64, device=dev)
    seg = torch.repeat_interleave(torch.arange(ln.numel(), device=dev), ln)
    start = torch.cumsum(ln, 0) - ln
    return off[seg] + (torch.arange(tot, device=dev) - start[seg])


WALK_HOP_BITS = 24         # bits of a node's packed word that count the steps to its pointer (a chain of 2^24 k-mers or more is refused)
WALK_CHUNK = 1 << 26       # local nodes per exchange of the doubling / of the chain nucleotides (bounds the temporaries: ~40 B per node)
WALK_START_CHUNK = 1 << 22  # start de-edges per fetch of their chains


def _rounds_of(n_local: int, chunk: int, dev) -> int:
    """chunks the rank with the most elements needs: every rank runs that many (collective) rounds"""
    t = torch.tensor([-(-n_local // chunk)], dtype=torch.int64, device=dev)
    _AR(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def torch_walks(engine, k: int, rank: int, world: int, dev, kmers_per_rank, a2a=None, all_reduce=None):
    global _A2A, _AR
    saved = (_A2A, _AR)
    _A2A, _AR = a2a or _a2a, all_reduce or dist.all_reduce
    try:
        return _torch_walks(engine, k, rank, world, dev, kmers_per_rank)
    finally:
        _A2A, _AR = saved


def _torch_walks(engine, k: int, rank: int, world: int, dev, kmers_per_rank):
    """Unitigs of a graph whose k-mer file stays sharded (collective). Every rank holds its bucket range of
    {k-mer file, InOutMask bytes}; a walk of the reference (debruijn_graph_constructor.hpp:264-273) would change rank at every step, so
    nothing is walked:
      1. every oriented non-junction k-mer learns its successor node and whether that is a junction k-mer (ONE lookup exchange: the
         canonical successor k-mer travels to its owner, a node id comes back), every start de-edge its first node (a second one);
      2. the chains of non-junction k-mers are ranked by pointer doubling — per round one exchange of node ids and one of (pointer,
         hops, last chain k-mer, end node), ceil(log2(longest chain)) rounds at most; what never finishes lies on perfect loops.
         (Rows may be read in the state of this round or of the one before — a pointer only ever moves ahead along its chain, the
         hops with it — so the rounds run in chunks of WALK_CHUNK nodes and the temporaries stay bounded.)
      3. a chain k-mer x is hops(x^1) steps behind the head tail(x^1)^1 of its chain (the reverse strand went through the same
         doubling): it sends its outgoing nucleotide there, and the owner of the head lays the chain's nucleotides out in order;
      4. the owner of a start de-edge fetches length, end node and nucleotides of the chain behind it and assembles, keeps or drops the
         unitig exactly as the single-GPU route does (engine.shard_unitigs).
    Returns this rank's kept unitigs (k-mer-file order of their start k-mers: concatenated in rank order they are the reference's edge
    list), the local ranks of its k-mers on perfect loops, and the number of doubling rounds."""
    import os
    import time
    nw = (k + 31) // 32
    first = [0]
    for c in kmers_per_rank:
        first.append(first[-1] + int(c))
    n_mine = int(kmers_per_rank[rank])
    base = 2 * first[rank]
    bounds = torch.tensor([2 * f for f in first[1:]], dtype=torch.int64, device=dev)
    t_last = [time.perf_counter()]

    def mark(what):  # SMX_DEBUG: wall time of every phase on rank 0
        if os.environ.get("SMX_DEBUG") and rank == 0:
            _sync(dev)
            now = time.perf_counter()
            print(f"[dist] walks: {what} {1e3 * (now - t_last[0]):.0f} ms", flush=True)
            t_last[0] = now

    def owner_of(nodes):
        return torch.bucketize(nodes, bounds, right=True)

    def lookup(starts: bool, first_item: int = 0, n_items: int = -1):
        """-> (tags of this rank's requests, node each one leads to, is that a junction k-mer) in the order the library grouped them;
        first_item / n_items: the requests of that range of oriented nodes (start de-edges) only (collective: every rank its own range)"""
        if n_items < 0:
            recs, tags, counts = _GUARD(dev, "successor requests of the shard", engine.walk_requests, starts, k, world, dev)
        else:
            recs, tags, counts = _GUARD(dev, "successor requests of the shard", engine.walk_requests, starts, k, world, dev, first_item, n_items)
        recv, rcounts = _A2A(recs, [c * nw for c in counts], rank, world, dev)
        del recs
        n_recv = sum(rcounts) // nw
        reply = _GUARD(dev, "lookup in the shard", engine.shard_lookup, recv, n_recv, dev)
        del recv
        back, _ = _A2A(reply.contiguous(), [c // nw for c in rcounts], rank, world, dev)
        del reply
        n = sum(counts)
        back = back[:n]

        def check():
            if n and bool((back < 0).any().item()):
                raise RuntimeError(f"{int((back < 0).sum().item())} successor k-mers are in no shard: the k-mer file and the masks disagree")
        _GUARD(dev, "successor lookups", check)
        junc = (back & 1).to(torch.bool)
        back >>= 1
        o = 0
        for p_, c in enumerate(counts):  # local rank at the owner -> global rank
            back[o:o + c] += first[p_]
            o += c
        back <<= 1
        back |= (tags >> 2) & 1
        return tags, back, junc

    # 1. successors of the chain k-mers, first nodes of the start de-edges. State of a local oriented node: ONE packed word + one byte
    #    (round 3 kept four int64 rows + two more int64 arrays per node: 110 B per owned k-mer; this is 18 B):
    #      word  F << 63 | T << 62 | id << 24 | hops      F: the end of the chain is known; T: this node IS the end (its successor is a
    #            open:      id = pointer, hops = steps to it    junction k-mer); id: 38 bits (2.7e11 nodes), hops: 24 bits (a chain of
    #            tail (T):  id = the junction node behind it    16.7 M k-mers or more is refused)
    #            finished:  id = the tail of its chain, hops = steps to the tail
    #      byte  bit 0 chain k-mer (non-junction), bit 1 its successor is a junction k-mer, bits 2-3 its outgoing nucleotide
    n2 = 2 * n_mine
    HB = WALK_HOP_BITS  # (24; tests make it small: chains at the limit and loops whose hop counts saturate, on inputs of a few thousand reads)
    FBIT, TBIT, IDM, HM = -(1 << 63), 1 << 62, (1 << (62 - HB)) - 1, (1 << HB) - 1
    if 2 * first[-1] > IDM:
        raise ValueError(f"{first[-1]} k-mers: node ids beyond {62 - HB} bits")
    # (the arrays of the size of the shard's node set live in the library's arena where the engine offers that — GpuEngine.state: after the sharded
    # count the arena holds ~2x the shard and gives nothing back, a 4.3 G-k-mer shard left torch 0 bytes of a 288 GB device, round 5 — and are torch's
    # own on the CPU doubles)
    _state = getattr(engine, "state", None)
    _handles = {}

    def big(name, n, dtype, zero):
        if _state is None:
            return torch.zeros(n, dtype=dtype, device=dev) if zero else torch.empty(n, dtype=dtype, device=dev)
        t, h = _state(n, dtype, dev)
        _handles[name] = h
        return t.zero_() if zero else t

    def free_big(*names):  # (the tensor views of what is freed must be gone)
        for nm in names:
            h = _handles.pop(nm, None)
            if h:
                engine.state_free(h)

    word = big("word", n2, torch.int64, True)
    flag = big("flag", n2, torch.uint8, True)
    # (range by range: the requests of WALK_CHUNK oriented nodes at a time — a k-mer record out and a node id back per request; all at once
    # the exchange buffers of a shard were 48 B per oriented node, the peak of the whole construction)
    for c in range(_rounds_of(n2, WALK_CHUNK, dev)):
        a = min(c * WALK_CHUNK, n2)
        tags, node, junc = lookup(False, a, min(WALK_CHUNK, n2 - a))
        xl = tags >> 4
        word[xl] = torch.where(junc, node << HB | (FBIT | TBIT), node << HB | 1)  # tails know their end node; the others: pointer, one step
        flag[xl] = (1 | (junc.to(torch.int64) << 1) | ((tags & 3) << 2)).to(torch.uint8)
        del node, junc, tags, xl
    n_cand = int(engine.walk_counts()[1])
    c_first = torch.empty(n_cand, dtype=torch.int64, device=dev)
    c_fj = torch.empty(n_cand, dtype=torch.bool, device=dev)
    for c in range(_rounds_of(n_cand, WALK_CHUNK, dev)):
        a = min(c * WALK_CHUNK, n_cand)
        ctags, cfirst, cjunc = lookup(True, a, min(WALK_CHUNK, n_cand - a))
        ci = ctags >> 4
        c_first[ci] = cfirst
        c_fj[ci] = cjunc
        del ctags, cfirst, cjunc, ci

    def is_open(w, f):
        return ((f & 1) != 0) & (w >= 0)

    mark("successor lookups")
    # 2. pointer doubling over the chains
    node_rounds = _rounds_of(n2, WALK_CHUNK, dev)
    prev, rounds = -1, 0
    too_long = False
    def chunks_of_nodes():
        # (every pass over the node array goes chunk by chunk: an elementwise expression over all 2 x |shard| words makes temporaries of that size —
        # 40 GiB each at the 2.7 G k-mers of a 62.5 M-read share, where the first run of this path at that size ran out of memory, round 5)
        for c_ in range(node_rounds):
            yield min(c_ * WALK_CHUNK, n2), min((c_ + 1) * WALK_CHUNK, n2)

    while True:
        tot = torch.zeros(1, dtype=torch.int64, device=dev)
        for a, b in chunks_of_nodes():
            tot += is_open(word[a:b], flag[a:b]).sum()
        _AR(tot)
        tot = int(tot.item())
        if tot == 0 or tot == prev:  # every round ends at least one k-mer of every open chain: what is left runs in circles
            break
        prev = tot
        rounds += 1
        mark(f"round {rounds}: {tot} open")
        for c in range(node_rounds):
            a, b = min(c * WALK_CHUNK, n2), min((c + 1) * WALK_CHUNK, n2)
            act = is_open(word[a:b], flag[a:b]).nonzero().squeeze(1) + a
            mine_w = word[act]
            tg = (mine_w >> HB) & IDM
            wp = _remote_rows(tg, owner_of(tg), word.unsqueeze(0), base, rank, world, dev)[:, 0]
            p_tail = (wp & TBIT) != 0                 # the target ends its chain: it is the tail, no step is added
            p_fin = wp < 0
            hops = (mine_w & HM) + torch.where(p_tail, torch.zeros_like(wp), wp & HM)
            # Only a node that FINISHES this round has a chain length to overflow. A node on a perfect loop never finishes and its hop count
            # doubles every round (2^r after r rounds): once an ordinary chain needs ~24 rounds, every plasmid in the input tripped the
            # check although no real chain was that long. Open nodes saturate at HM instead — sticky: a node that finishes with
            # HM or more hops is refused, so a saturated count can never pass for a real one.
            too_long = too_long or (act.numel() > 0 and bool((p_fin & (hops >= HM)).any().item()))
            hops = hops.clamp_(max=HM)
            nid = torch.where(p_tail, tg, (wp >> HB) & IDM)
            word[act] = torch.where(p_fin, torch.full_like(wp, FBIT), torch.zeros_like(wp)) | (nid << HB) | (hops & HM)
            del act, mine_w, tg, wp, p_tail, p_fin, hops, nid

    def check_hops():
        if too_long:
            raise RuntimeError(f"a chain of 2^{HB} k-mers or more: beyond the packed hop count of the distributed walks")
    _GUARD(dev, "chain lengths", check_hops)
    left = [is_open(word[a:b], flag[a:b]).nonzero().squeeze(1) + a for a, b in chunks_of_nodes()]
    left = torch.cat(left) if left else torch.empty(0, dtype=torch.int64, device=dev)
    loop_local = torch.unique(left >> 1) if left.numel() else torch.empty(0, dtype=torch.int64, device=dev)
    del left
    def done_of(a, b):  # finished chain k-mers among the nodes [a, b) (computed where it is needed: one byte per node less to hold)
        return ((flag[a:b] & 1) != 0) & (word[a:b] < 0)

    mark("doubling")
    # 3. every chain k-mer to the head of its chain. A node is a head when the reverse strand's node of its k-mer is a tail; the heads'
    #    bookkeeping (chain length, offset of its nucleotides, end node) is kept per HEAD (hidx: their local nodes, ascending), not per node
    hidx = []
    pair_chunk = max(2, WALK_CHUNK // 2 * 2)  # (whole k-mers per chunk: the two nodes of a k-mer are looked at together)
    for a in range(0, n2, pair_chunk):
        b = min(a + pair_chunk, n2)
        rev_tail = ((word[a:b] & TBIT) != 0).view(-1, 2).flip(1).reshape(-1)
        hidx.append((done_of(a, b) & rev_tail).nonzero().squeeze(1) + a)
        del rev_tail
    hidx = torch.cat(hidx) if hidx else torch.empty(0, dtype=torch.int64, device=dev)
    hw = word[hidx]
    hlen = torch.where((hw & TBIT) != 0, torch.zeros_like(hw), hw & HM) + 1  # k-mers of the chain (a head that is its own tail: 1)
    del hw
    hoff = torch.cumsum(hlen, 0) - hlen
    hend = torch.full_like(hlen, -1)
    total = int(hlen.sum().item()) if hidx.numel() else 0
    bases = big("bases", max(total, 1), torch.uint8, True)
    n_heads = hidx.numel()

    def head_slot(local_nodes):
        """ordinal among this rank's heads of local nodes that must be heads (-> slots, all found)"""
        if n_heads == 0:
            return torch.zeros_like(local_nodes), local_nodes.numel() == 0
        slot = torch.searchsorted(hidx, local_nodes).clamp_(max=n_heads - 1)
        return slot, bool((hidx[slot] == local_nodes).all().item())

    def tail_of(nodes, w):  # the tail of the chain of finished nodes (w = their words)
        return torch.where((w & TBIT) != 0, nodes, (w >> HB) & IDM)

    n_got_all, bad_head = 0, False
    for c in range(node_rounds):
        a, b = min(c * WALK_CHUNK, n2), min((c + 1) * WALK_CHUNK, n2)
        xs = done_of(a, b).nonzero().squeeze(1) + a
        xr = xs ^ 1
        wr = word[xr]
        head = tail_of(xr + base, wr) ^ 1
        steps_back = torch.where((wr & TBIT) != 0, torch.zeros_like(wr), wr & HM)  # x is that many k-mers behind the head of its chain
        payload = (steps_back << 2) | ((flag[xs].to(torch.int64) >> 2) & 3)
        # the tails also tell the head which junction node ends the chain (payload: end node << 2 | 3 marks it: no nucleotide code 3 + huge)
        ws = word[xs]
        tl = ((ws & TBIT) != 0).nonzero().squeeze(1)
        head = torch.cat([head, head[tl]])
        payload = torch.cat([payload, -(((ws[tl] >> HB) & IDM) + 1)])  # negative: "the end node of your chain is -(payload) - 1"
        del wr, steps_back, ws, tl
        order, counts = _by_owner(owner_of(head), world)
        msg = torch.stack([head[order], payload[order]], 1).reshape(-1).contiguous()
        del xs, xr, head, payload, order
        got, rcounts = _A2A(msg, [2 * c_ for c_ in counts], rank, world, dev)
        del msg
        n_got = sum(rcounts) // 2
        got = got[:2 * n_got].reshape(-1, 2)
        if n_got:
            slot, ok_ = head_slot(got[:, 0] - base)
            bad_head = bad_head or not ok_
            is_end = got[:, 1] < 0
            e_sl = slot[is_end]
            hend[e_sl] = -got[:, 1][is_end] - 1
            n_sl, n_pl = slot[~is_end], got[:, 1][~is_end]
            pos = hoff[n_sl] + (n_pl >> 2)
            pos.clamp_(0, max(total, 1) - 1)
            bases[pos] = (n_pl & 3).to(torch.uint8)
            n_got_all += int(n_pl.numel())
            del slot, is_end, e_sl, n_sl, n_pl, pos
        del got

    def placed():
        if bad_head:
            raise RuntimeError("a chain nucleotide arrived at a k-mer that heads no chain")
        if n_got_all != total:
            raise RuntimeError(f"{n_got_all} chain nucleotides arrived for chains of {total} k-mers")
        if n_heads and bool((hend < 0).any().item()):
            raise RuntimeError("a chain whose end node never reached its head")
    _GUARD(dev, "chain nucleotides at the heads", placed)
    del flag, word
    free_big("word", "flag")  # (the chain nucleotides stay until the chains have been fetched)

    mark("chain nucleotides to the heads")
    # 4. the chains behind this rank's start de-edges
    q = (~c_fj).nonzero().squeeze(1)
    steps = big("steps", max(n_cand, 1), torch.int64, True)[:n_cand]
    last = big("last", max(n_cand, 1), torch.int64, False)[:n_cand]
    last.copy_(c_first)
    boff = big("boff", n_cand + 1, torch.int64, True)
    pieces, have, headless = [], 0, False
    for c in range(_rounds_of(q.numel(), WALK_START_CHUNK, dev)):
        qc = q[c * WALK_START_CHUNK:(c + 1) * WALK_START_CHUNK]
        tq = c_first[qc]
        order, counts = _by_owner(owner_of(tq), world)
        asks, rcounts = _A2A(tq[order].contiguous(), counts, rank, world, dev)
        n_asks = sum(rcounts)
        slot, ok_ = head_slot(asks[:n_asks] - base)
        headless = headless or not ok_
        a_len, a_end, a_off = hlen[slot], hend[slot], hoff[slot]
        if not ok_:  # (reported below, on every rank; nothing may be indexed with a wrong slot's length meanwhile)
            a_len = torch.zeros_like(a_len)
        rows, _ = _A2A(torch.stack([a_len, a_end], 1).reshape(-1).contiguous(), [2 * c_ for c_ in rcounts], rank, world, dev)
        seg_of = torch.repeat_interleave(torch.arange(world, device=dev), torch.tensor(rcounts, dtype=torch.int64, device=dev))
        per_rank = torch.zeros(world, dtype=torch.int64, device=dev)
        if n_asks:
            per_rank.index_add_(0, seg_of, a_len)
        flat = bases[_ragged(a_off, a_len, dev)] if n_asks else torch.empty(0, dtype=torch.uint8, device=dev)
        mine, rc2 = _A2A(flat.contiguous(), [int(v) for v in per_rank.tolist()], rank, world, dev)
        rows = rows[:2 * qc.numel()].reshape(-1, 2)
        qo = qc[order]
        steps[qo] = rows[:, 0]
        last[qo] = rows[:, 1]
        boff[qo] = have + torch.cumsum(rows[:, 0], 0) - rows[:, 0]
        headless = headless or (qc.numel() > 0 and bool((rows[:, 0] <= 0).any().item()))
        pieces.append(mine[:sum(rc2)])
        have += sum(rc2)
        del asks, slot, a_len, a_end, a_off, rows, flat, mine

    def check_chains():
        if headless:
            raise RuntimeError("a start de-edge leads to a k-mer that heads no chain")
    _GUARD(dev, "chains behind the start de-edges", check_chains)
    del hidx, hlen, hoff, hend, bases
    free_big("bases")
    # the fetched chains in one array (piece after piece into place: a torch.cat would hold them twice)
    my_bases = big("my_bases", max(have, 1), torch.uint8, have == 0)
    at = 0
    while pieces:
        pc = pieces.pop(0)
        my_bases[at:at + pc.numel()] = pc
        at += pc.numel()
        del pc
    del pieces
    # (no torch.cuda.empty_cache() here: VRAM that one allocator has just released is not safe for the next one to take at once on this
    # stack — arena_trim in csrc/smx_ctx.hpp has the measurements)
    _sync(dev)
    mark("chains of the start de-edges")
    if os.environ.get("SMX_DEBUG") and n_cand:
        print(f"[dist] walks: rank {rank}: {n_cand} start de-edges, steps max {int(steps.max().item())} sum {int(steps.sum().item())}, "
              f"{my_bases.numel()} nucleotides fetched", flush=True)
    unitigs = _GUARD(dev, "unitigs of the shard", engine.shard_unitigs, first[rank], steps, last, boff, my_bases, dev)
    del steps, last, boff, my_bases
    free_big("steps", "last", "boff", "my_bases")
    mark("unitigs")
    return unitigs, loop_local, rounds


