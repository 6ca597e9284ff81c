"""One-process-per-GPU hetmers over a PREFIX-SHARDED table (RCCL through torch.distributed).

Rank r owns a contiguous range of the globally sorted table (cut between k-mers, i.e. on a
k-mer prefix boundary).  The reference has no distributed mode; its nearest analogue is the
prefix-subtree task decomposition of `small_window` (src/lib/PloidyPlot.c:1040-1084).

Data path per run (see DESIGN.md "Multi-GPU"):
  1. pass 1 on every shard (window scan of the suffix-side positions: always shard local);
  2. request filter (hash proof, k <= 85): one all_gather of the candidate block maps (each rank contributes the
     words its k-mer range covers: two-bit map of 30 id bits, 29 from 8 ranks on: 256 / 128 MB over all ranks) while
     the map-independent half of the filter (the request partition) runs -- a request whose target block holds no
     candidate of pass 2 is dropped before it is sent (12 in 13);
  3. ONE exchange: every surviving request rc(kmer) goes to the rank that owns it: grouped by destination on the
     device, the per-destination counts go from the router into an all_to_all_single without a host round trip, the
     host reads what it sends and receives in one copy, then the records follow in a second all_to_all_single;
  4. look-ups of the received requests and pass 2, both queued without a host wait;
  5. ONE all_reduce(SUM, int64[1001*501 + 1 + 2*world]) of the per-GPU 2-D histograms with the symmetry proof appended
     (missing count + one 128-bit XOR-fingerprint slot per rank, written on the device in stream order).
  k > 85 takes the same steps with the counted kernels (no block map: step 2 falls away).
  condition_sharded: a RAW table (canonical k-mers, untrimmed) is trimmed and closed under reverse complement across the
  ranks first (histogram of the closed table -> balanced splitters, entries + complements exchanged, sorted per rank).

torch is plumbing here: device buffers and collectives.  The compute is the C-ABI engine
(`engine.Engine`); `engine_factory` lets the CPU test-suite substitute a numpy stand-in so the
world_size>1 orchestration is exercised under gloo without a GPU.
"""

from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from . import engine as _engine

PLOT_CELLS = _engine.PLOT_CELLS


class NotSymmetric(RuntimeError):
    """The sharded path needs a reverse-complement closed table (what Symmex produces)."""


class TorchEngine:
    """Adapter: `engine.Engine` with torch tensors instead of raw device pointers."""

    def __init__(self, device: torch.device):
        assert device.type == "cuda", "the HIP engine needs a GPU (no CPU fallback)"
        self.device = device
        index = device.index if device.index is not None else torch.cuda.current_device()
        stream = torch.cuda.current_stream(device).cuda_stream
        self.e = _engine.Engine(index, stream)

    def bind(self, k, keys, counts, index=None, first_entry=0):
        """index: the table's FastK prefix index (int64[2^24] on the device, entries up to every 3-byte prefix of the WHOLE
        table; first_entry = number of this shard's first entry in it) -- what a .ktab stub carries, libfastk.c:841"""
        self._keep = (keys, counts)
        self.words = (k + 31) // 32
        self.e.bind(k, counts.numel(), keys.data_ptr(), counts.data_ptr())
        # what the engine is bound to: hetmers_sharded(prebound=True) checks it (a run that failed the symmetry proof
        # leaves rank 0's engine on the GATHERED table, see _general_on_rank0) and binds again when it does not match
        self._bound = (keys.data_ptr(), counts.numel())
        if index is not None:
            # the engine reads 2^24 int64 words from this pointer: anything else is an out-of-bounds read on the device
            if not (index.dtype == torch.int64 and index.numel() == 1 << 24 and index.is_contiguous()
                    and index.device == keys.device):
                raise ValueError("prefix index: need a contiguous int64 tensor of 2^24 words on the table's device "
                                 "(entries up to every 3-byte prefix, libfastk.c:841)")
            self.e.set_prefix_index(index.data_ptr(), 3, first_entry)
            # The engine turns the 2^24 int64 words into its own 32-bit directory by a kernel on ITS stream (which need not be
            # torch's current one).  bind is set-up, not a step: wait for it here, so that the caller's 134 MB tensor is the
            # caller's again when bind returns -- nothing of it is kept alive by this object (round 5 kept a reference).
            torch.cuda.synchronize(self.device)

    def rebind(self, k, keys, counts):
        """bind again after the engine was left on another table (rank 0 after a failed symmetry proof).  The prefix index of the
        first bind is not kept: the engine builds a directory of its own in pass 1, as for any table that comes without one."""
        self.bind(k, keys, counts)

    def pass1(self, symcheck, exchange=True, world=1):
        # nobody to exchange block maps with: the finest map (32 id bits) costs nothing but its memset.  Exchanged maps:
        # 30 id bits (256 MB over all ranks); from 8 ranks on 29 -- a rank's share of the requests is small enough that
        # the coarser map costs nothing (profiles/r03_rank_share_forced_exchange.txt: 3.21 vs 3.30 ms at 1/8 of the table)
        # and the all_gather, whose size does not shrink with the number of ranks, is halved
        self.e.set_blockmap_bits((29 if world >= 8 else 0) if exchange else 32)
        self.e.pass1(symcheck)

    def nreq(self):
        return self.e.nreq()

    def record_words(self):
        return self.e.record_words()

    def route(self, splitters, nranks, send, counts_out=None):
        """counts_out: int64 device tensor [nranks] -- the per-rank counts stay on the device (returns None, no host wait)"""
        cap = send.numel() // self.e.record_words()
        if counts_out is not None:
            self.e.route_device(splitters, nranks, send.data_ptr(), cap, counts_out.data_ptr())
            return None
        return self.e.route(splitters, nranks, send.data_ptr(), cap)

    def apply(self, recv, nrecv, wait=True):
        return self.e.apply(recv.data_ptr(), nrecv, wait)

    def proof_into(self, dst):
        """dst: int64 tensor of >= 4 words on the device <- (missing, fingerprint residue x 2, replayed step found other
        counts than it was queued with), in stream order"""
        self.e.proof_into(dst.data_ptr())

    def proof_tail(self, tail, nslots, slot):
        """tail: int64 tensor of 3 + 2 nslots words (the end of the reduction buffer) <- the whole proof tail, one launch"""
        self.e.proof_tail(tail.data_ptr(), nslots, slot)

    def set_replay(self, on):
        self.e.set_replay(on)

    def replay_state(self):
        return self.e.replay_state()

    def replay_done(self, ok):
        self.e.replay_done(ok)

    def apply_own(self):
        return self.e.apply_own()

    def blockmap(self):
        return self.e.blockmap()

    def blockmap_copy(self, word_lo, nw, dst):
        self.e.blockmap_copy(word_lo, nw, dst.data_ptr())

    def presort(self):
        self.e.presort()

    def merge_maps(self, parts, width, wlo, wlen, full):
        self.e.merge_maps(parts.data_ptr(), width, wlo, wlen, full.data_ptr())

    def run_single(self, plot, symcheck):
        """the whole computation on this one shard in ONE call of the C ABI (smg_engine_run): pass 1, look-ups, symmetry
        proof, pass 2 -- and, when the proof fails, the general all-positions path.  On a table the engine has run before
        nothing is read back between the launches (one host wait per run)."""
        return self.e.run(plot.data_ptr(), symcheck)

    def run_general(self, k, keys, counts, plot):
        """the assumption-free all-positions path on a whole table (what one GPU does when the proof fails)"""
        self.e.bind(k, counts.numel(), keys.data_ptr(), counts.data_ptr())
        self._bound = None                    # (not the shard any more: a later prebound=True call binds again)
        return self.e.run(plot.data_ptr(), "none")

    def filter(self, full_map=None):
        return self.e.filter(None if full_map is None else full_map.data_ptr())

    def symhash(self):
        return self.e.symhash()

    # ---- conditioning across shards (condition_sharded) ----
    def trim(self, ethresh):
        return self.e.condition(ethresh, True, False)

    def symm_hist(self, bits):
        return self.e.symm_hist(bits)

    def symm_route(self, splitters, nranks, send):
        return self.e.symm_route(splitters, nranks, send.data_ptr(), send.numel() // (self.words + 1))

    def symm_finish(self, recv, nrecv):
        return self.e.symm_finish(recv.data_ptr(), nrecv)

    def nels(self):
        return self.e.table()[0]

    def pass2(self, plot):
        self.e.pass2(plot.data_ptr())

    def stats(self):
        return self.e.stats()


def shard_bounds(n: int, world: int, keys_first_word=None):
    """Even cut points [0..n] for `world` shards.  Cuts fall between k-mers, which is all the
    window scan needs when the cut is also a window-block boundary; `fix_cut` moves them."""
    return [(n * r) // world for r in range(world + 1)]


def fix_cut(keys_u64: np.ndarray, words: int, k: int, cut: int) -> int:
    """Move a cut forward to the next window-block boundary (entries that share their first
    k//2 bases must stay on one rank)."""
    n = len(keys_u64) // words
    if cut <= 0 or cut >= n:
        return min(max(cut, 0), n)
    p0 = k // 2
    kw = keys_u64.reshape(n, words)

    def prefix(i):
        out = []
        for w in range(words):
            bases = min(32, max(0, p0 - 32 * w))
            if bases == 0:
                break
            out.append(int(kw[i, w]) >> (64 - 2 * bases))
        return tuple(out)

    while cut < n and prefix(cut) == prefix(cut - 1):
        cut += 1
    return cut


def blockmap_ranges(splitters, words: int, world: int, bits: int, scale: int = 1):
    """Word ranges (first word, length) of the candidate block map that the k-mer ranges of the ranks cover.
    Block id = leading `bits` bits of a k-mer; rank r holds ids [id(first_r), id(first_{r+1})], so neighbours
    share their boundary word (the receiver ORs the ranges together) -- unless the splitter is a CUT VALUE with nothing
    below its id bits (a table generated or cut on key-space boundaries): every k-mer of rank r then lies below id(first_{r+1}),
    and if that id starts a map word the two ranges do not meet.  scale = 32-bit words per 32 block ids (2 for the engine's
    two-bit map: 64 bits per group)."""
    ids = [0] + [int(splitters[(r - 1) * words]) >> (64 - bits) for r in range(1, world)] + [(1 << bits) - 1]
    last = []
    for r in range(world):
        nxt = ids[r + 1]
        if r + 1 < world:
            sp = [int(splitters[r * words + w]) for w in range(words)]
            clean = (sp[0] & ((1 << (64 - bits)) - 1)) == 0 and all(x == 0 for x in sp[1:])
            if clean and nxt > ids[r]:
                nxt -= 1                       # (all k-mers of rank r are below the splitter: its last id is the one in front)
        last.append(nxt)
    wlo = [(ids[r] >> 5) * scale for r in range(world)]
    wlen = [max(((last[r] >> 5) + 1) * scale - wlo[r], 1) for r in range(world)]
    return wlo, wlen


def ranges_tile_the_map(wlo, wlen, nwords: int) -> bool:
    """the ranks' word ranges are equally long, disjoint and cover the map in rank order: the all_gather can write the map itself"""
    w = wlen[0]
    return all(x == w for x in wlen) and all(wlo[r] == r * w for r in range(len(wlo))) and w * len(wlo) == nwords


def _general_on_rank0(k, keys, counts, sizes, eng, plot, group, rank, world, words):
    """The table failed the symmetry proof: the shards cannot help each other (a pair's mirror image may be
    missing), so rank 0 collects the table and runs the general all-positions path, like a single GPU would; every
    rank gets the plot.  The reference gives an answer for such a table (PloidyPlot.c scans every position of every
    k-mer), so must the sharded run."""
    dev = keys.device
    if world > 1:
        if rank == 0:
            ks, cs = [keys], [counts]
            for r in range(1, world):
                if sizes[r]:
                    kr = torch.empty(sizes[r] * words, dtype=keys.dtype, device=dev)
                    cr = torch.empty(sizes[r], dtype=counts.dtype, device=dev)
                    dist.recv(kr, src=r, group=group)
                    dist.recv(cr, src=r, group=group)
                    ks.append(kr); cs.append(cr)
            keys, counts = torch.cat(ks), torch.cat(cs)
        elif counts.numel():
            dist.send(keys.contiguous(), dst=0, group=group)
            dist.send(counts.contiguous(), dst=0, group=group)
    if rank == 0:
        eng._general_keep = (keys, counts)
        eng.run_general(k, keys, counts, plot)
    else:
        plot.zero_()
    if world > 1:
        dist.broadcast(plot, src=0, group=group)


def _scratch(eng, name: str, nwords: int, dev) -> torch.Tensor:
    """an int64 device buffer of at least `nwords` words that belongs to the engine object and is reused by every step"""
    pool = getattr(eng, "_scratch_pool", None)
    if pool is None:
        pool = eng._scratch_pool = {}
    t = pool.get(name)
    if t is None or t.numel() < nwords or t.device != dev:
        t = pool[name] = torch.empty(int(nwords * 1.25) + 16, dtype=torch.int64, device=dev)
    return t[:nwords]


def symm_splitters(hist: np.ndarray, bits: int, world: int, words: int) -> np.ndarray:
    """Balanced splitters for the CLOSED table from the summed histogram of smg_engine_symm_hist (entries, then
    complements, per leading `bits` k-mer bits): rank r starts at the first bin in front of which the closed table
    holds r / world of its entries; a rank for which nothing is left gets the all-ones k-mer.  The same rule as the
    in-process driver (smg_multi.hpp)."""
    nb = 1 << bits
    tot = hist[:nb].astype(np.int64) + hist[nb:].astype(np.int64)
    before = np.concatenate([[0], np.cumsum(tot)[:-1]])
    total = int(tot.sum())
    out = np.zeros((max(world - 1, 0), words), dtype=np.uint64)
    for r in range(1, world):
        b = int(np.searchsorted(before, (total // world) * r, side="left"))
        if b >= nb:
            out[r - 1, :] = np.uint64(0xFFFFFFFFFFFFFFFF)
        else:
            out[r - 1, 0] = np.uint64(b) << np.uint64(64 - bits)
    return out.reshape(-1)


def condition_sharded(k: int, keys: torch.Tensor, counts: torch.Tensor, ethresh: int = 0, trim: bool = False,
                      symm: bool = True, engine_factory=TorchEngine, group=None, eng=None):
    """Condition a table that is spread over the ranks (any cut between k-mers will do: the entries are dealt out
    again): drop the entries below `ethresh` (Logex 'A[e-]') and / or close the table under reverse complement
    (Symmex) -- what the reference shells out for, PloidyPlot.c:1381-1414, without a size limit.  Collective.

    Returns (eng, splitters): the engine OWNS this rank's shard of the conditioned table afterwards (sorted, one
    entry per k-mer, ranges given by `splitters`); hand both to hetmers_sharded(k, None, None, eng=eng,
    splitters=splitters)."""
    if not symm:      # (checked before anything is bound, trimmed or exchanged: the caller's table and engine stay as they were)
        raise ValueError("condition_sharded(symm=False): trim a closed table with the engine's own condition()")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = keys.device
    words = (k + 31) // 32
    rw = words + 1
    if eng is None:
        eng = engine_factory(dev)
    eng.bind(k, keys, counts)
    n = counts.numel()
    if trim:
        n = eng.trim(ethresh)
    bits = max(2, min(12, 2 * (k // 2)))
    hist = torch.from_numpy(eng.symm_hist(bits).astype(np.int64)).to(dev)
    if world > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    splitters = symm_splitters(hist.cpu().numpy(), bits, world, words)
    send = torch.empty(max(2 * n, 1) * rw, dtype=torch.int64, device=dev)
    send_counts = eng.symm_route(splitters, world, send)
    if world > 1:
        sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
        rcnt = torch.empty_like(sc)
        dist.all_to_all_single(rcnt, sc, group=group)
        recv_counts = [int(v) for v in rcnt.cpu().tolist()]
        nrecv = sum(recv_counts)
        recv = torch.empty(max(nrecv, 1) * rw, dtype=torch.int64, device=dev)
        dist.all_to_all_single(recv[: nrecv * rw], send[: 2 * n * rw],
                               output_split_sizes=[c * rw for c in recv_counts],
                               input_split_sizes=[c * rw for c in send_counts], group=group)
    else:
        recv, nrecv = send, 2 * n
    eng.symm_finish(recv, nrecv)
    eng._splitter_cache = None
    return eng, splitters


def hetmers_sharded(k: int, keys: torch.Tensor, counts: torch.Tensor, symcheck: str = "hash",
                    engine_factory=TorchEngine, group=None, eng=None, fallback: bool = True, splitters=None,
                    prebound: bool = False, sizes=None):
    """Run hetmers on this rank's shard; returns (plot int64[1001*501] on the shard's device,
    summed over all ranks, and a stats dict).  Collective: every rank must call it.

    keys   : int64 tensor viewing the shard's uint64 k-mer words (n * ceil(k/32)), sorted
    counts : int16 tensor viewing the shard's uint16 counts (n)
    eng    : an engine from a previous call on the same shard (its device buffers are reused;
             allocation is set-up cost, not part of a step)
    fallback: a table that fails the symmetry proof is collected on rank 0 and run through the general path
             (False: raise NotSymmetric instead)
    keys = counts = None with `eng` and `splitters` from condition_sharded: the engine owns the shard already
    prebound: `eng` is bound to exactly these tensors already (eng.bind, possibly with the table's prefix index): the
             table is not bound again, so what the engine knows about it (index directory, first / last k-mer) stays.
             The k-mers must not change while bound (counts may).  If the engine is found bound to something else -- the
             run before failed the symmetry proof and left rank 0 on the gathered table -- the shard is bound again.
    splitters: lower bounds of the k-mer ranges of ranks 1..world-1 (W words each), e.g. the cut values a table was
             generated or cut by; `sizes` (entries per rank) with them saves the all_gather of the shard sizes that the
             fall-back on rank 0 needs
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    # SMG_FORCE_EXCHANGE=1 (tests): run the collectives of the exchange even in a one-rank group, so that the
    # real backend (RCCL) sees every call of the multi-rank protocol on a single-GPU box
    exchange = world > 1 or (dist.is_initialized() and os.environ.get("SMG_FORCE_EXCHANGE") == "1")
    words = (k + 31) // 32
    owned = keys is None
    if owned:
        if eng is None or splitters is None:
            raise ValueError("hetmers_sharded without a table needs the engine and the splitters of condition_sharded")
        dev = eng.device
        n = eng.nels()
        fallback = False                      # (a table this module closed itself cannot fail the proof)
    else:
        dev = keys.device
        n = counts.numel()
        if eng is None:
            eng = engine_factory(dev)
            prebound = False
        if prebound and getattr(eng, "_bound", (keys.data_ptr(), n)) != (keys.data_ptr(), n):
            if hasattr(eng, "rebind"):
                eng.rebind(k, keys, counts)
            else:
                eng.bind(k, keys, counts)
            eng._splitter_cache = None
        elif not prebound:
            eng.bind(k, keys, counts)

    # splitters = first k-mer of ranks 1..world-1 (an empty shard inherits its successor's).  They depend
    # on the table only: an engine that is reused on the same shard (bench.py) keeps them.
    tag = (0 if owned else keys.data_ptr(), n, world, rank)
    cached = getattr(eng, "_splitter_cache", None)
    if splitters is not None:
        splitters = np.ascontiguousarray(splitters, dtype=np.uint64).reshape(-1)
        sizes = [int(v) for v in sizes] if sizes is not None else None
        if sizes is None and not owned and fallback and exchange:
            # the caller's splitters come without shard sizes, and the general path on rank 0 needs them: one
            # all_gather, as in the uncached branch (without it rank 0 raised while the others sat in dist.send)
            mine = torch.tensor([n], dtype=torch.int64, device=dev)
            allv = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allv, mine, group=group)
            sizes = [int(v) for v in torch.cat(allv).cpu().tolist()]
    elif cached is not None and cached[0] == tag:
        splitters, sizes = cached[1], cached[2]
    elif exchange:
        first = torch.full((words,), -1, dtype=torch.int64, device=dev)       # all ones = +inf
        if n > 0:
            first.copy_(keys[:words])
        # one all_gather: the first k-mer and the shard size of every rank
        mine = torch.cat([first, torch.tensor([n], dtype=torch.int64, device=dev)])
        allv = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine, group=group)
        host = torch.stack(allv).cpu().numpy()
        firsts = [host[r, :words].view(np.uint64).copy() for r in range(world)]
        sizes = [int(host[r, words]) for r in range(world)]
        for r in range(world - 2, -1, -1):
            if sizes[r] == 0:
                firsts[r] = firsts[r + 1]
        splitters = np.concatenate(firsts[1:]) if world > 1 else np.zeros(0, np.uint64)
        eng._splitter_cache = (tag, splitters, sizes)
    else:
        splitters, sizes = np.zeros(0, np.uint64), [n]

    if not exchange and hasattr(eng, "run_single"):
        # one shard, nobody to talk to: the engine runs the table in one call (and takes the general path by itself when
        # the symmetry proof fails -- the reference answers for any sorted table)
        plot = torch.empty(PLOT_CELLS, dtype=torch.int64, device=dev)
        st = eng.run_single(plot, symcheck)
        if st["path"] != 1 and not fallback:
            raise NotSymmetric("table is not closed under reverse complement with equal counts; "
                               "run the single-GPU engine (general path) or condition the table")
        if st["path"] != 1:
            eng._splitter_cache = None
        st.update(rank=rank, world=world, shard_nels=n, sent=0, received=st.get("nemitted", 0), engine=eng)
        return plot, st

    # Replay (round 5): a step on the table of the step before is queued WITHOUT a host read in the middle -- the engine takes
    # the counts it needs between its phases from its record of that step, this driver the per-rank send / receive counts
    # from its own; the device compares all of them with what this step produces and the verdict rides with the proof words
    # in the one all_reduce.  A mismatch anywhere makes every rank run the step again the plain way.
    rkey = (tag, symcheck, world)
    rec = getattr(eng, "_replay_rec", None)
    # OFF unless SMG_REPLAY=1: on ONE GPU with every collective forced it does not pay (2.72 against 2.64 ms per step at an
    # eighth of the 1 Gbp table, profiles/r05_rank_share_forced_exchange.txt: what three host reads cost is what the checks of
    # the replayed step cost) -- it is there for runs over a fabric, where a host read in the middle of a step also waits
    # for the slowest rank, and has never run on more than one device.
    can_replay = (exchange and symcheck == "hash" and hasattr(eng, "set_replay") and os.environ.get("SMG_REPLAY") == "1"
                  and not owned)
    if can_replay:
        # A step is replayed by ALL ranks or by none: a replaying rank splits the exchange by its recorded counts, a plain one by
        # this step's -- if the counts moved, the split sizes would disagree across the ranks, which RCCL answers with a hang, not
        # with an error.  rec["all"] says that EVERY rank left the recorded step with a record (summed in that step's all_reduce).
        if rec is None or rec["key"] != rkey or not rec["all"]:
            eng.set_replay(False)              # (no counts of ours to go with the engine's: drop its record too)
            rec = None
        eng.set_replay(True)
    eng.pass1(symcheck, exchange, world)
    replay = bool(can_replay and rec is not None and (eng.replay_state() & 1))
    if can_replay and rec is not None and not replay:
        rec = None
    rw = eng.record_words()
    nreq = eng.nreq()
    both = None

    if exchange:
        bits, nwords = eng.blockmap()
        if bits:
            wlo, wlen = blockmap_ranges(splitters, words, world, bits, nwords // (((1 << bits) + 31) >> 5))
            width = max(wlen)
            # the three map buffers live as long as the engine (a fresh 128 MB torch.zeros per step is a memset and
            # an allocator round trip; the merge kernel overwrites every word of `full`)
            mk = (bits, world, width)
            bufs = getattr(eng, "_map_bufs", None)
            if bufs is None or bufs[0] != mk:
                bufs = (mk, torch.zeros(width, dtype=torch.int32, device=dev),
                        torch.empty(world * width, dtype=torch.int32, device=dev),
                        torch.empty(nwords, dtype=torch.int32, device=dev))
                eng._map_bufs = bufs
            _, mine, parts, full = bufs
            eng.blockmap_copy(wlo[rank], wlen[rank], mine)
            direct = ranges_tile_the_map(wlo, wlen, nwords)
            # (shards cut on key-space boundaries -- bench.py's -- cover equal, disjoint word ranges: the gather IS the map)
            work = dist.all_gather_into_tensor(full if direct else parts, mine, group=group, async_op=True)
            eng.presort()          # the map-independent half of the filter runs while the maps are in flight
            work.wait()
            if not direct:
                # ranges of neighbours share their boundary word: the merge ORs them (one launch)
                eng.merge_maps(parts, width, wlo, wlen, full)
            nreq = eng.filter(full)
        # (the exchange buffers live as long as the engine and only ever grow: a step allocates nothing)
        send = _scratch(eng, "send", max(nreq, 1) * rw, dev)
        # grouped on the device; the per-destination totals go from the router into the count exchange without a host
        # round trip, and the host reads what it sends and what it receives in ONE copy
        both = _scratch(eng, "both", 2 * world, dev)
        sc, rcnt = both[:world], both[world:]
        eng.route(splitters, world, send, counts_out=sc)
        dist.all_to_all_single(rcnt, sc, group=group)
        if replay:
            send_counts, recv_counts = rec["send"], rec["recv"]          # (checked on the device against `both` below)
        else:
            hb = both.cpu().tolist()
            send_counts, recv_counts = [int(v) for v in hb[:world]], [int(v) for v in hb[world:]]
        nrecv = sum(recv_counts)
        recv = _scratch(eng, "recv", max(nrecv, 1) * rw, dev)
        dist.all_to_all_single(recv[: nrecv * rw], send[: nreq * rw],
                               output_split_sizes=[c * rw for c in recv_counts],
                               input_split_sizes=[c * rw for c in send_counts], group=group)
        missing = eng.apply(recv, nrecv, wait=False)       # queued: the count of missing complements stays on the device
    else:
        nrecv = nreq
        missing = eng.apply_own()          # every complement is local: no routing, no copy

    # pass 2 runs before the symmetry proof is known (its result is discarded when the proof fails): the proof
    # words ride at the end of the histogram buffer, so ONE all_reduce carries both (sums wrap mod 2^64)
    # Layout behind the plot: [missing count] + two words per rank + [replayed steps that found other counts, replayed steps].  The
    # fingerprint is an XOR over the table, which a SUM all_reduce cannot combine -- so every rank writes its 128-bit residue
    # into ITS OWN two words (zeros elsewhere), the sum hands every rank all residues, and the XOR over the ranks is taken on
    # the host.
    nslot = world if exchange else 1
    nproof = 3 + 2 * nslot
    # (+ 1 word behind the proof: how many ranks hold a record that the NEXT step could be replayed from)
    buf = _scratch(eng, "plot", PLOT_CELLS + nproof + 1, dev)     # (pass 2 clears the plot itself; the proof words below)
    plot = buf[:PLOT_CELLS]
    eng.pass2(plot)
    me = rank if exchange else 0
    if missing is None and hasattr(eng, "proof_tail"):
        # the engine lays the whole tail out on the device, in stream order, in one launch: no host round trip between the
        # look-ups and the all_reduce (a replayed step's verdict on its counts and on the router's totals included)
        eng.proof_tail(buf[PLOT_CELLS: PLOT_CELLS + nproof], nslot, me)
    elif missing is None:
        buf[PLOT_CELLS: PLOT_CELLS + nproof].zero_()
        # the engine writes (missing, residue word 0, residue word 1, replay verdict) on the device, in stream order: no host
        # round trip between the look-ups and the all_reduce.  Rank r's words go to [0] (summed) and to ITS slot.
        tmp = _scratch(eng, "proof", 4, dev)
        eng.proof_into(tmp)
        buf[PLOT_CELLS] = tmp[0]
        buf[PLOT_CELLS + 1 + 2 * me: PLOT_CELLS + 3 + 2 * me] = tmp[1:3]
        if replay:
            # ... and this driver's part of the verdict: the counts it split the exchange by are the ones the router produced
            buf[PLOT_CELLS + nproof - 2] = tmp[3] + (both != rec["both"]).any().to(torch.int64)
            buf[PLOT_CELLS + nproof - 1] = 1
    else:
        fpw = eng.symhash()
        proof = np.zeros(nproof, dtype=np.uint64)
        proof[0] = missing
        proof[1 + 2 * me] = fpw[0] ^ fpw[2]
        proof[2 + 2 * me] = fpw[1] ^ fpw[3]
        buf[PLOT_CELLS: PLOT_CELLS + nproof] = torch.from_numpy(proof.view(np.int64).copy()).to(dev)
    # this rank's vote on replaying the next step: it holds a record (a replayed step keeps the one it ran from)
    have_rec = bool(can_replay and both is not None and (replay or (eng.replay_state() & 2)))
    buf[PLOT_CELLS + nproof:].fill_(1 if have_rec else 0)
    if exchange:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    pv = buf[PLOT_CELLS:].cpu().numpy().view(np.uint64)
    all_have = int(pv[nproof]) == (world if exchange else 1)
    pv = pv[:nproof]
    replay_bad, replayed = int(pv[nproof - 2]) != 0, int(pv[nproof - 1]) != 0       # (sums over the ranks: the same everywhere)
    symmetric = pv[0] == 0
    if symcheck == "hash":
        symmetric = symmetric and int(np.bitwise_xor.reduce(pv[1:nproof - 2:2])) == 0 and int(np.bitwise_xor.reduce(pv[2:nproof - 2:2])) == 0
    if can_replay:
        if replay:
            eng.replay_done(not replay_bad and bool(symmetric))
            rec["all"] = all_have
        if replay_bad or (not symmetric and replayed):
            # some rank ran from a record that no longer holds (or the table stopped being closed under a replayed step:
            # the plain path decides): forget the records and run the step again
            eng._replay_rec = None
            eng.set_replay(False)
            return hetmers_sharded(k, keys, counts, symcheck=symcheck, engine_factory=engine_factory, group=group, eng=eng,
                                   fallback=fallback, splitters=splitters, prebound=True, sizes=sizes)
        if symmetric and not replay and have_rec:
            # (the receive counts are other ranks' send totals: every rank vouches for its own on the device, and the verdicts
            #  are summed -- a replayed step in which ANY count moved is run again by all)
            eng._replay_rec = {"key": rkey, "send": send_counts, "recv": recv_counts, "both": both.clone(), "all": all_have}
    if not symmetric:
        if not fallback:
            raise NotSymmetric("table is not closed under reverse complement with equal counts; "
                               "run the single-GPU engine (general path) or condition the table")
        _general_on_rank0(k, keys, counts, sizes, eng, plot, group, rank, world if exchange else 1, words)
        eng._splitter_cache = None          # (rank 0's engine is bound to the gathered table now)
        eng._replay_rec = None
    st = eng.stats()
    st.update(rank=rank, world=world, shard_nels=n, sent=nreq if exchange else 0, received=nrecv, engine=eng, replayed=replay)
    if not symmetric:
        st["path"] = 2
    return plot.clone(), st           # (the reduction buffer belongs to the engine: the caller gets its own 4 MB)
