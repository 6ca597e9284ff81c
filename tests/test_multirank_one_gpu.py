"""`bench.py --gpus N` with the REAL engine and N > 1 ranks -- on the one GPU a test box has.

RCCL refuses two ranks on one device, so the ranks of these tests (N processes, each with its own HIP context on cuda:0 and
its own engine) talk through `gloo`, with every collective of `torch.distributed` staged through host memory by the shim below
(test infrastructure: the product code is untouched and calls `dist.*` on device tensors exactly as it does under RCCL).
What runs is everything of the N-rank bench step but the fabric: per-rank generation of the shard (bench.make_shard), the
summed prefix index handed to the engine with the shard's first entry, cut-value splitters, the 30-bit / (8 ranks:) 29-bit
candidate map gathered straight into place, the device-side router, the exchange with uneven splits, the look-ups of
received requests, pass 2, the proof tail in the one all_reduce -- and the fall-back to rank 0 for a table that fails the proof.
The summed plot must equal the one-engine run on the whole table (which tests/test_gpu_parity.py pins to the reference).
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stage_collectives_through_host():
    """gloo moves host memory: copy in, run the collective, copy out.  `.cpu()` waits for the current stream -- the engine's --
    so the order of the device work around a collective is what RCCL's stream semantics give."""
    import torch
    import torch.distributed as d
    real = {n: getattr(d, n) for n in ("all_reduce", "all_gather", "all_gather_into_tensor", "all_to_all_single", "send", "recv",
                                        "broadcast")}

    class Done:
        def __init__(self, fin):
            self.fin = fin

        def wait(self):
            self.fin()

    def all_reduce(t, op=d.ReduceOp.SUM, group=None, async_op=False):
        h = t.cpu()
        real["all_reduce"](h, op=op, group=group)
        t.copy_(h)

    def all_gather(lst, t, group=None, async_op=False):
        h = t.cpu()
        hl = [torch.empty_like(h) for _ in lst]
        real["all_gather"](hl, h, group=group)
        for a, b in zip(lst, hl):
            a.copy_(b)

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        def fin():
            h = inp.cpu()
            ho = torch.empty(out.shape, dtype=out.dtype)
            real["all_gather_into_tensor"](ho, h, group=group)
            out.copy_(ho)
        if async_op:
            return Done(fin)          # (the work the caller queues meanwhile is in front of the copy: as under RCCL)
        fin()

    def all_to_all_single(out, inp, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        h = inp.cpu()
        ho = torch.empty(out.shape, dtype=out.dtype)
        real["all_to_all_single"](ho, h, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)
        out.copy_(ho)

    def send(t, dst, group=None, tag=0):
        real["send"](t.cpu(), dst=dst, group=group, tag=tag)

    def recv(t, src=None, group=None, tag=0):
        h = torch.empty(t.shape, dtype=t.dtype)
        real["recv"](h, src=src, group=group, tag=tag)
        t.copy_(h)

    def broadcast(t, src, group=None, async_op=False):
        h = t.cpu()
        real["broadcast"](h, src=src, group=group)
        t.copy_(h)

    for n, f in (("all_reduce", all_reduce), ("all_gather", all_gather), ("all_gather_into_tensor", all_gather_into_tensor),
                 ("all_to_all_single", all_to_all_single), ("send", send), ("recv", recv), ("broadcast", broadcast)):
        setattr(d, n, f)


def _worker(rank, world, port, workload, G, k, steps, replay, drop, q, table=None):
    try:
        import faulthandler
        faulthandler.dump_traceback_later(150, exit=True)          # a rank that hangs says where, and ends
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if replay:
            os.environ["SMG_REPLAY"] = "1"
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        _stage_collectives_through_host()
        import bench
        from smudgeplot_amd import sharded
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        if table is not None:
            # the parent generated the table and hands it over in shared host memory; this rank takes its key range of it
            # (the generators read counts back from the device -- polyploid_table_wide some hundred times -- and with eight
            #  processes taking turns on one GPU each of those waits its turn: minutes for a 2e7-entry table.  Per-rank
            #  generation on the device is what the 2- and 3-rank cases do; tests/test_bench_shards.py has it for all workloads)
            from smudgeplot_amd import synth_device
            tk, tc, tL = table

            def from_parent(workload_, G_, k_, dev_, seed=1, key_range=None):
                keep = synth_device._in_range16((tk[:, 0] >> 48) & 0xFFFF, key_range) if key_range is not None else slice(None)
                return tk[keep].to(dev_), tc[keep].to(dev_), tL, "a table generated by the parent process"
            bench.make_table = from_parent
        sh = bench.make_shard(workload, G, k, dev, rank, world)
        keys, cnt = sh["keys"].reshape(-1), sh["counts"]
        sizes, first = list(sh["sizes"]), sh["first_entry"]
        W = (k + 31) // 32
        index = sh["index"]
        if drop is not None and rank == drop[0]:
            # break the symmetry: this rank's entry #drop[1] goes (its complement stays, on whatever rank holds it)
            j = drop[1]
            keys = torch.cat([keys[: j * W], keys[(j + 1) * W:]]).contiguous()
            cnt = torch.cat([cnt[:j], cnt[j + 1:]]).contiguous()
        if drop is not None:
            sizes[drop[0]] -= 1
            index = None                     # (the index of the undamaged table does not describe this one)
            if rank > drop[0]:
                first -= 1
        eng = sharded.TorchEngine(dev)
        eng.bind(k, keys, cnt, index=index, first_entry=first)
        out = []
        for _ in range(steps):
            plot, st = sharded.hetmers_sharded(k, keys, cnt, symcheck="hash", eng=eng, prebound=True, splitters=sh["splitters"],
                                               sizes=sizes)
            torch.cuda.synchronize()
            out.append((plot.cpu().numpy().copy(), st["path"], st["sent"], st["received"], bool(st.get("replayed"))))
        par = bench.parity_against_golden(workload, G, k, sh["n_total"], sh["hk"], sh["hc"], plot)
        q.put((rank, None, sizes, sh["hk"], sh["hc"], out, par["ok"], eng.blockmap()[0]))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                    # noqa: BLE001 -- the parent must see WHY a rank died, not a queue timeout
        import traceback
        q.put((rank, "%r\n%s" % (e, traceback.format_exc()), None, 0, 0, None, None, 0))


def _run(world, workload, G, k, steps=2, replay=False, drop=None, table=None):
    import gc
    import torch
    import torch.multiprocessing as mp
    # the ranks share this process's GPU: what earlier tests of the session left in torch's cache (the full-size tables) goes back
    gc.collect()
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, workload, G, k, steps, replay, drop, q, table)) for r in range(world)]
    for p in procs:
        p.start()
    res, err = [], None
    try:
        while len(res) < world and err is None:
            r = q.get(timeout=180)
            if r[1] is not None:          # the first rank that raises ends the run: the others sit in a collective it never joins
                err = "rank %d: %s" % (r[0], r[1])
            res.append(r)
    finally:
        for p in procs:
            if err is None:
                p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert err is None, err
    return sorted(res, key=lambda t: t[0])


def _whole_table(workload, G, k):
    """the one-engine run on the whole table in this process: (plot, entries, hk, hc, keys [n, W], counts)"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from smudgeplot_amd import engine, sharded, synth_device
    dev = torch.device("cuda", 0)
    keys, cnt, L, _ = bench.make_table(workload, G, k, dev)
    W = (k + 31) // 32
    hk, hc = synth_device.table_hash(keys, cnt)
    _whole_table.L = L
    return dev, keys.reshape(-1, W), cnt, hk, hc


def _shared(keys, cnt):
    """the table in shared host memory, for ranks that take their key range of it instead of generating it (_worker)"""
    return keys.cpu().share_memory_(), cnt.cpu().share_memory_(), _whole_table.L


CASES = [
    # world, workload, genome, k
    (2, "uniform", 40_000_000, 31),        # 30-bit map, two halves of the key space
    (8, "uniform", 20_000_000, 31),        # what the driver's --gpus 8 run is: 29-bit map, eight slices
    (4, "octoploid", 8_000_000, 31),       # XCD-synchronous probing of exchanged requests
    (8, "hexaploid", 8_000_000, 51),       # two-word k-mers, 16-byte requests
    (3, "repeats", 20_000_000, 31),        # a world that does not divide the key space evenly; deferred entries
]


@pytest.mark.parametrize("world,workload,G,k", CASES)
def test_bench_step_with_real_engines_on_n_ranks(world, workload, G, k):
    import torch
    from smudgeplot_amd import engine, sharded
    if world >= 4:
        # (generated once, here: see _worker)
        dev, keys, cnt, hk, hc = _whole_table(workload, G, k)
        res = _run(world, workload, G, k, steps=2, table=_shared(keys, cnt))
    else:
        res = _run(world, workload, G, k, steps=2)
        dev, keys, cnt, hk, hc = _whole_table(workload, G, k)
    n = cnt.numel()
    plot, st = sharded.hetmers_sharded(k, keys.reshape(-1), cnt, symcheck="hash")
    torch.cuda.synchronize()
    want = plot.cpu().numpy()
    assert st["path"] == 1 and want.sum() > 0
    sizes = res[0][2]
    assert sum(sizes) == n and all(s > 0 for s in sizes)
    sent = [0, 0]
    for rank, _, rs, rhk, rhc, out, par_ok, bits in res:
        assert rs == sizes and (rhk, rhc) == (hk, hc)               # the ranks' shards ARE this table (checksum summed over them)
        assert bits == (29 if world >= 8 else 30)
        assert par_ok is None                                       # (no golden at this size)
        for step, (p, path, s, r, replayed) in enumerate(out):
            assert path == 1 and not replayed
            assert np.array_equal(p, want), (rank, step)
            sent[step] += s - r
    assert sent == [0, 0]                                           # every request that left a rank arrived at one
    assert sum(o[2] for r in res for o in [r[5][0]]) > 0            # ... and some did


def test_replayed_steps_with_real_engines_on_n_ranks():
    import torch
    from smudgeplot_amd import sharded
    world, workload, G, k = 4, "uniform", 20_000_000, 31
    dev, keys, cnt, hk, hc = _whole_table(workload, G, k)
    res = _run(world, workload, G, k, steps=4, replay=True, table=_shared(keys, cnt))
    plot, st = sharded.hetmers_sharded(k, keys.reshape(-1), cnt, symcheck="hash")
    want = plot.cpu().numpy()
    flags = None
    for rank, _, rs, rhk, rhc, out, par_ok, bits in res:
        assert (rhk, rhc) == (hk, hc)
        for p, path, s, r, replayed in out:
            assert path == 1 and np.array_equal(p, want)
        f = [o[4] for o in out]
        assert flags is None or f == flags                          # replay is a collective decision
        flags = f
    assert flags[0] is False and any(flags[1:]), flags              # the first step records, a later one runs from the record


def test_fallback_to_rank_0_with_real_engines():
    """one entry missing on rank 1: the proof fails on every rank, rank 0 collects the shards and runs the general path"""
    import torch
    from smudgeplot_amd import engine, sharded
    world, workload, G, k = 2, "uniform", 400_000, 31
    dev, keys, cnt, hk, hc = _whole_table(workload, G, k)
    from smudgeplot_amd import synth_device
    cut = np.uint64(synth_device.key_range_of(1, world)[0]) << np.uint64(48)
    first1 = int(torch.searchsorted(keys[:, 0].contiguous() ^ (-2 ** 63), torch.tensor([int(cut) - 2 ** 63], device=dev)).item())
    j = 5
    res = _run(world, workload, G, k, steps=2, drop=(1, j))
    g = first1 + j
    k2 = torch.cat([keys[:g], keys[g + 1:]]).contiguous()
    c2 = torch.cat([cnt[:g], cnt[g + 1:]]).contiguous()
    plot, st = sharded.hetmers_sharded(k, k2.reshape(-1), c2, symcheck="hash")
    torch.cuda.synchronize()
    assert st["path"] == 2
    want = plot.cpu().numpy()
    for rank, _, rs, rhk, rhc, out, par_ok, bits in res:
        assert rs[1] == res[0][2][1] and sum(rs) == cnt.numel() - 1
        for p, path, s, r, replayed in out:
            assert path == 2 and np.array_equal(p, want), rank
