#!/usr/bin/env python3
"""bench.py -- k-mers/s through hetmers (k=31) on N MI355X GPUs, with roofline + CPU baseline.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched under
torch.distributed.run with one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

A "step" = one complete hetmers computation (pass 1 with the symmetry proof -> complement look-ups /
exchange -> pass 2 -> histogram reduction) over the whole device-resident table.  The table is handed
to the engine once, before the timed region, together with its FastK prefix index (what the stub of a
.ktab file carries; the generator stands in for the file and supplies it: config.directory, --no-index).
Workload at N=1 is BASELINE.json configs[2]: "Synthetic diploid 1 Gbp, 50x cov, k=31" generated on device
(smudgeplot_amd/synth_device.py); --genome scales it, --workload picks the repeats / octoploid (configs[3]) /
hexaploid k=51 (configs[4]) stand-ins (stated in config.workload).
N>1: STRONG scaling -- the same table, prefix-sharded across the ranks (every rank generates its own shard, none holds
the whole table); one all_to_all of the complement requests and one all_reduce of the 2-D histogram per step.
After the timed loop the plot of the last step is rendered as .smu text and compared with the REFERENCE binary's output
for this very table (tests/golden/bench_<workload>.smu; "parity" in the JSON line); a mismatch exits non-zero.
Then, still outside the timed region, the END-TO-END comparison of SURVEY.md section 8d (ii) / BASELINE.md section 3 is run in
this very process ("e2e" in the JSON line): a ~1e9-entry table of the bench's own generator is written as FastK files, the
drop-in executable (smudgeplot_amd/bin/hetmers) and the reference binary (oracle/_ref/hetmers_ref) run on those IDENTICAL files,
wall clock around each process, and the two .smu files are compared byte for byte.  "cpu_baseline" is that reference run.

value = table entries (k-mers) x steps / wall time, wall = max over ranks between barriers.
"""

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from smudgeplot_amd import engine, ktab, sharded, synth, synth_device  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
def alg_bytes_per_kmer_pass(k):
    """TBYTE = ceil(k/4)+2 read per pass + 1 degree byte written/read: 11 B at k=31 (22 B/k-mer over both
    passes), 16 B at k=51"""
    return (k + 3) // 4 + 2 + 1


WORKLOADS = ("uniform", "repeats", "octoploid", "hexaploid")


def default_k(workload: str) -> int:
    return 51 if workload == "hexaploid" else 31


def default_genome(workload: str, k: int = 31) -> int:
    """haploid genome size: BASELINE configs[2] is 1 Gbp; configs[3] "8 haplotypes x 0.2 Gbp (scaled to fit)" (SURVEY.md
    section 8d); configs[4] is a 10 Gbp hexaploid on 8 GPUs -- 4e8 bp of six haplotypes give one GPU 1.4e9 two-word entries,
    about an eighth of that table"""
    return {"uniform": 10 ** 9, "repeats": 10 ** 9, "octoploid": 2 * 10 ** 8, "hexaploid": 4 * 10 ** 8}[workload]


def make_table(workload: str, G: int, k: int, dev, seed: int = 1, key_range=None):
    """-> (keys, counts, L, description): the conditioned (trimmed, rc-closed) table of a workload, generated on `dev`.
    key_range = synth_device.key_range_of(rank, world): only this rank's prefix shard of that table (N > 1: no rank ever
    holds the whole table; the shards of all ranks, in rank order, are the table of N = 1 entry for entry)"""
    kr = {} if key_range is None else {"key_range": key_range}
    if workload in ("uniform", "repeats"):
        L = 10
        rep = 0.05 if workload == "repeats" else 0.0
        if k <= 31:
            keys, cnt = synth_device.diploid_table(G, k=k, het=0.01, cov=50.0, L=L, seed=seed, device=dev, repeats=rep, **kr)
        else:                                  # two-word k-mers (BASELINE configs[4] is k=51); keys is [n, 2].  ONE generator whatever
            # the size and the number of ranks (chunk by chunk of the key space), so that the shards of N ranks are the table of N = 1
            keys, cnt = synth_device.polyploid_table_wide(G, ploidy=2, rates=(0.01,), cov_hap=25.0, k=k, L=L, seed=seed, device=dev, **kr)
        desc = f"synthetic diploid {G:.3g} bp, 50x, 1% het, k={k}, L={L}" + \
               (", 5% of the genome repeats (dispersed, tandem, homopolymer)" if rep else "")
    elif workload == "octoploid":
        assert k <= 31
        L = 8
        keys, cnt = synth_device.polyploid_table_graded(G, ploidy=8, cov_hap=14.0, k=k, L=L, seed=3 + seed, device=dev, **kr)
        desc = (f"synthetic octoploid {G:.3g} bp x 8 haplotypes, graded divergences (variant sets carried by 1, 2, 3 and 4 of 8 "
                f"haplotypes at 0.1 / 0.15 / 0.1 / 0.2 %), 14x per haplotype, k={k}, L={L}")
    else:
        assert 33 <= k <= 64
        L = 5
        keys, cnt = synth_device.polyploid_table_wide(G, ploidy=6, cov_hap=10.0, k=k, L=L, seed=4 + seed, device=dev, **kr)
        desc = (f"synthetic hexaploid {G:.3g} bp x 6 haplotypes, graded divergences (variant sets carried by 1, 2 and 3 of 6 "
                f"haplotypes at 0.1 / 0.15 / 0.2 %), 10x per haplotype, k={k}, L={L}")
    return keys, cnt, L, desc


def make_shard(workload: str, G: int, k: int, dev, rank: int, world: int, with_index: bool = True, group=None):
    """This rank's prefix shard of a bench table and what the sharded run needs to know about the whole of it -- collective when
    world > 1 (any backend: the CPU test-suite runs it under gloo).  Every rank GENERATES its own shard (synth_device
    key_range_of: equal slices of the key space, cut on window-block boundaries), so no rank ever holds the table.
    -> dict: keys, counts, L, desc, sizes (entries of every rank), n_total, first_entry (number of this shard's first entry in
    the whole table), index (the WHOLE table's FastK prefix index, int64[2^24]: bucket counts summed over the ranks -- what the
    stub of a .ktab carries, libfastk.c:841 -- or None), splitters (the cut values: lower bound of the k-mer range of ranks 1 ..
    world-1, W words each; None for one rank), hk / hc (synth_device.table_hash of the whole table: the shards' sums)."""
    words = (k + 31) // 32
    key_range = synth_device.key_range_of(rank, world) if world > 1 else None
    keys, cnt, L, desc = make_table(workload, G, k, dev, key_range=key_range)
    n_local = cnt.numel()
    kw0 = keys if keys.dim() == 1 else keys[:, 0]          # the word that holds the window-block prefix
    sizes = [n_local]
    if world > 1:
        mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
        allv = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine, group=group)
        sizes = [int(v) for v in torch.cat(allv).cpu().tolist()]
    first_entry = sum(sizes[:rank])
    # A FastK table comes with its prefix index (entries up to every 3-byte prefix: the stub of the .ktab file,
    # libfastk.c:841); the generator stands in for the table file, so it supplies the index too -- the engine takes it
    # as its look-up directory (the `hetmers` executable hands over the index it read from the stub the same way)
    index = None
    if with_index:
        per = torch.bincount((kw0 >> 40) & 0xFFFFFF, minlength=1 << 24)
        if world > 1:
            dist.all_reduce(per, op=dist.ReduceOp.SUM, group=group)
        index = torch.cumsum(per, 0)
        del per
    splitters = None
    if world > 1:
        splitters = np.zeros((world - 1, words), dtype=np.uint64)
        for r in range(1, world):
            splitters[r - 1, 0] = np.uint64(synth_device.key_range_of(r, world)[0]) << np.uint64(48)
        splitters = splitters.reshape(-1)
    # checksum of the table (summed over the ranks): what the golden .smu of the parity block was made on
    hk, hc = synth_device.table_hash(keys, cnt, first_entry=first_entry)
    if world > 1:
        hv = torch.tensor([hk - (1 << 64) if hk >> 63 else hk, hc - (1 << 64) if hc >> 63 else hc], dtype=torch.int64, device=dev)
        dist.all_reduce(hv, op=dist.ReduceOp.SUM, group=group)
        hk, hc = [int(v) & 0xFFFFFFFFFFFFFFFF for v in hv.cpu().tolist()]
    return {"keys": keys, "counts": cnt, "L": L, "desc": desc, "sizes": sizes, "n_total": sum(sizes), "first_entry": first_entry,
            "index": index, "splitters": splitters, "hk": hk, "hc": hc}


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def golden_key(workload: str, G: int, k: int) -> str:
    """name of a bench table in tests/golden/bench_tables.json: the workload at its default size and k, or
    <workload>_k<k>_<genome>Mbp for the other lines that are kept under profiles/ (k = 30, k = 51 diploid)"""
    if int(G) == default_genome(workload) and int(k) == default_k(workload):
        return workload
    return "%s_k%d_%dMbp" % (workload, int(k), int(G) // 10 ** 6)


def parity_against_golden(workload: str, G: int, k: int, n_total: int, hk: int, hc: int, plot) -> dict:
    """Compare the plot of the last timed step with what the REFERENCE binary (PloidyPlot.c, compiled from the reference's
    own sources) wrote for this very table: tests/golden/bench_<workload>.smu, made on the GPU box by
    tools/make_bench_goldens.py together with the table's checksum (tests/golden/bench_tables.json).  The table is
    identified by synth_device.table_hash (an order-sensitive checksum of every k-mer word and count, summed over the
    ranks), the result by the bytes of the .smu text (PloidyPlot.c:1603-1617).  ok = None: no golden for this size."""
    import hashlib
    th = synth_device.table_hash_text(n_total, hk, hc)
    out = {"vs": "reference binary golden", "table_hash": th, "ok": None}
    tj = os.path.join(GOLDEN_DIR, "bench_tables.json")
    if not os.path.exists(tj):
        out["reason"] = "tests/golden/bench_tables.json is missing"
        return out
    with open(tj) as f:
        g = json.load(f).get(golden_key(workload, G, k))
    if not g or int(g["genome"]) != int(G) or int(g["k"]) != int(k):
        out["reason"] = "no golden for this workload at this size and k (tests/golden/bench_tables.json lists the ones there are)"
        return out
    smu = engine.smu_text(plot.cpu().numpy().reshape(engine.PLOT_ROWS, engine.PLOT_COLS))
    out["smu_sha256"] = hashlib.sha256(smu.encode()).hexdigest()
    out["golden"] = "tests/golden/" + g["smu_file"]
    out["same_table"] = g["table_hash"] == th
    with open(os.path.join(GOLDEN_DIR, g["smu_file"])) as f:
        out["ok"] = bool(out["same_table"] and f.read() == smu)
    if not out["same_table"]:
        out["reason"] = "the generator produced another table than the one the golden was made on (%s)" % g["table_hash"]
    return out


def lib_hash() -> str:
    """identity of the engine's device code (sha256 of the gfx950 code object inside the library):
    profiles/hbm_traffic.json carries the one its counters were read on"""
    from smudgeplot_amd import codeobj
    return codeobj.code_object_hash()


REF_BIN = os.path.join(ROOT, "oracle", "_ref", "hetmers_ref")
OUR_BIN = os.path.join(ROOT, "smudgeplot_amd", "bin", "hetmers")


def e2e_genome(workload: str, G: int) -> int:
    """size of the end-to-end table: the north-star's "1e9-entry k=31 table" for the diploid workloads (4e8 bp -> 1.01e9
    entries; the reference takes about a minute on it at -T64), the bench table itself for the octoploid stand-in, 1.5e8 bp
    of the hexaploid one (the reference needs 140 s for the whole of it)"""
    cap = {"uniform": 4 * 10 ** 8, "repeats": 4 * 10 ** 8, "octoploid": 2 * 10 ** 8, "hexaploid": 15 * 10 ** 7}[workload]
    return min(int(G), cap)


def _scratch_dir(nbytes: int) -> str:
    """where the table files go: memory-backed /dev/shm when it has room for them twice over (the files are read from the page
    cache either way: both programs are timed warm), else the default temporary directory"""
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > 2 * nbytes + (8 << 30):
            return "/dev/shm"
    except OSError:
        pass
    return tempfile.gettempdir()


def end_to_end(workload: str, G: int, k: int, dev, runs: int = 2):
    """process start -> .smu closed, on the IDENTICAL table files, for the drop-in executable and for
    the reference binary, timed in THIS process on this box's host cores; the two .smu files are compared byte for byte.
    -> (e2e block, cpu_baseline block).  The reference binary is the judge and the baseline here, never part of the product
    path; without it (or without the executable) the bench fails loudly instead of printing a line without a baseline."""
    for path, what in ((REF_BIN, "the reference binary (python -c 'import __graft_entry__ as g; g.build()' builds it where "
                                 "/root/reference exists; it travels to the GPU box with the tree)"),
                       (OUR_BIN, "the drop-in executable (make -C smudgeplot_amd/csrc)")):
        if not (os.path.isfile(path) and os.access(path, os.X_OK)):
            raise SystemExit(f"bench.py: {os.path.relpath(path, ROOT)} is missing -- {what}; --no-cpu runs without the CPU comparison")
    cores = min(64, os.cpu_count() or 1)                        # (the reference clamps -T at 64, PloidyPlot.c:1279-1282)
    t0 = time.time()
    tk, tc, L, desc = make_table(workload, G, k, dev)           # the bench's own generator, the bench's own seed
    torch.cuda.synchronize()
    n = tc.numel()
    t_gen = time.time() - t0
    nbytes = n * ((k + 3) // 4 - 3 + 2) + (1 << 27)
    with tempfile.TemporaryDirectory(prefix="smg_e2e", dir=_scratch_dir(nbytes)) as d:
        t0 = time.time()
        written = synth_device.write_table_from_device(os.path.join(d, "t"), tk, tc, k, nparts=4)
        t_write = time.time() - t0
        del tk, tc
        torch.cuda.empty_cache()

        def timed(argv, out):
            p = os.path.join(d, out + ".smu")
            if os.path.exists(p):
                os.remove(p)
            with open(os.path.join(d, out + ".err"), "w") as err:      # (files, not pipes: a pipe is read until the GPU worker
                t0 = time.perf_counter()                               #  process has closed its end too, hetmers_main.c)
                r = subprocess.run(argv, cwd=d, stdin=subprocess.DEVNULL, stdout=err, stderr=err)
                dt = time.perf_counter() - t0
            if r.returncode != 0:
                raise SystemExit("bench.py: %s failed (exit %d): %s" % (" ".join(argv), r.returncode,
                                                                        open(os.path.join(d, out + ".err")).read()[-2000:]))
            return dt, open(p, "rb").read()

        ours = [timed([OUR_BIN, f"-e{L}", "-T4", "-ogpu", "t.ktab"], "gpu") for _ in range(runs)]
        ref_s, ref_smu = timed([REF_BIN, f"-e{L}", f"-T{cores}", "-oref", "t.ktab"], "ref")
    best = min(dt for dt, _ in ours)
    same = all(smu == ref_smu for _, smu in ours)
    e2e = {"entries": int(n), "k": k, "table": f"{desc}: {n} entries, {written} bytes in 4 part files + stub (format F, ibyte 3)",
           "hetmers_wall_s": round(best, 3), "hetmers_runs_s": [round(dt, 3) for dt, _ in ours],
           "reference_wall_s": round(ref_s, 3), "cores": cores, "smu_identical": bool(same), "smu_bytes": len(ref_smu),
           "speedup": round(ref_s / best, 1),
           "what": "wall clock around each process (exec -> exit), same files, page cache warm from the write; hetmers = "
                   "smudgeplot_amd/bin/hetmers -T4 (two processes: hetmers_main.c), reference = oracle/_ref/hetmers_ref "
                   f"-T{cores}; generated in {t_gen:.1f} s, written in {t_write:.1f} s"}
    cpu = {"value": n / ref_s, "unit": "k-mers/s", "cores": cores, "kind": "reference",
           "sample": f"reference hetmers -T{cores}, wall clock of the whole process on a {n}-entry table of the bench's generator "
                     f"(genome {G} bp, k={k}): {ref_s:.1f} s, .smu {'identical to' if same else 'DIFFERENT from'} the drop-in's "
                     "on the same files (the e2e block)"}
    return e2e, cpu


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome", type=float, default=0, help="haploid genome size in bases (default: the workload's, 1e9 for uniform)")
    ap.add_argument("--k", type=int, default=0, help="default 31 (hexaploid: 51)")
    ap.add_argument("--symcheck", default="hash", choices=["exact", "hash"])
    ap.add_argument("--e2e-genome", type=float, default=0, help="genome size of the end-to-end / CPU-baseline table (default: "
                                                                "4e8 bp = 1.01e9 entries for the diploid workloads)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the end-to-end comparison with the reference binary (e2e, cpu_baseline)")
    ap.add_argument("--no-index", action="store_true", help="do not hand the table's prefix index to the engine (pass 1 builds "
                                                            "a directory of its own, as in rounds 1-3)")
    ap.add_argument("--workload", default="uniform", choices=list(WORKLOADS),
                    help="uniform: BASELINE configs[2] (uniform random genome); repeats: 5 %% of the genome are dispersed / "
                         "tandem repeats and homopolymer runs (exercises kf_bigfix and the repeat tail of the plot); "
                         "octoploid: stand-in for configs[3] (8 graded haplotypes, k=31); hexaploid: for configs[4] (6, k=51)")
    args = ap.parse_args()
    if not args.k:
        args.k = default_k(args.workload)
    if not args.genome:
        args.genome = default_genome(args.workload, args.k)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force = os.environ.get("SMG_FORCE_EXCHANGE") == "1"     # diagnostics: the N > 1 protocol in a one-rank group
    if world > 1 or force:
        if force and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    # ---- workload: every rank generates ITS prefix shard only (the generators cut the key space; no rank holds the table)
    G = int(args.genome)
    sh = make_shard(args.workload, G, args.k, dev, rank, world, with_index=not args.no_index)
    keys, cnt, L, desc = sh["keys"], sh["counts"], sh["L"], sh["desc"]
    n_local, n_total, sizes, first_entry = cnt.numel(), sh["n_total"], sh["sizes"], sh["first_entry"]
    index, splitters, hk, hc = sh["index"], sh["splitters"], sh["hk"], sh["hc"]
    del sh
    keys = keys.reshape(-1)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()

    eng_stats = []
    eng = sharded.TorchEngine(dev)        # device buffers are allocated once and reused by every step
    eng.bind(args.k, keys, cnt, index=index, first_entry=first_entry)       # the table is handed over once, like a file is read once
    del index

    def step():
        plot, st = sharded.hetmers_sharded(args.k, keys, cnt, symcheck=args.symcheck, eng=eng, prebound=True,
                                           splitters=splitters, sizes=sizes if world > 1 else None)
        st.pop("engine", None)
        eng_stats.append(st)
        return plot

    for _ in range(args.warmup):
        step()
    eng_stats.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plot = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline of the dominant kernel, from HIP events recorded by the engine on its stream
    ms = {key: float(np.mean([s[key] for s in eng_stats])) for key in ("ms_pass1", "ms_rclookup", "ms_pass2")}
    nreq = float(np.mean([s.get("nemitted", s["nrequests"]) for s in eng_stats]))   # requests pass 1 emitted
    nkept = float(np.mean([s["nrequests"] for s in eng_stats]))                     # ... and the filter kept
    ms_filter = float(np.mean([s.get("ms_filter", 0.0) for s in eng_stats]))
    # Algorithmic bytes per launch: the two scan kernels move B_alg/2 = 11 B per
    # entry each (10 B record + 1 B degree/code); the look-up phase moves one 8-byte k-mer + 2-byte
    # count per request (per entry in exact mode, where every complement is looked up).
    alg = {"ms_pass1": n_local * float(alg_bytes_per_kmer_pass(args.k)),
           "ms_pass2": n_local * float(alg_bytes_per_kmer_pass(args.k)),
           "ms_rclookup": (nreq if args.symcheck == "hash" else n_local) * 10.0}
    # the dominant KERNEL: pass 1 and pass 2 are one launch each; the look-up phase is a chain of
    # short launches (compact, 4 radix passes, in-order look-ups), each well below pass 1
    tf = lambda b: "true" if b else "false"                                   # noqa: E731
    # (the name as rocprofv3 prints it: <W, RW, ODD, KF, VAR>; VAR = 2 is the hot form -- hash proof, the table's prefix index
    #  as directory, the 32-bit two-bit candidate map of a run that exchanges no maps: fast_pass1 in smg_hetmers.hip)
    var = 2 if (args.symcheck == "hash" and not args.no_index and not (world > 1 or force) and args.k >= 24) else 1
    if args.k <= 32:
        p1 = "kf_pass1_d<1, %d, %s, %s, %d>" % (1 if args.symcheck == "hash" else 2, tf(args.k & 1), tf(17 <= args.k), var)
    elif args.k <= 64:
        p1 = "kf_pass1_d<2, %d, %s, false, %d>" % (2 if args.symcheck == "hash" else 3, tf(args.k & 1), var)
    else:
        p1 = "kf_pass1<3>"
    single = {"ms_pass1": p1, "ms_pass2": "kf_pass2<%d>" % ((args.k + 31) // 32)}
    dom = max(single, key=ms.get)
    achieved = alg[dom] / (ms[dom] * 1e-3) / 1e9 if ms[dom] > 0 else 0.0
    traffic, traffic_src = None, None
    tj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tj):
        with open(tj) as f:
            t = json.load(f)
        t = t.get("k%d%s" % (args.k, "" if args.workload == "uniform" else "_" + args.workload), {})
        if dom in t.get("bytes_per_entry", {}):
            # counters are read in separate rocprofv3 passes (tools/r04_measure.sh), so the figure belongs to the build it
            # was read on: another build of the engine gets no traffic figure instead of a stale one
            if t.get("code_object_sha256_16") == lib_hash():
                traffic = t["bytes_per_entry"][dom] * n_local
                traffic_src = t.get("source")
            else:
                traffic_src = "stale: profiles/hbm_traffic.json was measured on code object %s, this library carries %s" % (
                    t.get("code_object_sha256_16"), lib_hash())

    e2e_failure = None
    if rank == 0:
        parity = parity_against_golden(args.workload, G, args.k, n_total, hk, hc, plot)
        # the end-to-end comparison and the CPU baseline: rank 0 of the single-GPU run only (the reference takes about a minute)
        cpu = e2e = None
        if not (args.no_cpu or world > 1):
            try:
                e2e, cpu = end_to_end(args.workload, int(args.e2e_genome) if args.e2e_genome else e2e_genome(args.workload, G), args.k, dev)
            except SystemExit as ex:
                # (a missing reference binary or a failing program: the line of the timed region is still printed -- with the
                #  error in place of the block -- and the process then FAILS with the message: loud, but no measurement is lost)
                e2e_failure = str(ex)
                e2e = {"error": e2e_failure, "smu_identical": None}
        value = n_total * args.steps / dt
        out = {
            "metric": "k-mers/sec through hetmers (k=%d)" % args.k,
            "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64" if args.k <= 32 else "u64x%d" % ((args.k + 31) // 32), "data": "synthetic",
            "config": {"workload": desc + f": {n_total} table entries (conditioned, rc-closed)",
                       "symcheck": args.symcheck, "sharding": f"prefix x{world}",
                       "directory": "pass 1 builds its own" if args.no_index else
                                    "the table's FastK prefix index (2^24 buckets, generated with the table as a .ktab stub carries it)"},
            "roofline": {"bound": "hbm", "kernel": single[dom],
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg[dom], "kernel_ms": ms,
                         # ms_pass1 brackets kf_pass1_d AND the two short launches behind it (kf_collect, kf_bigfix =
                         # "ms_bigfix" below); `achieved` is priced on the whole bracket, the kernel alone is:
                         "pass1_kernel_alone_ms": ms["ms_pass1"] - float(np.mean([s.get("ms_bigfix", 0.0) for s in eng_stats])),
                         "requests": {"emitted": nreq, "kept_by_filter": nkept, "ms_partition": ms_filter},
                         "deferred_entries": {"count": float(np.mean([s.get("nbig", 0) for s in eng_stats])),
                                              "ms_bigfix": float(np.mean([s.get("ms_bigfix", 0.0) for s in eng_stats]))},
                         "lookup_phase_GBps": alg["ms_rclookup"] / (ms["ms_rclookup"] * 1e-3) / 1e9
                         if ms["ms_rclookup"] > 0 else 0.0,
                         # whole job against B_alg(k) = 2 x (ceil(k/4) + 2) + 2 bytes per k-mer (22 at k = 31, 32 at k = 51)
                         "whole_job_frac_of_the_B_alg_roofline":
                             (n_total * 2.0 * alg_bytes_per_kmer_pass(args.k) / (dt / args.steps)) / 1e9 / (HBM_PEAK_GBS * world)},
            "cpu_baseline": cpu,
            # process start -> .smu closed on identical table files: the drop-in executable next to the reference binary
            "e2e": e2e,
            # the plot of the last TIMED step against the reference binary's .smu of this very table (outside the timed region)
            "parity": parity,
            "pairs_in_plot": int(plot.sum().item()),
            "hbm_peak_allocated_by_torch_GB": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),     # (the generator's peak)
        }
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0 and e2e_failure is not None:
        raise SystemExit(e2e_failure)
    if rank == 0 and e2e is not None and not e2e["smu_identical"]:
        raise SystemExit("bench.py: the drop-in executable's .smu differs from the reference binary's on the end-to-end table")
    if rank == 0 and parity["ok"] is False:
        raise SystemExit("bench.py: the plot of the timed run differs from the reference binary's .smu of this table (%s)"
                         % parity.get("golden"))


if __name__ == "__main__":
    main()
