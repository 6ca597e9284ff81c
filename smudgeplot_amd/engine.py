"""ctypes binding of the C ABI in include/smg_hetmers.h (libsmg_hetmers.so, built in-tree).

This is the reference-side binding a maintainer would add if smudgeplot called the engine
in-process instead of exec'ing the `hetmers` binary.  PyTorch is used only
as plumbing by callers (device tensors, streams, torch.distributed); nothing here imports it.

There is no CPU fallback: if the shared library is missing, or no HIP device is usable, the
calls raise `EngineError`.
"""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

SMAX, FMAX = 1000, 500
PLOT_ROWS, PLOT_COLS = SMAX + 1, FMAX + 1
PLOT_CELLS = PLOT_ROWS * PLOT_COLS

SYM_EXACT, SYM_HASH, SYM_NONE = 0, 1, 2
_SYM = {"exact": SYM_EXACT, "hash": SYM_HASH, "none": SYM_NONE}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMG_LIB", os.path.join(_HERE, "libsmg_hetmers.so"))   # SMG_LIB: tuning builds only
BIN_PATH = os.path.join(_HERE, "bin", "hetmers")


class EngineError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"smg_hetmers error {code}: {msg}")
        self.code = code


class TableView(C.Structure):
    _fields_ = [("kmer", C.c_int32), ("ibyte", C.c_int32), ("nparts", C.c_int32),
                ("minval", C.c_int32), ("nels", C.c_int64),
                ("part_data", C.POINTER(C.c_void_p)), ("part_nels", C.POINTER(C.c_int64)),
                ("prefix_index", C.POINTER(C.c_int64))]


class Opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("symcheck", C.c_int32), ("verbose", C.c_int32),
                ("condition", C.c_int32), ("ethresh", C.c_int32), ("ngpus", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("nels", C.c_int64), ("npairs", C.c_int64), ("nrequests", C.c_int64),
                ("path", C.c_int32), ("key_words", C.c_int32),
                ("ms_h2d", C.c_double), ("ms_decode", C.c_double), ("ms_pass1", C.c_double),
                ("ms_rclookup", C.c_double), ("ms_pass2", C.c_double), ("ms_total", C.c_double),
                ("nemitted", C.c_int64), ("ms_filter", C.c_double), ("nbig", C.c_int64), ("ms_bigfix", C.c_double)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


EXPORTS = [
    "smg_hetmers_run", "smg_hetmers_run_source", "smg_device_count", "smg_engine_create", "smg_engine_destroy",
    "smg_engine_decode", "smg_engine_bind", "smg_engine_set_prefix_index", "smg_engine_condition", "smg_engine_run", "smg_engine_pass1",
    "smg_engine_nreq", "smg_engine_record_words", "smg_engine_route", "smg_engine_route_device", "smg_engine_apply",
    "smg_engine_apply_own", "smg_engine_blockmap", "smg_engine_blockmap_copy", "smg_engine_filter",
    "smg_engine_presort", "smg_engine_merge_maps", "smg_engine_set_blockmap_bits",
    "smg_engine_symhash", "smg_engine_pass2", "smg_engine_stats", "smg_engine_proof",
    "smg_engine_set_replay", "smg_engine_replay_state", "smg_engine_replay_done", "smg_engine_proof_tail",
    "smg_engine_symm_hist", "smg_engine_symm_route", "smg_engine_symm_finish", "smg_engine_table",
    "smg_engine_extract", "smg_hetmers_extract", "smg_free", "smg_condition_table", "smg_version",
]

_lib = None


def _share_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so.7 / HSA runtime.  Two HIP runtimes in one
    process do not work (the second one finds no GPU), and stream handles must come from the
    runtime that launches on them.  Both copies carry the same soname, so whichever is loaded
    first serves everybody: if torch is installed but not imported yet, map ITS copy first so a
    later `import torch` and this library share one runtime.  Without torch (the plain-C
    `hetmers` executable) the system ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """dlopen the in-tree library; raises EngineError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise EngineError(-1, f"{LIB_PATH} is not built (run `python -c 'import __graft_entry__ as g; "
                              f"g.build()'` or `make -C smudgeplot_amd/csrc`)")
    lib = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    err = (C.c_char_p, C.c_size_t)
    lib.smg_version.restype = C.c_char_p
    lib.smg_device_count.restype = i32
    lib.smg_hetmers_run.argtypes = [C.POINTER(TableView), C.POINTER(Opts), vp, C.POINTER(Stats), *err]
    lib.smg_engine_create.restype = vp
    lib.smg_engine_create.argtypes = [i32, vp, *err]
    lib.smg_engine_destroy.argtypes = [vp]
    lib.smg_engine_decode.argtypes = [vp, i32, i32, i64, vp, vp, *err]
    lib.smg_engine_bind.argtypes = [vp, i32, i64, vp, vp, *err]
    lib.smg_engine_set_prefix_index.argtypes = [vp, vp, i32, i64, *err]
    lib.smg_engine_condition.argtypes = [vp, i32, i32, i32, C.POINTER(i64), *err]
    lib.smg_engine_run.argtypes = [vp, i32, vp, C.POINTER(Stats), *err]
    lib.smg_engine_pass1.argtypes = [vp, i32, *err]
    lib.smg_engine_nreq.restype = i64
    lib.smg_engine_nreq.argtypes = [vp]
    lib.smg_engine_record_words.argtypes = [vp]
    lib.smg_engine_route.argtypes = [vp, vp, i32, vp, i64, C.POINTER(i64), *err]
    lib.smg_engine_route_device.argtypes = [vp, vp, i32, vp, i64, vp, *err]
    lib.smg_engine_apply.argtypes = [vp, vp, i64, C.POINTER(i64), *err]
    lib.smg_engine_apply_own.argtypes = [vp, C.POINTER(i64), *err]
    lib.smg_engine_blockmap.argtypes = [vp, C.POINTER(i32), C.POINTER(i64)]
    lib.smg_engine_blockmap_copy.argtypes = [vp, i64, i64, vp, *err]
    lib.smg_engine_filter.argtypes = [vp, vp, C.POINTER(i64), *err]
    lib.smg_engine_presort.argtypes = [vp, *err]
    lib.smg_engine_set_blockmap_bits.argtypes = [vp, i32]
    lib.smg_engine_merge_maps.argtypes = [vp, vp, i64, i32, C.POINTER(i64), C.POINTER(i64), vp, *err]
    lib.smg_engine_symhash.argtypes = [vp, C.POINTER(C.c_uint64), *err]
    lib.smg_engine_proof.argtypes = [vp, vp, *err]
    lib.smg_engine_proof_tail.argtypes = [vp, vp, i32, i32, *err]
    lib.smg_engine_set_replay.argtypes = [vp, i32]
    lib.smg_engine_replay_state.argtypes = [vp]
    lib.smg_engine_replay_done.argtypes = [vp, i32, *err]
    lib.smg_engine_symm_hist.argtypes = [vp, i32, vp, *err]
    lib.smg_engine_symm_route.argtypes = [vp, vp, i32, vp, i64, C.POINTER(i64), *err]
    lib.smg_engine_symm_finish.argtypes = [vp, vp, i64, C.POINTER(i64), *err]
    lib.smg_engine_table.argtypes = [vp, C.POINTER(i64), C.POINTER(vp), C.POINTER(vp)]
    lib.smg_engine_pass2.argtypes = [vp, vp, *err]
    lib.smg_engine_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.smg_engine_extract.argtypes = [vp, vp, vp, i64, C.POINTER(i64), *err]
    lib.smg_hetmers_extract.argtypes = [C.POINTER(TableView), C.POINTER(Opts), vp, vp, C.POINTER(vp),
                                        C.POINTER(i64), C.POINTER(i32), C.POINTER(Stats), *err]
    lib.smg_free.argtypes = [vp]
    lib.smg_condition_table.argtypes = [C.POINTER(TableView), C.POINTER(Opts), C.POINTER(vp), C.POINTER(vp),
                                        C.POINTER(i64), C.POINTER(i32), *err]
    _lib = lib
    return lib


def _check(rc: int, buf) -> None:
    if rc != 0:
        raise EngineError(rc, buf.value.decode(errors="replace"))


def device_count() -> int:
    return int(load_library().smg_device_count())


COND_TRIM, COND_SYMM = 1, 2


def hetmers_run(table, device: int = 0, symcheck: str = "exact", verbose: int = 0, condition: int = 0,
                ethresh: int = 0, ngpus: int = 0):
    """Host FastK table (`ktab.KTable`) -> (plot int64[1001,501], stats dict).
    condition: COND_TRIM | COND_SYMM to trim to count >= ethresh / symmetrise on the device first
    (what the reference delegates to Logex / Symmex, PloidyPlot.c:1381-1414).

    In-process equivalent of running the `hetmers` executable on a conditioned table
    (reference: main(), src/lib/PloidyPlot.c:1433-1575).
    """
    lib = load_library()
    tv, keep = _table_view(table)
    opts = Opts(device, _SYM[symcheck], verbose, condition, ethresh, ngpus)
    plot = np.zeros(PLOT_CELLS, dtype=np.int64)
    st = Stats()
    buf = C.create_string_buffer(512)
    rc = lib.smg_hetmers_run(C.byref(tv), C.byref(opts), plot.ctypes.data, C.byref(st), buf, 512)
    _check(rc, buf)
    return plot.reshape(PLOT_ROWS, PLOT_COLS), st.asdict()


def _table_view(table):
    """`ktab.KTable` -> (TableView, objects that must stay alive while it is used)"""
    kb = (table.k + 3) >> 2
    pb = kb + 2 - table.ibyte
    n = table.nels
    rec = np.empty((n, pb), dtype=np.uint8)
    rec[:, : kb - table.ibyte] = table.packed[:, table.ibyte:]
    rec[:, kb - table.ibyte:] = np.ascontiguousarray(table.counts.astype("<u2")).view(np.uint8).reshape(n, 2)
    parts, offs = [], 0
    for pn in table.part_nels:
        parts.append(np.ascontiguousarray(rec[offs: offs + int(pn)]))
        offs += int(pn)
    ptrs = (C.c_void_p * max(1, len(parts)))(*[p.ctypes.data for p in parts])
    pn = np.ascontiguousarray(table.part_nels, dtype=np.int64)
    idx = np.ascontiguousarray(table.index, dtype=np.int64)
    tv = TableView(table.k, table.ibyte, len(parts), table.minval, n,
                   C.cast(ptrs, C.POINTER(C.c_void_p)),
                   pn.ctypes.data_as(C.POINTER(C.c_int64)),
                   idx.ctypes.data_as(C.POINTER(C.c_int64)))
    return tv, (parts, ptrs, pn, idx)


def hetmers_extract(table, labels: dict, device: int = 0, symcheck: str = "hash", condition: int = 0,
                    ethresh: int = 0):
    """In-process equivalent of `extract_kmer_pairs` (reference: src/lib/PloidyList.c).
    labels: {(covB, covA): "<a>A<b>B"} as read from a .sma file.
    -> (plot, {smudge name: list of text lines as the reference prints them (order unspecified)})"""
    lib = load_library()
    tv, keep = _table_view(table)
    names = sorted(set(labels.values()))
    lab = np.zeros(PLOT_CELLS, dtype=np.uint16)
    for (cb, ca), name in labels.items():
        lab[(ca + cb) * PLOT_COLS + cb] = names.index(name) + 1
    opts = Opts(device, _SYM[symcheck], 0, condition, ethresh, 0)
    plot = np.zeros(PLOT_CELLS, dtype=np.int64)
    st = Stats()
    buf = C.create_string_buffer(512)
    recs, nrec, rw = C.c_void_p(), C.c_int64(0), C.c_int(0)
    rc = lib.smg_hetmers_extract(C.byref(tv), C.byref(opts), lab.ctypes.data, plot.ctypes.data, C.byref(recs),
                                 C.byref(nrec), C.byref(rw), C.byref(st), buf, 512)
    _check(rc, buf)
    n, w = int(nrec.value), int(rw.value)
    arr = np.ctypeslib.as_array(C.cast(recs, C.POINTER(C.c_uint64)), shape=(max(n, 1) * w,))[: n * w].copy().reshape(n, w)
    lib.smg_free(recs)
    out = {name: [] for name in names}
    dna = "acgt"
    for row in arr:
        meta = int(row[w - 1])
        pos, alt, label = meta & 0xFF, (meta >> 8) & 3, meta >> 16
        bases = []
        for q in range(table.k):
            bases.append((int(row[q >> 5]) >> (62 - 2 * (q & 31))) & 3)
        out[names[label - 1]].append("".join(dna[b] for b in bases[:pos]) + f"({dna[bases[pos]]}/{dna[alt]})"
                                     + "".join(dna[b] for b in bases[pos + 1:]) + "\n")
    return plot.reshape(PLOT_ROWS, PLOT_COLS), out


def smu_text(plot: np.ndarray) -> str:
    """The `.smu` writer (reference: PloidyPlot.c:1603-1617): rows `covB\\tcovA\\tfreq`, sum
    ascending then covB ascending, zero cells skipped, covB == 500 never printed."""
    out = []
    s_idx, m_idx = np.nonzero(plot[:, :FMAX])
    for s, m in zip(s_idx.tolist(), m_idx.tolist()):
        out.append(f"{m}\t{s - m}\t{int(plot[s, m])}\n")
    return "".join(out)


@dataclass
class _Buf:
    ptr: int


class Engine:
    """Device-resident engine object (phase-level API).  Device pointers are plain ints, e.g.
    `tensor.data_ptr()` of torch tensors on the same device."""

    def __init__(self, device: int = 0, stream: int = 0):
        self.lib = load_library()
        self._buf = C.create_string_buffer(512)
        self.h = self.lib.smg_engine_create(device, stream or None, self._buf, 512)
        if not self.h:
            raise EngineError(-1, self._buf.value.decode(errors="replace"))
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.smg_engine_destroy(self.h)
            self.h = None

    __del__ = close

    def bind(self, k: int, nels: int, keys_ptr: int, counts_ptr: int):
        _check(self.lib.smg_engine_bind(self.h, k, nels, keys_ptr, counts_ptr, self._buf, 512), self._buf)

    def set_prefix_index(self, index_ptr: int, ibyte: int = 3, first_entry: int = 0):
        """the FastK prefix index of the bound table (int64[2^(8 ibyte)] on the device): the engine's look-up directory"""
        _check(self.lib.smg_engine_set_prefix_index(self.h, index_ptr, ibyte, first_entry, self._buf, 512), self._buf)

    def decode(self, k: int, ibyte: int, nels: int, records_ptr: int, index_ptr: int):
        _check(self.lib.smg_engine_decode(self.h, k, ibyte, nels, records_ptr, index_ptr, self._buf, 512),
               self._buf)

    def condition(self, ethresh: int, trim: bool, symm: bool) -> int:
        n = C.c_int64(0)
        _check(self.lib.smg_engine_condition(self.h, ethresh, int(trim), int(symm), C.byref(n), self._buf, 512),
               self._buf)
        return int(n.value)

    # ---- symmetrising across shards (smg_hetmers.h) ----------------------------------------
    def symm_hist(self, bits: int) -> np.ndarray:
        """int64[2 << bits]: this shard's entries, then their reverse complements, per leading `bits` k-mer bits"""
        h = np.zeros(2 << bits, dtype=np.int64)
        _check(self.lib.smg_engine_symm_hist(self.h, bits, h.ctypes.data, self._buf, 512), self._buf)
        return h

    def symm_route(self, splitters: np.ndarray, nranks: int, send_ptr: int, capacity: int):
        counts = (C.c_int64 * nranks)()
        sp = np.ascontiguousarray(splitters, dtype=np.uint64)
        _check(self.lib.smg_engine_symm_route(self.h, sp.ctypes.data if sp.size else None, nranks, send_ptr,
                                              capacity, counts, self._buf, 512), self._buf)
        return [int(c) for c in counts]

    def symm_finish(self, recv_ptr: int, nrecv: int) -> int:
        n = C.c_int64(0)
        _check(self.lib.smg_engine_symm_finish(self.h, recv_ptr, nrecv, C.byref(n), self._buf, 512), self._buf)
        return int(n.value)

    def table(self):
        """(entries, device pointer of the k-mers, device pointer of the counts) of the engine's current table"""
        n, pk, pc = C.c_int64(0), C.c_void_p(0), C.c_void_p(0)
        self.lib.smg_engine_table(self.h, C.byref(n), C.byref(pk), C.byref(pc))
        return int(n.value), pk.value or 0, pc.value or 0

    def run(self, plot_ptr: int, symcheck: str = "exact") -> dict:
        st = Stats()
        _check(self.lib.smg_engine_run(self.h, _SYM[symcheck], plot_ptr, C.byref(st), self._buf, 512),
               self._buf)
        return st.asdict()

    # ---- sharded phases -------------------------------------------------------------------
    def pass1(self, symcheck: str = "hash"):
        _check(self.lib.smg_engine_pass1(self.h, _SYM[symcheck], self._buf, 512), self._buf)

    def nreq(self) -> int:
        return int(self.lib.smg_engine_nreq(self.h))

    def record_words(self) -> int:
        return int(self.lib.smg_engine_record_words(self.h))

    def route(self, splitters: np.ndarray, nranks: int, send_ptr: int, capacity: int):
        counts = (C.c_int64 * nranks)()
        sp = np.ascontiguousarray(splitters, dtype=np.uint64)
        _check(self.lib.smg_engine_route(self.h, sp.ctypes.data if sp.size else None, nranks, send_ptr,
                                         capacity, counts, self._buf, 512), self._buf)
        return [int(c) for c in counts]

    def route_device(self, splitters: np.ndarray, nranks: int, send_ptr: int, capacity: int, counts_ptr: int):
        """route with the per-rank counts left on the device (int64[nranks] at counts_ptr): no host wait"""
        sp = np.ascontiguousarray(splitters, dtype=np.uint64)
        _check(self.lib.smg_engine_route_device(self.h, sp.ctypes.data if sp.size else None, nranks, send_ptr,
                                                capacity, counts_ptr, self._buf, 512), self._buf)

    def apply(self, recv_ptr: int, nrecv: int, wait: bool = True):
        """look-ups of received requests; wait=False queues them and returns None (the count of missing complements
        stays on the device: proof_into)"""
        if not wait:
            _check(self.lib.smg_engine_apply(self.h, recv_ptr, nrecv, None, self._buf, 512), self._buf)
            return None
        missing = C.c_int64(0)
        _check(self.lib.smg_engine_apply(self.h, recv_ptr, nrecv, C.byref(missing), self._buf, 512),
               self._buf)
        return int(missing.value)

    def proof_into(self, dst_ptr: int):
        """device uint64[4] at dst_ptr <- (missing complements, fingerprint residue word 0, word 1, 1 if a replayed step did not
        find the counts it was queued with), in stream order.  FOUR words since round 5: a caller's buffer must hold them."""
        _check(self.lib.smg_engine_proof(self.h, dst_ptr, self._buf, 512), self._buf)

    def proof_tail(self, tail_ptr: int, nslots: int, slot: int):
        """device uint64[3 + 2 nslots] at tail_ptr <- the proof words of this shard laid out for the final all_reduce (one launch)"""
        _check(self.lib.smg_engine_proof_tail(self.h, tail_ptr, nslots, slot, self._buf, 512), self._buf)

    def set_replay(self, on: bool):
        """queue the phase calls of a step from the counts of the step before (smg_hetmers.h: smg_engine_set_replay)"""
        self.lib.smg_engine_set_replay(self.h, int(bool(on)))

    def replay_state(self) -> int:
        return int(self.lib.smg_engine_replay_state(self.h))

    def replay_done(self, ok: bool):
        _check(self.lib.smg_engine_replay_done(self.h, int(bool(ok)), self._buf, 512), self._buf)

    def apply_own(self) -> int:
        missing = C.c_int64(0)
        _check(self.lib.smg_engine_apply_own(self.h, C.byref(missing), self._buf, 512), self._buf)
        return int(missing.value)

    def blockmap(self):
        """(id_bits, nwords) of the candidate block map of the last pass 1; id_bits 0 = none"""
        bits, nw = C.c_int32(0), C.c_int64(0)
        self.lib.smg_engine_blockmap(self.h, C.byref(bits), C.byref(nw))
        return int(bits.value), int(nw.value)

    def blockmap_copy(self, word_lo: int, nw: int, dst_ptr: int):
        _check(self.lib.smg_engine_blockmap_copy(self.h, word_lo, nw, dst_ptr, self._buf, 512), self._buf)

    def presort(self):
        _check(self.lib.smg_engine_presort(self.h, self._buf, 512), self._buf)

    def set_blockmap_bits(self, bits: int):
        self.lib.smg_engine_set_blockmap_bits(self.h, bits)

    def merge_maps(self, parts_ptr: int, width: int, word_lo, nwords_of, full_ptr: int):
        """gathered word ranges of len(word_lo) shards -> the whole map at full_ptr (one kernel launch)"""
        n = len(word_lo)
        lo = (C.c_int64 * n)(*[int(v) for v in word_lo])
        ln = (C.c_int64 * n)(*[int(v) for v in nwords_of])
        _check(self.lib.smg_engine_merge_maps(self.h, parts_ptr, width, n, lo, ln, full_ptr, self._buf, 512), self._buf)

    def filter(self, map_ptr=None) -> int:
        kept = C.c_int64(0)
        _check(self.lib.smg_engine_filter(self.h, map_ptr, C.byref(kept), self._buf, 512), self._buf)
        return int(kept.value)

    def symhash(self):
        out = (C.c_uint64 * 4)()
        _check(self.lib.smg_engine_symhash(self.h, out, self._buf, 512), self._buf)
        return [int(v) for v in out]

    def pass2(self, plot_ptr: int):
        _check(self.lib.smg_engine_pass2(self.h, plot_ptr, self._buf, 512), self._buf)

    def stats(self) -> dict:
        st = Stats()
        self.lib.smg_engine_stats(self.h, C.byref(st))
        return st.asdict()
