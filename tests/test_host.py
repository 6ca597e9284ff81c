"""CPU tests of the host side: format F round trips, the C table loader behind the drop-in
executable, argv / exit-code behaviour of `hetmers`, and that the C-ABI library loads and
exports every symbol include/smg_hetmers.h declares (no compute calls: there is no GPU here)."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import HETMERS_BIN, LIB, ROOT
from smudgeplot_amd import engine, ktab, synth


@pytest.mark.parametrize("k,ibyte,nparts", [(31, 1, 1), (31, 2, 3), (21, 1, 2), (51, 2, 4), (65, 1, 1)])
def test_ktab_roundtrip(k, ibyte, nparts, tmp_path):
    packed, cnt = synth.adversarial_table(k, 500, 4, seed=k + ibyte)
    ktab.write_ktab(str(tmp_path / "x.ktab"), k, packed, cnt, ibyte=ibyte, nparts=nparts)
    t = ktab.read_ktab(str(tmp_path / "x"))
    assert t.k == k and t.ibyte == ibyte and t.nparts == nparts
    assert np.array_equal(t.packed, packed) and np.array_equal(t.counts, cnt)
    assert t.index[-1] == len(cnt) and int(t.part_nels.sum()) == len(cnt)


def test_symmetric_generator_is_closed_under_revcomp():
    packed, cnt = synth.adversarial_table(32, 800, 4, seed=9, low_complexity=50, dense=1)
    rc = ktab.revcomp_packed(packed, 32)
    v, r = ktab._as_void(packed), ktab._as_void(rc)
    j = np.searchsorted(v, r)
    assert (v[j] == r).all() and (cnt[j] == cnt).all()


def test_u64_helpers_agree_with_packed():
    packed, cnt = synth.adversarial_table(31, 300, 4, seed=2)
    keys = ktab.packed_to_u64(packed)
    assert (np.diff(keys.astype(np.float64)) >= 0).all()
    assert np.array_equal(ktab.u64_to_packed(keys, 31), packed)
    assert np.array_equal(ktab.u64_to_packed(ktab.revcomp_u64(keys, 31), 31), ktab.revcomp_packed(packed, 31))
    k2, c2 = synth.diploid_table_u64(5000, k=31, seed=3)
    assert (k2[1:] > k2[:-1]).all()
    j = np.searchsorted(k2, ktab.revcomp_u64(k2, 31))
    assert (k2[j] == ktab.revcomp_u64(k2, 31)).all() and (c2[j] == c2).all()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "smg_hetmers.h")).read()
    declared = set(re.findall(r"\b(smg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = engine.load_library()
    for name in declared:
        assert getattr(lib, name) is not None, name
    assert declared == set(engine.EXPORTS)
    assert b"gfx950" in lib.smg_version()
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True).stdout
    for name in declared:
        assert f" T {name}" in out


def run(args, cwd, stdin=None):
    return subprocess.run([HETMERS_BIN, *args], cwd=cwd, capture_output=True, text=True, input=stdin)


def test_hetmers_usage_and_argument_errors(tmp_path):
    r = run([], tmp_path)
    assert r.returncode == 1 and r.stderr.startswith("\nUsage: hetmers  [-v] [-T<int(4)>] [-P<dir(/tmp)>]")
    assert "      -P: Place all temporary files in directory -P.\n" in r.stderr
    r = run(["a", "b"], tmp_path)
    assert r.returncode == 1 and "Usage: hetmers" in r.stderr
    r = run(["-x", "t"], tmp_path)
    assert r.returncode == 1 and r.stderr == "hetmers: -x is an illegal option\n"
    r = run(["-eabc", "t"], tmp_path)
    assert r.returncode == 1 and r.stderr == "hetmers: -e 'abc' argument is not an integer\n"
    r = run(["-e0", "t"], tmp_path)
    assert r.returncode == 1 and r.stderr == "hetmers: Error-mer threshold must be positive (0)\n"
    r = run(["-T-3", "t"], tmp_path)
    assert r.returncode == 1 and r.stderr == "hetmers: Number of threads must be positive (-3)\n"
    r = run(["-T100", "-klfs", "missing"], tmp_path)
    assert r.returncode == 1
    assert r.stderr == ("hetmers: Warning, only 64 threads will be used\n"
                        "hetmers: Cannot open k-mer table missing\n")


def test_hetmers_table_errors_and_prompt(tmp_path):
    packed, cnt = synth.adversarial_table(31, 200, 4, seed=4)
    ktab.write_ktab(str(tmp_path / "t"), 31, packed, cnt, ibyte=1, nparts=2)
    # existing .smu + "y" => reuse, exit 0, nothing recomputed
    (tmp_path / "out.smu").write_text("1\t2\t3\n")
    r = run(["-e4", "-oout", "t.ktab"], tmp_path, stdin="yes\n")
    assert r.returncode == 0
    assert r.stdout == "\n  Found het-table out.smu, use it? "
    assert r.stderr == "\n  Using the found het-table, done\n"
    assert (tmp_path / "out.smu").read_text() == "1\t2\t3\n"
    # default output name = source minus .ktab
    (tmp_path / "t.smu").write_text("x")
    r = run(["-e4", "t.ktab"], tmp_path, stdin="Y\n")
    assert r.returncode == 0 and "Found het-table t.smu" in r.stdout
    os.remove(tmp_path / "t.smu"); os.remove(tmp_path / "out.smu")
    # missing part
    os.rename(tmp_path / ".t.ktab.2", tmp_path / "hidden")
    r = run(["-e4", "-oout", "t"], tmp_path)
    assert r.returncode == 1 and r.stderr == "hetmers: Table part ./.t.ktab.2 is missing ?\n"
    os.rename(tmp_path / "hidden", tmp_path / ".t.ktab.2")
    # k mismatch between stub and part
    raw = bytearray((tmp_path / ".t.ktab.2").read_bytes()); raw[0] = 30
    (tmp_path / ".t.ktab.2").write_bytes(bytes(raw))
    r = run(["-e4", "-oout", "t"], tmp_path)
    assert r.returncode == 1
    assert r.stderr == "hetmers: Table part ./.t.ktab.2 does not have k-mer length matching stub ?\n"


def test_hetmers_conditioning_decision_and_no_cpu_fallback(tmp_path):
    """Without a GPU the executable must fail loudly AFTER the host-side table probe."""
    if engine.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu tests")
    packed, cnt = synth.adversarial_table(31, 200, 6, seed=5)
    ktab.write_ktab(str(tmp_path / "t"), 31, packed, cnt, ibyte=1)
    r = run(["-e6", "-v", "-oout", "t"], tmp_path)
    assert r.returncode == 1
    assert r.stderr.startswith("\n  The input table is trimmed and symmetric\n"
                               "\n  Starting to count covariant pairs\n")
    assert r.stderr.endswith("hetmers: no HIP device available (this engine has no CPU fallback)\n")
    assert not (tmp_path / "out.smu").exists()
    # untrimmed table: same decision as the reference; conditioning now happens on the device, so
    # without a GPU the run fails there (loudly), not in a Logex shell-out
    r = run(["-e9", "-v", "-T3", "-oout", "t"], tmp_path)
    assert r.returncode == 1
    assert "  The input table is untrimmed yet symmetric\n" in r.stderr
    assert "  Trimming k-mers in table with count < 9\n" in r.stderr
    assert r.stderr.endswith("hetmers: no HIP device available (this engine has no CPU fallback)\n")
    # SMUDGEPLOT_USE_FASTK_TOOLS=1: the reference's shell-outs, same command strings (Logex is not installed)
    r = subprocess.run([HETMERS_BIN, "-e9", "-v", "-T3", "-oout", "t"], cwd=tmp_path, capture_output=True, text=True,
                       env=dict(os.environ, SMUDGEPLOT_USE_FASTK_TOOLS="1"))
    assert r.returncode == 1
    assert "  Trimming k-mers in table with count < 9\n" in r.stderr
    assert r.stderr.endswith("hetmers: Command 'Logex -T3 '.trim=A[9-]' t' failed\n")
    # canonical-only table: not symmetric
    rc = ktab.revcomp_packed(packed, 31)
    canon = np.array([bytes(a) <= bytes(b) for a, b in zip(packed, rc)])
    ktab.write_ktab(str(tmp_path / "c"), 31, packed[canon], cnt[canon], ibyte=1)
    r = run(["-e6", "-v", "-oout", "c"], tmp_path)
    assert r.returncode == 1
    assert "  The input table is trimmed but not symmetric\n" in r.stderr
    assert "  Making table symmetric\n" in r.stderr
    with pytest.raises(engine.EngineError):
        engine.Engine(0)


def test_trim_probe_over_several_threads_sees_a_single_low_count(tmp_path):
    """the count scan of the probe runs on several threads above 1e6 entries: one count below the threshold, in the range
    of the last thread (or the first, or nowhere), must give the reference's decision (PloidyPlot.c:1169-1197)"""
    if engine.device_count() > 0:
        pytest.skip("a GPU is visible: the run would go on")
    k = 31
    keys, cnt = synth.diploid_table_u64(600000, k=k, seed=21, het_frac=0.01, cov=40, L=8)
    assert len(cnt) > 1_000_000
    for where, want in ((None, "trimmed"), (len(cnt) - 7, "untrimmed"), (3, "untrimmed")):
        c = cnt.copy()
        if where is not None:
            c[where] = 5
        synth.write_u64_table(str(tmp_path / "t"), keys, c, k, ibyte=2, nparts=3)
        r = run(["-e8", "-v", "-T8", "-oout", "t"], tmp_path)
        assert r.returncode == 1
        assert f"  The input table is {want}" in r.stderr, (where, r.stderr)


def test_blockmap_ranges_cover_the_map_and_share_only_boundary_words():
    """sharded.blockmap_ranges: the word ranges of the ranks' k-mer ranges tile the candidate block map"""
    from smudgeplot_amd import sharded
    rng = np.random.default_rng(5)
    for bits in (4, 12, 30):
        nwords = ((1 << bits) + 31) >> 5
        for world in (1, 2, 3, 8):
            firsts = np.sort(rng.integers(0, 2 ** 63, size=world - 1, dtype=np.uint64) << np.uint64(1))
            if world > 2:
                firsts[1] = firsts[0]                       # an empty shard inherits its successor's first k-mer
            wlo, wlen = sharded.blockmap_ranges(firsts, 1, world, bits)
            assert wlo[0] == 0 and wlo[-1] + wlen[-1] == nwords
            for r in range(world):
                assert wlen[r] >= 1 and 0 <= wlo[r] and wlo[r] + wlen[r] <= nwords
                if r:
                    prev_end = wlo[r - 1] + wlen[r - 1]     # neighbours overlap in exactly their boundary word
                    assert prev_end - 1 == wlo[r]
                    # the id of rank r's first k-mer falls into that shared word
                    assert (int(firsts[r - 1]) >> (64 - bits)) >> 5 == wlo[r]
            # the engine's two-bit map: 64 bits per 32 block ids, the same ranges in units of two words
            wlo2, wlen2 = sharded.blockmap_ranges(firsts, 1, world, bits, 2)
            assert wlo2 == [2 * v for v in wlo] and wlen2 == [2 * v for v in wlen]


def test_hostile_stub_is_refused_before_any_record_is_touched(tmp_path):
    """A stub that claims k = 300 (records wider than the loader's one-record scratch buffers), a part header whose
    entry count overflows the size check, and a part that shrinks after the probe's open: all are errors with the
    loader's message and exit code 1 -- never a crash, never a table that is silently called "trimmed"."""
    import struct
    packed, cnt = synth.adversarial_table(31, 200, 4, seed=8)
    ktab.write_ktab(str(tmp_path / "t"), 31, packed, cnt, ibyte=1, nparts=1)
    stub = bytearray((tmp_path / "t.ktab").read_bytes())
    part = bytearray((tmp_path / ".t.ktab.1").read_bytes())
    # k = 300 in stub and part (consistent, so only the bound can stop it)
    (tmp_path / "big.ktab").write_bytes(struct.pack("<i", 300) + bytes(stub[4:]))
    (tmp_path / ".big.ktab.1").write_bytes(struct.pack("<i", 300) + bytes(part[4:]))
    r = run(["-e4", "-obig", "big"], tmp_path)
    assert r.returncode == 1 and r.stderr == "hetmers: Table file ./big.ktab is truncated or not a FastK table\n"
    # a part that claims 2^62 entries: 12 + n * pbyte overflows int64
    (tmp_path / "ovf.ktab").write_bytes(bytes(stub))
    (tmp_path / ".ovf.ktab.1").write_bytes(bytes(part[:4]) + struct.pack("<q", 1 << 62) + bytes(part[12:]))
    r = run(["-e4", "-oovf", "ovf"], tmp_path)
    assert r.returncode == 1 and r.stderr == "hetmers: Table file ./.ovf.ktab.1 is truncated or not a FastK table\n"
    r = run(["-e4", "-oneg", "ovf"], tmp_path)
    assert r.returncode == 1
    (tmp_path / ".ovf.ktab.1").write_bytes(bytes(part[:4]) + struct.pack("<q", -5) + bytes(part[12:]))
    r = run(["-e4", "-oneg", "ovf"], tmp_path)
    assert r.returncode == 1 and "truncated or not a FastK table" in r.stderr


def test_code_object_hash_and_traffic_file():
    """bench.py quotes `roofline.traffic` only for the gfx950 code object the PMC passes were read on
    (smudgeplot_amd/codeobj.py, profiles/hbm_traffic.json): the hash must be computable without a GPU, and every entry of
    the traffic file must say which code object it belongs to"""
    import json
    import warnings
    from smudgeplot_amd import codeobj
    h = codeobj.code_object_hash()
    assert len(h) == 16 and int(h, 16) >= 0
    assert h == codeobj.code_object_hash()                       # deterministic
    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
        doc = json.load(f)
    assert {"k31", "k51", "k31_repeats", "k31_octoploid", "k51_hexaploid"} <= set(doc)
    for key, ent in doc.items():
        assert len(ent["code_object_sha256_16"]) == 16 and ent["bytes_per_entry"]["ms_pass1"] > 0, key
        if ent["code_object_sha256_16"] != h:
            warnings.warn(f"profiles/hbm_traffic.json[{key}] was measured on another code object: bench.py will report traffic = null")


@pytest.mark.parametrize("kind", ["octoploid", "hexaploid_k51", "diploid_k51_chunked"])
def test_polyploid_generators_make_conditioned_tables_the_reference_accepts(kind, tmp_path):
    """the bench's polyploid stand-ins (synth_device.polyploid_table_graded / polyploid_table_wide), run on the CPU device at
    toy size: sorted, duplicate free, closed under reverse complement with equal counts -- the REFERENCE binary takes the
    table as trimmed and symmetric (it would shell out to Logex / Symmex and die otherwise) and its .smu is the numpy oracle's;
    the device-side FastK writer round-trips through the numpy reader"""
    import torch
    import brute
    from conftest import REF_BIN
    from smudgeplot_amd import engine, ktab, synth_device
    if kind == "octoploid":
        k, L = 31, 8
        tk, tc = synth_device.polyploid_table_graded(25000, ploidy=8, cov_hap=14.0, k=k, L=L, seed=4, device="cpu")
    elif kind == "hexaploid_k51":
        k, L = 51, 5
        tk, tc = synth_device.polyploid_table_wide(15000, ploidy=6, cov_hap=10.0, k=k, L=L, seed=5, device="cpu", max_chunk=4e4)
    else:
        k, L = 51, 10
        tk, tc = synth_device.polyploid_table_wide(20000, ploidy=2, rates=(0.01,), cov_hap=25.0, k=k, L=L, seed=1, device="cpu", max_chunk=2e4)
    n = tc.numel()
    kw = tk.numpy().view(np.uint64).reshape(n, -1)
    cnt = tc.numpy().view(np.uint16)
    assert cnt.min() >= L
    lt = (kw[1:, 0] > kw[:-1, 0])
    for w in range(1, kw.shape[1]):
        lt |= (kw[1:, 0] == kw[:-1, 0]) & (kw[1:, w] > kw[:-1, w])
    assert lt.all()                                               # strictly increasing
    packed = np.ascontiguousarray(np.ascontiguousarray(kw.astype(">u8")).view(np.uint8).reshape(n, -1)[:, : (k + 3) // 4])
    synth_device.write_table_from_device(str(tmp_path / "t"), tk, tc, k, nparts=3)
    T = ktab.read_ktab(str(tmp_path / "t"))
    assert np.array_equal(T.packed, packed) and np.array_equal(T.counts, cnt)
    want = engine.smu_text(brute.hetmers_plot(packed, cnt, k))
    assert want.count("\n") > 20
    if os.path.exists(REF_BIN):
        r = subprocess.run([REF_BIN, f"-e{L}", "-T2", "-oref", "t.ktab"], cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert (tmp_path / "ref.smu").read_text() == want
