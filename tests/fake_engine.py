"""Numpy stand-in for `smudgeplot_amd.sharded.TorchEngine` -- TEST INFRASTRUCTURE ONLY.

It implements the same phase protocol (bind / pass1 / blockmap / filter / route / apply / symhash / pass2, and the
general-path fallback) on CPU tensors with brute-force numpy, so that the world_size>1 orchestration of
`sharded.hetmers_sharded` (splitter exchange, block-map all_gather, all_to_all of the complement requests,
symmetry proof + histogram all_reduce, gather for the fallback) can run under the `gloo` backend without a GPU.
Any k <= 128: a k-mer is handled as a Python integer of 64 * W bits (left aligned, like the engine's W words).
The product path never imports this module.
"""
import numpy as np
import torch

from smudgeplot_amd import ktab

SMAX, FMAX = 1000, 500
PLOT_COLS = FMAX + 1
PLOT_CELLS = (SMAX + 1) * PLOT_COLS
M64 = (1 << 64) - 1


def _mix(z):
    z &= M64
    z ^= z >> 30; z = (z * 0xbf58476d1ce4e5b9) & M64
    z ^= z >> 27; z = (z * 0x94d049bb133111eb) & M64
    return z ^ (z >> 31)


class NumpyEngine:
    def __init__(self, device):
        self.device = device
        self._rp_want, self._rp_rec, self._rp_active, self._rp_bad = False, None, False, 0

    # ---- replay of the phase calls (smg_engine_set_replay): the stand-in does all the work every time, but hands out the
    #      RECORDED counts and reports a difference through the fourth proof word, like the engine -------------------------
    def set_replay(self, on):
        self._rp_want = bool(on)
        if not on:
            self._rp_rec = None

    def replay_state(self):
        return (1 if self._rp_active else 0) | (2 if self._rp_rec is not None else 0)

    def replay_done(self, ok):
        if self._rp_active and not ok:
            self._rp_rec = None
        self._rp_active = False

    # ---- helpers on integer k-mers ------------------------------------------------------------------
    def _ints(self, words_u64):
        w = np.asarray(words_u64, dtype=np.uint64).reshape(-1, self.W)
        out = np.empty(len(w), dtype=object)
        for i in range(len(w)):
            v = 0
            for j in range(self.W):
                v = (v << 64) | int(w[i, j])
            out[i] = v
        return out

    def _words(self, x):
        return [(x >> (64 * (self.W - 1 - j))) & M64 for j in range(self.W)]

    def _rc(self, x):
        k, bits = self.k, 64 * self.W
        v = x >> (bits - 2 * k)                       # right aligned
        r = 0
        for _ in range(k):
            r = (r << 2) | (3 - (v & 3))
            v >>= 2
        return r << (bits - 2 * k)

    def bind(self, k, keys, counts):
        self.k = k
        self.W = (k + 31) // 32
        self.keys = self._ints(keys.cpu().numpy().view(np.uint64))
        self.cnt = counts.cpu().numpy().view(np.uint16).astype(np.int64)
        self.n = len(self.cnt)

    def record_words(self):
        return self.W + 1

    # ---- conditioning across shards (sharded.condition_sharded) --------------------------------------------
    def nels(self):
        return self.n

    def trim(self, ethresh):
        keep = self.cnt >= ethresh
        self.keys, self.cnt = self.keys[keep], self.cnt[keep]
        self.n = len(self.cnt)
        return self.n

    def symm_hist(self, bits):
        h = np.zeros(2 << bits, dtype=np.int64)
        sh = 64 * self.W - bits
        for x in self.keys:
            h[x >> sh] += 1
            h[(1 << bits) + (self._rc(x) >> sh)] += 1
        return h

    def symm_route(self, splitters, nranks, send):
        """two records per entry (itself; its complement with bit 16 of the count word), grouped by destination"""
        sp = list(self._ints(np.asarray(splitters, dtype=np.uint64))) if len(splitters) else []
        rec = []
        for x, c in zip(self.keys, self.cnt):
            rec.append((x, int(c)))
            rec.append((self._rc(x), int(c) | (1 << 16)))
        dest = np.array([sum(1 for s_ in sp if t >= s_) for t, _ in rec], dtype=np.int64)
        out = []
        for r in range(nranks):
            for (t, m), d in zip(rec, dest):
                if d == r:
                    out.extend(self._words(t)); out.append(m)
        if out:
            send[: len(out)] = torch.from_numpy(np.array(out, dtype=np.uint64).view(np.int64))
        return [int((dest == r).sum()) for r in range(nranks)]

    def symm_finish(self, recv, nrecv):
        rw = self.W + 1
        rec = recv[: nrecv * rw].cpu().numpy().view(np.uint64).reshape(-1, rw)
        best = {}
        ks = self._ints(rec[:, : self.W].reshape(-1)) if len(rec) else []
        for x, m in zip(ks, rec[:, self.W]):
            m = int(m)
            cur = best.get(x)
            if cur is None or ((cur >> 16) & 1 and not (m >> 16) & 1):      # an entry beats a complement
                best[x] = m
        xs = sorted(best)
        self.keys = np.array(xs, dtype=object)
        self.cnt = np.array([best[x] & 0xFFFF for x in xs], dtype=np.int64)
        self.n = len(xs)
        return self.n

    def pass1(self, symcheck, exchange=True, world=1):
        k, keys, cnt, n = self.k, self.keys, self.cnt, self.n
        bits = 64 * self.W
        p0 = k // 2
        s_all = np.zeros(n, np.int64)
        s_hi = np.zeros(n, np.int64)
        pa, pb, ph = [], [], []
        for p in range(p0, k):
            mask = ~(3 << (bits - 2 - 2 * p))
            groups = {}
            for i in range(n):
                groups.setdefault(keys[i] & mask, []).append(i)
            hi = int(p != k - 1 - p)
            for g in groups.values():
                for x in range(len(g)):
                    for y in range(x + 1, len(g)):
                        a, b = g[x], g[y]
                        if cnt[a] + cnt[b] <= SMAX:
                            s_all[a] += 1; s_all[b] += 1
                            s_hi[a] += hi; s_hi[b] += hi
                            pa.append(a); pb.append(b); ph.append(hi)
        self.pa = np.array(pa, np.int64); self.pb = np.array(pb, np.int64); self.ph = np.array(ph, np.int64)
        self.s_all = s_all
        self.P = np.zeros(n, np.int64)
        self.rc = [self._rc(x) for x in keys]
        emit = np.ones(n, bool) if symcheck == "exact" else s_hi > 0
        req = []
        for i in np.flatnonzero(emit):
            req.extend(self._words(self.rc[i]))
            req.append(int(cnt[i]) | (int(s_hi[i] > 0) << 16))
        self.req = np.array(req, dtype=np.uint64)
        # canonical XOR fingerprint (any function of (min(x, rc x), count) will do: the two members of a class cancel)
        f = 0
        for i in range(n):
            x, r = keys[i], self.rc[i]
            if x == r:
                continue
            c = min(x, r)
            f ^= _mix(_mix(c & M64) ^ _mix(c >> 64) ^ _mix(int(cnt[i])))
        self.fp = [f, 0, 0, 0]
        self.symcheck = symcheck
        self.missing = 0
        self._rp_active = bool(self._rp_want and self._rp_rec is not None and symcheck == "hash" and exchange)
        self._rp_bad = int(self._rp_active and self._rp_rec["emitted"] != self.nreq())
        self._emitted = self.nreq()

    def nreq(self):
        return len(self.req) // (self.W + 1)

    # request filter: same protocol as the engine (block id = leading id_bits bits of the k-mer)
    def blockmap(self):
        if self.symcheck != "hash" or self.k > 85:
            return 0, 0
        bits = min(14, 2 * (self.k // 2))      # (the engine uses 30 bits; any width the ranks agree on works)
        return bits, ((1 << bits) + 31) >> 5

    def _own_map(self):
        bits, nwords = self.blockmap()
        m = np.zeros(nwords, np.uint32)
        for i in np.flatnonzero(self.s_all == 1):
            b = self.keys[i] >> (64 * self.W - bits)
            m[b >> 5] |= np.uint32(1 << (b & 31))
        return m

    def blockmap_copy(self, word_lo, nw, dst):
        dst[:nw] = torch.from_numpy(self._own_map()[word_lo: word_lo + nw].view(np.int32).copy())

    def presort(self):
        pass

    def merge_maps(self, parts, width, wlo, wlen, full):
        p = parts.cpu().numpy().view(np.uint32)
        m = np.zeros(full.numel(), np.uint32)
        for r in range(len(wlo)):
            m[wlo[r]: wlo[r] + wlen[r]] |= p[r * width: r * width + wlen[r]]
        full.copy_(torch.from_numpy(m.view(np.int32)))

    def filter(self, full_map=None):
        bits, _ = self.blockmap()
        m = self._own_map() if full_map is None else full_map.cpu().numpy().view(np.uint32)
        rw = self.W + 1
        rec = self.req.reshape(-1, rw)
        keep = np.zeros(len(rec), bool)
        for i in range(len(rec)):
            b = int(rec[i, 0]) >> (64 - bits)
            keep[i] = (int(m[b >> 5]) >> (b & 31)) & 1
        self.dropped = int((~keep).sum())
        self.req = rec[keep].reshape(-1)
        kept = len(self.req) // rw
        if self._rp_active:
            if kept != self._rp_rec["kept"]:       # (the engine would have queued everything with the recorded count)
                self._rp_bad = 1
                want = self._rp_rec["kept"]
                pad = np.zeros(max(want - kept, 0) * rw, dtype=np.uint64)
                self.req = np.concatenate([self.req, pad])[: want * rw]
            return self._rp_rec["kept"]
        if self._rp_want and self.symcheck == "hash":
            self._rp_rec = {"emitted": self._emitted, "kept": kept}
        return kept

    def route(self, splitters, nranks, send, counts_out=None):
        rw = self.W + 1
        rec = self.req.reshape(-1, rw)
        sp = self._ints(np.asarray(splitters, dtype=np.uint64)) if len(splitters) else []
        tgt = self._ints(rec[:, : self.W].reshape(-1)) if len(rec) else []
        dest = np.array([sum(1 for s in sp if t >= s) for t in tgt], dtype=np.int64)
        order = np.argsort(dest, kind="stable")
        out = rec[order].reshape(-1)
        send[: len(out)] = torch.from_numpy(out.view(np.int64).copy())
        counts = [int((dest == r).sum()) for r in range(nranks)]
        if counts_out is not None:
            counts_out.copy_(torch.tensor(counts, dtype=torch.int64))
            return None
        return counts

    def _apply(self, rec):
        rw = self.W + 1
        rec = np.asarray(rec, dtype=np.uint64).reshape(-1, rw)
        index = {x: i for i, x in enumerate(self.keys)}
        bad = 0
        tgt = self._ints(rec[:, : self.W].reshape(-1)) if len(rec) else []
        for t, meta in zip(tgt, rec[:, self.W]):
            j = index.get(t)
            meta = int(meta)
            if j is None or self.cnt[j] != (meta & 0xFFFF):
                bad += 1
            elif (meta >> 16) & 1:
                self.P[j] = 1
        return bad

    def apply(self, recv, nrecv, wait=True):
        self.missing = self._apply(recv[: nrecv * (self.W + 1)].cpu().numpy().view(np.uint64))
        return self.missing if wait else None

    def proof_into(self, dst):
        w = np.array([self.missing, self.fp[0] ^ self.fp[2], self.fp[1] ^ self.fp[3], self._rp_bad], dtype=np.uint64)
        dst[:4] = torch.from_numpy(w.view(np.int64).copy())

    def apply_own(self):
        return self._apply(self.req)

    def symhash(self):
        return list(self.fp)

    def pass2(self, plot):
        a, b = self.pa, self.pb
        out = np.zeros(PLOT_CELLS, np.int64)
        if len(a):
            keep = (self.s_all[a] == 1) & (self.s_all[b] == 1) & (self.P[a] == 0) & (self.P[b] == 0)
            a, b, w = a[keep], b[keep], 1 + self.ph[keep]
            s = self.cnt[a] + self.cnt[b]
            m = np.minimum(self.cnt[a], self.cnt[b])
            np.add.at(out, s * PLOT_COLS + m, w)
        plot.copy_(torch.from_numpy(out))

    def run_general(self, k, keys, counts, plot):
        """what one GPU does with a table that fails the proof: the oracle's all-positions count"""
        import brute
        W = (k + 31) // 32
        kw = keys.cpu().numpy().view(np.uint64).reshape(-1, W)
        packed = np.ascontiguousarray(kw.astype(">u8")).view(np.uint8).reshape(len(kw), 8 * W)[:, : (k + 3) // 4]
        cnt = counts.cpu().numpy().view(np.uint16)
        plot.copy_(torch.from_numpy(brute.hetmers_plot(np.ascontiguousarray(packed), cnt, k).reshape(-1)))

    def stats(self):
        return {"nels": self.n, "path": 1}
