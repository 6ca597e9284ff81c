"""Numpy stand-in for `smudgeplot_amd.sharded.TorchEngine` -- TEST INFRASTRUCTURE ONLY.

It implements the same phase protocol (bind / pass1 / route / apply / symhash / pass2) on CPU
tensors with brute-force numpy, so that the world_size>1 orchestration of
`sharded.hetmers_sharded` (splitter exchange, all_to_all of the complement requests, symmetry
proof all_reduce, histogram all_reduce) can run under the `gloo` backend without a GPU.
k <= 32 only (one 64-bit word per k-mer).  The product path never imports this module.
"""
import numpy as np
import torch

from smudgeplot_amd import ktab

SMAX, FMAX = 1000, 500
PLOT_COLS = FMAX + 1
PLOT_CELLS = (SMAX + 1) * PLOT_COLS


def _mix(z):
    z = z.astype(np.uint64)
    z ^= z >> np.uint64(30); z *= np.uint64(0xbf58476d1ce4e5b9)
    z ^= z >> np.uint64(27); z *= np.uint64(0x94d049bb133111eb)
    return z ^ (z >> np.uint64(31))


class NumpyEngine:
    def __init__(self, device):
        self.device = device

    def bind(self, k, keys, counts):
        assert k <= 32
        self.k = k
        self.keys = keys.cpu().numpy().view(np.uint64).copy()
        self.cnt = counts.cpu().numpy().view(np.uint16).astype(np.int64)
        self.n = len(self.cnt)

    def record_words(self):
        return 2

    def pass1(self, symcheck):
        k, keys, cnt, n = self.k, self.keys, self.cnt, self.n
        p0 = k // 2
        s_all = np.zeros(n, np.int64)
        s_hi = np.zeros(n, np.int64)
        pa, pb, ph = [], [], []
        for p in range(p0, k):
            m = keys & ~(np.uint64(3) << np.uint64(62 - 2 * p))
            order = np.argsort(m, kind="stable")
            ms = m[order]
            for d in (1, 2, 3):
                if n <= d:
                    continue
                same = ms[d:] == ms[:-d]
                a, b = order[:-d][same], order[d:][same]
                ok = cnt[a] + cnt[b] <= SMAX
                a, b = a[ok], b[ok]
                hi = int(p != k - 1 - p)
                np.add.at(s_all, a, 1); np.add.at(s_all, b, 1)
                np.add.at(s_hi, a, hi); np.add.at(s_hi, b, hi)
                pa.append(a); pb.append(b); ph.append(np.full(len(a), hi, np.int64))
        self.pa = np.concatenate(pa) if pa else np.zeros(0, np.int64)
        self.pb = np.concatenate(pb) if pb else np.zeros(0, np.int64)
        self.ph = np.concatenate(ph) if ph else np.zeros(0, np.int64)
        self.s_all = s_all
        self.P = np.zeros(n, np.int64)
        emit = np.ones(n, bool) if symcheck == "exact" else s_hi > 0
        rc = ktab.revcomp_u64(keys, k)
        self.req = np.stack([rc[emit], (cnt[emit] | ((s_hi[emit] > 0).astype(np.int64) << 16)).astype(np.uint64)],
                            axis=1).reshape(-1)
        # signed canonical fingerprint (any function that cancels over {x, rc(x)} pairs will do)
        h = _mix(np.minimum(keys, rc) ^ _mix(cnt.astype(np.uint64)))
        sign = np.where(keys < rc, 1, np.where(keys > rc, -1, 0)).astype(np.int64)
        with np.errstate(over="ignore"):
            f = np.sum(h.view(np.int64) * sign, dtype=np.int64) if n else np.int64(0)
        self.fp = [int(np.uint64(np.int64(f))), 0, 0, 0]

        self.symcheck = symcheck

    def nreq(self):
        return len(self.req) // 2

    # request filter: same protocol as the engine (block id = leading id_bits bits of the k-mer)
    def blockmap(self):
        if self.symcheck != "hash":
            return 0, 0
        bits = min(14, 2 * (self.k // 2))      # (the engine uses up to 30 bits = 128 MB; any width the ranks agree on works)
        return bits, ((1 << bits) + 31) >> 5

    def _own_map(self):
        bits, nwords = self.blockmap()
        m = np.zeros(nwords, np.uint32)
        ids = (self.keys[self.s_all == 1] >> np.uint64(64 - bits)).astype(np.int64)
        np.bitwise_or.at(m, ids >> 5, (np.uint32(1) << (ids & 31).astype(np.uint32)))
        return m

    def blockmap_copy(self, word_lo, nw, dst):
        dst[:nw] = torch.from_numpy(self._own_map()[word_lo: word_lo + nw].view(np.int32).copy())

    def presort(self):
        pass

    def filter(self, full_map=None):
        bits, _ = self.blockmap()
        m = self._own_map() if full_map is None else full_map.cpu().numpy().view(np.uint32)
        rec = self.req.reshape(-1, 2)
        ids = (rec[:, 0] >> np.uint64(64 - bits)).astype(np.int64)
        keep = ((m[ids >> 5] >> (ids & 31).astype(np.uint32)) & np.uint32(1)).astype(bool)
        self.dropped = int((~keep).sum())
        self.req = rec[keep].reshape(-1)
        return len(self.req) // 2

    def route(self, splitters, nranks, send):
        rec = self.req.reshape(-1, 2)
        dest = np.searchsorted(np.asarray(splitters, dtype=np.uint64), rec[:, 0], side="right")
        order = np.argsort(dest, kind="stable")
        out = rec[order].reshape(-1)
        send[: len(out)] = torch.from_numpy(out.view(np.int64).copy())
        return [int((dest == r).sum()) for r in range(nranks)]

    def _apply(self, rec):
        rec = rec.reshape(-1, 2)
        j = np.searchsorted(self.keys, rec[:, 0])
        jj = np.minimum(j, max(self.n - 1, 0))
        found = (j < self.n) & (self.keys[jj] == rec[:, 0]) if self.n else np.zeros(len(rec), bool)
        c = (rec[:, 1] & np.uint64(0xFFFF)).astype(np.int64)
        good = found & (self.cnt[jj] == c) if self.n else found
        flag = ((rec[:, 1] >> np.uint64(16)) & np.uint64(1)).astype(bool)
        self.P[jj[good & flag]] = 1
        return int((~good).sum())

    def apply(self, recv, nrecv):
        return self._apply(recv[: nrecv * 2].cpu().numpy().view(np.uint64))

    def apply_own(self):
        return self._apply(self.req)

    def symhash(self):
        return list(self.fp)

    def pass2(self, plot):
        a, b = self.pa, self.pb
        keep = (self.s_all[a] == 1) & (self.s_all[b] == 1) & (self.P[a] == 0) & (self.P[b] == 0)
        a, b, w = a[keep], b[keep], 1 + self.ph[keep]
        s = self.cnt[a] + self.cnt[b]
        m = np.minimum(self.cnt[a], self.cnt[b])
        out = np.zeros(PLOT_CELLS, np.int64)
        np.add.at(out, s * PLOT_COLS + m, w)
        plot.copy_(torch.from_numpy(out))

    def stats(self):
        return {"nels": self.n, "path": 1}
