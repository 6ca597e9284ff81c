"""Identity of the engine's DEVICE code: sha256 of the gfx950 code object inside libsmg_hetmers.so (the clang offload bundle).
Host-side changes of the library leave it alone; any change of a kernel moves it.  profiles/hbm_traffic.json carries the value
its counters were read on, and bench.py reports no traffic figure for another one."""
import hashlib
import os
import struct

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsmg_hetmers.so")


def code_object_hash(path: str = LIB) -> str:
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    h = hashlib.sha256()
    found = False
    i = data.find(magic)
    while i >= 0:
        n = struct.unpack_from("<Q", data, i + 24)[0]
        o = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            triple = data[o + 24: o + 24 + tl]
            o += 24 + tl
            if b"gfx950" in triple:
                h.update(data[i + off: i + off + size])
                found = True
        i = data.find(magic, i + 1)
    if not found:                      # (no bundle found: fall back to the whole file, which still identifies the build)
        h.update(data)
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(code_object_hash())
