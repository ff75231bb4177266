""".agc v3 container reader (test/diagnostic aid): streams -> list of (metadata, payload).
Layout per src/common/archive.cpp:142-169, 280-293 and io.h:371-380."""
import struct


def _num(b, p):
    n = b[p]
    v = int.from_bytes(b[p + 1:p + 1 + n], "big") if n else 0
    return v, p + 1 + n


def parse(data):
    fs = struct.unpack("<Q", data[-8:])[0]
    p = len(data) - 8 - fs
    n_streams, p = _num(data, p)
    streams = {}
    order = []
    for _ in range(n_streams):
        e = data.index(b"\0", p)
        name = data[p:e].decode("latin1")
        p = e + 1
        n_parts, p = _num(data, p)
        raw_size, p = _num(data, p)
        parts = []
        for _i in range(n_parts):
            off, p = _num(data, p)
            size, p = _num(data, p)
            meta, q = _num(data, off)
            parts.append((meta, bytes(data[q:q + size]), off))
        streams[name] = parts
        order.append(name)
    return streams, order


def diff(a, b, limit=10):
    """human-readable differences between two archives (bytes)"""
    sa, oa = parse(a)
    sb, ob = parse(b)
    out = []
    if oa != ob:
        out.append(f"stream order/names differ: {len(oa)} vs {len(ob)}; first diff: "
                   f"{next(((x, y) for x, y in zip(oa, ob) if x != y), (oa[len(ob):][:3], ob[len(oa):][:3]))}")
    for name in oa:
        if name not in sb:
            out.append(f"stream {name} missing in b")
            continue
        pa, pb = sa[name], sb[name]
        if len(pa) != len(pb):
            out.append(f"stream {name}: {len(pa)} vs {len(pb)} parts")
        for i, (x, y) in enumerate(zip(pa, pb)):
            if x[0] != y[0] or x[1] != y[1]:
                out.append(f"stream {name} part {i}: meta {x[0]} vs {y[0]}, size {len(x[1])} vs {len(y[1])}, offsets {x[2]} vs {y[2]}")
            elif x[2] != y[2]:
                out.append(f"stream {name} part {i}: same bytes, file offset {x[2]} vs {y[2]}")
        if len(out) >= limit:
            break
    return out[:limit]
