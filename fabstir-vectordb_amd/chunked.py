"""The reference's chunked on-disk index format, read and written for the GPU index.

Files under `<path>/` (src/hybrid/persistence.rs:188-277 save_index_chunked, :497-693 load_index_chunked):

    manifest.json          Manifest v3 (src/core/chunk.rs:236-306), pretty JSON
    chunks/chunk-N.cbor    VectorChunk{chunk_id, start_idx, end_idx, vectors: map<VectorId -> [f32]>} (:36-42)
    hnsw_nodes.cbor        [HNSWNode{id, vector, level, neighbors: [[VectorId]], is_deleted}] (src/hnsw/core.rs:48-56)
    timestamps.cbor        {timestamps: map<VectorId -> RFC 3339 string>} (persistence.rs:106-109)
    metadata.cbor          HybridMetadata{version, config, counts, timestamp, ivf_trained} (:56-67)

CBOR is what `serde_cbor = "0.11"` (Cargo.toml:17; the crate is not in the tree) emits for those derives, restated
from RFC 8949 and serde's data model: structs are maps keyed by field name, `VectorId([u8; 32])` is an array of 32
unsigned integers (also when it is a map key), `Option::None` is null, `usize` the shortest unsigned form, and a
float is written in the shortest of f16 / f32 (/ f64) that holds it exactly.  The codec is pinned by the RFC's
Appendix A examples (tests/test_chunked_format.py); there is no file written by the Rust code on this box, so
byte-for-byte agreement with serde_cbor itself is unpinned — the decoder accepts every definite- and
indefinite-length form, whatever the writer chose.

Loading reproduces the reference's reconstruction: the graph comes back node for node from hnsw_nodes.cbor, the
entry point from the manifest, and every chunk vector whose display id (`vec_<8 hex>`) is not in the manifest's
node map is re-assigned to its nearest centroid — on the GPU, with the reference's arithmetic (`IVFIndex.assign`,
the same kernel the search's coarse stage uses) — and appended to that list in chunk order.  Vector ids are 32
bytes on disk and u64 row ids in the engine: the row id is the first 8 bytes, little endian (as
session.VectorId.row_id), and the loader returns the table back to the 32-byte ids.
"""
import json
import os
import re
import struct
from datetime import datetime, timezone

import numpy as np

from .engine import FvdbError
from .index import HybridIndex
from .session import blake3

MANIFEST_VERSION = 3     # src/core/chunk.rs:30
METADATA_VERSION = 1     # src/hybrid/persistence.rs:15
CHUNK_SIZE = 10000       # persistence.rs:190


class PersistenceError(Exception):
    """src/hybrid/persistence.rs:17-42; `.kind` is the variant name."""

    def __init__(self, kind, msg):
        super().__init__(f"{kind}: {msg}")
        self.kind = kind


# ------------------------------------------------------------------------------------------------------
# CBOR (RFC 8949), the subset serde_cbor produces plus every form a decoder must accept
# ------------------------------------------------------------------------------------------------------
class F32(float):
    """A float that came from / goes to an f32 field (serde's serialize_f32: f16 if exact, else f32)."""


def _head(major, n):
    if n < 24:
        return bytes([major << 5 | n])
    if n < 1 << 8:
        return bytes([major << 5 | 24, n])
    if n < 1 << 16:
        return bytes([major << 5 | 25]) + struct.pack(">H", n)
    if n < 1 << 32:
        return bytes([major << 5 | 26]) + struct.pack(">I", n)
    return bytes([major << 5 | 27]) + struct.pack(">Q", n)


def _enc_float(x, single):
    if x != x:
        return b"\xf9\x7e\x00"
    if x in (float("inf"), float("-inf")):
        return b"\xf9\x7c\x00" if x > 0 else b"\xf9\xfc\x00"
    with np.errstate(over="ignore"):
        h = np.float16(x)
        if float(h) == x:
            return b"\xf9" + struct.pack(">e", float(h))
        if single or float(np.float32(x)) == x:
            return b"\xfa" + struct.pack(">f", x)
    return b"\xfb" + struct.pack(">d", x)


def encode_f32_array(a):
    """[f32] as serde_cbor writes a Vec<f32>: array header, then f16 where exact, f32 otherwise (vectorised)."""
    a = np.ascontiguousarray(a, np.float32).ravel()
    with np.errstate(over="ignore", invalid="ignore"):
        h = a.astype(np.float16)
        short = (h.astype(np.float32) == a) | ~np.isfinite(a)
    # non-finite values take serde_cbor's canonical f16 forms
    hb = h.view(np.uint16).copy()
    hb[np.isnan(a)] = 0x7E00
    size = np.where(short, 3, 5)
    off = np.concatenate(([0], np.cumsum(size)))
    out = np.zeros(int(off[-1]), np.uint8)
    so, lo = off[:-1][short], off[:-1][~short]
    out[so] = 0xF9
    out[so + 1] = hb[short] >> 8
    out[so + 2] = hb[short] & 0xFF
    out[lo] = 0xFA
    wb = a[~short].view(np.uint32)
    for j in range(4):
        out[lo + 1 + j] = (wb >> (24 - 8 * j)) & 0xFF
    return _head(4, a.size) + out.tobytes()


def cbor_encode(o):
    if o is None:
        return b"\xf6"
    if o is True:
        return b"\xf5"
    if o is False:
        return b"\xf4"
    if isinstance(o, (int, np.integer)):
        o = int(o)
        return _head(0, o) if o >= 0 else _head(1, -1 - o)
    if isinstance(o, F32) or isinstance(o, np.float32):
        return _enc_float(float(o), True)
    if isinstance(o, (float, np.floating)):
        return _enc_float(float(o), False)
    if isinstance(o, VectorIdBytes):
        return _head(4, 32) + b"".join(_head(0, b) for b in o)
    if isinstance(o, bytes):
        return _head(2, len(o)) + o
    if isinstance(o, str):
        b = o.encode("utf-8")
        return _head(3, len(b)) + b
    if isinstance(o, np.ndarray) and o.dtype == np.float32:
        return encode_f32_array(o)
    if isinstance(o, (list, tuple)):
        return _head(4, len(o)) + b"".join(cbor_encode(x) for x in o)
    if isinstance(o, dict):
        return _head(5, len(o)) + b"".join(cbor_encode(k) + cbor_encode(v) for k, v in o.items())
    if isinstance(o, PairMap):
        return _head(5, len(o.pairs)) + b"".join(cbor_encode(k) + cbor_encode(v) for k, v in o.pairs)
    raise TypeError(f"cannot encode {type(o)}")


class VectorIdBytes(bytes):
    """32 bytes written as serde writes `VectorId([u8; 32])`: an array of 32 unsigned integers."""


class PairMap:
    """A map kept as (key, value) pairs in file order (keys may be arrays, which are legal CBOR map keys)."""

    def __init__(self, pairs):
        self.pairs = pairs


class _Dec:
    def __init__(self, data):
        self.b = bytes(data)
        self.u8 = np.frombuffer(self.b, np.uint8)
        self.p = 0
        self.depth = 0

    def _need(self, n):
        if self.p + n > len(self.b):
            raise ValueError("unexpected end of CBOR input")

    def _arg(self, info):
        if info < 24:
            return info
        if info > 27:
            raise ValueError("reserved CBOR additional information")
        n = 1 << (info - 24)
        self._need(n)
        v = int.from_bytes(self.b[self.p:self.p + n], "big")
        self.p += n
        return v

    def _floats(self, n):
        """n float items starting at p -> np.float32 array, or None if the items are not all floats."""
        p, u8 = self.p, self.u8
        if n == 0 or p >= len(u8):
            return None
        t = u8[p]
        if t == 0xFA and p + 5 * n <= len(u8) and np.all(u8[p:p + 5 * n:5] == 0xFA):
            w = u8[p:p + 5 * n].reshape(n, 5)[:, 1:]
            self.p += 5 * n
            return np.ascontiguousarray(w).view(">f4").ravel().astype(np.float32)
        if t == 0xF9 and p + 3 * n <= len(u8) and np.all(u8[p:p + 3 * n:3] == 0xF9):
            w = u8[p:p + 3 * n].reshape(n, 3)[:, 1:]
            self.p += 3 * n
            return np.ascontiguousarray(w).view(">f2").ravel().astype(np.float32)
        if t not in (0xF9, 0xFA, 0xFB):
            return None
        out, q = np.empty(n, np.float32), p  # mixed widths
        b = self.b
        for i in range(n):
            if q >= len(b):
                raise ValueError("unexpected end of CBOR input")
            t = b[q]
            if t == 0xFA:
                out[i] = struct.unpack_from(">f", b, q + 1)[0]
                q += 5
            elif t == 0xF9:
                out[i] = struct.unpack_from(">e", b, q + 1)[0]
                q += 3
            elif t == 0xFB:
                out[i] = struct.unpack_from(">d", b, q + 1)[0]  # `as f32`
                q += 9
            else:
                return None
        self.p = q
        return out

    def item(self):
        self.depth += 1
        if self.depth > 128:  # serde_cbor's recursion limit
            raise ValueError("CBOR nesting deeper than 128")
        try:
            return self._item()
        finally:
            self.depth -= 1

    def _item(self):
        self._need(1)
        ib = self.b[self.p]
        self.p += 1
        major, info = ib >> 5, ib & 31
        if major == 7:
            if info == 20:
                return False
            if info == 21:
                return True
            if info in (22, 23):
                return None
            if info == 25:
                self._need(2)
                v = struct.unpack_from(">e", self.b, self.p)[0]
                self.p += 2
                return F32(v)
            if info == 26:
                self._need(4)
                v = struct.unpack_from(">f", self.b, self.p)[0]
                self.p += 4
                return F32(v)
            if info == 27:
                self._need(8)
                v = struct.unpack_from(">d", self.b, self.p)[0]
                self.p += 8
                return v
            if info == 31:
                raise ValueError("unexpected CBOR break")
            return self._arg(info)  # simple value
        if major == 6:  # tag: the content is what serde sees
            self._arg(info)
            return self.item()
        indefinite = info == 31
        n = None if indefinite else self._arg(info)
        if major == 0:
            return n
        if major == 1:
            return -1 - n
        if major in (2, 3):
            if indefinite:
                parts = []
                while self.b[self.p] != 0xFF:
                    parts.append(self.item())
                self.p += 1
                raw = b"".join(x if isinstance(x, bytes) else x.encode() for x in parts)
            else:
                self._need(n)
                raw = self.b[self.p:self.p + n]
                self.p += n
            return raw if major == 2 else raw.decode("utf-8")
        if major == 4:
            if not indefinite:
                if n > len(self.b) - self.p:
                    raise ValueError("CBOR array longer than its input")
                f = self._floats(n)
                if f is not None:
                    return f
                return [self.item() for _ in range(n)]
            out = []
            while True:
                self._need(1)
                if self.b[self.p] == 0xFF:
                    self.p += 1
                    return out
                out.append(self.item())
        pairs = []
        while True:
            if indefinite:
                self._need(1)
                if self.b[self.p] == 0xFF:
                    self.p += 1
                    break
            elif len(pairs) == n:
                break
            k = self.item()
            pairs.append((k, self.item()))
        if all(isinstance(k, (str, int)) for k, _ in pairs):
            return dict(pairs)
        return PairMap(pairs)


def cbor_decode(data):
    """One CBOR item from `data`; anything malformed raises ValueError."""
    d = _Dec(data)
    try:
        v = d.item()
    except (IndexError, struct.error, UnicodeDecodeError, OverflowError, MemoryError, AttributeError, TypeError) as e:
        raise ValueError(f"malformed CBOR: {e}") from e
    if d.p != len(d.b):
        raise ValueError("trailing bytes after the CBOR item")  # serde_cbor::from_slice rejects them too
    return v


def plain(v):
    """Decoded CBOR -> plain JSON-like Python values (serde_json::Value on the reference's side)."""
    if isinstance(v, np.ndarray):
        return [float(x) for x in v]
    if isinstance(v, F32):
        return float(v)
    if isinstance(v, list):
        return [plain(x) for x in v]
    if isinstance(v, dict):
        return {k: plain(x) for k, x in v.items()}
    return v


# ------------------------------------------------------------------------------------------------------
# ids, timestamps
# ------------------------------------------------------------------------------------------------------
def _id_bytes(v, what="VectorId"):
    """[u8; 32] as decoded (list of ints, or a byte string) -> bytes."""
    if isinstance(v, (bytes, bytearray)):
        b = bytes(v)
    elif isinstance(v, np.ndarray):  # 32 values that happened to look like floats cannot be an id
        raise PersistenceError("Deserialization", f"{what}: expected 32 integers")
    else:
        try:
            b = bytes(v)
        except (TypeError, ValueError) as e:
            raise PersistenceError("Deserialization", f"{what}: {e}") from e
    if len(b) != 32:
        raise PersistenceError("Deserialization", f"{what}: expected 32 bytes, got {len(b)}")
    return b


def display_id(b):
    return "vec_" + b[:4].hex()  # src/core/types.rs:32-34


def row_id(b):
    return struct.unpack("<Q", b[:8])[0]


def default_id_bytes(rid):
    """32-byte id for a row that has none on record: the row id, zero extended (round-trips through row_id)."""
    return struct.pack("<Q", int(rid)) + bytes(24)


_RFC3339 = re.compile(r"^(\d{4})-(\d\d)-(\d\d)[Tt ](\d\d):(\d\d):(\d\d)(?:\.(\d+))?(Z|z|[+-]\d\d:\d\d)$")


def parse_timestamp(s):
    """chrono's DateTime<Utc> as serde writes it (RFC 3339) -> seconds since the epoch (f64, the host's clock)."""
    m = _RFC3339.match(s) if isinstance(s, str) else None
    if not m:
        raise PersistenceError("Deserialization", f"invalid timestamp {s!r}")
    y, mo, d, h, mi, sec = (int(m.group(i)) for i in range(1, 7))
    t = datetime(y, mo, d, h, mi, sec, tzinfo=timezone.utc).timestamp()
    if m.group(7):
        t += int(m.group(7)[:9].ljust(9, "0")) / 1e9
    z = m.group(8)
    if z not in ("Z", "z"):
        off = int(z[1:3]) * 3600 + int(z[4:6]) * 60
        t -= off if z[0] == "+" else -off
    return t


def format_timestamp(t):
    """seconds -> `2025-01-02T03:04:05.123456789Z` (chrono prints 0, 3, 6 or 9 fractional digits)."""
    sec = int(np.floor(t))
    ns = int(round((t - sec) * 1e9))
    if ns >= 10 ** 9:
        sec, ns = sec + 1, ns - 10 ** 9
    base = datetime.fromtimestamp(sec, tz=timezone.utc).strftime("%Y-%m-%dT%H:%M:%S")
    if ns == 0:
        frac = ""
    elif ns % 10 ** 6 == 0:
        frac = f".{ns // 10 ** 6:03d}"
    elif ns % 10 ** 3 == 0:
        frac = f".{ns // 10 ** 3:06d}"
    else:
        frac = f".{ns:09d}"
    return base + frac + "Z"


# ------------------------------------------------------------------------------------------------------
# storage (the reference's S5Storage get/put, src/core/storage.rs) — a directory or a dict
# ------------------------------------------------------------------------------------------------------
class DirStorage:
    def __init__(self, root):
        self.root = root

    def put(self, path, data):
        full = os.path.join(self.root, path)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        with open(full, "wb") as f:
            f.write(data)

    def get(self, path):
        try:
            with open(os.path.join(self.root, path), "rb") as f:
                return f.read()
        except FileNotFoundError:
            return None


class MemoryStorage(dict):
    def put(self, path, data):
        self[path] = bytes(data)


# ------------------------------------------------------------------------------------------------------
# Manifest (src/core/chunk.rs:236-306)
# ------------------------------------------------------------------------------------------------------
def manifest_from_json(text):
    """Manifest::from_json: required fields as serde demands them, version <= MANIFEST_VERSION."""
    try:
        m = json.loads(text)
    except (ValueError, UnicodeDecodeError) as e:
        raise PersistenceError("Deserialization", f"Failed to parse manifest: {e}") from e
    if not isinstance(m, dict):
        raise PersistenceError("Deserialization", "Failed to parse manifest: expected a map")
    for field, typ in (("version", int), ("chunk_size", int), ("total_vectors", int), ("chunks", list)):
        if field not in m:
            raise PersistenceError("Deserialization", f"Failed to parse manifest: missing field `{field}`")
        if not isinstance(m[field], typ) or isinstance(m[field], bool):
            raise PersistenceError("Deserialization", f"Failed to parse manifest: invalid type for `{field}`")
    for field in ("hnsw_structure", "ivf_structure"):  # Option<_> without a default: null allowed, absent allowed
        m.setdefault(field, None)
    m.setdefault("deleted_vectors", None)
    m.setdefault("schema", None)
    if m["version"] > MANIFEST_VERSION:
        raise PersistenceError("IncompatibleVersion", f"expected <= {MANIFEST_VERSION}, found {m['version']}")
    for c in m["chunks"]:
        for field in ("chunk_id", "vector_count", "byte_size", "vector_id_range"):
            if not isinstance(c, dict) or field not in c:
                raise PersistenceError("Deserialization", f"Failed to parse manifest: chunk missing `{field}`")

    def bad(what):
        return PersistenceError("Deserialization", f"Failed to parse manifest: {what}")

    def is_int(v):
        return isinstance(v, int) and not isinstance(v, bool)

    # HNSWManifest (src/core/chunk.rs:152-160): entry_point [u8; 32], layers Vec<LayerMetadata>, node_chunk_map
    hs = m["hnsw_structure"]
    if hs is not None:
        if not isinstance(hs, dict):
            raise bad("hnsw_structure: expected a map")
        for field, typ in (("entry_point", list), ("layers", list), ("node_chunk_map", dict)):
            if field not in hs:
                raise bad(f"hnsw_structure: missing field `{field}`")
            if not isinstance(hs[field], typ):
                raise bad(f"hnsw_structure: invalid type for `{field}`")
        ep = hs["entry_point"]
        if len(ep) != 32 or not all(is_int(b) and 0 <= b <= 255 for b in ep):
            raise bad("hnsw_structure.entry_point: expected 32 bytes")
    # IVFManifest (:179-185): centroids Vec<Vec<f32>> (rectangular for set_trained), cluster_assignments usize -> chunks
    iv = m["ivf_structure"]
    if iv is not None:
        if not isinstance(iv, dict):
            raise bad("ivf_structure: expected a map")
        for field, typ in (("centroids", list), ("cluster_assignments", dict)):
            if field not in iv:
                raise bad(f"ivf_structure: missing field `{field}`")
            if not isinstance(iv[field], typ):
                raise bad(f"ivf_structure: invalid type for `{field}`")
        width = None
        for c in iv["centroids"]:
            if not isinstance(c, list) or not all(isinstance(v, (int, float)) and not isinstance(v, bool) for v in c):
                raise bad("ivf_structure.centroids: expected arrays of numbers")
            if width is not None and len(c) != width:
                raise bad("ivf_structure.centroids: rows of different lengths")
            width = len(c)
        for key in iv["cluster_assignments"]:
            if not (isinstance(key, str) and key.isascii() and key.isdigit()):  # a usize map key as serde_json writes it
                raise bad(f"ivf_structure.cluster_assignments: invalid cluster id {key!r}")
    dv = m["deleted_vectors"]
    if dv is not None and not isinstance(dv, list):
        raise bad("deleted_vectors: expected an array")
    return m


def manifest_validate(m):
    """Manifest::validate (:271-289): chunk ids unique."""
    ids = [c["chunk_id"] for c in m["chunks"]]
    for i, c in enumerate(ids):
        if c in ids[:i]:
            raise PersistenceError("ChunkOverlap", f"Duplicate chunk ID: {c}")


# ------------------------------------------------------------------------------------------------------
# save_index_chunked (persistence.rs:188-277)
# ------------------------------------------------------------------------------------------------------
def _config_of(c):
    """HybridConfig as serde writes it (src/hybrid/core.rs:37-46, Duration as whole seconds :49-67)."""
    return {"recent_threshold": int(c["recent_threshold"]),
            "hnsw_config": {"max_connections": c["max_connections"], "max_connections_layer_0": c["max_connections_layer_0"],
                            "ef_construction": c["ef_construction"], "seed": c.get("hnsw_seed")},
            "ivf_config": {"n_clusters": c["n_clusters"], "n_probe": c["n_probe"], "train_size": c["train_size"],
                           "max_iterations": c["max_iterations"], "seed": c.get("ivf_seed")},
            "migration_batch_size": c["migration_batch_size"], "auto_migrate": bool(c["auto_migrate"]),
            "min_ivf_training_size": c["min_ivf_training_size"]}


def snapshot_of(index):
    """Everything save_index_chunked writes, copied out of a HybridIndex (lists come back from HBM in list order)."""
    hnsw, ivf = index.hnsw(), index.ivf()
    nids, nlev, noff, nnb = hnsw.export_graph()
    trained = ivf.is_trained()
    tids, ts = index.export_timestamps()
    return {"config": dict(index.config), "recent_count": index.recent_count(), "historical_count": index.historical_count(),
            "ivf_trained": index.is_ivf_trained(),
            "node_ids": nids, "node_levels": nlev, "node_offsets": noff, "node_neighbors": nnb,
            "node_vectors": [hnsw.get_vector_by_id(i) for i in nids.tolist()],
            "node_deleted": [hnsw.is_deleted(i) for i in nids.tolist()], "entry_point": hnsw.entry_point(),
            "centroids": ivf.get_centroids() if trained else None,
            "lists": [ivf.export_list(c) for c in range(index.n_clusters)] if trained else [],
            "timestamp_ids": tids, "timestamps": ts}


def save_index_chunked(index, storage, path, id_table=None, now=0.0, chunk_size=CHUNK_SIZE):
    """Write `index` (a HybridIndex) under `path` in the reference's chunked layout; returns the manifest dict.
    `id_table`: u64 row id -> 32-byte VectorId (rows not in it get default_id_bytes)."""
    if not path:
        raise PersistenceError("InvalidData", "Path cannot be empty")
    return write_snapshot(snapshot_of(index), storage, path, id_table, now, chunk_size)


def write_snapshot(snap, storage, path, id_table=None, now=0.0, chunk_size=CHUNK_SIZE):
    """save_index_chunked's steps 1-9 over a snapshot (see snapshot_of for the fields)."""
    if not path:
        raise PersistenceError("InvalidData", "Path cannot be empty")
    if isinstance(storage, str):
        storage = DirStorage(storage)
    id_table = id_table or {}
    vid = lambda r: VectorIdBytes(id_table.get(int(r)) or default_id_bytes(r))  # noqa: E731
    total = snap["recent_count"] + snap["historical_count"]
    manifest = {"version": MANIFEST_VERSION, "chunk_size": chunk_size, "total_vectors": total, "chunks": [],
                "hnsw_structure": None, "ivf_structure": None}
    if total == 0:
        storage.put(f"{path}/manifest.json", json.dumps(manifest, indent=2).encode())
        return manifest
    # 1. every vector: live graph nodes, then every row of every inverted list (:279-311)
    nids, nlev = np.asarray(snap["node_ids"], np.uint64), np.asarray(snap["node_levels"], np.uint32)
    noff, nnb = np.asarray(snap["node_offsets"], np.uint64), np.asarray(snap["node_neighbors"], np.uint64)
    nvec, ndel = snap["node_vectors"], [bool(x) for x in snap["node_deleted"]]
    all_ids = [i for i, dead in zip(nids.tolist(), ndel) if not dead]
    all_vecs = [v for v, dead in zip(nvec, ndel) if not dead]
    trained = snap["centroids"] is not None
    n_clusters = snap["config"]["n_clusters"]
    ivf_deleted, nonempty = [], []
    for c, (rows, ids, live) in enumerate(snap["lists"]):
        ids, live = np.asarray(ids, np.uint64), np.asarray(live, bool)
        if ids.size:
            nonempty.append(c)
        all_ids += ids.tolist()
        all_vecs += list(rows)
        ivf_deleted += ids[~live].tolist()
    # 2-3. chunks of `chunk_size` in that order (:313-338), one CBOR file each (:340-375)
    n_chunks = 0
    for s in range(0, len(all_ids), chunk_size):
        e = min(s + chunk_size, len(all_ids))
        keys = [vid(r) for r in all_ids[s:e]]
        chunk = {"chunk_id": f"chunk-{n_chunks}", "start_idx": s, "end_idx": min(s + chunk_size - 1, len(all_ids) - 1),
                 "vectors": PairMap([(k, np.asarray(v, np.float32)) for k, v in zip(keys, all_vecs[s:e])])}
        data = cbor_encode(chunk)
        storage.put(f"{path}/chunks/chunk-{n_chunks}.cbor", data)
        manifest["chunks"].append({"chunk_id": f"chunk-{n_chunks}", "cid": None, "vector_count": e - s, "byte_size": len(data),
                                   "vector_id_range": [list(keys[0]), list(keys[-1])]})
        n_chunks += 1
    # the reference's placeholder node -> chunk rule (:455-475): chunk (len("vec_xxxxxxxx") mod n_chunks)
    chunk_of = manifest["chunks"][12 % n_chunks]["chunk_id"] if n_chunks else "chunk-0"
    # 4. graph summary (:377-405)
    entry = snap["entry_point"]
    layers = [int(np.sum(nlev >= l)) for l in range(int(nlev.max()) + 1)] if nids.size else [0]
    manifest["hnsw_structure"] = {
        "entry_point": list(vid(entry) if entry is not None else VectorIdBytes(blake3(b"placeholder"))),
        "layers": [{"layer_id": l, "node_count": c} for l, c in enumerate(layers)],
        "node_chunk_map": {display_id(vid(r)): chunk_of for r in nids.tolist()}}
    # 5. centroids + which chunks hold each cluster's rows (:407-453)
    manifest["ivf_structure"] = {
        "centroids": [[float(x) for x in row] for row in snap["centroids"]] if trained else [],
        "cluster_assignments": {str(c): ([chunk_of] if c in nonempty else []) for c in range(n_clusters)} if trained else {}}
    deleted = [display_id(vid(r)) for r, dead in zip(nids.tolist(), ndel) if dead] + [display_id(vid(r)) for r in ivf_deleted]
    if deleted:
        manifest["deleted_vectors"] = deleted
    storage.put(f"{path}/manifest.json", json.dumps(manifest, indent=2).encode())
    # 7. timestamps
    storage.put(f"{path}/timestamps.cbor", cbor_encode({"timestamps": PairMap(
        [(vid(r), format_timestamp(t)) for r, t in zip(np.asarray(snap["timestamp_ids"]).tolist(),
                                                       np.asarray(snap["timestamps"]).tolist())])}))
    # 8. every node with its links, deleted ones too
    nodes, slot = [], 0
    for i, r in enumerate(nids.tolist()):
        nb = []
        for _ in range(int(nlev[i]) + 1):
            nb.append([vid(x) for x in nnb[int(noff[slot]):int(noff[slot + 1])].tolist()])
            slot += 1
        nodes.append({"id": vid(r), "vector": np.asarray(nvec[i], np.float32), "level": int(nlev[i]), "neighbors": nb,
                      "is_deleted": ndel[i]})
    storage.put(f"{path}/hnsw_nodes.cbor", cbor_encode(nodes))
    # 9. metadata
    storage.put(f"{path}/metadata.cbor", cbor_encode(
        {"version": METADATA_VERSION, "config": _config_of(snap["config"]), "recent_count": snap["recent_count"],
         "historical_count": snap["historical_count"], "total_vectors": total, "timestamp": format_timestamp(now),
         "ivf_trained": bool(snap["ivf_trained"])}))
    return manifest


# ------------------------------------------------------------------------------------------------------
# load_index_chunked (persistence.rs:497-693)
# ------------------------------------------------------------------------------------------------------
def _get(storage, path, what):
    data = storage.get(path)
    if data is None:
        raise PersistenceError("MissingComponent", what)
    return data


def _struct(v, what, fields):
    if not isinstance(v, dict):
        raise PersistenceError("Deserialization", f"{what}: expected a map")
    for f in fields:
        if f not in v:
            raise PersistenceError("Deserialization", f"{what}: missing field `{f}`")
    return v


def _decode(data, what):
    try:
        return cbor_decode(data)
    except (ValueError, IndexError, struct.error, UnicodeDecodeError) as e:
        raise PersistenceError("Deserialization", f"{what}: {e}") from e


def _pairs(v, what):
    if isinstance(v, PairMap):
        return v.pairs
    if isinstance(v, dict) and not v:
        return []
    raise PersistenceError("Deserialization", f"{what}: expected a map keyed by VectorId")


def _vec(v, what):
    if isinstance(v, np.ndarray):
        return v
    if isinstance(v, list) and all(isinstance(x, (int, float)) and not isinstance(x, bool) for x in v):
        return np.asarray(v, np.float32)  # an integer-valued item is not a float to serde, but an empty list is fine
    raise PersistenceError("Deserialization", f"{what}: expected an array of floats")


def read_chunk(data):
    """VectorChunk::from_cbor -> (chunk_id, start_idx, end_idx, [32-byte ids], rows f32 [n, d]) in file order."""
    c = _struct(_decode(data, "Failed to parse chunk"), "VectorChunk", ("chunk_id", "start_idx", "end_idx", "vectors"))
    pairs = _pairs(c["vectors"], "VectorChunk.vectors")
    ids = [_id_bytes(k) for k, _ in pairs]
    rows = [_vec(v, "VectorChunk.vectors") for _, v in pairs]
    return c["chunk_id"], c["start_idx"], c["end_idx"], ids, rows


def load_index_chunked(ctx, storage, path, now=0.0, **config):
    """Open an index saved in the reference's chunked layout.  `config` are HybridIndex keyword arguments (the
    reference also takes the config from the caller, :497).  Returns (HybridIndex, {row id: 32-byte VectorId})."""
    if isinstance(storage, str):
        storage = DirStorage(storage)
    raw = storage.get(f"{path}/manifest.json")
    if raw is None:
        raise PersistenceError("MissingComponent", "manifest.json")
    try:
        text = raw.decode("utf-8")
    except UnicodeDecodeError as e:
        raise PersistenceError("Deserialization", f"Invalid UTF-8 in manifest: {e}") from e
    manifest = manifest_from_json(text)
    index = HybridIndex(ctx, **config)
    table = {}
    if manifest["total_vectors"] == 0:
        return index, table
    meta = _struct(_decode(_get(storage, f"{path}/metadata.cbor", "metadata.cbor"), "metadata"), "HybridMetadata",
                   ("version", "config", "recent_count", "historical_count", "total_vectors", "timestamp"))
    if meta["version"] > METADATA_VERSION:
        raise PersistenceError("IncompatibleVersion", f"expected <= {METADATA_VERSION}, found {meta['version']}")
    # 5. every chunk, in manifest order
    all_ids, all_rows = [], []
    for cm in manifest["chunks"]:
        _, _, _, ids, rows = read_chunk(_get(storage, f"{path}/chunks/{cm['chunk_id']}.cbor", f"chunk {cm['chunk_id']}"))
        all_ids += ids
        all_rows += rows
    for b in all_ids:
        table[row_id(b)] = b
    # 6. the graph, node for node
    hs = manifest["hnsw_structure"]
    raw = storage.get(f"{path}/hnsw_nodes.cbor")
    if raw is not None:
        nodes = _decode(raw, "Failed to deserialize HNSW nodes")
        if not isinstance(nodes, list):
            raise PersistenceError("Deserialization", "Failed to deserialize HNSW nodes: expected an array")
        seen, nid, nvec, nlev, nnb, ndel = set(), [], [], [], [], []
        for n in nodes:  # restore_node: a later node with the same id replaces the earlier one (HashMap insert)
            _struct(n, "HNSWNode", ("id", "vector", "level", "neighbors"))
        for n in reversed(nodes):
            b = _id_bytes(n["id"])
            if b in seen:
                continue
            seen.add(b)
            table[row_id(b)] = b
            nid.append(row_id(b))
            nvec.append(_vec(n["vector"], "HNSWNode.vector"))
            nlev.append(int(n["level"]))
            nnb.append([[row_id(_id_bytes(x)) for x in layer] for layer in n["neighbors"]])
            ndel.append(bool(n.get("is_deleted", False)))
        nid, nvec, nlev, nnb, ndel = nid[::-1], nvec[::-1], nlev[::-1], nnb[::-1], ndel[::-1]
        if nid:
            if hs is None:
                raise PersistenceError("InvalidData", "hnsw_nodes.cbor without hnsw_structure: no entry point to search from")
            entry = row_id(_id_bytes(hs["entry_point"], "entry_point"))
            known = set(nid)
            if entry not in known:
                raise PersistenceError("InvalidData", "entry point is not a stored node")
            off, flat = [0], []
            for lv, layers in zip(nlev, nnb):
                for l in range(lv + 1):
                    if l < len(layers):
                        flat += [x for x in layers[l] if x in known]  # a link to a missing node is never followed
                    off.append(len(flat))
            if any(v.size != nvec[0].size or v.size == 0 for v in nvec):
                raise PersistenceError("HNSWError", "Failed to restore node: vectors of different dimensions")
            try:
                index.hnsw().restore(np.asarray(nid, np.uint64), np.stack(nvec), np.asarray(nlev, np.uint32),
                                     np.asarray(off, np.uint64), np.asarray(flat, np.uint64), entry)
            except FvdbError as e:
                raise PersistenceError("HNSWError", f"Failed to restore node: {e}") from e
            h = index.hnsw()
            for r, dead in zip(nid, ndel):
                if dead:
                    h.mark_deleted(r)
    # 7. lists: every chunk vector not in the graph's node map goes to its nearest centroid (:591-640)
    iv = manifest["ivf_structure"]
    if iv is not None:
        _struct(iv, "IVFManifest", ("centroids", "cluster_assignments"))
        cents = [np.asarray(c, np.float32) for c in iv["centroids"]]
        if cents:
            if len(cents) != index.n_clusters:
                raise PersistenceError("InvalidData", f"{len(cents)} centroids for n_clusters = {index.n_clusters}")
            index.ivf().set_trained(np.stack(cents))
        clusters = set()
        for key in iv["cluster_assignments"]:
            c = int(key)
            if c < 0 or c >= index.n_clusters:
                raise PersistenceError("InvalidData", f"Invalid cluster ID: {c}")
            clusters.add(c)
        in_graph = set(hs["node_chunk_map"]) if hs is not None else set()
        keep = [i for i, b in enumerate(all_ids) if display_id(b) not in in_graph]
        if keep and clusters:
            if not cents:
                raise PersistenceError("IVFError", "Failed to find cluster: index not trained")
            dim = len(cents[0])
            for i in keep:  # find_cluster's dimension check (src/ivf/core.rs:493-507)
                if all_rows[i].size != dim:
                    raise PersistenceError("IVFError", f"Failed to find cluster: Dimension mismatch: expected {dim}, "
                                                       f"got {all_rows[i].size}")
            x = np.stack([all_rows[i] for i in keep])
            rid = np.asarray([row_id(all_ids[i]) for i in keep], np.uint64)
            ivf = index.ivf()
            cl = ivf.assign(x)  # GPU, the reference's arithmetic (find_cluster, src/ivf/core.rs:373-386)
            sel = np.isin(cl, np.asarray(sorted(clusters), cl.dtype))
            if sel.any():
                n_ok, n_failed = ivf.batch_insert(rid[sel], x[sel])
                if n_failed:
                    raise PersistenceError("IVFError", "Failed to insert to inverted list: duplicate vector")
    # 8-9. timestamps, from_parts
    ts = _struct(_decode(_get(storage, f"{path}/timestamps.cbor", "timestamps.cbor"), "timestamps"), "SerializableTimestamps",
                 ("timestamps",))
    pairs = _pairs(ts["timestamps"], "timestamps")
    tid = np.asarray([row_id(_id_bytes(k)) for k, _ in pairs], np.uint64)
    tval = np.asarray([parse_timestamp(v) for _, v in pairs], np.float64)
    for k, _ in pairs:
        table.setdefault(row_id(_id_bytes(k)), _id_bytes(k))
    try:
        index.from_parts(tid, tval, meta["recent_count"], meta["historical_count"], bool(meta.get("ivf_trained", False)))
    except FvdbError as e:
        raise PersistenceError("InvalidData", f"Failed to reconstruct index: {e}") from e
    # 10. deleted ids are stored as display strings and hashed again on the way in (:675-682) — best effort
    for s in manifest.get("deleted_vectors") or []:
        try:
            index.delete(row_id(blake3(str(s).encode("utf-8"))), now)
        except Exception:  # noqa: BLE001 — "ignore errors if vector doesn't exist"
            pass
    return index, table
