"""VectorDbSession / REST search surface over the GPU hybrid index.

Mirrors the observable behaviour of the reference's two public surfaces for the hot path:
  * Node binding  bindings/node/src/session.rs  (search :203-336, add_vectors :340-432, delete)
  * REST handler  src/api/rest.rs               (search :599-677)
Kept: f64 -> f32 narrowing of inputs (bindings/node/src/utils.rs:6-8), dimension latch and mismatch
errors, first-call initialisation with the first <= 10 vectors (session.rs:365-378), score =
1/(1+distance) computed in f32 (session.rs:291,328; rest.rs:653), default threshold 0.0, results in
ascending distance, `_originalId` round trip, ids shown as `vec_<8 hex of BLAKE3>` when no original id
exists (src/core/types.rs:32-34), metadata filters with 3x oversampling (the filter language in metadata_filter.py;
the oversampled search itself is HybridIndex::search_with_filter of the C++ host mirror), saveToS5 /
loadUserVectors over the chunked on-disk format (chunked.py).  Beyond SURVEY §8's rows and kept only because the
surface tests exercise them: setSchema (metadata_schema.py), deleteByMetadata, updateMetadata.  Not built (out of
scope): the S5 network back end, native metadata types.

`blake3()` below is written from the published BLAKE3 specification (the `blake3` crate is not
available here); it is pinned by the official known-answer vector for the empty input.
"""
import struct

import numpy as np

from .index import HybridIndex
from .metadata_schema import MetadataSchema, SchemaError
from .metadata_filter import FilterError, MetadataFilter

# ---------------------------------------------------------------------------------------------
# BLAKE3 (hash mode, 32-byte output) — VectorId::from_string (src/core/types.rs:19-22)
# ---------------------------------------------------------------------------------------------
_IV = (0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19)
_PERM = (2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
_CHUNK_START, _CHUNK_END, _PARENT, _ROOT = 1, 2, 4, 8
_M32 = 0xFFFFFFFF


def _rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & _M32


def _g(s, a, b, c, d, mx, my):
    s[a] = (s[a] + s[b] + mx) & _M32
    s[d] = _rotr(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & _M32
    s[b] = _rotr(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b] + my) & _M32
    s[d] = _rotr(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & _M32
    s[b] = _rotr(s[b] ^ s[c], 7)


def _compress(cv, block_words, counter, block_len, flags):
    s = list(cv) + list(_IV[:4]) + [counter & _M32, (counter >> 32) & _M32, block_len, flags]
    m = list(block_words)
    for r in range(7):
        _g(s, 0, 4, 8, 12, m[0], m[1])
        _g(s, 1, 5, 9, 13, m[2], m[3])
        _g(s, 2, 6, 10, 14, m[4], m[5])
        _g(s, 3, 7, 11, 15, m[6], m[7])
        _g(s, 0, 5, 10, 15, m[8], m[9])
        _g(s, 1, 6, 11, 12, m[10], m[11])
        _g(s, 2, 7, 8, 13, m[12], m[13])
        _g(s, 3, 4, 9, 14, m[14], m[15])
        if r < 6:
            m = [m[p] for p in _PERM]
    for i in range(8):
        s[i] ^= s[i + 8]
        s[i + 8] ^= cv[i]
    return s


def _words(block):
    return struct.unpack("<16I", block.ljust(64, b"\0"))


def _chunk_output(chunk, counter):
    """(cv, last block words, last block len, flags-without-ROOT) of one <=1024-byte chunk."""
    cv = _IV
    blocks = [chunk[i:i + 64] for i in range(0, len(chunk), 64)] or [b""]
    for i, blk in enumerate(blocks[:-1]):
        flags = _CHUNK_START if i == 0 else 0
        cv = tuple(_compress(cv, _words(blk), counter, 64, flags)[:8])
    last = blocks[-1]
    flags = (_CHUNK_START if len(blocks) == 1 else 0) | _CHUNK_END
    return cv, _words(last), len(last), flags, counter


def blake3(data: bytes) -> bytes:
    chunks = [data[i:i + 1024] for i in range(0, len(data), 1024)] or [b""]
    if len(chunks) == 1:
        cv, w, blen, flags, ctr = _chunk_output(chunks[0], 0)
        out = _compress(cv, w, ctr, blen, flags | _ROOT)
        return struct.pack("<8I", *out[:8])
    # binary tree over chunk chaining values: left subtrees are the largest power of two
    def subtree_cv(lo, hi):
        if hi - lo == 1:
            cv, w, blen, flags, ctr = _chunk_output(chunks[lo], lo)
            return tuple(_compress(cv, w, ctr, blen, flags)[:8])
        n = hi - lo
        left = 1 << ((n - 1).bit_length() - 1)
        l, r = subtree_cv(lo, lo + left), subtree_cv(lo + left, hi)
        return tuple(_compress(_IV, l + r, 0, 64, _PARENT)[:8])

    n = len(chunks)
    left = 1 << ((n - 1).bit_length() - 1)
    l, r = subtree_cv(0, left), subtree_cv(left, n)
    out = _compress(_IV, l + r, 0, 64, _PARENT | _ROOT)
    return struct.pack("<8I", *out[:8])


class VectorId:
    """32-byte BLAKE3 of the string id (src/core/types.rs:9-43)."""

    def __init__(self, s):
        self.bytes = blake3(s.encode("utf-8"))

    def to_string(self):
        return "vec_" + self.bytes[:4].hex()  # `vec_<8 hex>` (src/core/types.rs:32-34)

    def row_id(self):
        return struct.unpack("<Q", self.bytes[:8])[0]  # u64 row id handed to the engine


def js_array_to_vec_f32(v):
    """bindings/node/src/utils.rs:6-8: `x as f32` per element (round to nearest even)."""
    return np.asarray(v, dtype=np.float64).astype(np.float32)


class SessionError(Exception):
    pass


class VectorDbSession:
    """bindings/node/src/session.rs VectorDBSession — search()/addVectors()/deleteVector() on the GPU index."""

    def __init__(self, ctx, now=0.0, storage=None, session_id="session", **hybrid_config):
        self.ctx = ctx
        self.storage = storage                              # get/put store (chunked.DirStorage / MemoryStorage)
        self.session_id = session_id                        # path prefix of save_to_s5 (:636-642)
        self.index = HybridIndex(ctx, **hybrid_config)   # HybridConfig::default unless overridden
        self.vector_dimension = None                        # latched by the first addVectors (:345-357)
        self.metadata = {}                                  # VectorId string -> metadata (with _originalId)
        self._rows = {}                                     # u64 row id -> VectorId string
        self.now = now
        self.destroyed = False
        self.schema = None                                  # MetadataSchema or None (set_schema)

    # -- add_vectors: session.rs:340-432 -------------------------------------------------------
    def add_vectors(self, vectors):
        """vectors: iterable of dicts {"id": str, "vector": [float], "metadata": any}."""
        if self.destroyed:
            raise SessionError("Session already destroyed")
        vectors = list(vectors)
        if vectors:
            first_dim = len(vectors[0]["vector"])
            if self.vector_dimension is not None and first_dim != self.vector_dimension:
                raise SessionError(f"Vector dimension mismatch: expected {self.vector_dimension}, got {first_dim}")
            if self.vector_dimension is None:
                self.vector_dimension = first_dim
        if not self.index.is_initialized() and vectors:
            training = np.stack([js_array_to_vec_f32(v["vector"]) for v in vectors[:10]])  # first 10 (:367-371)
            self.index.initialize(training)
        for inp in vectors:
            if self.schema is not None:  # :388-392, before anything of this vector is stored
                try:
                    self.schema.validate(inp.get("metadata", {}))
                except SchemaError as e:
                    raise SessionError(f"Schema validation failed for vector '{inp['id']}': {e}") from e
            vid = VectorId(inp["id"])
            vec = js_array_to_vec_f32(inp["vector"])
            if vec.size != self.vector_dimension:
                raise SessionError(f"Vector dimension mismatch: expected {self.vector_dimension}, got {vec.size}")
            try:
                self.index.insert(vid.row_id(), vec, now=self.now)
            except Exception as e:  # index errors surface as strings (bindings/node/src/error.rs:61-65)
                raise SessionError(f"Failed to add vector: {e}") from e
            md = inp.get("metadata", {})
            if isinstance(md, dict):
                md = dict(md)
                md["_originalId"] = inp["id"]
            else:
                md = {"_originalId": inp["id"], "_userMetadata": md}
            self.metadata[vid.to_string()] = md
            self._rows[vid.row_id()] = vid.to_string()

    addVectors = add_vectors

    # -- search: session.rs:203-336 ------------------------------------------------------------------
    def search(self, query_vector, k, options=None):
        if self.destroyed:
            raise SessionError("Session already destroyed")
        options = options or {}
        q = js_array_to_vec_f32(query_vector)
        if self.vector_dimension is not None and q.size != self.vector_dimension:
            raise SessionError(
                f"Query vector dimension mismatch: expected {self.vector_dimension} dimensions, got {q.size}")
        threshold = np.float32(options.get("threshold", 0.0))
        flt = None
        if options.get("filter") is not None:  # session.rs:233-247
            try:
                flt = MetadataFilter.from_json(options["filter"])
            except FilterError as e:
                raise SessionError(f"Invalid filter: {e}") from e
        # with a filter: 3 k candidates, keep those whose metadata match, truncate — HybridIndex::search_with_filter of
        # the host mirror (src/hybrid/core.rs:513-549); this side only answers "does this id's metadata match"
        matches = None
        if flt is not None:
            def matches(rid):
                key = self._rows.get(rid)
                return key in self.metadata and flt.matches(self.metadata[key])
        try:
            res = self.index.search_with_filter(q.reshape(1, -1), int(k), matches, now=self.now)  # defaults: ef 50, nprobe 10
        except Exception as e:
            raise SessionError(f"Search failed: {e}") from e
        out = []
        ids, ds = res[0]
        pairs = list(zip(ids.tolist(), ds.tolist()))
        for rid, d in pairs:
            score = np.float32(1.0) / (np.float32(1.0) + np.float32(d))
            if not score >= threshold:
                continue
            key = self._rows.get(rid, f"row_{rid}")
            md = dict(self.metadata.get(key, {}))
            rid_out = key
            if isinstance(md.get("_originalId"), str):
                rid_out = md.pop("_originalId")
                if "_userMetadata" in md:
                    md = md.pop("_userMetadata")
            out.append({"id": rid_out, "score": float(score), "metadata": md})
        return out

    def delete_vector(self, id):
        vid = VectorId(id)
        if self.destroyed:
            raise SessionError("Session already destroyed")
        try:
            self.index.delete(vid.row_id(), self.now)
        except Exception as e:
            raise SessionError(f"Failed to delete vector: {e}") from e
        self.metadata.pop(vid.to_string(), None)  # :464-467

    deleteVector = delete_vector

    # -- delete_by_metadata: session.rs:489-578 ----------------------------------------------------
    def delete_by_metadata(self, flt):
        """Soft-delete every vector whose metadata matches the simple field filter (`matches_filter` below).
        Returns {"deleted_count", "deleted_ids"} with the user's original ids."""
        if self.destroyed:
            raise SessionError("Session already destroyed")
        matching = []
        for key, md in self.metadata.items():
            if matches_filter(md, flt):
                oid = md.get("_originalId") if isinstance(md, dict) else None
                matching.append((key, oid if isinstance(oid, str) else key))
        ok = 0
        for _, oid in matching:  # batch_delete (src/hybrid/core.rs:968-986): failures are counted, not raised
            try:
                self.index.delete(VectorId(oid).row_id(), self.now)
                ok += 1
            except Exception:  # noqa: BLE001
                pass
        done = matching[:ok]  # the first `successful` entries, as the reference takes them (:553-558)
        for key, _ in done:
            self.metadata.pop(key, None)
        return {"deleted_count": ok, "deleted_ids": [oid for _, oid in done]}

    deleteByMetadata = delete_by_metadata

    # -- update_metadata: session.rs:581-632 -------------------------------------------------------
    def update_metadata(self, id, metadata):
        if self.destroyed:
            raise SessionError("Session already destroyed")
        key = VectorId(id).to_string()
        if self.schema is not None:  # :592-598
            try:
                self.schema.validate(metadata)
            except SchemaError as e:
                raise SessionError(f"Schema validation failed for vector '{id}': {e}") from e
        if key not in self.metadata:
            raise SessionError(f"Vector with id '{id}' does not exist")
        if isinstance(metadata, dict):
            md = dict(metadata)
            md["_originalId"] = id
        else:
            md = {"_originalId": id, "_userMetadata": metadata}
        self.metadata[key] = md

    updateMetadata = update_metadata

    def set_schema(self, schema_json):
        """session.rs:742-765: a schema in serde's JSON shape (metadata_schema.py), or None to switch validation off."""
        if self.destroyed:
            raise SessionError("Session already destroyed")
        if schema_json is None:
            self.schema = None
            return
        try:
            self.schema = MetadataSchema.from_json(schema_json)
        except ValueError as e:
            raise SessionError(f"Invalid schema format: {e}") from e

    setSchema = set_schema

    def vacuum(self):
        """session.rs:793-810: physically remove soft-deleted vectors; returns VacuumStats."""
        if self.destroyed:
            raise SessionError("Session already destroyed")
        try:
            return self.index.vacuum()
        except Exception as e:
            raise SessionError(f"Vacuum failed: {e}") from e

    # -- save_to_s5 / load_user_vectors: session.rs:636-697, :99-198 ----------------------------------
    def save_to_s5(self):
        """Chunked index under `session_id/` plus `metadata_map.cbor`; returns the path identifier."""
        from . import chunked
        if self.destroyed:
            raise SessionError("Session already destroyed")
        if self.storage is None:
            raise SessionError("Failed to save to S5: no storage configured")
        ids = {rid: VectorId(md["_originalId"]).bytes for rid, key in self._rows.items()
               for md in [self.metadata.get(key, {})] if isinstance(md.get("_originalId"), str)}
        try:
            chunked.save_index_chunked(self.index, self.storage, self.session_id, id_table=ids, now=self.now)
        except chunked.PersistenceError as e:
            raise SessionError(f"Failed to save index: {e}") from e
        self.storage.put(f"{self.session_id}/metadata_map.cbor", chunked.cbor_encode(self.metadata))
        if self.schema is not None:  # :680-692
            import json
            self.storage.put(f"{self.session_id}/schema.json", json.dumps(self.schema.to_json()).encode())
        return self.session_id

    saveToS5 = save_to_s5

    def load_user_vectors(self, cid, options=None):
        """Replace the index with the one saved under `cid/` (HybridConfig::default, :118) and the metadata map with
        `cid/metadata_map.cbor` (cleared when absent, :139-157)."""
        from . import chunked
        if self.destroyed:
            raise SessionError("Session already destroyed")
        if self.storage is None:
            raise SessionError("Failed to load from S5: no storage configured")
        try:
            index, table = chunked.load_index_chunked(self.ctx, self.storage, cid, now=self.now)
        except chunked.PersistenceError as e:
            if e.kind == "MissingComponent":
                raise SessionError(f"Missing index component: {e}") from e
            if e.kind == "IncompatibleVersion":
                raise SessionError(f"Incompatible index version: {e}") from e
            raise SessionError(f"Failed to load index: {e}") from e
        self.index = index
        self._rows = {rid: chunked.display_id(b) for rid, b in table.items()}
        data = self.storage.get(f"{cid}/metadata_map.cbor")
        if data is None:
            self.metadata = {}
        else:
            try:
                md = chunked.cbor_decode(data)
            except ValueError as e:
                raise SessionError(f"Failed to deserialize metadata: {e}") from e
            if not isinstance(md, dict):
                raise SessionError("Failed to deserialize metadata: expected a map")
            self.metadata = chunked.plain(md)
        raw = self.storage.get(f"{cid}/schema.json")  # :159-196: replaced by the saved one, or cleared
        if raw is None:
            self.schema = None
        else:
            import json
            try:
                self.schema = MetadataSchema.from_json(json.loads(raw.decode("utf-8")))
            except (ValueError, UnicodeDecodeError) as e:
                raise SessionError(f"Failed to deserialize schema: {e}") from e

    loadUserVectors = load_user_vectors

    def get_stats(self):
        # session.rs:699-722: vector_count is the ACTIVE count (soft-deleted rows excluded)
        if self.destroyed:
            raise SessionError("Session already destroyed")
        hnsw, ivf = self.index.hnsw(), self.index.ivf()
        hnsw_deleted = hnsw.node_count() - hnsw.active_count()
        ivf_deleted = ivf.total_vectors() - ivf.active_count()
        return {"vector_count": hnsw.active_count() + ivf.active_count(), "index_type": "hybrid",
                "hnsw_vector_count": self.index.recent_count(), "ivf_vector_count": self.index.historical_count(),
                "hnsw_deleted_count": hnsw_deleted, "ivf_deleted_count": ivf_deleted,
                "total_deleted_count": hnsw_deleted + ivf_deleted}

    def destroy(self):
        self.destroyed = True
        self.index = None


def matches_filter(metadata, flt):
    """bindings/node/src/session.rs:831-889: every filter field must be present (dotted paths descend) and equal —
    or, when the metadata value is an array, contain the filter value.  A non-object filter matches nothing, an
    empty one everything."""
    from .metadata_filter import json_eq
    if not isinstance(flt, dict):
        return False
    for key, want in flt.items():
        cur = metadata
        for part in (key.split(".") if "." in key else [key]):
            if isinstance(cur, dict) and part in cur:
                cur = cur[part]
            else:  # Value::get(&str) finds nothing in arrays and scalars
                return False
        if isinstance(cur, list):
            if not any(json_eq(x, want) for x in cur):
                return False
        elif not json_eq(cur, want):
            return False
    return True


def rest_search(index, request, id_of_row=None, now=0.0):
    """POST /api/v1/search (src/api/rest.rs:599-677) for one request dict {"vector": [...], "k": int,
    "options": {...}}.  Like the handler, the hybrid search runs with the default config (the handler builds a
    HybridSearchConfig but calls `hybrid_index.search(&vector, k)`, :631-634)."""
    import time
    vec = np.asarray(request["vector"], dtype=np.float32)
    if vec.size == 0:
        raise ValueError("Vector cannot be empty")  # validate_vector (:741-746) -> 400
    k = int(request["k"])
    opts = request.get("options") or {}
    t0 = time.perf_counter()
    res = index.search(vec.reshape(1, -1), k, now=now)
    ids, ds = res[0]
    results = []
    for rid, d in zip(ids.tolist(), ds.tolist()):
        score = float(np.float32(1.0) / (np.float32(1.0) + np.float32(d)))
        item = {"id": id_of_row(rid) if id_of_row else f"row_{rid}", "distance": float(np.float32(d)), "score": score}
        if opts.get("include_metadata", False):
            item["metadata"] = {}
        results.append(item)
    if opts.get("score_threshold") is not None:
        results = [r for r in results if r["score"] >= opts["score_threshold"]]
    sr, sh = opts.get("search_recent", True), opts.get("search_historical", True)
    return {"results": results, "search_time_ms": (time.perf_counter() - t0) * 1e3,
            "indices_searched": 2 if (sr and sh) else 1, "partial_results": False}


def rest_insert_vector(index, request, now=0.0, rows=None):
    """POST /api/v1/vectors (src/api/rest.rs:392-446) for {"id": str, "vector": [...], "metadata": any}: the vector
    goes in with timestamp = now, so it lands in the recent (HNSW) index.  Returns (201, body); an empty vector raises
    ValueError (400), an index error RuntimeError (500) with the handler's message.  `rows`: dict filled with
    row id -> request id, for rest_search's id_of_row."""
    from .chunked import format_timestamp
    vec = np.asarray(request["vector"], dtype=np.float32)
    if vec.size == 0:
        raise ValueError("Vector cannot be empty")
    vid = VectorId(request["id"])
    try:
        index.insert_with_timestamp(vid.row_id(), vec, now, now)
    except Exception as e:
        raise RuntimeError(f"Failed to add vector to index: {e}") from e
    if rows is not None:
        rows[vid.row_id()] = request["id"]
    # chrono's to_rfc3339() writes UTC as "+00:00" (its serde form, used on disk, writes "Z")
    return 201, {"id": request["id"], "index": "recent", "timestamp": format_timestamp(now)[:-1] + "+00:00"}


def rest_batch_insert(index, request, now=0.0, rows=None):
    """POST /api/v1/vectors/batch (src/api/rest.rs:449-531): per-vector outcome, never an error as a whole."""
    ok, errors = 0, []
    for v in request["vectors"]:
        vec = np.asarray(v["vector"], dtype=np.float32)
        if vec.size == 0:
            errors.append({"id": v["id"], "error": "Vector cannot be empty"})
            continue
        vid = VectorId(v["id"])
        try:
            index.insert_with_timestamp(vid.row_id(), vec, now, now)
        except Exception as e:  # noqa: BLE001
            errors.append({"id": v["id"], "error": f"Index error: {e}"})
            continue
        if rows is not None:
            rows[vid.row_id()] = v["id"]
        ok += 1
    return {"successful": ok, "failed": len(errors), "errors": errors}
