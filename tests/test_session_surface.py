"""Session / REST surface above the seam (bindings/node/src/session.rs, src/api/rest.rs) and the
from-spec BLAKE3 behind VectorId (src/core/types.rs:9-43)."""
import numpy as np
import pytest

import fvdb_import

fv = fvdb_import.load()


def _inp(n):
    return bytes(i % 251 for i in range(n))


def test_blake3_known_answers():
    # official BLAKE3 test vectors (test_vectors.json: input byte i = i % 251), first 32 output bytes
    assert fv.blake3(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    assert fv.blake3(_inp(1)).hex() == "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213"
    assert fv.blake3(_inp(1024)).hex().startswith("42214739f095a406")   # exactly one chunk
    assert fv.blake3(_inp(1025)).hex().startswith("d00278ae47eb27b3")   # two chunks -> parent node
    assert fv.blake3(_inp(2049)).hex().startswith("5f4d72f40d7a5f82")   # three chunks, unbalanced tree


def test_vector_id_display_form():
    a, b = fv.VectorId("a"), fv.VectorId("b")
    assert a.to_string().startswith("vec_") and len(a.to_string()) == 12  # `vec_<8 hex>` src/core/types.rs:32-34
    assert a.to_string() != b.to_string() and a.row_id() != b.row_id()
    assert fv.VectorId("a").bytes == a.bytes


def test_js_array_narrowing_is_round_to_nearest_f32():
    # bindings/node/src/utils.rs:6-8 `v as f32`
    v = fv.session.js_array_to_vec_f32([0.1, 1.0 + 2.0**-30, 16777217.0])
    assert v.dtype == np.float32 and v[0] == np.float32(0.1) and v[1] == np.float32(1.0) and v[2] == np.float32(16777216.0)


@pytest.mark.gpu
def test_session_topk_and_scores():
    # bindings/node/test/test-topk-bug.js:25-83 (k=3/10/100 on 20 vectors -> 3/10/20 hits),
    # bindings/node/test/session.test.js:161-163 (score in [0,1]), _originalId round trip
    ctx = fv.Context(0)
    s = fv.VectorDbSession(ctx)
    s.add_vectors([{"id": f"doc-{i}", "vector": [float(i), 1.0, 0.5], "metadata": {"n": i}} for i in range(20)])
    assert s.get_stats()["vector_count"] == 20
    for k, want in ((3, 3), (10, 10), (100, 20)):
        assert len(s.search([0.0, 1.0, 0.5], k)) == want
    res = s.search([0.0, 1.0, 0.5], 5)
    assert res[0]["id"] == "doc-0" and res[0]["score"] == 1.0 and res[0]["metadata"] == {"n": 0}
    assert all(0.0 <= r["score"] <= 1.0 for r in res)
    assert [r["score"] for r in res] == sorted((r["score"] for r in res), reverse=True)
    d1 = np.float32(1.0)  # doc-1 is at distance 1 -> score 1/(1+1) in f32
    assert res[1]["score"] == float(np.float32(1.0) / (np.float32(1.0) + d1))
    assert len(s.search([0.0, 1.0, 0.5], 20, {"threshold": 0.3})) == 3  # scores 1, 1/2, 1/3 pass 0.3
    with pytest.raises(fv.session.SessionError):
        s.search([0.0, 1.0], 3)               # session.rs:216-223
    with pytest.raises(fv.session.SessionError):
        s.add_vectors([{"id": "bad", "vector": [1.0, 2.0]}])
    with pytest.raises(fv.session.SessionError):
        s.add_vectors([{"id": "doc-1", "vector": [9.0, 9.0, 9.0]}])   # duplicate id
    s.delete_vector("doc-0")
    assert s.search([0.0, 1.0, 0.5], 1)[0]["id"] == "doc-1"
    # REST shape (src/api/rest.rs:116-130, 650-676)
    r = fv.rest_search(s.index, {"vector": [2.0, 1.0, 0.5], "k": 3, "options": {"score_threshold": 0.4}})
    assert set(r) == {"results", "search_time_ms", "indices_searched", "partial_results"}
    assert r["indices_searched"] == 2 and r["partial_results"] is False
    assert [x["distance"] for x in r["results"]] == [0.0, 1.0, 1.0][:len(r["results"])]
    assert all(x["score"] >= 0.4 and set(x) == {"id", "distance", "score"} for x in r["results"])
    s.destroy()
    with pytest.raises(fv.session.SessionError):
        s.search([0.0, 1.0, 0.5], 1)


def test_session_simple_filter_semantics():
    # bindings/node/src/session.rs:831-889 (matches_filter / get_field_value / values_match)
    mf = fv.session.matches_filter
    md = {"category": "tech", "views": 10, "score": 1.5, "tags": ["ai", "ml"], "user": {"id": "123", "lvl": 2}, "n": None}
    assert mf(md, {}) and not mf(md, "tech") and not mf(md, None) and not mf(md, ["category"])
    assert mf(md, {"category": "tech"}) and not mf(md, {"category": "art"}) and not mf(md, {"missing": 1})
    assert mf(md, {"category": "tech", "views": 10}) and not mf(md, {"category": "tech", "views": 11})   # AND
    assert mf(md, {"user.id": "123"}) and not mf(md, {"user.id": 123}) and not mf(md, {"user.name": "x"})
    assert mf(md, {"tags": "ai"}) and not mf(md, {"tags": "db"}) and not mf(md, {"tags": ["ai", "ml"]})  # membership
    assert mf(md, {"views": 10}) and not mf(md, {"views": 10.0}) and mf(md, {"score": 1.5})             # Value equality
    assert mf(md, {"n": None}) and not mf(md, {"tags.0": "ai"}) and mf(md, {"user": {"id": "123", "lvl": 2}})


@pytest.mark.gpu
def test_session_delete_by_metadata_update_metadata_and_stats():
    # bindings/node/src/session.rs:447-632, :699-722; bindings/node/test (deleteByMetadata / updateMetadata suites)
    ctx = fv.Context(0)
    s = fv.VectorDbSession(ctx)
    s.add_vectors([{"id": f"doc-{i}", "vector": [float(i), 1.0, 0.5], "metadata": {"n": i, "tag": ["even"] if i % 2 == 0 else ["odd"],
                                                                                 "user": {"id": str(i % 3)}}} for i in range(12)])
    st = s.get_stats()
    assert st["vector_count"] == 12 and st["total_deleted_count"] == 0 and st["index_type"] == "hybrid"
    r = s.delete_by_metadata({"tag": "odd", "user.id": "1"})      # doc-1, doc-7
    assert r == {"deleted_count": 2, "deleted_ids": ["doc-1", "doc-7"]}
    assert s.delete_by_metadata({"tag": "odd", "user.id": "1"}) == {"deleted_count": 0, "deleted_ids": []}
    st = s.get_stats()
    assert st["vector_count"] == 10 and st["hnsw_deleted_count"] + st["ivf_deleted_count"] == st["total_deleted_count"] == 2
    assert st["hnsw_vector_count"] + st["ivf_vector_count"] == 12          # soft delete: the rows are still stored
    assert all(x["id"] not in ("doc-1", "doc-7") for x in s.search([1.0, 1.0, 0.5], 12))
    s.delete_vector("doc-0")
    assert "vec_" + fv.blake3(b"doc-0")[:4].hex() not in s.metadata and s.get_stats()["vector_count"] == 9
    s.update_metadata("doc-2", {"n": 200, "fresh": True})
    top = s.search([2.0, 1.0, 0.5], 1)[0]
    assert top["id"] == "doc-2" and top["metadata"] == {"n": 200, "fresh": True}
    s.update_metadata("doc-4", "just a string")
    assert s.search([4.0, 1.0, 0.5], 1)[0] == {"id": "doc-4", "score": 1.0, "metadata": "just a string"}
    with pytest.raises(fv.session.SessionError):
        s.update_metadata("doc-0", {"n": 1})    # deleted -> metadata gone -> "does not exist"
    with pytest.raises(fv.session.SessionError):
        s.update_metadata("never-added", {})
    v = s.vacuum()                                                 # session.rs:793-810
    assert v["total_removed"] == 3 and v["hnsw_removed"] + v["ivf_removed"] == 3
    st = s.get_stats()
    assert st["total_deleted_count"] == 0 and st["vector_count"] == 9 and s.search([2.0, 1.0, 0.5], 1)[0]["id"] == "doc-2"
    assert s.delete_by_metadata({}) ["deleted_count"] == 9        # an empty filter matches everything left
    assert s.get_stats()["vector_count"] == 0 and s.search([4.0, 1.0, 0.5], 3) == []


def test_metadata_schema():
    # src/core/schema.rs (its unit tests :227-293) and bindings/node/test/schema-validation.test.js
    from fabstir_vectordb_amd.metadata_schema import FieldType, MetadataSchema, SchemaError
    assert [FieldType.from_json(j).type_name() for j in ("String", "Number", "Boolean", {"Array": "String"}, {"Object": {}})] == \
        ["String", "Number", "Boolean", "Array<String>", "Object"]
    ok = lambda t, v: FieldType.from_json(t).validate_value("test", v)  # noqa: E731

    def bad(t, v):
        with pytest.raises(SchemaError) as e:
            ok(t, v)
        return str(e.value)

    ok("String", "hello"), ok("String", None), ok("Number", 123), ok("Number", 123.45), ok("Boolean", True), ok("Boolean", False)
    ok({"Array": "String"}, ["a", "b", None]), ok({"Object": {"name": "String"}}, {"name": "x", "extra": 1})
    assert bad("String", 123) == "Invalid type for field 'test': expected String, found Number"
    assert bad("Number", "123") == "Invalid type for field 'test': expected Number, found String"
    assert bad("Number", True).endswith("found Boolean") and bad("Boolean", "true").endswith("found String")
    assert bad({"Array": "String"}, ["a", 123, "c"]) == \
        "Invalid array element at index 1 in field 'test': expected String, found Number"
    assert bad({"Array": "String"}, "a") == "Invalid type for field 'test': expected Array<String>, found String"
    assert bad({"Object": {"name": "String"}}, {"name": 5}) == "Invalid type for field 'test.name': expected String, found Number"
    assert bad({"Array": {"Array": "Number"}}, [[1, "x"]]).startswith("Invalid array element at index 1 in field 'test[0]'")
    s = MetadataSchema.from_json({"fields": {"title": "String", "views": "Number", "published": "Boolean", "tags": {"Array": "String"}},
                                  "required": ["title", "views"]})
    s.validate({"title": "Valid Document", "views": 500, "published": True, "tags": ["valid", "test"]})
    s.validate({"title": "Minimal Document", "views": 50, "unknown": object()})  # optional / unknown fields are not checked
    for md, msg in (({"views": 1}, "Missing required field: title"), ({"title": "T", "views": "not-a-number"},
                    "expected Number, found String"), ([1], "Invalid type for field 'metadata': expected Object, found Array")):
        with pytest.raises(SchemaError) as e:
            s.validate(md)
        assert msg in str(e.value)
    assert MetadataSchema.from_json(s.to_json()).to_json() == s.to_json()
    for j in ({"fields": {"a": "string"}, "required": []}, {"fields": {}}, {"required": []}, {"fields": {"a": {"type": "string"}}, "required": []}):
        with pytest.raises(ValueError):
            MetadataSchema.from_json(j)


@pytest.mark.gpu
def test_rest_insert_and_batch_insert():
    # src/api/rest.rs:392-531 (insert_vector -> 201 {"id", "index": "recent", "timestamp"}; batch_insert counts per
    # vector), then /search finds them (:599-677)
    ctx = fv.Context(0)
    ix = fv.HybridIndex(ctx)
    ix.initialize([[float(i), 0.0, 1.0] for i in range(12)])
    rows, now = {}, 1764325230.5
    code, body = fv.rest_insert_vector(ix, {"id": "a", "vector": [1.0, 0.0, 1.0], "metadata": {}}, now=now, rows=rows)
    assert code == 201 and body == {"id": "a", "index": "recent", "timestamp": "2025-11-28T10:20:30.500+00:00"}
    with pytest.raises(ValueError, match="Vector cannot be empty"):
        fv.rest_insert_vector(ix, {"id": "e", "vector": []}, now=now)
    with pytest.raises(RuntimeError, match="Failed to add vector to index"):
        fv.rest_insert_vector(ix, {"id": "a", "vector": [1.0, 0.0, 1.0]}, now=now)     # duplicate
    r = fv.rest_batch_insert(ix, {"vectors": [{"id": "b", "vector": [2.0, 0.0, 1.0]}, {"id": "c", "vector": []},
                                              {"id": "a", "vector": [9.0, 9.0, 9.0]}, {"id": "d", "vector": [3.0, 0.0]},
                                              {"id": "f", "vector": [4.0, 0.0, 1.0]}]}, now=now, rows=rows)
    assert (r["successful"], r["failed"]) == (2, 3) and [e["id"] for e in r["errors"]] == ["c", "a", "d"]
    assert r["errors"][0]["error"] == "Vector cannot be empty" and r["errors"][1]["error"].startswith("Index error: ")
    assert ix.recent_count() == 3 and ix.historical_count() == 0
    out = fv.rest_search(ix, {"vector": [2.1, 0.0, 1.0], "k": 5}, id_of_row=rows.get, now=now)
    assert [x["id"] for x in out["results"]] == ["b", "a", "f"]
