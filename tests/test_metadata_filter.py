"""Metadata filter + oversampled filtered search of the session surface.

CPU cases restate the reference's own unit tests (src/core/metadata_filter.rs:380-616) and the behaviours its
integration tests pin (tests/integration/search_filter_tests.rs); the GPU case runs a filtered session search.
"""
import numpy as np
import pytest

import fvdb_import

fv = fvdb_import.load()
mf = fv.metadata_filter
F = mf.MetadataFilter


def test_equals_string_and_number():  # :381-421
    f = F("equals", field="category", value="technology")
    assert f.matches({"category": "technology", "views": 1000})
    assert not f.matches({"category": "sports", "views": 1000})
    f = F("equals", field="version", value=1)
    assert f.matches({"version": 1, "name": "test"})
    assert not f.matches({"version": 2, "name": "test"})
    assert not f.matches({"version": 1.0})  # serde_json: 1 != 1.0
    assert not F("equals", field="flag", value=True).matches({"flag": 1})


def test_in_and_range():  # :424-454
    f = F("in", field="status", values=["active", "pending"])
    assert f.matches({"status": "active"}) and not f.matches({"status": "archived"})
    f = F("range", field="age", min=18.0, max=65.0, min_inclusive=True, max_inclusive=True)
    assert f.matches({"age": 25}) and not f.matches({"age": 17}) and f.matches({"age": 18}) and f.matches({"age": 65})
    assert not f.matches({"age": "25"}) and not f.matches({"name": "x"})
    g = F.from_json({"score": {"$gt": 40, "$lt": 100}})
    assert g.matches({"score": 41}) and not g.matches({"score": 40}) and not g.matches({"score": 100})


def test_combinators_nested_and_arrays():  # :456-535
    f = F("and", filters=[F("equals", field="category", value="technology"), F("equals", field="published", value=True)])
    assert f.matches({"category": "technology", "published": True})
    assert not f.matches({"category": "technology", "published": False})
    f = F("or", filters=[F("equals", field="status", value="urgent"),
                         F("range", field="priority", min=8.0, max=None, min_inclusive=True, max_inclusive=True)])
    assert f.matches({"status": "urgent", "priority": 5}) and f.matches({"status": "normal", "priority": 9})
    assert not f.matches({"status": "normal", "priority": 3})
    assert F("and", filters=[]).matches({}) and not F("or", filters=[]).matches({})
    assert F("equals", field="user.id", value="123").matches({"user": {"id": "123", "name": "Alice"}})
    assert F("equals", field="tags", value="ai").matches({"tags": ["ai", "ml", "nlp"]})
    assert not F("equals", field="tags", value="db").matches({"tags": ["ai", "ml"]})


def test_from_json():  # :538-615
    assert F.from_json({"category": "technology"}).matches({"category": "technology"})
    assert F.from_json({"status": {"$in": ["active", "pending"]}}).matches({"status": "active"})
    assert F.from_json({"age": {"$gte": 18, "$lte": 65}}).matches({"age": 25})
    f = F.from_json({"$and": [{"category": "technology"}, {"published": True}]})
    assert f.matches({"category": "technology", "published": True})
    f = F.from_json({"category": "tech", "year": 2024})  # several fields: implicit AND
    assert f.kind == "and" and f.matches({"category": "tech", "year": 2024}) and not f.matches({"category": "tech", "year": 2023})
    with pytest.raises(mf.UnsupportedOperator):
        F.from_json({"$invalid": "test"})
    with pytest.raises(mf.UnsupportedOperator):
        F.from_json({"age": {"$regex": "x"}})
    with pytest.raises(mf.InvalidSyntax):
        F.from_json({"age": {"$gte": 1, "$gt": 2}})
    with pytest.raises(mf.InvalidSyntax):
        F.from_json({"age": {}})
    with pytest.raises(mf.InvalidSyntax):
        F.from_json({"$and": {"a": 1}})
    with pytest.raises(mf.InvalidSyntax):
        F.from_json(["not", "an", "object"])
    assert mf.get_field({"user": {"id": "123", "profile": {"email": "test@example.com"}}}, "user.profile.email") == "test@example.com"
    assert mf.get_field({"user": {"id": "123"}}, "user.missing") is mf._MISSING


def test_search_with_filter_oversamples_three_k_and_truncates():  # src/hybrid/core.rs:513-549
    ranked = [(i, float(i)) for i in range(100)]
    md = {i: {"even": i % 2 == 0, "n": i} for i in range(100) if i != 4}  # id 4 has no metadata: dropped
    asked = []

    def search(k):
        asked.append(k)
        return ranked[:k]

    f = F.from_json({"even": True})
    got = mf.search_with_filter(search, 5, f, md.get)
    assert asked == [15] and [g[0] for g in got] == [0, 2, 6, 8, 10]
    # fewer than k matches among the 3k candidates: a short list (the reference does not search deeper)
    got = mf.search_with_filter(search, 5, F.from_json({"n": {"$gte": 13}}), md.get)
    assert [g[0] for g in got] == [13, 14]
    assert mf.search_with_filter(search, 5, None, md.get) == ranked[:5] and asked[-1] == 5


@pytest.mark.gpu
def test_session_filtered_search():
    ctx = fv.Context(0)
    try:
        s = fv.session.VectorDbSession(ctx)
        rng = np.random.default_rng(5)
        base = rng.standard_normal((60, 16))
        s.add_vectors([{"id": f"doc{i}", "vector": base[i].tolist(),
                        "metadata": {"category": "tech" if i % 3 == 0 else "other", "rank": i}} for i in range(60)])
        q = base[9].tolist()
        plain = s.search(q, 5)
        assert plain[0]["id"] == "doc9"
        got = s.search(q, 5, {"filter": {"category": "tech"}})
        assert got and got[0]["id"] == "doc9" and all(r["metadata"]["category"] == "tech" for r in got)
        # the filtered list is the unfiltered 3k list, filtered, truncated
        wide = s.search(q, 15)
        want = [r["id"] for r in wide if r["metadata"]["category"] == "tech"][:5]
        assert [r["id"] for r in got] == want
        got = s.search(q, 5, {"filter": {"$and": [{"category": "tech"}, {"rank": {"$gte": 30}}]}})
        assert all(r["metadata"]["rank"] >= 30 for r in got)
        with pytest.raises(fv.session.SessionError, match="Invalid filter"):
            s.search(q, 5, {"filter": {"$nope": 1}})
    finally:
        ctx.close()
