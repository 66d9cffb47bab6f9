"""Metadata filter of the search surface (src/core/metadata_filter.rs:27-375) and the oversampled filtered search
(src/hybrid/core.rs:513-549).  Host-side, above the seam: the GPU path returns 3 k neighbours, this keeps the first k
whose metadata match.

Filter language (MongoDB-like JSON): {"field": value} equality (a metadata array matches if it CONTAINS the value),
{"field": {"$in": [...]}}, {"field": {"$gte"/"$gt"/"$lte"/"$lt": number}}, {"$and": [...]}, {"$or": [...]}, several
fields in one object = implicit AND, dotted paths for nested objects.  JSON equality is serde_json's: integers and
floats are different values (1 != 1.0), booleans are not numbers.
"""


class FilterError(ValueError):
    pass


class InvalidSyntax(FilterError):
    def __init__(self, msg):
        super().__init__(f"Invalid filter syntax: {msg}")


class UnsupportedOperator(FilterError):
    def __init__(self, op):
        super().__init__(f"Unsupported operator: {op}")


def json_eq(a, b):
    """serde_json::Value equality."""
    if isinstance(a, bool) or isinstance(b, bool):
        return isinstance(a, bool) and isinstance(b, bool) and a == b
    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
        return type(a) is type(b) and a == b  # PosInt/NegInt vs Float never compare equal
    if isinstance(a, list) and isinstance(b, list):
        return len(a) == len(b) and all(json_eq(x, y) for x, y in zip(a, b))
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(json_eq(a[k], b[k]) for k in a)
    return type(a) is type(b) and a == b


def as_f64(v):
    """serde_json Value::as_f64: numbers only."""
    if isinstance(v, bool) or not isinstance(v, (int, float)):
        return None
    return float(v)


_MISSING = object()


def get_field(metadata, path):
    """Dotted path lookup (:360-375); `_MISSING` when absent."""
    cur = metadata
    for part in path.split("."):
        if not isinstance(cur, dict) or part not in cur:
            return _MISSING
        cur = cur[part]
    return cur


class MetadataFilter:
    """kind in {"equals", "in", "range", "and", "or"} (:33-58)."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    # ---- parsing (:85-247) ----
    @staticmethod
    def from_json(value):
        if not isinstance(value, dict):
            raise InvalidSyntax("Filter must be a JSON object")
        if "$and" in value:
            return MetadataFilter._combinator("and", "$and", value["$and"])
        if "$or" in value:
            return MetadataFilter._combinator("or", "$or", value["$or"])
        for key in value:
            if key.startswith("$"):
                raise UnsupportedOperator(key)
        if len(value) == 1:
            (field, fv), = value.items()
            return MetadataFilter._field(field, fv)
        return MetadataFilter("and", filters=[MetadataFilter._field(f, v) for f, v in value.items()])

    @staticmethod
    def _combinator(kind, name, arr):
        if not isinstance(arr, list):
            raise InvalidSyntax(f"{name} must be an array")
        return MetadataFilter(kind, filters=[MetadataFilter.from_json(f) for f in arr])

    @staticmethod
    def _field(field, value):
        if not isinstance(value, dict):
            return MetadataFilter("equals", field=field, value=value)
        if "$in" in value:
            if not isinstance(value["$in"], list):
                raise InvalidSyntax("$in value must be an array")
            return MetadataFilter("in", field=field, values=value["$in"])
        gte, gt = as_f64(value.get("$gte")), as_f64(value.get("$gt"))
        lte, lt = as_f64(value.get("$lte")), as_f64(value.get("$lt"))
        if gte is not None and gt is not None:
            raise InvalidSyntax("Cannot use both $gte and $gt in the same range filter")
        if lte is not None and lt is not None:
            raise InvalidSyntax("Cannot use both $lte and $lt in the same range filter")
        lo, lo_inc = (gte, True) if gte is not None else ((gt, False) if gt is not None else (None, True))
        hi, hi_inc = (lte, True) if lte is not None else ((lt, False) if lt is not None else (None, True))
        if lo is not None or hi is not None:
            return MetadataFilter("range", field=field, min=lo, max=hi, min_inclusive=lo_inc, max_inclusive=hi_inc)
        for key in value:
            if key.startswith("$") and key not in ("$in", "$gte", "$gt", "$lte", "$lt"):
                raise UnsupportedOperator(key)
        if not value:
            raise InvalidSyntax(f"Empty object for field '{field}' - must specify a value or operator")
        return MetadataFilter("equals", field=field, value=value)  # a nested object compared as a whole

    # ---- evaluation (:265-338) ----
    def matches(self, metadata):
        k = self.kind
        if k == "equals":
            fv = get_field(metadata, self.field)
            if fv is _MISSING:
                return False
            if isinstance(fv, list):
                return any(json_eq(x, self.value) for x in fv)
            return json_eq(fv, self.value)
        if k == "in":
            fv = get_field(metadata, self.field)
            return fv is not _MISSING and any(json_eq(fv, v) for v in self.values)
        if k == "range":
            fv = get_field(metadata, self.field)
            num = None if fv is _MISSING else as_f64(fv)
            if num is None:
                return False
            lo_ok = self.min is None or (num >= self.min if self.min_inclusive else num > self.min)
            hi_ok = self.max is None or (num <= self.max if self.max_inclusive else num < self.max)
            return lo_ok and hi_ok
        if k == "and":
            return all(f.matches(metadata) for f in self.filters)  # empty AND matches everything
        if k == "or":
            return any(f.matches(metadata) for f in self.filters)  # empty OR matches nothing
        raise AssertionError(k)


def search_with_filter(search, k, flt, metadata_of):
    """src/hybrid/core.rs:513-549.  `search(k)` -> [(id, distance)] ascending; `metadata_of(id)` -> dict or None.
    No filter: plain search.  Else: 3 k candidates, keep those that have metadata and match, truncate to k."""
    if flt is None:
        return search(k)
    out = []
    for vid, dist in search(3 * k):
        md = metadata_of(vid)
        if md is not None and flt.matches(md):
            out.append((vid, dist))
    return out[:k]
