"""Optional metadata schema of a session (src/core/schema.rs), in serde's JSON shape:

    {"fields": {"title": "String", "views": "Number", "ok": "Boolean", "tags": {"Array": "String"},
                "author": {"Object": {"name": "String"}}}, "required": ["title", "views"]}

(bindings/node/test/schema-validation.test.js:52-61).  null passes every type; error texts are the reference's.
"""


class SchemaError(Exception):
    """src/core/schema.rs:13-33; `.kind` in {"MissingField", "InvalidType", "InvalidArrayElement"}."""

    def __init__(self, kind, msg, expected=None, found=None):
        super().__init__(msg)
        self.kind, self.expected, self.found = kind, expected, found


def _invalid_type(field, expected, found):
    return SchemaError("InvalidType", f"Invalid type for field '{field}': expected {expected}, found {found}", expected, found)


def _type_name_of_value(v):  # get_value_type_name (:216-225)
    if v is None:
        return "Null"
    if isinstance(v, bool):
        return "Boolean"
    if isinstance(v, (int, float)):
        return "Number"
    if isinstance(v, str):
        return "String"
    if isinstance(v, list):
        return "Array"
    return "Object"


class FieldType:
    """String | Number | Boolean | Array(FieldType) | Object({name: FieldType}) (:36-47)."""

    def __init__(self, kind, inner=None):
        self.kind, self.inner = kind, inner

    @staticmethod
    def from_json(j):
        if isinstance(j, str) and j in ("String", "Number", "Boolean"):
            return FieldType(j)
        if isinstance(j, dict) and len(j) == 1:
            (k, v), = j.items()
            if k == "Array":
                return FieldType("Array", FieldType.from_json(v))
            if k == "Object" and isinstance(v, dict):
                return FieldType("Object", {name: FieldType.from_json(t) for name, t in v.items()})
        raise ValueError(f"unknown field type {j!r}")

    def to_json(self):
        if self.kind == "Array":
            return {"Array": self.inner.to_json()}
        if self.kind == "Object":
            return {"Object": {k: t.to_json() for k, t in self.inner.items()}}
        return self.kind

    def type_name(self):  # :50-58
        return f"Array<{self.inner.type_name()}>" if self.kind == "Array" else self.kind

    def validate_value(self, field, value):  # :61-140
        if value is None:
            return
        bad = _invalid_type(field, self.type_name(), _type_name_of_value(value))
        if self.kind == "String":
            if not isinstance(value, str):
                raise bad
        elif self.kind == "Number":
            if isinstance(value, bool) or not isinstance(value, (int, float)):
                raise bad
        elif self.kind == "Boolean":
            if not isinstance(value, bool):
                raise bad
        elif self.kind == "Array":
            if not isinstance(value, list):
                raise bad
            for i, el in enumerate(value):
                if el is None:
                    continue
                try:
                    self.inner.validate_value(f"{field}[{i}]", el)
                except SchemaError as e:
                    if e.kind != "InvalidType":
                        raise
                    raise SchemaError("InvalidArrayElement", f"Invalid array element at index {i} in field '{field}': "
                                      f"expected {e.expected}, found {e.found}", e.expected, e.found) from e
        else:
            if not isinstance(value, dict):
                raise bad
            for key, t in self.inner.items():
                if key in value:
                    t.validate_value(f"{field}.{key}", value[key])


class MetadataSchema:
    def __init__(self, fields=None, required=None):
        self.fields, self.required = dict(fields or {}), list(required or [])

    @staticmethod
    def from_json(j):
        """serde_json::from_value::<MetadataSchema> (:143-149): both members are required."""
        if not isinstance(j, dict) or not isinstance(j.get("fields"), dict) or not isinstance(j.get("required"), list) \
                or not all(isinstance(r, str) for r in j["required"]):
            raise ValueError("expected {\"fields\": {...}, \"required\": [...]}")
        return MetadataSchema({k: FieldType.from_json(v) for k, v in j["fields"].items()}, list(dict.fromkeys(j["required"])))

    def to_json(self):
        return {"fields": {k: t.to_json() for k, t in self.fields.items()}, "required": list(self.required)}

    def validate(self, metadata):  # :170-197
        if not isinstance(metadata, dict):
            raise _invalid_type("metadata", "Object", _type_name_of_value(metadata))
        for r in self.required:
            if r not in metadata:
                raise SchemaError("MissingField", f"Missing required field: {r}")
        for name, t in self.fields.items():
            if name in metadata:
                t.validate_value(name, metadata[name])
