"""A small JSON-schema validator (the subset the reference's tests/json_schema.json uses: type, properties, items,
required, minimum, maximum, minItems, uniqueItems, additionalProperties) -- `jsonschema`, which the reference's
tests/test_transcribe.py:287-296 calls, is not installed in this image.  TEST INFRASTRUCTURE."""
import json

_TYPES = {
    "object": lambda v: isinstance(v, dict),
    "array": lambda v: isinstance(v, list),
    "string": lambda v: isinstance(v, str),
    "boolean": lambda v: isinstance(v, bool),
    "integer": lambda v: isinstance(v, int) and not isinstance(v, bool),
    "number": lambda v: isinstance(v, (int, float)) and not isinstance(v, bool),
    "null": lambda v: v is None,
}
_KNOWN = {"type", "properties", "items", "required", "minimum", "maximum", "minItems", "uniqueItems",
          "additionalProperties", "description", "title", "$schema"}


class SchemaError(AssertionError):
    pass


def validate(instance, schema, path="$"):
    unknown = set(schema) - _KNOWN
    if unknown:
        raise SchemaError(f"{path}: schema keywords this validator does not implement: {sorted(unknown)}")
    t = schema.get("type")
    if t is not None:
        kinds = t if isinstance(t, list) else [t]
        if not any(_TYPES[k](instance) for k in kinds):
            raise SchemaError(f"{path}: {instance!r} is not of type {t}")
    if isinstance(instance, (int, float)) and not isinstance(instance, bool):
        if "minimum" in schema and instance < schema["minimum"]:
            raise SchemaError(f"{path}: {instance} < minimum {schema['minimum']}")
        if "maximum" in schema and instance > schema["maximum"]:
            raise SchemaError(f"{path}: {instance} > maximum {schema['maximum']}")
    if isinstance(instance, dict):
        for key in schema.get("required", []):
            if key not in instance:
                raise SchemaError(f"{path}: missing required property {key!r}")
        props = schema.get("properties", {})
        for key, value in instance.items():
            if key in props:
                validate(value, props[key], f"{path}.{key}")
            elif schema.get("additionalProperties") is False:
                raise SchemaError(f"{path}: unexpected property {key!r}")
            elif isinstance(schema.get("additionalProperties"), dict):
                validate(value, schema["additionalProperties"], f"{path}.{key}")
    if isinstance(instance, list):
        if "minItems" in schema and len(instance) < schema["minItems"]:
            raise SchemaError(f"{path}: {len(instance)} items < minItems {schema['minItems']}")
        if schema.get("uniqueItems"):
            seen = [json.dumps(x, sort_keys=True, default=float) for x in instance]
            if len(set(seen)) != len(seen):
                raise SchemaError(f"{path}: items are not unique")
        if isinstance(schema.get("items"), dict):
            for k, item in enumerate(instance):
                validate(item, schema["items"], f"{path}[{k}]")
    return True
