#!/usr/bin/env python3
"""tests/golden/json_schema.json = the reference's own result schema (/root/reference/tests/json_schema.json, the file
its tests/test_transcribe.py:287-296 validates every JSON output against), taken over as a fixture so that the GPU
tests -- which run where /root/reference does not exist -- can hold transcribe()'s dictionary against it.
Build container only."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/tests/json_schema.json"

if __name__ == "__main__":
    schema = json.load(open(SRC))
    with open(os.path.join(HERE, "json_schema.json"), "w") as f:
        json.dump(schema, f, indent=1, sort_keys=True)
    print("wrote json_schema.json:", len(schema["properties"]), "top-level properties")
