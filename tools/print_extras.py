#!/usr/bin/env python
"""Headline value and the `extra.*` legs of bench.py lines: `python tools/print_extras.py a.json b.json ...`"""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    e = d.get("extra", {})
    print(f, d["value"], d["ms_per_step"], {k: v.get("value") for k, v in e.items() if isinstance(v, dict) and "value" in v})
