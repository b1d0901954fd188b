#!/usr/bin/env python3
"""Prints the Rust `unsafe extern "C"` block for every entry point of include/qdrant_amd.h (the block INTEGRATION.md §2 carries;
tests/test_integration_doc.py checks that the two stay equal in name, arity and types)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_integration_doc import _c_type  # noqa: E402

text = open(os.path.join(ROOT, "include", "qdrant_amd.h")).read()
text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
print('unsafe extern "C" {')
for m in re.finditer(r"QMX_API\s+([\w\s\*]+?)\b(qmx_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
    ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
    params = []
    if args and args != "void":
        for a in args.split(","):
            mm = re.match(r"(.*?)(\w+)$", a.strip())
            pname = mm.group(2)
            if pname in ("in", "type", "ref", "fn"):
                pname += "_"
            params.append("%s: %s" % (pname, _c_type(mm.group(1).strip())))
    line = "    fn %s(%s)" % (name, ", ".join(params))
    if ret != "void":
        line += " -> %s" % _c_type(ret)
    line += ";"
    # wrap at ~130 columns
    out, cur = [], ""
    for tok in re.split(r"(?<=,) ", line):
        if len(cur) + len(tok) > 128 and cur:
            out.append(cur.rstrip())
            cur = " " * (8 + len(name)) + tok + " "
        else:
            cur += tok + " "
    out.append(cur.rstrip())
    print("\n".join(out))
print("}")
