#!/usr/bin/env python3
"""Counts the tests by marker (`pytest --collect-only`) and writes them where the docs quote them: between the markers
<!-- test-counts --> ... <!-- /test-counts --> of README.md and DESIGN.md.  Run it when tests were added; nothing else edits those numbers."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def count(marker):
    out = subprocess.run([sys.executable, "-m", "pytest", "tests", "--collect-only", "-q", "-m", marker], cwd=ROOT, capture_output=True, text=True).stdout
    m = re.search(r"(\d+)(?:/\d+)? tests? collected|(\d+) selected", out)
    if m:
        return int(m.group(1) or m.group(2))
    m = re.search(r"(\d+)/(\d+) tests collected", out)
    return int(m.group(1)) if m else sum(1 for line in out.splitlines() if "::" in line)


def main():
    cpu, gpu = count("not gpu"), count("gpu")
    text = "%d CPU tests (`pytest tests -m 'not gpu'`), %d GPU tests (`pytest tests -m gpu`)" % (cpu, gpu)
    for name in ("README.md", "DESIGN.md"):
        path = os.path.join(ROOT, name)
        s = open(path).read()
        new, n = re.subn(r"<!-- test-counts -->.*?<!-- /test-counts -->", "<!-- test-counts -->%s<!-- /test-counts -->" % text, s, flags=re.S)
        if n and new != s:
            open(path, "w").write(new)
        print(name, "updated" if n and new != s else ("unchanged" if n else "no marker"))
    print(text)


if __name__ == "__main__":
    main()
