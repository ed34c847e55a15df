#!/usr/bin/env python3
"""Every BBGPU_* / BB_* environment variable the product reads, from the sources: one markdown table (variable, where it is read, what the
source line says about it) written between the markers of INTEGRATION.md.  `--check` exits non-zero when the committed table is stale
(tests/test_host_boundary.py runs it), so the table cannot drift from the code.

    python tools/gen_env_table.py          # rewrite the table in INTEGRATION.md
    python tools/gen_env_table.py --check"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["barretenberg_amd/csrc", "barretenberg_amd/shim", "barretenberg_amd/bbgpu.py", "barretenberg_amd/sharding.py", "barretenberg_amd/plonk.py", "bench.py"]
BEGIN, END = "<!-- env-table:begin (tools/gen_env_table.py) -->", "<!-- env-table:end -->"
PAT = re.compile(r'(?:getenv|environ\.get|environ\.setdefault|setenv)\(\s*"((?:BBGPU|BB|GPU_MAX)_[A-Z0-9_]+)"')


def scan():
    found = {}
    files = []
    for s in SOURCES:
        p = os.path.join(ROOT, s)
        if os.path.isdir(p):
            files += [os.path.join(p, f) for f in sorted(os.listdir(p)) if f.endswith((".hip", ".hpp", ".h", ".cpp", ".py"))]
        else:
            files.append(p)
    for f in files:
        for ln, text in enumerate(open(f, errors="replace"), 1):
            for m in PAT.finditer(text):
                c = text.split("//", 1)[1].strip() if "//" in text else (text.split("#", 1)[1].strip() if f.endswith(".py") and "#" in text else "")
                c = re.sub(r"\s+", " ", c).replace("|", "/")
                found.setdefault(m.group(1), []).append((os.path.relpath(f, ROOT), c[:160]))  # file only: line numbers would make the table stale with every edit
    return found


def table():
    rows = ["| variable | read at | what the source says |", "|---|---|---|"]
    for name, uses in sorted(scan().items()):
        where = ", ".join("`%s`" % f for f in sorted({u[0] for u in uses}))
        note = next((u[1] for u in uses if u[1]), "")
        rows.append("| `%s` | %s | %s |" % (name, where, note))
    return "\n".join(rows)


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    doc = open(path).read()
    if BEGIN not in doc or END not in doc:
        sys.exit("INTEGRATION.md lacks the env-table markers")
    head, rest = doc.split(BEGIN, 1)
    _, tail = rest.split(END, 1)
    new = head + BEGIN + "\n" + table() + "\n" + END + tail
    if "--check" in sys.argv:
        if new != doc:
            sys.exit("INTEGRATION.md: the environment-variable table is stale -- run python tools/gen_env_table.py")
        return
    open(path, "w").write(new)
    print("wrote the table of %d variables into INTEGRATION.md" % len(scan()))


if __name__ == "__main__":
    main()
