#!/usr/bin/env python3
"""INTEGRATION.md section 5's environment table, generated from the source: the `// ENV name | default | meaning` lines of
minarrow_amd/csrc/ma_env.hpp, cross-checked against the MINARROW_HIP_* names the product's sources actually read. `--check` only
compares (exit 1 when INTEGRATION.md is stale or a variable is undocumented / documented but unread); without it the table between
the `<!-- env-table:begin -->` / `<!-- env-table:end -->` markers is rewritten."""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "minarrow_amd" / "csrc"


def documented():
    rows = []
    for line in (CSRC / "ma_env.hpp").read_text().splitlines():
        m = re.match(r"// ENV (MINARROW_HIP_\w+) \| (.*?) \| (.*)$", line)
        if m:
            rows.append(m.groups())
    return rows


def read_by_sources():
    names = {}
    for path in sorted(list(CSRC.glob("*.hip")) + [p for p in CSRC.glob("*.hpp") if p.name != "ma_env.hpp"] + list((ROOT / "minarrow_amd").glob("*.py"))):
        for i, line in enumerate(path.read_text().splitlines(), 1):
            for name in re.findall(r"\"(MINARROW_HIP_[A-Z_0-9]+)\"", line):
                names.setdefault(name, f"{path.relative_to(ROOT)}:{i}")
    return names


def table():
    used = read_by_sources()
    out = ["| variable | default | meaning | read at |", "|---|---|---|---|"]
    for name, default, meaning in documented():
        out.append(f"| `{name}` | {default} | {meaning} | `{used.get(name, '?')}` |")
    return "\n".join(out)


def main():
    used, doc = read_by_sources(), [r[0] for r in documented()]
    problems = [f"{n} is read ({used[n]}) but has no ENV line in ma_env.hpp" for n in used if n not in doc]
    problems += [f"{n} has an ENV line but no source reads it" for n in doc if n not in used]
    text = (ROOT / "INTEGRATION.md").read_text()
    begin, end = "<!-- env-table:begin -->", "<!-- env-table:end -->"
    if begin not in text or end not in text:
        problems.append("INTEGRATION.md has no env-table markers")
        new = text
    else:
        new = text[:text.index(begin) + len(begin)] + "\n" + table() + "\n" + text[text.index(end):]
    if "--check" in sys.argv:
        if new != text:
            problems.append("INTEGRATION.md's environment table is stale: run tools/gen_env_table.py")
        for p in problems:
            print(p, file=sys.stderr)
        return 1 if problems else 0
    for p in problems:
        print("warning:", p, file=sys.stderr)
    (ROOT / "INTEGRATION.md").write_text(new)
    print(f"wrote the environment table: {len(doc)} variables")
    return 0


if __name__ == "__main__":
    sys.exit(main())
