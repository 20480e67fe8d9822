#!/usr/bin/env python3
"""Remove the preprocessor blocks of macros that are never defined any more (`#ifdef M` / `#ifndef M` / `#if defined(M)` with #else and
#endif, nested) from a source file, keeping the text of the branch that is live when M is undefined.
    tools/unifdef_lite.py FILE MACRO_PREFIX [MACRO_PREFIX ...]     (rewrites FILE in place)
Used in round 5 to strip the *_EXP_* timing ablations whose conclusions are recorded in DESIGN.md."""
import re
import sys


def strip(lines, prefixes):
    out, stack = [], []            # stack entries: None (foreign conditional) or [keeping: bool]
    def dead(name):
        return any(name.startswith(p) for p in prefixes)
    def emitting():
        return all(e is None or e[0] for e in stack)
    for ln in lines:
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|else|elif|endif)\b\s*(.*)", ln)
        if not m:
            if emitting(): out.append(ln)
            continue
        kind, rest = m.group(1), m.group(2).strip()
        if kind in ("ifdef", "ifndef"):
            name = rest.split()[0] if rest else ""
            if dead(name):
                stack.append([kind == "ifndef"])
                continue
            if emitting(): out.append(ln)
            stack.append(None)
        elif kind == "if":
            md = re.fullmatch(r"defined\s*\(?\s*(\w+)\s*\)?", rest)
            if md and dead(md.group(1)):
                stack.append([False])
                continue
            if emitting(): out.append(ln)
            stack.append(None)
        elif kind in ("else", "elif"):
            if stack and stack[-1] is not None:
                assert kind == "else", "elif on a stripped macro"
                stack[-1][0] = not stack[-1][0]
                continue
            if emitting(): out.append(ln)
        else:
            e = stack.pop()
            if e is not None: continue
            if emitting(): out.append(ln)
    assert not stack
    return out


if __name__ == "__main__":
    path, prefixes = sys.argv[1], sys.argv[2:]
    src = open(path).read().splitlines(keepends=True)
    open(path, "w").writelines(strip(src, prefixes))
