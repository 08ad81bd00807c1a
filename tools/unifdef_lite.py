#!/usr/bin/env python3
"""Resolve a fixed set of preprocessor conditionals in a source file and drop the dead branches (a minimal `unifdef`): used in round 6 to
take the ablation / A-B branches that no build of the library selects out of tmx_k_column.hip -- what was removed is kept as
tools/experiments/column_kernel_ablation_switches.patch.
usage: unifdef_lite.py file -DNAME[=VALUE] ... -UNAME ...     (writes the result to stdout)
Handled: #ifdef NAME, #ifndef NAME, #if NAME, #if NAME == N, #if !NAME, with #else / #endif; everything else passes through."""
import re, sys
src = open(sys.argv[1]).read().split("\n")
defs, undefs = {}, set()
for a in sys.argv[2:]:
    if a.startswith("-D"):
        k, _, v = a[2:].partition("="); defs[k] = int(v) if v else 1
    elif a.startswith("-U"):
        undefs.add(a[2:])
def known(expr):
    """value of the condition if it only involves known names, else None"""
    e = expr.strip()
    m = re.fullmatch(r"(!?)\s*(\w+)", e)
    if m and (m.group(2) in defs or m.group(2) in undefs):
        v = defs.get(m.group(2), 0); return (not v) if m.group(1) else bool(v)
    m = re.fullmatch(r"(\w+)\s*==\s*(\d+)", e)
    if m and (m.group(1) in defs or m.group(1) in undefs):
        return defs.get(m.group(1), 0) == int(m.group(2))
    return None
out, stack = [], []      # stack entries: [resolved (None = unknown, passes through), currently_emitting, in_else]
emitting = lambda: all(s[1] for s in stack)
for line in src:
    st = line.strip()
    m = re.match(r"#\s*(ifdef|ifndef|if|else|endif|elif)\b(.*)", st)
    if not m:
        if emitting(): out.append(line)
        continue
    kw, rest = m.group(1), m.group(2).split("//")[0].strip()
    if kw in ("ifdef", "ifndef", "if"):
        if kw == "if": val = known(rest)
        else:
            name = rest.split()[0] if rest else ""
            val = None if name not in defs and name not in undefs else ((name in defs) == (kw == "ifdef"))
        if val is None:
            if emitting(): out.append(line)
            stack.append([None, True, False])
        else:
            stack.append([val, val, False])
    elif kw == "elif":
        if stack[-1][0] is None:
            if emitting(): out.append(line)
        else: raise SystemExit("#elif on a resolved conditional is not handled: " + line)
    elif kw == "else":
        if stack[-1][0] is None:
            if emitting(): out.append(line)
        else: stack[-1][1] = not stack[-1][0]
    else:
        top = stack.pop()
        if top[0] is None and emitting(): out.append(line)
sys.stdout.write("\n".join(out))
