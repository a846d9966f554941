#!/usr/bin/env python3
"""Extract every call the reference's scripts make into its `nerf` package.

Run in the build container (the reference is only read as text, never imported):

    python tests/golden/make_callsites.py /root/reference tests/golden/callsites.json

For train.py and eval.py it collects the `from nerf.<module> import <name>` statements and every
call whose callee is one of those names, and records for each call the number of positional
arguments and the keyword names.  tests/test_callsites.py binds each record against this
package's function of the same name, so a signature that the reference's scripts could not call
fails on the CPU.
"""
import ast
import json
import os
import sys


def callsites(path: str):
    with open(path, "r") as handle:
        tree = ast.parse(handle.read(), filename=path)
    origin = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "nerf":
            for alias in node.names:
                origin[alias.asname or alias.name] = (node.module, alias.name)
    imports = sorted({f"{m}.{n}" for m, n in origin.values()})
    calls = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in origin:
            if any(isinstance(a, ast.Starred) for a in node.args) or any(k.arg is None for k in node.keywords):
                raise SystemExit(f"{path}:{node.lineno}: starred call, cannot pin its signature")
            module, name = origin[node.func.id]
            calls.append({"file": os.path.basename(path), "line": node.lineno, "module": module, "name": name,
                          "positional": len(node.args), "keywords": [k.arg for k in node.keywords]})
    calls.sort(key=lambda c: (c["file"], c["line"]))
    return imports, calls


def main():
    ref, out = sys.argv[1], sys.argv[2]
    result = {"source": "AST of train.py and eval.py of josedelrey/nerf-pytorch", "imports": {}, "calls": []}
    for script in ("train.py", "eval.py"):
        imports, calls = callsites(os.path.join(ref, script))
        result["imports"][script] = imports
        result["calls"] += calls
    with open(out, "w") as handle:
        json.dump(result, handle, indent=1)
        handle.write("\n")
    print(f"{len(result['calls'])} call sites -> {out}")


if __name__ == "__main__":
    main()
