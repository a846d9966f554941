"""Kernel names as rocprofv3 prints them -> a short key, `function<template,arguments>`.

rocprofv3 reports demangled names such as
    void (anonymous namespace)::nerf_fwd_kernel<true, false>((anonymous namespace)::FwdArgs)
bench.py looks counters up by function name plus the LEADING template arguments it cares about, so a kernel
that gains a trailing template parameter keeps matching; summarize_pmc.py writes the key next to the full name.
"""
import re


def kernel_key(name: str) -> str:
    """`nerf_fwd_kernel<true,false>` for the example above; names without template arguments come back bare."""
    s = name.strip()
    if s.startswith("void "):
        s = s[5:]
    # cut the argument list: the first '(' at template depth 0 that is not the "(anonymous namespace)" qualifier
    s = s.replace("(anonymous namespace)::", "")
    depth, end = 0, len(s)
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            end = i
            break
    s = s[:end].strip()
    m = re.match(r"^([\w:]+?)(?:<(.*)>)?$", s)
    if not m:
        return s
    base = m.group(1).split("::")[-1]
    if m.group(2) is None:
        return base
    return base + "<" + ",".join(a.strip() for a in split_template_args(m.group(2))) + ">"


def split_template_args(args: str):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch in "<(":
            depth += 1
        elif ch in ">)":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def matches(name: str, function: str, leading_args=()) -> bool:
    """True when `name` (full or key form) is `function` with template arguments that START with leading_args."""
    key = kernel_key(name)
    base, _, rest = key.partition("<")
    if base != function:
        return False
    args = [a.strip() for a in split_template_args(rest[:-1])] if rest else []
    return list(args[:len(leading_args)]) == [str(a) for a in leading_args]
