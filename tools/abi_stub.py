#!/usr/bin/env python3
"""Generate the ctypes stub of include/uspace_hip.h's structs (the block INTEGRATION.md shows between its
`abi-stub` markers).  `python tools/abi_stub.py` prints it; `--write` refreshes INTEGRATION.md in place.
tests/test_host_logic.py checks that this output, INTEGRATION.md and uspace_amd/_hip.py agree with the header."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "uspace_hip.h")
BEGIN, END = "<!-- abi-stub:begin -->", "<!-- abi-stub:end -->"

_CTYPES = {"int": "ctypes.c_int", "float": "ctypes.c_float", "long": "ctypes.c_long", "size_t": "ctypes.c_size_t",
           "double": "ctypes.c_double"}


def parse_structs(text=None):
    """[(struct name, [(field name, ctypes expression)])] in header order."""
    if text is None:
        text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"^(?:const\s+)?(\w+)\s*(\*?)\s*(?:const\s+)?(\w+)\s*(?:\[(\d+)\])?$", decl)
            if not fm:
                raise ValueError(f"cannot parse field {decl!r} of {name}")
            ctype, star, fname, arr = fm.groups()
            if star:
                expr = "ctypes.c_void_p"
            else:
                expr = _CTYPES[ctype]
                if arr:
                    expr = f"{expr} * {arr}"
            fields.append((fname, expr))
        out.append((name, fields))
    return out


def stub():
    lines = ["import ctypes", ""]
    for name, fields in parse_structs():
        lines.append(f"class {name}(ctypes.Structure):")
        lines.append("    _fields_ = [")
        for fname, expr in fields:
            lines.append(f'        ("{fname}", {expr}),')
        lines.append("    ]")
        lines.append("")
    return "\n".join(lines).rstrip() + "\n"


def main():
    text = stub()
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "INTEGRATION.md")
        s = open(p).read()
        a, b = s.index(BEGIN), s.index(END)
        s = s[:a] + BEGIN + "\n```python\n" + text + "```\n" + s[b:]
        open(p, "w").write(s)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
