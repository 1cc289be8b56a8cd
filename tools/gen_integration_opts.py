#!/usr/bin/env python
"""Regenerate the ctypes mirror of cvxpnpl_opts_t in INTEGRATION.md (between the opts:begin / opts:end markers) from
cvxpnpl_amd/_lib.py, so that the document, the binding and the header move in lock-step (tests/test_capi_exports.py checks
all three against each other).  python tools/gen_integration_opts.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvxpnpl_amd import _lib  # noqa: E402

rows, line = [], "    _fields_ = ["
for name, typ in _lib.Opts._fields_:
    item = f'("{name}", C.{typ.__name__}), '
    if len(line) + len(item) > 118:
        rows.append(line.rstrip())
        line = "                "
    line += item
rows.append(line.rstrip().rstrip(",") + "]")
block = ("<!-- opts:begin -->\n```python\nclass Opts(C.Structure):            # cvxpnpl_opts_t -- generated from cvxpnpl_amd/_lib.py (tools/gen_integration_opts.py)\n"
         + "\n".join(rows) + "\n\n"
         "_L.cvxpnpl_default_opts.argtypes = [C.POINTER(Opts)]   # ALWAYS initialise with cvxpnpl_default_opts: it fills in struct_size,\n"
         "_L.cvxpnpl_opts_size.restype = C.c_size_t              # and the library refuses a block of any other size (return -1)\n"
         "assert _L.cvxpnpl_opts_size() == C.sizeof(Opts), \"this mirror is out of date with include/cvxpnpl_amd.h\"\n```\n")
p = os.path.join(ROOT, "INTEGRATION.md")
s = open(p).read()
a, b = s.index("<!-- opts:begin -->"), s.index("<!-- opts:end -->")
open(p, "w").write(s[:a] + block + s[b:])
print("INTEGRATION.md: Opts block regenerated,", len(_lib.Opts._fields_), "fields")
