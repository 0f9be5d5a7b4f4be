"""Helper of tests/test_abi.py (run as a child process so that a crash is a test failure, not the end of the test run): every
`int zk_*(zk_ctx* | zk_srs* | zk_transcript*, ...)` of the C ABI called with a null first argument and zeros everywhere else.
Prints {name: return code} as JSON.  No GPU is touched: a null handle must be refused before anything else happens."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_plonk_amd import _lib  # noqa: E402

L = _lib.lib()
res = {}
for name, (rt, args) in sorted(_lib.SYMBOLS.items()):
    if rt is not ctypes.c_int or not args or args[0] is not ctypes.c_void_p or name == "zk_ctx_create":
        continue
    vals = [0 if a in (ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int64, ctypes.c_size_t, ctypes.c_uint8) else None for a in args]
    print(name, file=sys.stderr, flush=True)            # the last name on stderr is the one that crashed
    res[name] = getattr(L, name)(*vals)
print(json.dumps(res))
