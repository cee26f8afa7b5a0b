"""Static instruction counts of the kernels in a hipcc -save-temps .s file (whole kernel bodies: prologue + loop + epilogue), by
class.  usage: isa_count.py file.s [substring ...]   (test infrastructure; no GPU needed)"""
import re
import sys
from collections import Counter

src = open(sys.argv[1]).read()
want = sys.argv[2:]
for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\n\s*s_endpgm", src, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want and not any(w in name for w in want):
        continue
    c = Counter()
    for line in body.splitlines():
        t = line.strip()
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        op = t.split()[0]
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else \
            "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else "other"
        c[kind] += 1
        if op.endswith("_f64") or "_f64_" in op:
            c["f64"] += 1
        if op in ("s_waitcnt", "s_nop"):
            c["wait/nop"] += 1
    print(name[:70].ljust(72), dict(c))
