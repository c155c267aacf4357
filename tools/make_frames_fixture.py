#!/usr/bin/env python3
"""Turn the reference's tools/frames.js (a DATA file: output of demod_2000's JavaScript
debug dumper, ref: src/demod-2000.c:380-414) into tests/golden/frames_2000.json.

Each record: 245 magnitudes m[j-5 .. j+16+224-1], the 14 message bytes the reference
demodulated from them (phase-corrected retry), the message length in bits and the
description.  Run in the build container only (needs /root/reference)."""
import json, re, sys
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tools/frames.js"
out = sys.argv[2] if len(sys.argv) > 2 else "tests/golden/frames_2000.json"
recs = []
for line in open(src):
    m = re.match(r'frames\.push\(\{"descr": "([^"]*)", "mag": \[([0-9,]*)\], "fix1": (-?\d+), "fix2": (-?\d+), '
                 r'"bits": (\d+), "hex": "([^"]*)"\}\);', line.strip())
    if not m:
        continue
    hexs = "".join(re.findall(r'\\x([0-9a-f]{2})', m.group(6)))
    recs.append({"descr": m.group(1), "mag": [int(x) for x in m.group(2).split(",")],
                 "fix": [int(m.group(3)), int(m.group(4))], "bits": int(m.group(5)), "hex": hexs})
json.dump(recs, open(out, "w"), separators=(",", ":"))
print(len(recs), "frames ->", out)
