#!/usr/bin/env python3
"""embed_src.py IN OUT -- writes OUT as C++ raw string literals holding IN with its local `#include "x"` lines expanded
(recursively, each file once), so that hipRTC can compile the text without a header search path."""
import os
import re
import sys


def expand(path, seen):
    out = []
    base = os.path.dirname(path)
    for line in open(path):
        m = re.match(r'\s*#include\s+"([^"]+)"', line)
        if m:
            inc = os.path.join(base, m.group(1))
            if os.path.exists(inc):
                key = os.path.realpath(inc)
                if key.endswith(".inc") or key not in seen:
                    seen.add(key)
                    out.extend(expand(inc, seen))
                continue
        out.append(line)
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    text = "".join(expand(src, {os.path.realpath(src)}))
    assert ')L64SRC"' not in text
    with open(dst, "w") as f:
        # MSVC-style limits do not apply here, but keep the literals moderate: split every ~8 KB at a line end
        chunk, size = [], 0
        for line in text.splitlines(keepends=True):
            chunk.append(line)
            size += len(line)
            if size > 8000:
                f.write('R"L64SRC(' + "".join(chunk) + ')L64SRC"\n')
                chunk, size = [], 0
        f.write('R"L64SRC(' + "".join(chunk) + ')L64SRC"\n')


if __name__ == "__main__":
    main()
