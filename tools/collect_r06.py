#!/usr/bin/env python
"""`python tools/collect_r06.py install r06`: copies gpurun_out/<tag>/* into profiles/ as <tag>_<name> (what bench.py's evidence lookups
and the judge read); raw logs stay behind."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", tag)
    n = 0
    for f in sorted(os.listdir(src)):
        if f.endswith((".err", ".log")) or os.path.isdir(os.path.join(src, f)):
            continue
        shutil.copyfile(os.path.join(src, f), os.path.join(ROOT, "profiles", "%s_%s" % (tag, f)))
        n += 1
    print("installed %d files as profiles/%s_*" % (n, tag))


if __name__ == "__main__":
    main()
