"""Tiny docopt-compatible parser for the reference's `--flag=<value>` command lines (docopt is not installed here).
Flag names, defaults and the usage text are the reference's (run_infer_tile.py:1-23, run_infer_wsi.py:1-37)."""
import re
import sys


def parse(doc, argv=None, version=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    opts = {}
    for m in re.finditer(r"^\s+(--[a-z_]+)(=<[a-z]+>)?\s+.*?(?:\[default: (.*?)\])?$", doc.split("Options:")[1], re.M):
        flag, has_val, default = m.group(1), m.group(2), m.group(3)
        opts[flag] = (default if has_val else False)
    if "-h" in argv or "--help" in argv:
        print(doc)
        sys.exit(0)
    if "--version" in argv:
        print(version)
        sys.exit(0)
    i = 0
    while i < len(argv):
        a = argv[i]
        k, eq, v = a.partition("=")
        if k not in opts:
            print(doc)
            sys.exit("unknown option %s" % a)
        if opts[k] is False or opts[k] is True:
            opts[k] = True
        elif eq:
            opts[k] = v
        else:
            i += 1
            opts[k] = argv[i]
        i += 1
    return opts
