#!/bin/bash
# developer helper: compile one csrc/*.hip unit to gfx950 assembly in /tmp and print per-kernel register / scratch statistics
# usage: scripts/dev_kstat.sh conv_wgrad_wino.hip [extra hipcc flags]
U=$1; shift
T=/tmp/kstat_$$; mkdir -p $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --offload-device-only -S -o $T/k.s "$@" /root/repo/cerberus_amd/csrc/$U || exit 1
python3 - $T/k.s <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n  - |\Z)", s[s.index("amdhsa.kernels"):], re.S):
    body = m.group(2)
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", body) or [None, "?"])[1]
    print("%-70s vgpr %s agpr %s sgpr %s scratch %s spills %s lds %s" % (m.group(1)[:70], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("group_segment_fixed_size")))
PY
cp $T/k.s /tmp/last_kstat.s; rm -rf $T
