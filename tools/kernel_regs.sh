#!/bin/bash
# Register / scratch / LDS / occupancy report of every kernel of one translation unit (compiled to ISA text, nothing linked):
#   tools/kernel_regs.sh conv_igemm [extra -D flags]
# Prints: kernel name, VGPRs (arch + acc), SGPRs, scratch bytes, static LDS bytes, occupancy (waves per SIMD).
set -e
unit=$1; shift
src=$(dirname "$0")/../mrfp_amd/csrc/$unit.hip
out=/tmp/regs_$unit.s
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S "$@" "$src" -o "$out"
python3 - "$out" <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    def g(k):
        r = re.search(r"\.amdhsa_%s (\S+)" % k, body)
        return r.group(1) if r else "?"
    # the human-readable comment block after the kernel has the occupancy
    c = re.search(re.escape(name) + r".*?; Occupancy: (\d+)", txt[m.end():m.end() + 4000], re.S)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"^void mrfp::", "", dem)[:110]
    print("%-112s vgpr %-4s acc_off %-4s sgpr %-4s scratch %-5s lds %-6s occ %s" % (
        dem, g("next_free_vgpr"), g("accum_offset"), g("next_free_sgpr"), g("private_segment_fixed_size"),
        g("group_segment_fixed_size"), c.group(1) if c else "?"))
PY
