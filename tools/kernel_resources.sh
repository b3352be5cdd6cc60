#!/bin/bash
# Prints VGPRs / SGPRs / LDS bytes / scratch bytes of every gfx950 kernel in a hipcc object file (no GPU needed).
# Usage: tools/kernel_resources.sh mi-fieldcalc_amd/csrc/mifc_stencil_split.o [name filter]
set -e
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" "$T/fat.bin"
$LLVM/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$T/fat.bin" --output="$T/k.co"
$LLVM/llvm-readelf --notes "$T/k.co" | python3 -c '
import re, sys, subprocess
txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
for blk in txt.split("- .agpr_count")[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = g("name")
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = name.replace("mifc::(anonymous namespace)::", "").replace("(mifc::SRowsParams)", "").replace("(mifc::(anonymous namespace)::RowsParams)", "")
    if flt in name:
        rows.append((name, g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
print("%-70s %5s %5s %7s %7s" % ("kernel", "vgpr", "sgpr", "lds", "scratch"))
for r in sorted(rows):
    print("%-70s %5s %5s %7s %7s" % r)
' "${2:-}"
rm -rf "$T"
