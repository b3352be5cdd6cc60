#!/usr/bin/env python3
"""Instruction histogram of the innermost (largest backward-branch) loop of a kernel in an llvm-objdump -d listing.
    tools/isa_histogram.py listing.s <mangled-name substring> [cells per loop iteration]
Prints the mnemonics by count and a cycle estimate per cell (wave64 on a 16-lane SIMD: 4 cycles per fp32 / int VALU op,
8 per fp64 op, 16 per transcendental f64; a rough issue model, not a simulator)."""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    cells = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*%s.*>:" % re.escape(key), l))
    end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i])), len(lines))
    body = []
    for l in lines[start + 1:end]:
        m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            body.append((int(m.group(3), 16), m.group(1), m.group(2)))
    addr_index = {a: i for i, (a, _, _) in enumerate(body)}
    best = None
    for i, (a, op, args) in enumerate(body):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"(-?\d+)\s*$", args)
            if not m:
                continue
            # objdump prints the simm16 word offset; target = a + 4 + 4*off
            off = int(m.group(1))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt < a and tgt in addr_index:
                span = i - addr_index[tgt]
                if best is None or span > best[2]:
                    best = (addr_index[tgt], i, span)
    if best is None:
        print("no backward branch found; whole kernel")
        lo, hi = 0, len(body) - 1
    else:
        lo, hi = best[0], best[1]
    hist = collections.Counter(op for _, op, _ in body[lo:hi + 1])
    total = sum(hist.values())
    valu = {k: v for k, v in hist.items() if k.startswith("v_")}
    f64 = sum(v for k, v in valu.items() if "f64" in k and not k.startswith("v_cvt"))
    cvt64 = sum(v for k, v in valu.items() if "f64" in k and k.startswith("v_cvt"))
    trans64 = sum(v for k, v in valu.items() if k in ("v_rcp_f64_e32", "v_rsq_f64_e32", "v_sqrt_f64_e32", "v_rcp_f64_e64"))
    trans32 = sum(v for k, v in valu.items() if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32", k))
    nvalu = sum(valu.values())
    cyc = 4 * (nvalu - f64 - trans32) + 8 * (f64 - trans64) + 16 * trans64 + 16 * trans32
    print("loop of %d instructions (%d listed in the kernel): VALU %d (f64 arithmetic %d, f64 converts %d, f64 transcendental %d, f32 transcendental %d), "
          "SALU %d, LDS %d, VMEM %d" % (total, len(body), nvalu, f64, cvt64, trans64, trans32, sum(v for k, v in hist.items() if k.startswith("s_")),
                                        sum(v for k, v in hist.items() if k.startswith("ds_")), sum(v for k, v in hist.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))))
    print("per cell (%g cells per iteration): VALU %.1f, issue cycles per wave %.1f" % (cells, nvalu / cells, cyc / cells))
    for k, v in hist.most_common(60):
        print("  %-28s %5d  %6.2f per cell" % (k, v, v / cells))


if __name__ == "__main__":
    main()
