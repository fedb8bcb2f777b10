"""Instruction statistics of the kernels in a gfx950 assembly listing (hipcc -save-temps / -S): registers, and per class the number of
instructions in the LARGEST loop of each kernel (the unrolled march loop of the stencil kernels) - the static counterpart of SQ_INSTS_VALU.

usage: python tools/asm_loop_stats.py file.s [name-filter]
"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
CLASSES = [("accvgpr", r"v_accvgpr"), ("dpp", r"_dpp"), ("cndmask", r"v_cndmask"), ("f64 add", r"v_add_f64"), ("f64 mul", r"v_mul_f64"), ("f64 fma", r"v_fma_f64"),
           ("f32 arith", r"v_(add|mul|fma|sub|pk_\w+)_f32"), ("cvt", r"v_cvt"), ("mov", r"v_mov_b"), ("readlane", r"v_(read|write)lane|v_readfirstlane"),
           ("other valu", r"v_"), ("vmem load", r"global_load|buffer_load"), ("vmem store", r"global_store|buffer_store"), ("lds", r"ds_"), ("salu", r"s_")]
i = 0
while i < len(text):
    m = re.match(r"^(_Z\w+):", text[i])
    if not m:
        i += 1
        continue
    name = m.group(1)
    j = i + 1
    while j < len(text) and not text[j].startswith(".Lfunc_end"):
        j += 1
    body = text[i:j]
    i = j
    if flt and flt not in name:
        continue
    labels = {}
    for k, line in enumerate(body):
        lm = re.match(r"^(\.LBB\d+_\d+):", line)
        if lm:
            labels[lm.group(1)] = k
    best = None
    for k, line in enumerate(body):
        bm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", line)
        if bm and bm.group(1) in labels and labels[bm.group(1)] < k:
            span = (labels[bm.group(1)], k)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    def count(lines):
        c = Counter()
        for line in lines:
            ins = line.strip().split(" ")[0].split("\t")[0]
            if not ins or ins.startswith((".", ";")) or ins.endswith(":"):
                continue
            for cname, pat in CLASSES:
                if re.match(pat, ins) or (cname == "dpp" and "_dpp" in line.split(";")[0]):
                    c[cname] += 1
                    break
        return c
    whole = count(body)
    loop = count(body[best[0]:best[1]]) if best else Counter()
    regs = {}
    for line in text[j:j + 400]:
        rm = re.match(r"\s*\.set\s+" + re.escape(name) + r"\.(num_vgpr|num_agpr|private_seg_size),\s*(\d+)", line)
        if rm:
            regs[rm.group(1)] = int(rm.group(2))
    valu = lambda c: sum(v for k, v in c.items() if k not in ("vmem load", "vmem store", "lds", "salu"))
    print(f"{name[:110]}\n   vgpr {regs.get('num_vgpr')} agpr {regs.get('num_agpr')} scratch {regs.get('private_seg_size')}   VALU whole {valu(whole)}  largest loop {valu(loop)}")
    print("   loop: " + ", ".join(f"{k} {loop[k]}" for k, _ in CLASSES if loop[k]))
