"""Static census of k_trace's node-step loop from the compiler's ISA (VERDICT r5 #3b): instructions per class in the inner loop of one kernel instantiation.
   hipcc ... --cuda-device-only -S -o trace.s gravit_amd/csrc/trace.hip ; python tools/step_census.py trace.s [kernel-substring]
The inner loop = the loop (by LLVM's own block annotations) whose header block holds the node's v_cvt_f32_ubyte decodes.  Reported: the straight-line head of the
loop (fetch, decode, slab tests, sorting network: every lane at an inner node executes all of it) and the whole loop body (all paths: fast pushes, the spill path, the
pop, the parking check, the vote), per class: VALU, of which v_cvt / v_pk_fma / v_cndmask / v_cmp; SALU; LDS; VMEM; branches / waits / nops."""
import re, sys

def classify(op):
    if op.startswith("v_cmpx") or op.startswith("v_cmp"): return "valu", "v_cmp"
    if op.startswith("v_cvt_f32_ubyte"): return "valu", "v_cvt_ubyte"
    if op.startswith("v_pk_fma"): return "valu", "v_pk_fma"
    if op.startswith("v_cndmask"): return "valu", "v_cndmask"
    if op.startswith("v_"): return "valu", "other_valu"
    if op.startswith("ds_"): return "lds", None
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem", None
    if op.startswith(("s_cbranch", "s_branch")): return "branch", None
    if op.startswith(("s_waitcnt", "s_nop")): return "wait_nop", None
    if op.startswith("s_"): return "salu", None
    return "other", None

def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "k_traceILb0ELb1ELi0ELb0ELb1ELb0"
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    # blocks
    blocks, cur = [], None
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            cur = {"label": m.group(1), "note": m.group(2) or "", "ins": []}
            blocks.append(cur)
            continue
        if cur is None: continue
        if l.startswith(";") and cur["ins"] == []:
            cur["note"] += " " + l
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."): continue
        cur["ins"].append(t.split()[0])
    hdr = next(b for b in blocks if any(o.startswith("v_cvt_f32_ubyte") for o in b["ins"]))
    # the loop header of that block
    m = re.search(r"Header=(BB\d+_\d+)", hdr["note"])
    head_label = "." + ("L" + m.group(1) if m else hdr["label"][1:])
    if "Inner Loop Header" in hdr["note"] or "Loop Header" in hdr["note"]: head_label = hdr["label"]
    loop = [b for b in blocks if b["label"] == head_label or ("Header=" + head_label[2:]) in b["note"]]
    def census(bs):
        tot, sub = {}, {}
        for b in bs:
            for o in b["ins"]:
                c, s = classify(o)
                tot[c] = tot.get(c, 0) + 1
                if s: sub[s] = sub.get(s, 0) + 1
        return tot, sub
    print("kernel: %s" % lines[start].split(":")[0][:120])
    print("inner loop header %s, %d blocks in the loop" % (head_label, len(loop)))
    for name, bs in (("straight-line head (fetch + decode + slab tests + sorting network)", [hdr]), ("whole loop body, all paths", loop)):
        tot, sub = census(bs)
        print("  %s:" % name)
        print("    VALU %d (v_cvt_f32_ubyte %d, v_pk_fma_f32 %d, v_cndmask %d, v_cmp %d, other %d)  SALU %d  LDS %d  VMEM %d  branches %d  waits/nops %d" % (
            tot.get("valu", 0), sub.get("v_cvt_ubyte", 0), sub.get("v_pk_fma", 0), sub.get("v_cndmask", 0), sub.get("v_cmp", 0), sub.get("other_valu", 0),
            tot.get("salu", 0), tot.get("lds", 0), tot.get("vmem", 0), tot.get("branch", 0), tot.get("wait_nop", 0)))
    meta = "\n".join(lines)
    mm = re.search(r"\.name:\s+\S*" + re.escape(want) + r"\S*\n(.*?)\.wavefront_size", meta, re.S)
    if mm:
        g = lambda k: re.search(k + r":\s+(\d+)", mm.group(1)).group(1)
        print("  registers: %s VGPRs (%s spilled), %s SGPRs (%s spilled), scratch %s B" % (g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"), g(r"\.sgpr_count"), g(r"\.sgpr_spill_count"), g(r"\.private_segment_fixed_size")))

main()
