#!/usr/bin/env python3
"""dev tool: per-opcode histogram of the hot march loop of a kernel, from the hipcc assembly listing.

    python tools/isa_histogram.py [kernel-substring ...]    (default: the strict kernels without media, with media, with the noise tables, and the fast one)

Compiles csrc/rrt_hip.hip with the build's own flags + -save-temps into a scratch directory, finds the
outermost loop of each requested kernel (LLVM annotates every block with its loop header), splits its blocks
into the STRAIGHT path (blocks reached by fall-through or by the loop's own control flow) and SIDE blocks
(targets of a forward `s_cbranch_vccnz`: the guarded, practically never taken paths), and prints instruction
counts by class and opcode.  The listing is the evidence for the "instructions per RK4 step" figures in
DESIGN.md; the dynamic cross-check is SQ_INSTS_VALU / (waves x steps) from profiles/*_summary.json.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from relativisticraytracer_amd import build  # noqa: E402


def listing():
    d = tempfile.mkdtemp(prefix="rrt_isa_")
    cmd = [build.hipcc_path()] + [f for f in build.HIPCC_FLAGS if f != "-shared"] + os.environ.get("RRT_ISA_FLAGS", "").split() + ["-save-temps", "-c"] + build.SOURCES + ["-o", os.path.join(d, "x.o")]
    subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = [x for x in os.listdir(d) if x.endswith("gfx950.s")][0]
    return open(os.path.join(d, f)).read().split("\n")


def demangled(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def classify(op):
    if op.startswith("v_"):
        if re.match(r"v_(rsq|rcp|sqrt|exp|log|sin|cos)_", op): return "VALU transcendental (half rate)"
        if re.match(r"v_cmp|v_cmpx", op): return "VALU compare"
        if re.match(r"v_cndmask|v_mov|v_readfirstlane|v_readlane|v_writelane", op): return "VALU select/move"
        if re.match(r"v_(fma|fmac|mad|mac)_", op): return "VALU fma"
        if re.match(r"v_mul_", op): return "VALU mul"
        if re.match(r"v_(add|sub|subrev)_f", op): return "VALU add/sub f32"
        return "VALU other"
    if op.startswith("s_"):
        if "branch" in op: return "SALU branch"
        if op in ("s_nop", "s_waitcnt"): return "wait/nop"
        return "SALU"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    if op.startswith("ds_"): return "LDS"
    return "other"


def analyse(src, name, pretty):
    start = [i for i, l in enumerate(src) if l.startswith(name + ":")][0]
    end = [i for i in range(start, len(src)) if src[i].strip().startswith(".Lfunc_end")][0]
    body = src[start:end]
    res = {}
    for l in src[end:end + 80]:
        m = re.search(r"; (NumVgprs|TotalNumSgprs|Occupancy|ScratchSize|codeLenInByte)\s*[:=]\s*(\d+)", l)
        if m: res[m.group(1)] = int(m.group(2))
    # blocks
    blocks, cur = [], None
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            cur = {"label": m.group(1), "comment": m.group(2) or "", "ins": []}
            blocks.append(cur)
            continue
        t = l.split(";")[0].strip()
        if cur is not None and not t and l.strip().startswith(";") and not cur["ins"]:
            cur["comment"] += " " + l.strip()            # loop annotations follow the label on comment lines
        if cur is not None and t and not t.startswith("."):
            cur["ins"].append(t)
            if re.match(r"s_c?branch", t):                   # a block ends at every branch (anonymous continuation block)
                cur = {"label": cur["label"].split("+")[0] + "+%d" % (len(blocks)), "comment": cur["comment"], "ins": []}
                blocks.append(cur)
    # outermost loop with the most instructions
    hdrs = collections.Counter()
    for b in blocks:
        m = re.search(r"Header=(BB\d+_\d+) Depth=1", b["comment"])
        if m: hdrs[m.group(1)] += len(b["ins"])
    for b in blocks:
        if re.search(r"Loop Header: Depth=1", b["comment"]): hdrs[b["label"][2:]] += len(b["ins"])
    if not hdrs:
        print(f"{pretty}: no loop found"); return
    hdr = hdrs.most_common(1)[0][0]
    loop = [b for b in blocks if b["label"].split("+")[0] == ".L" + hdr or re.search(r"Header=%s Depth=1" % hdr, b["comment"])]
    labels = {b["label"] for b in loop}
    # STRAIGHT path = what a wavefront executes in a typical iteration: walk from the header, conditional
    # branches fall through (LLVM lays the likely successor out next and moves guarded / expect-false paths
    # out of line), unconditional branches are followed, until control returns to the header.
    order = {b["label"]: k for k, b in enumerate(blocks)}
    # Round 3: the loop has TWO straight paths -- the wave-uniform vacuum step (r >= 30 in every lane: h = 0.3 as literals,
    # recognisable by the folded h/2 = 0x3e19999a) and the generic step.  The generic path is the fall-through walk; the
    # vacuum path takes a conditional branch exactly when its target is the block that holds the vacuum step's literals.
    VAC_LIT = "0x3e19999a"
    vac_entry = set()
    for k, b in enumerate(blocks):
        if b in loop and not b["label"].count("+"):
            head = []
            for bb in blocks[k:k + 3]:
                if bb is not b and not bb["label"].startswith(b["label"] + "+"):
                    break
                head += bb["ins"]
            if any(VAC_LIT in t for t in head[:8]):
                vac_entry.add(b["label"])

    def walk(take_vacuum):
        path, seen = [], set()
        cur_i = order[".L" + hdr]
        while True:
            b = blocks[cur_i]
            if b["label"] in seen or b["label"] not in labels: break
            seen.add(b["label"]); path.append(b)
            last = b["ins"][-1] if b["ins"] else ""
            m = re.match(r"s_branch\s+(\.LBB\d+_\d+)", last)
            mc = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", last)
            if m:
                if m.group(1) == ".L" + hdr: break
                cur_i = order[m.group(1)]
            elif mc and take_vacuum and mc.group(1) in vac_entry:
                cur_i = order[mc.group(1)]
            elif mc and "execz" in last and cur_i + 1 < len(blocks) and any(
                    re.match(r"v_(rsq|sqrt)_f32", t) for t in blocks[cur_i + 1]["ins"]):
                cur_i = order[mc.group(1)]                   # an in-line rejected-seed fall-back: skipped when no lane needs it
            else:
                cur_i += 1
                if cur_i >= len(blocks): break
        return path, seen

    # Round 6: the vacuum steps run in a NESTED loop (rrt_kernels.h: vacuum_run), its body written out RRT_VAC_INNER times.  Its
    # straight path: from the inner header until control returns to it; conditional branches fall through (rejected seeds, small
    # radii, exits are the unlikely successors) EXCEPT a `s_cbranch_vccz` whose target is a stub of <= 2 instructions -- that is the
    # "no lane is beyond r = 250" skip of the escape test's dot product, taken on all but a ray's last few steps.
    inner = None
    for b in blocks:
        if re.search(r"Inner Loop Header: Depth=2", b["comment"]) and not b["label"].count("+"):
            members = [x for x in blocks if x["label"].split("+")[0] == b["label"] or re.search(r"Header=%s Depth=2" % b["label"][2:], x["comment"])]
            if sum(len(x["ins"]) for x in members) > 250 and any(VAC_LIT in t for x in members for t in x["ins"]):
                inner = (b["label"], {x["label"] for x in members})
    nested = []
    if inner:
        ihdr, ilabels = inner
        first_of = {}
        for k, bb in enumerate(blocks):
            first_of.setdefault(bb["label"].split("+")[0], k)
        cur_i, seen_i = order[ihdr], set()
        while True:
            b = blocks[cur_i]
            if b["label"] in seen_i or b["label"] not in ilabels: break
            seen_i.add(b["label"]); nested.append(b)
            last = b["ins"][-1] if b["ins"] else ""
            m = re.match(r"s_branch\s+(\.LBB\d+_\d+)", last)
            mz = re.match(r"s_cbranch_vccz\s+(\.LBB\d+_\d+)", last)
            if m:
                if m.group(1) == ihdr: break
                cur_i = order[m.group(1)]
            elif mz and mz.group(1) in order and len(blocks[order[mz.group(1)]]["ins"]) <= 2 and not re.match(r"s_c?branch", (blocks[order[mz.group(1)]]["ins"] or [""])[-1]):
                cur_i = order[mz.group(1)]
            else:
                cur_i += 1
                if cur_i >= len(blocks): break
    straight, seen = walk(False)
    vacuum, vseen = walk(True) if vac_entry else ([], set())
    if nested:
        seen |= {b["label"] for b in nested}
    sideb = [b for b in loop if b["label"] not in seen and b["label"] not in vseen]
    print(f"== {pretty}")
    print(f"   registers: {res.get('NumVgprs')} VGPR, {res.get('TotalNumSgprs')} SGPR, occupancy {res.get('Occupancy')} waves/SIMD, "
          f"scratch {res.get('ScratchSize')} B, code {res.get('codeLenInByte')} B")
    sections = []
    if nested:
        m = re.search(r"#define RRT_VAC_INNER (\d+)", open(os.path.join(ROOT, "relativisticraytracer_amd", "csrc", "rrt_kernels.h")).read())
        n_steps = int(m.group(1)) if m else 2
        flags = os.environ.get("RRT_ISA_FLAGS", "")
        mf = re.search(r"-DRRT_VAC_INNER=(\d+)", flags)
        if mf: n_steps = max(1, int(mf.group(1)))
        ins_n = [t.split()[0] for b in nested for t in b["ins"]]
        valu_n = sum(1 for o in ins_n if o.startswith("v_"))
        mov_n = sum(1 for o in ins_n if o.startswith("v_mov"))
        print(f"   VACUUM LOOP (nested, body written out {n_steps}x): straight path {len(nested)} blocks, {len(ins_n)} instructions, {valu_n} VALU ({mov_n} v_mov) "
              f"= {valu_n / n_steps:.1f} VALU per RK4 step")
        sections.append((f"vacuum loop, straight path of one trip = {n_steps} RK4 steps (every lane at r >= 30: h = 0.3 folded, no zone tests)", nested))
    if vacuum and [b["label"] for b in vacuum] != [b["label"] for b in straight]:
        sections.append(("VACUUM path of the march loop (every lane at r >= 30: one RK4 step, h = 0.3 folded, no zone tests)", vacuum))
    sections.append(("generic path of the march loop (one RK4 step" + (", media blocks included" if len(loop) > 60 else "") + ")", straight))
    sections.append(("blocks of the loop off those paths (guarded fall-backs: rejected seeds, a stage radius < 1; lanes leaving)", sideb))
    if os.environ.get("RRT_ISA_DUMP") and os.environ["RRT_ISA_DUMP"] in pretty:      # the vacuum path's listing (or the generic one)
        for b in (nested or vacuum or straight):
            print("      " + b["label"])
            for t in b["ins"]:
                print("         " + t)
    for title, bl in sections:
        ins = [t.split()[0] for b in bl for t in b["ins"]]
        cls = collections.Counter(classify(o) for o in ins)
        ops = collections.Counter(o for o in ins)
        valu = sum(v for k, v in cls.items() if k.startswith("VALU"))
        slots = valu + cls.get("VALU transcendental (half rate)", 0)
        print(f"   {title}: {len(bl)} blocks, {len(ins)} instructions, {valu} VALU ({slots} issue slots counting half-rate ops twice)")
        for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
            print(f"      {k:34s} {v:5d}")
        if bl is not sideb:
            print("      opcodes: " + ", ".join(f"{o} {n}" for o, n in ops.most_common(40)))
    print()


def main():
    src = listing()
    names = [l.split(":")[0] for l in src if re.match(r"^_ZN12_GLOBAL__N_1\w+:", l)]
    dm = demangled(names)
    want = sys.argv[1:] or ["raymarch_pixels<true, 0, false, 0>", "raymarch_pixels<false, 0, false, 0>",
                            "raymarch_pixels<true, 1, false, 0>", "raymarch_pixels<true, 2, false, 0>",
                            "raymarch_pixels<true, 0, false, 2>", "raymarch_pixels<true, 2, false, 2>", "raymarch_pixels<true, 0, false, 1>"]
    for w in want:
        for n in names:
            if w in dm[n]:
                analyse(src, n, dm[n].split("(")[1].split("::")[-1] if "anonymous" in dm[n] else dm[n])


if __name__ == "__main__":
    main()
