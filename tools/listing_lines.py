"""Instructions per source line of one kernel, from a listing made with
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S -gline-tables-only --cuda-device-only -o trace.s aic_trace.hip
usage: python tools/listing_lines.py trace.s <mangled kernel name or substring> [--ranges a-b:name,c-d:name,...] [--top N]
Static counts (a divergent branch skipped by a whole wave costs nothing at run time): read them next to the phase counters.
The listing's `.loc` comments carry the inline stack (`file:line:col @[ caller:line:col @[ ... ] ]`): every instruction is attributed BOTH to the
innermost line (`--ranges` over helper bodies: lvl_init, lvl_next, lm_interpolated_light ... count every inlined copy together) and, with
`--outer-ranges`, to the OUTERMOST line of the main source file -- the statement of the kernel body it was inlined into -- so that ranges over the
kernel body (the SHADE section, ENTER, NEWRAY ...) hold everything those sections execute, helpers included. --sections splits the stream by the
markers `; AIC_SECTION <name>` (asm comments the kernel emits with -DAIC_SECTION_MARKS) instead, which follow code position and perturb the build."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        if op.startswith(("s_load", "s_buffer_load", "s_store")):
            return "smem"
        if op.startswith(("s_waitcnt", "s_nop", "s_sleep")):
            return "wait"
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
            return "branch"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    ranges, outer_ranges, top, sections = [], [], 25, False
    args = sys.argv[3:]
    i = 0
    while i < len(args):
        if args[i] in ("--ranges", "--outer-ranges"):
            for part in args[i + 1].split(","):
                r, nm = part.split(":")
                a, b = r.split("-")
                (ranges if args[i] == "--ranges" else outer_ranges).append((int(a), int(b), nm))
            i += 2
        elif args[i] == "--top":
            top = int(args[i + 1])
            i += 2
        elif args[i] == "--sections":
            sections = True
            i += 1
        else:
            raise SystemExit("unknown argument " + args[i])
    per_line = collections.defaultdict(collections.Counter)
    per_outer = collections.defaultdict(collections.Counter)
    per_section = collections.defaultdict(collections.Counter)
    inside, cur_line, cur_file, cur_sec, cur_outer = False, 0, 1, "-", 0
    main_re = re.compile(r"aic_trace\.hip:(\d+):\d+")
    total = collections.Counter()
    loc_re = re.compile(r"\s*\.loc\s+(\d+)\s+(\d+)")
    label_re = re.compile(r"^([A-Za-z_.$][\w.$]*):")
    with open(path, errors="replace") as f:
        for line in f:
            if not inside:
                m = label_re.match(line)
                if m and name in m.group(1) and not m.group(1).startswith("."):
                    inside = True
                continue
            if line.startswith(".Lfunc_end") or line.lstrip().startswith(".end_amdhsa_kernel"):
                break
            m = loc_re.match(line)
            if m:
                cur_file, cur_line = int(m.group(1)), int(m.group(2))
                hits = main_re.findall(line)
                cur_outer = int(hits[-1]) if hits else (cur_line if cur_file == 0 else 0)
                continue
            s = line.strip()
            if s.startswith("; AIC_SECTION"):
                cur_sec = s.split()[2]
                continue
            if not s or s.startswith((";", ".", "//")) or label_re.match(line):
                continue
            op = s.split()[0]
            k = classify(op)
            per_line[(cur_file, cur_line)][k] += 1
            per_outer[cur_outer][k] += 1
            per_section[cur_sec][k] += 1
            total[k] += 1
    if not inside:
        raise SystemExit("kernel not found: " + name)
    kinds = ["valu", "salu", "lds", "vmem", "smem", "branch", "wait", "other"]
    print("total      " + "  ".join(f"{k} {total[k]}" for k in kinds), " all", sum(total.values()))
    if sections:
        for sec, c in per_section.items():
            print(f"section {sec:14s} " + "  ".join(f"{k} {c[k]}" for k in kinds), " all", sum(c.values()))
    if ranges:
        for a, b, nm in ranges:
            c = collections.Counter()
            for (fl, ln), cc in per_line.items():
                if fl == 0 and a <= ln <= b:
                    c.update(cc)
            print(f"lines {a}-{b} {nm:18s} " + "  ".join(f"{k} {c[k]}" for k in kinds), " all", sum(c.values()))
    if outer_ranges:
        covered = collections.Counter()
        for a, b, nm in outer_ranges:
            c = collections.Counter()
            for ln, cc in per_outer.items():
                if a <= ln <= b:
                    c.update(cc)
            covered.update(c)
            print(f"outer {a}-{b} {nm:18s} " + "  ".join(f"{k} {c[k]}" for k in kinds), " all", sum(c.values()))
        rest = collections.Counter(total)
        rest.subtract(covered)
        print(f"outer (elsewhere)            " + "  ".join(f"{k} {rest[k]}" for k in kinds), " all", sum(rest.values()))
    rows = sorted(per_line.items(), key=lambda kv: -sum(kv[1].values()))[:top]
    for (fl, ln), c in rows:
        print(f"file {fl} line {ln:5d}  " + "  ".join(f"{k} {c[k]}" for k in kinds if c[k]), " all", sum(c.values()))


if __name__ == "__main__":
    main()
