#!/usr/bin/env python3
"""Static check of the trace kernels' ISA: nothing may read (or overwrite) the destination register of a voxel lookup that an
inline-asm statement has issued until the `s_waitcnt vmcnt(0)` that follows it.

The stepping loop of csrc/aic_trace.hip issues its `global_load_ushort` lookups from inline assembly and waits for them in a later
statement, so that bookkeeping overlaps the load. The compiler does not know the destination is pending: a register copy it places
between the two statements (e.g. to satisfy a tied operand) reads the old value. Round 4 met exactly that (profiles/r04_experiments.txt B);
this scan runs over every instantiation of trace_image_kernel and is part of the CPU test suite (tests/test_kernel_isa.py).

usage: python tools/check_pending_loads.py [file.s]   (without a file: compiles csrc/aic_trace.hip to assembly first)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def compile_to_asm(extra_flags=()):
    out = os.path.join(tempfile.mkdtemp(prefix="aic_isa_"), "aic_trace.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-S",
           *extra_flags, os.path.join(ROOT, "all_is_cubes_amd", "csrc", "aic_trace.hip"), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(path):
    """Returns (kernels scanned, lookups seen, list of violations)."""
    problems, kernels, lookups = [], 0, 0
    name, pending = None, {}
    for ln, line in enumerate(open(path), 1):
        if line.startswith("_ZN3aic18trace_image_kernel") and line.rstrip().endswith(("E:", "E: ")) or (line.startswith("_ZN3aic18trace_image_kernel") and ":" in line.split(";")[0]):
            name, pending = line.split(":")[0], {}
            kernels += 1
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            name, pending = None, {}
            continue
        code = line.split(";")[0].strip()
        if not code or code.endswith(":") or code.startswith("."):
            continue
        op = code.split()[0]
        if op.startswith("s_waitcnt") and "vmcnt(0)" in code:
            pending = {}
            continue
        operands = code[len(op):]
        parts = [p.strip() for p in operands.split(",")]
        if op == "global_load_ushort":
            lookups += 1
            dst = regs_of(parts[0])
            for r in dst:
                pending[r] = ln
            srcs = regs_of(",".join(parts[1:]))
            hit = srcs & set(pending) - dst
            if hit:
                problems.append((name, ln, code, sorted(hit)))
            continue
        if pending and (op.startswith("v_") or op.startswith("ds_") or op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_")):
            hit = regs_of(operands) & set(pending)
            if hit:
                problems.append((name, ln, code, sorted(hit)))
    return kernels, lookups, problems


def kernel_resources(path):
    """{kernel symbol: dict(vgprs, scratch_bytes, lds_bytes, wg_threads)} from the .amdhsa_* directives of every trace_image_kernel in the assembly."""
    out, name = {}, None
    for line in open(path):
        t = line.strip()
        if t.startswith(".amdhsa_kernel "):
            sym = t.split()[1]
            name = sym if sym.startswith("_ZN3aic18trace_image_kernel") else None
            if name:
                out[name] = {}
        elif t.startswith(".end_amdhsa_kernel"):
            name = None
        elif name and t.startswith(".amdhsa_"):
            parts = t.split()
            if len(parts) == 2 and parts[1].lstrip("-").isdigit():
                key = {".amdhsa_next_free_vgpr": "vgprs", ".amdhsa_private_segment_fixed_size": "scratch_bytes", ".amdhsa_group_segment_fixed_size": "lds_bytes"}.get(parts[0])
                if key:
                    out[name][key] = int(parts[1])
    # threads per workgroup (the launch bound) from the metadata block: "- .max_flat_workgroup_size: N" ... ".name: symbol"
    size = None
    for line in open(path):
        t = line.strip().lstrip("- ")
        if t.startswith(".max_flat_workgroup_size:"):
            size = int(t.split(":")[1])
        elif t.startswith(".name:") and size is not None:
            sym = t.split(":", 1)[1].strip()
            if sym in out:
                out[sym]["wg_threads"] = size
            size = None
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else compile_to_asm()
    kernels, lookups, problems = scan(path)
    for name, ln, code, regs in problems:
        print(f"{name}: line {ln}: `{code}` touches v{regs} while a lookup into it is in flight")
    print(f"{kernels} kernels, {lookups} lookups, {len(problems)} violations")
    return 1 if problems or not kernels else 0


if __name__ == "__main__":
    sys.exit(main())
