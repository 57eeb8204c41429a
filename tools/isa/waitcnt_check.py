"""Static check of gfx950 assembly (hipcc -S): does any instruction read a VGPR whose VMEM load may still be in flight?

Straight-line analysis per basic block (state cleared at labels: no false positives from joins, misses cross-block hazards):
VMEM loads in issue order with their destination registers; `s_waitcnt vmcnt(N)` retires all but the youngest N VMEM ops
(stores count in vmcnt on gfx9 and are kept as entries without destinations); any later instruction that reads or overwrites a
register of a still-outstanding load is reported.  Written in round 4 to test the hypothesis that the SLP-vectorised build's
wrong results under memory contention are an s_waitcnt bug around 64-bit (v_pk_*) operands.
    python tools/isa/waitcnt_check.py file.s [kernel-name-substring]
"""
import re
import sys

REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check(path, want=""):
    text = open(path).read()
    total = 0
    for m in re.finditer(r'^(_Z\w+):\s*; @\1\n(.*?)^\.Lfunc_end', text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want not in name:
            continue
        pending = []          # [(line no, set of dest regs)] in issue order
        for ln, line in enumerate(body.split("\n")):
            l = line.split(";")[0].strip()
            if not l:
                continue
            if l.endswith(":"):
                pending = []
                continue
            op = l.split()[0]
            args = l[len(op):]
            if op == "s_waitcnt":
                mm = re.search(r'vmcnt\((\d+)\)', args)
                if mm:
                    n = int(mm.group(1))
                    pending = pending[len(pending) - n:] if n else []
                elif "vmcnt" not in args and re.fullmatch(r'\s*\d+\s*', args or ""):
                    pending = []
                continue
            if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_barrier")):
                if op != "s_barrier":
                    pending = []
                continue
            is_vmem = op.startswith(("buffer_", "global_", "flat_", "scratch_", "tbuffer_"))
            ops_ = [a.strip() for a in args.split(",")]
            if is_vmem:
                is_load = "load" in op or "atomic" in op
                lds = " lds" in l
                dest = regs(ops_[0]) if (is_load and not lds) else set()
                srcs = set()
                for a in (ops_[1:] if (is_load and not lds) else ops_):
                    srcs |= regs(a)
                for pl, pd in pending:
                    if pd & (srcs | dest):
                        total += 1
                        print(f"{name[:60]} line {ln}: `{l}` touches {sorted(pd & (srcs | dest))} of the load at line {pl} still in flight")
                pending.append((ln, dest))
                continue
            used = set()
            for a in ops_:
                used |= regs(a)
            for pl, pd in pending:
                if pd & used:
                    total += 1
                    print(f"{name[:60]} line {ln}: `{l}` touches {sorted(pd & used)} of the load at line {pl} still in flight")
    print(path, "violations:", total)
    return total


if __name__ == "__main__":
    check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
