"""Register / spill counts and the instruction order of one K-loop iteration of conv_wino_kernel.
usage: python tools/isa_wino.py [variant]      (reads c2m_amd/lib/conv_wino[_variant].o; scratch files under gpurun_out/isa)"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else ""
    obj = os.path.join(ROOT, "c2m_amd", "lib", "conv_wino" + ("_" + variant if variant else "") + ".o")
    work = os.path.join(ROOT, "gpurun_out", "isa")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    shutil.copy(obj, os.path.join(work, "k.o"))
    subprocess.run([LLVM + "llvm-objdump", "-d", "--offloading", "k.o"], cwd=work, capture_output=True)
    co = [f for f in os.listdir(work) if "amdgcn" in f][0]
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], cwd=work, capture_output=True, text=True).stdout
    at = notes.index(".name:           _Z16conv_wino_kernel")
    for key, blk in (("group_segment_fixed_size", notes[at - 1200:at]), ("vgpr_count", notes[at:at + 600]),
                     ("vgpr_spill_count", notes[at:at + 600]), ("sgpr_spill_count", notes[at:at + 600])):
        m = re.findall(r"\." + key + r":\s+(\d+)", blk)
        print(key, m[-1] if m else "?")
    asm = subprocess.run([LLVM + "llvm-objdump", "-d", co], cwd=work, capture_output=True, text=True).stdout
    lines = [ln.split("//")[0].strip() for ln in asm.splitlines()]
    st = [i for i, ln in enumerate(lines) if "conv_wino_kernel" in ln and ln.endswith(">:")][0]
    en = [i for i, ln in enumerate(lines) if "wino_filter_kernel" in ln and ln.endswith(">:")][0]
    ins = [ln for ln in lines[st:en] if ln and not ln.endswith(":") and not ln.startswith(".")]
    idx = [i for i, x in enumerate(ins) if x.startswith("v_mfma")]
    print("instructions", len(ins), "mfma", len(idx), "scratch ops", sum("scratch_" in x for x in ins))
    body, out, n = ins[idx[31] + 1:idx[63] + 1], [], 0
    for x in body:
        op = x.split()[0]
        if op.startswith("v_mfma"):
            n += 1
            out.append(f"M{n}")
        elif op.startswith("s_waitcnt"):
            out.append("[" + x.replace("s_waitcnt ", "") + "]")
        elif op == "s_barrier":
            out.append("BARRIER")
        elif op.startswith("ds_read"):
            out.append("dr")
        elif op.startswith("ds_write"):
            out.append("dw")
        elif op.startswith("global_load"):
            out.append("GL")
        elif op.startswith("buffer_load"):
            out.append("DMA")
        elif op.startswith("scratch_"):
            out.append("SCR")
        elif op.startswith("v_"):
            out.append("v")
        elif op.startswith("s_cbranch"):
            out.append("br")
        else:
            out.append("s")
    print(" ".join(out))


if __name__ == "__main__":
    main()
