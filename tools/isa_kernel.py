"""Register / LDS / scratch use of every kernel of one object file whose name contains a pattern.
usage: python tools/isa_kernel.py conv_igemm conv_gather_nc8        (reads c2m_amd/lib/<file>.o; scratch under gpurun_out/isa_k)"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    obj = os.path.join(ROOT, "c2m_amd", "lib", sys.argv[1] + ".o")
    pat = sys.argv[2]
    work = os.path.join(ROOT, "gpurun_out", "isa_k")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    shutil.copy(obj, os.path.join(work, "k.o"))
    subprocess.run([LLVM + "llvm-objdump", "-d", "--offloading", "k.o"], cwd=work, capture_output=True)
    co = [f for f in os.listdir(work) if "amdgcn" in f][0]
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], cwd=work, capture_output=True, text=True).stdout
    ents = notes.split("- .agpr_count:")
    for e in ents[1:]:
        name = re.search(r"\.name:\s+(\S+)", e)
        if not name or pat not in name.group(1):
            continue
        dem = name.group(1)
        f = lambda k: (re.findall(r"\." + k + r":\s+(\d+)", e) or ["?"])[0]
        print(f"{dem[:90]:90s} vgpr {f('vgpr_count')} agpr {e.split()[0]} sgpr {f('sgpr_count')} lds {f('group_segment_fixed_size')} "
              f"scratch {f('private_segment_fixed_size')} spill {f('vgpr_spill_count')}")
    if len(sys.argv) > 3:      # dump the disassembly of the matching kernels
        asm = subprocess.run([LLVM + "llvm-objdump", "-d", co], cwd=work, capture_output=True, text=True).stdout
        open(os.path.join(work, "k.s"), "w").write(asm)
        print("disassembly:", os.path.join(work, "k.s"))


if __name__ == "__main__":
    main()
