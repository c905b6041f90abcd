"""Kernel resource table (VGPRs, AGPRs, SGPRs, LDS, scratch) of a hipcc object file or shared library built for gfx950, read from the code object's notes.
    python tools/kernel_resources.py ipc_amd/_obj/mf_numeric.hip.o [name filter]"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"


def resources(path):
    tmp = tempfile.mkdtemp(prefix="kres_")
    fat, elf = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.elf")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={elf}"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", elf], capture_output=True, text=True, check=True).stdout
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
        name = g("name").group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("ipcgpu::(anonymous namespace)::", "").replace("void ", "")
        dem = re.sub(r"\((?!anonymous).*", "", dem)
        out.append((dem, int(g("vgpr_count").group(1)), int(g("agpr_count").group(1)), int(g("sgpr_count").group(1)), int(g("group_segment_fixed_size").group(1)),
                    int(g("private_segment_fixed_size").group(1)), int(g("max_flat_workgroup_size").group(1))))
    return out


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%-44s %5s %5s %5s %7s %8s %5s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "wg"))
    for r in sorted(resources(sys.argv[1])):
        if flt in r[0]:
            print("%-44s %5d %5d %5d %7d %8d %5d" % r)
