"""Register / scratch / kernarg inventory of the kernels in one object file of csrc/_obj (no GPU needed).
  python tools/kernel_regs.py conv_wgrad [name filter]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
obj = os.path.join(ROOT, "pmt_learning_for_semantic_segmentation_and_disparity_amd", "csrc", "_obj", sys.argv[1] + ".o")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    subprocess.check_call([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
    t = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
rows = []
for k in re.split(r"\n\s+- \.agpr_count", t)[1:]:
    k = ".agpr_count" + k
    m = re.search(r"\.name:\s+(\S+)", k)
    if not m:
        continue
    g = lambda key: int(re.search(key + r":\s+(\d+)", k).group(1)) if re.search(key + r":\s+(\d+)", k) else -1
    rows.append((m.group(1), g(r"\.vgpr_count"), g(r"\.agpr_count"), g(r"\.sgpr_count"), g(r"\.private_segment_fixed_size"),
                 g(r"\.vgpr_spill_count"), g(r"\.kernarg_segment_size")))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
    n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")
    if flt in n:
        print("%-72s vgpr %3d agpr %3d sgpr %3d scratch %4d spill %3d kernarg %d" % ((n[:72],) + r[1:]))
