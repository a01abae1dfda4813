"""Registers, spills, LDS and scratch of every kernel of the library, from the compiler's own metadata:
    hipcc -S --cuda-device-only (the Makefile's flags) on each .hip file, then the amdhsa.kernels table of the assembly.
    python tools/kernel_resources.py [> profiles/r04_kernel_resources.txt]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "synthpy_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-munsafe-fp-atomics",
         "-I" + os.path.join(ROOT, "include"), "-I" + CS, "-S", "--cuda-device-only"] + sys.argv[1:]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return out[:len(names)]
    except OSError:
        return names


print("# hipcc -S --cuda-device-only with the Makefile's flags; amdhsa.kernels metadata.  VGPR = .vgpr_count (the unified count, AGPRs included);")
print("# 'spilled VGPR' with 0 bytes of scratch = parked in AGPRs (v_accvgpr_write / _read), not in memory")
print(f"# {'kernel':<70s} VGPR  AGPR  SGPR  spilled VGPR / SGPR   LDS (static)  scratch B/lane   waves/SIMD by VGPRs")
for src in ("trace.hip", "deposit.hip", "volume.hip", "field.hip", "beam.hip", "comm.hip"):
    with tempfile.TemporaryDirectory() as d:
        s = os.path.join(d, "a.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", s, os.path.join(CS, src)], check=True, capture_output=True)
        txt = open(s).read()
    blocks = txt.split("  - .agpr_count:")[1:]
    names = [re.search(r"\.name:\s+(\S+)", b).group(1) for b in blocks]
    for b, name in zip(blocks, demangle(names)):
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))
        agpr = int(re.match(r"\s*(\d+)", b).group(1))
        v = g("vgpr_count")
        waves = min(8, 512 // max(1, ((v + 7) // 8) * 8))  # .vgpr_count is the unified count (AGPRs included)
        name = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", "")).split("(")[0]
        print(f"{src[:-4] + ': ' + name:<72s} {v:4d}  {agpr:4d}  {g('sgpr_count'):4d}  {g('vgpr_spill_count'):6d} / {g('sgpr_spill_count'):<6d}      "
              f"{g('group_segment_fixed_size'):8d}  {g('private_segment_fixed_size'):8d}        {waves}")
