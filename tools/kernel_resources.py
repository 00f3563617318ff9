"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: VGPR / AGPR / scratch / occupancy / LDS per kernel.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -Iinclude -Rpass-analysis=kernel-resource-usage \
        skillful_nowcasting_amd/csrc/conv.hip -o /tmp/conv.o 2> /tmp/conv.ru
    python tools/kernel_resources.py /tmp/conv.ru
"""
import re
import subprocess
import sys


def field(block, key):
    m = re.search(re.escape(key) + r": (\d+)", block)
    return m.group(1) if m else "?"


def main(path):
    txt = open(path).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    for b in blocks:
        sym = b.split("\n")[0].strip().split()[0]
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:64]
        print(f"{name:64s} VGPR {field(b, 'VGPRs'):>4s} AGPR {field(b, 'AGPRs'):>4s} scratch {field(b, 'ScratchSize [bytes/lane]'):>4s} "
              f"waves/SIMD {field(b, 'Occupancy [waves/SIMD]'):>2s} LDS {field(b, 'LDS Size [bytes/block]'):>6s}")


if __name__ == "__main__":
    main(sys.argv[1])
