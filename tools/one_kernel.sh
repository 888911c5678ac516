#!/bin/bash
# Register / scratch / occupancy line (and the ISA in /tmp/one_kernel_<tag>.s) of ONE kernel of the fp64 build, without a GPU:
#   tools/one_kernel.sh 'k_backward<true, false, true, true, float>' [tag] [-D flags ...]
K=${1:-k_backward<true, false, true, true, float>}; TAG=${2:-k}; shift 2 2>/dev/null
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --offload-device-only -S -Rpass-analysis=kernel-resource-usage \
  "-DKERNEL=$K" "$@" tools/probes/one_kernel.hip -o /tmp/one_kernel_$TAG.s 2>/tmp/one_kernel_$TAG.err
python3 - "$TAG" <<'PY'
import re, subprocess, sys
err = open("/tmp/one_kernel_%s.err" % sys.argv[1]).read()
if "error:" in err:
    print(err[-3000:]); sys.exit(1)
for b in re.split(r"remark: [^\n]*Function Name: ", err)[1:]:
    name = b.split("\n")[0].split()[0]
    if not name.startswith(("_ZN5aoc64", "_ZN5aoc32")): continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if "<" not in dem: continue
    g = lambda k: int(m.group(1)) if (m := re.search(k + r": (\d+)", b)) else -1
    print("VGPR %d AGPR %d SGPR %d scratch %d waves/SIMD %d LDS %d  %s" % (g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), re.sub(r"\(.*", "", dem.replace("void ", ""))))
PY
