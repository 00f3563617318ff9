#!/bin/bash
# Round-4 PMC passes (separate rocprofv3 --pmc runs per counter set): the dominant class's biggest row (plain 3x3, 96 columns, 256-pixel
# tiles: g4.first at the generator pass's batch), the same layer through the wave-specialised kernel, and the phase launch of round 3.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/r4_pmc
EXTRA="" bash tools/pmc_conv.sh r4_pmc/plain bf16x3 "full g4.first" > gpurun_out/r4_pmc/plain.log 2>&1
EXTRA="--tune=-1,-1,7,-1" bash tools/pmc_conv.sh r4_pmc/ws bf16x3 "full g4.first" > gpurun_out/r4_pmc/ws.log 2>&1
EXTRA="--phases-only" bash tools/pmc_conv.sh r4_pmc/phase bf16x3 "full up_g4.first" > gpurun_out/r4_pmc/phase.log 2>&1
for t in plain ws phase; do echo "== $t"; cat gpurun_out/r4_pmc/$t/pmc_pass*.csv | grep -v "^kernel" | cut -c1-200; grep "fwd" gpurun_out/r4_pmc/$t/p1.log | tail -2; done
