#!/bin/bash
# Re-takes every committed measurement of the round for the build that is in the tree, in three gpurun calls (≈ 20 GPU-minutes),
# and copies the results under profiles/.  Run from the repo root in the build container after `make -C dau-convnet_amd/csrc all tuning`:
#
#   bash tools/refresh_artefacts.sh
#
# 1. bench lines of every workload, the rehearsals, rocprofv3 kernel stats and the PMC passes  (tools/steps_artefacts.txt, lines 1-33)
# 2. the PMC traffic file is keyed on the build id and bench.py quotes it only for the same build, so it is copied into profiles/
#    BEFORE the headline line is taken again (line 1), together with the full GPU suite and the fuzz soak (lines 34-35)
# 3. tools/collect_artefacts.py copies / renames the logs into profiles/r4_*, tools/design_table.py prints the table of DESIGN.md §6
# Afterwards: paste the table into DESIGN.md, update the `lib` id, the test counts and the headline numbers in DESIGN.md / README.md /
# profiles/r4_fuzz_soak_final_build.txt / profiles/INDEX.md, run the CPU suite, commit.
set -e
cd "$(dirname "$0")/.."
STEPS=tools/steps_artefacts.txt
rm -f gpurun_out/parity_margins.jsonl
bash tools/gpurun_retry.sh 1200 "bash tools/run_steps.sh < <(sed -n 1,33p $STEPS)"
cp gpurun_out/fa_ns_pmc_traffic.json profiles/r4_pmc_traffic.json
# (the headline line and its rocprofv3 statistics in ONE call = one box: the boxes differ by up to 6 %)
bash tools/gpurun_retry.sh 1200 "bash tools/run_steps.sh < <(sed -n '1p;27,31p;34,35p' $STEPS)"
python3 tools/collect_artefacts.py
python3 tools/design_table.py
