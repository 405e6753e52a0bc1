#!/bin/bash
# Diagnostic: builds and runs the instruction-cost microbenchmark (tests/tools_valu_rate3.hip -> build_tools/valu_rate3, git-ignored) on the GPU box;
# its table is what tests/tools_issue_budget.py prices the kernels' instruction streams with (profiles/r5_instruction_costs.txt).
set -e
mkdir -p build_tools
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o build_tools/valu_rate3 tests/tools_valu_rate3.hip
./build_tools/valu_rate3
