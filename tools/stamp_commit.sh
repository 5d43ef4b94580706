# Writes tools/.commit (the commit the working tree is at, "+dirty" when it differs from it) -- run in the build container right
# before a gpurun call whose outputs are committed as evidence: the GPU box's copy has no .git, tools/pmc_kernels.py reads this file.
cd "$(dirname "$0")/.." && h=$(git rev-parse --short HEAD) && { git diff --quiet HEAD -- . ':!profiles' ':!gpurun_out' || h="$h+dirty"; } && echo "$h" > tools/.commit && cat tools/.commit
