#!/bin/bash
# Experiment builds for tools/ab_kernels.py (timing ablations; they compute wrong numbers on purpose).
set -e
cd "$(dirname "$0")/.."
for spec in nosplit:-DNSA_ABL_NOSPLIT nosp:-DNSA_ABL_NOSOFTPLUS nope:-DNSA_ABL_NOPE nogather:-DNSA_ABL_NOGATHER \
            nogrid:-DNSA_ABL_NOGRID nomfma:-DNSA_ABL_NOMFMA wcache:-DNSA_EXP_WCACHE "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  NSA_BUILD_TAG=$tag NSA_EXTRA_HIPCC_FLAGS="$flags" python -m nicer_slam_amd.build > /dev/null &
done
wait
ls -la nicer_slam_amd/lib/
