#!/bin/bash
# The scaling series of BASELINE.md section 2 on a multi-GPU node (C2 at N = 1; C4 and C5 at N = 2, 4, 8): superseded by
# tools/first_8gpu_run.sh, which also runs the device-group tests first and both launcher-free forms of bench.py.
exec bash "$(dirname "$0")/first_8gpu_run.sh" "$@"
