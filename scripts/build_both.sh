#!/bin/bash
# builds libmmg.so and the -DMMG_TIMING library; exits non-zero on a compile error
python -c "
from multimodalgame_amd import build; build.build_library(force=True, verbose=False); build.build_timing_library(verbose=False)" 2>&1 | grep -E "error|warning: unused" | head -20
python -c "
from multimodalgame_amd import build; import sys; sys.exit(0 if build.check_library() else 1)"
