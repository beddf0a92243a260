#!/bin/bash
# Builds another copy of the library with extra compiler flags: scripts/build_variant.sh <name> [flags...]  -> build_exp/lib<name>.so
# (select it at run time with RT355_LIB=build_exp/lib<name>.so)
name=$1; shift
cd "$(dirname "$0")/.." && mkdir -p build_exp
python3 - "$name" "$@" <<'PY'
import sys
import __graft_entry__ as g
name, flags = sys.argv[1], sys.argv[2:]
g.build_library(f"build_exp/lib{name}.so", extra_flags=flags, obj_dir=f"build_exp/obj_{name}", force=True)
PY
