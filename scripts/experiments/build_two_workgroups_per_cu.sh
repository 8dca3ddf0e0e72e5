#!/bin/bash
# Builds lbdrn-msic_amd/liblbdrn_hip_two.so = the library with scripts/experiments/two_workgroups_per_cu.patch applied
# (the TWO mode of k_train_stream: two training workgroups per CU), in a scratch copy of csrc/ -- the tree
# itself is not touched.  Run the variant with
#     LBDRN_HIP_LIB=$PWD/lbdrn-msic_amd/liblbdrn_hip_two.so LBDRN_STREAM2=1 python3 bench.py --no-cpu-baseline --no-other-configs
# (LBDRN_STREAM2 unset: the same library runs the shipped one-workgroup-per-CU kernel -- the A/B of DESIGN.md 4.1).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/lbdrn-msic_amd" "$TMP/include"
cp -r "$ROOT/lbdrn-msic_amd/csrc" "$TMP/lbdrn-msic_amd/csrc"
cp "$ROOT/include/"*.h "$TMP/include/"
rm -f "$TMP"/lbdrn-msic_amd/csrc/*.o
(cd "$TMP" && git init -q . && git apply "$ROOT/scripts/experiments/two_workgroups_per_cu.patch")
python3 - "$TMP" "$ROOT" <<'PY'
import importlib.util, os, shutil, sys
tmp, root = sys.argv[1], sys.argv[2]
spec = importlib.util.spec_from_file_location("b", os.path.join(tmp, "lbdrn-msic_amd", "csrc", "build.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
out = m.build(force=True)
dst = os.path.join(root, "lbdrn-msic_amd", "liblbdrn_hip_two.so")
shutil.copy(out, dst)
print(dst)
PY
