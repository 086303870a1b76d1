#!/bin/bash
# Run HERE (where git is) before a gpurun that regenerates measured artefacts: records the commit the artefacts will be stamped with and
# the hash of its device sources.  Refuses when dumphfdl_amd/csrc has uncommitted changes -- a traffic record must name the code it ran.
cd "$(dirname "$0")/.."
if ! git diff --quiet HEAD -- dumphfdl_amd/csrc; then
	echo "stamp.sh: dumphfdl_amd/csrc differs from HEAD: commit first" >&2
	exit 1
fi
mkdir -p profiles/scripts
python - <<'PY'
import json, subprocess, sys
sys.path.insert(0, ".")
import bench
commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
json.dump(dict(commit=commit, csrc_sha16=bench.csrc_hash()), open("profiles/scripts/stamp.json", "w"))
print("stamped", commit, bench.csrc_hash())
PY
