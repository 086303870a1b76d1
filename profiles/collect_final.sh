#!/bin/bash
# copies the artefacts gpurun merged into gpurun_out/final/ (profiles/final_artifacts.sh) into the tracked profiles/ directory
cd "$(dirname "$0")/.."
R=${1:-r02}
for f in gpurun_out/final/*; do
	b=$(basename $f)
	case $b in
		fold_traffic_*.json) cp $f profiles/$b ;;
		bench.err) ;;
		*) cp $f profiles/${R}_final_$b ;;
	esac
done
ls profiles | grep ${R}_final | wc -l
