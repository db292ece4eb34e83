#!/bin/bash
# Build every micro-benchmark in this directory for gfx950 (binaries are git-ignored; they still travel with gpurun).
cd "$(dirname "$0")"
for f in *_probe.hip; do hipcc --offload-arch=gfx950 -O3 "$f" -o "${f%.hip}" 2>/dev/null && echo "built ${f%.hip}"; done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../sg_pr_amd/csrc tail_bench.hip -o tail_bench 2>/dev/null && echo "built tail_bench"
