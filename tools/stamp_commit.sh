#!/bin/bash
# Build container, before a profiling gpurun: record HEAD (the GPU box gets no .git) for the summaries' "commit" field.
cd "$(dirname "$0")/.." && { git rev-parse --short HEAD; git diff --quiet || echo "+uncommitted"; } | tr '\n' ' ' | sed 's/ *$//' > .commit_stamp && cat .commit_stamp && echo
